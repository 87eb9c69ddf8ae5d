// mhx_expr.cpp -- what an expression handed to mhx_set_function_expr IS.
//
// The reference's interface is "pass a lambda" (mcmc-fitting.lisp:1134-1137): the host shims walk
// the closure's body into a C-syntax expression and libmhx compiles it as written (mhx_rtc.cpp).
// Most closures of a fitting code are a polynomial background plus Gaussian or Lorentzian peaks,
// for which libmhx has kernels that do far better than the expression as written (peak skipping
// per window, the uniform-grid recurrence, parameters in scalar registers: 4x on BASELINE's
// config 2).  This file recognises such a body BELOW the C ABI, so that every host - the Lisp
// shim, Python, plain C - gets those kernels from the same text:
//
//     bg(x) + sum of peaks,   bg(x) = c0 + c1 x + c2 x^2 ...  (terms: key, key * x,
//                                     key * ipow(x, n), key * x * x, in any order of the factors)
//     Gaussian peak    a * exp(-S)              S = ipow(u, 2) | pow(u, 2.0) | u * u,
//     Lorentzian peak  a / (1 + S), a * (1 / (1 + S))         u = (x - mu) / w
//
// with a, mu, w, c_i bare parameter names, each used exactly once (-S also as -1 * S).  Anything
// else - cross terms, a cube, a key used twice, conditionals - is not of the shape and is
// compiled as written.  The enumerated model evaluates the same function with its own fused
// arithmetic: a few ulp from the closure's own rounding per point, inside the path's stated
// tolerance (1e-12 * sum |term|); mhx_set_expr_recognition(e, 0) keeps the text.
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>

#include "../../include/mhx.h"
#include "mhx_rtc.hpp"

namespace mhx {
namespace {

struct Node;
typedef std::shared_ptr<Node> NodeP;
struct Node {
  enum Kind { NUM, ID, NEG, ADD, SUB, MUL, DIV, CALL, OTHER } kind = OTHER;
  double num = 0.0;
  std::string name;         // ID, CALL
  std::vector<NodeP> kids;  // operands / arguments
};

NodeP mk(Node::Kind k) {
  NodeP n(new Node());
  n->kind = k;
  return n;
}

// recursive descent over the grammar rtc_prepare_expr admits; conditionals, comparisons and
// logic come out as OTHER nodes (never part of a recognised shape)
struct Parser {
  const std::string& s;
  size_t i = 0;
  bool bad = false;
  int depth = 0;
  explicit Parser(const std::string& str) : s(str) {}
  void ws() {
    while (i < s.size() && isspace((unsigned char)s[i])) ++i;
  }
  bool eat(char c) {
    ws();
    if (i < s.size() && s[i] == c) {
      ++i;
      return true;
    }
    return false;
  }
  bool peek2(const char* op) {
    ws();
    return s.compare(i, strlen(op), op) == 0;
  }
  NodeP other(NodeP a, NodeP b = nullptr, NodeP c = nullptr) {
    NodeP n = mk(Node::OTHER);
    n->kids.push_back(a);
    if (b) n->kids.push_back(b);
    if (c) n->kids.push_back(c);
    return n;
  }
  NodeP ternary() {
    if (++depth > 200) {
      bad = true;
      return mk(Node::OTHER);
    }
    NodeP c = lor();
    if (eat('?')) {
      NodeP a = ternary();
      if (!eat(':')) bad = true;
      NodeP b = ternary();
      c = other(c, a, b);
    }
    --depth;
    return c;
  }
  NodeP lor() {
    NodeP a = land();
    while (peek2("||")) {
      i += 2;
      a = other(a, land());
    }
    return a;
  }
  NodeP land() {
    NodeP a = cmp();
    while (peek2("&&")) {
      i += 2;
      a = other(a, cmp());
    }
    return a;
  }
  NodeP cmp() {
    NodeP a = add();
    for (;;) {
      ws();
      if (peek2("<=") || peek2(">=") || peek2("==") || peek2("!=")) {
        i += 2;
        a = other(a, add());
      } else if (i < s.size() && (s[i] == '<' || s[i] == '>')) {
        ++i;
        a = other(a, add());
      } else {
        return a;
      }
    }
  }
  NodeP add() {
    NodeP a = mul();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '+' || s[i] == '-')) {
        const char op = s[i++];
        NodeP n = mk(op == '+' ? Node::ADD : Node::SUB);
        n->kids = {a, mul()};
        a = n;
      } else {
        return a;
      }
    }
  }
  NodeP mul() {
    NodeP a = unary();
    for (;;) {
      ws();
      if (i < s.size() && (s[i] == '*' || s[i] == '/')) {
        const char op = s[i++];
        NodeP n = mk(op == '*' ? Node::MUL : Node::DIV);
        n->kids = {a, unary()};
        a = n;
      } else {
        return a;
      }
    }
  }
  NodeP unary() {
    ws();
    if (i < s.size() && s[i] == '-') {
      ++i;
      NodeP k = unary();
      if (k->kind == Node::NUM) {  // a negative literal
        k->num = -k->num;
        return k;
      }
      NodeP n = mk(Node::NEG);
      n->kids = {k};
      return n;
    }
    if (i < s.size() && s[i] == '+') {
      ++i;
      return unary();
    }
    if (i < s.size() && s[i] == '!' && !peek2("!=")) {
      ++i;
      return other(unary());
    }
    return primary();
  }
  NodeP primary() {
    ws();
    if (i >= s.size()) {
      bad = true;
      return mk(Node::OTHER);
    }
    const unsigned char c = (unsigned char)s[i];
    if (c == '(') {
      ++i;
      NodeP n = ternary();
      if (!eat(')')) bad = true;
      return n;
    }
    if (isdigit(c) || c == '.') {
      char* end = nullptr;
      const double v = strtod(s.c_str() + i, &end);
      if (end == s.c_str() + i) {
        bad = true;
        return mk(Node::OTHER);
      }
      i = (size_t)(end - s.c_str());
      NodeP n = mk(Node::NUM);
      n->num = v;
      return n;
    }
    if (isalpha(c) || c == '_') {
      size_t j = i;
      while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
      const std::string id = s.substr(i, j - i);
      i = j;
      if (eat('(')) {
        NodeP n = mk(Node::CALL);
        n->name = id;
        if (!eat(')')) {
          do n->kids.push_back(ternary());
          while (eat(','));
          if (!eat(')')) bad = true;
        }
        return n;
      }
      NodeP n = mk(Node::ID);
      n->name = id;
      return n;
    }
    bad = true;
    return mk(Node::OTHER);
  }
};

bool same(const NodeP& a, const NodeP& b) {
  if (a->kind != b->kind || a->kids.size() != b->kids.size()) return false;
  if (a->kind == Node::OTHER) return false;  // never compared structurally
  if (a->kind == Node::NUM && a->num != b->num) return false;
  if ((a->kind == Node::ID || a->kind == Node::CALL) && a->name != b->name) return false;
  for (size_t k = 0; k < a->kids.size(); ++k)
    if (!same(a->kids[k], b->kids[k])) return false;
  return true;
}

struct Shape {
  const std::set<std::string>& keys;
  explicit Shape(const std::set<std::string>& k) : keys(k) {}
  bool is_key(const NodeP& n) const { return n->kind == Node::ID && keys.count(n->name) != 0; }
  static bool is_x(const NodeP& n) { return n->kind == Node::ID && n->name == "x"; }
  static bool is_num(const NodeP& n, double v) { return n->kind == Node::NUM && n->num == v; }
  static bool is_call(const NodeP& n, const char* f, size_t args) {
    return n->kind == Node::CALL && n->name == f && n->kids.size() == args;
  }
  // (mu, w) of u = (x - mu) / w
  bool reduced(const NodeP& n, std::string* mu, std::string* w) const {
    if (n->kind != Node::DIV || !is_key(n->kids[1])) return false;
    const NodeP& d = n->kids[0];
    if (d->kind != Node::SUB || !is_x(d->kids[0]) || !is_key(d->kids[1])) return false;
    *mu = d->kids[1]->name;
    *w = n->kids[1]->name;
    return true;
  }
  bool squared(const NodeP& n, std::string* mu, std::string* w) const {
    if ((is_call(n, "ipow", 2) || is_call(n, "pow", 2)) && is_num(n->kids[1], 2.0))
      return reduced(n->kids[0], mu, w);
    if (n->kind == Node::MUL && same(n->kids[0], n->kids[1])) return reduced(n->kids[0], mu, w);
    return false;
  }
  bool neg_squared(const NodeP& n, std::string* mu, std::string* w) const {
    if (n->kind == Node::NEG) return squared(n->kids[0], mu, w);
    if (n->kind == Node::MUL) {
      if (is_num(n->kids[0], -1.0)) return squared(n->kids[1], mu, w);
      if (is_num(n->kids[1], -1.0)) return squared(n->kids[0], mu, w);
      // -u * u: C's unary minus binds to the first factor
      if (n->kids[0]->kind == Node::NEG && same(n->kids[0]->kids[0], n->kids[1]))
        return reduced(n->kids[1], mu, w);
      if (n->kids[1]->kind == Node::NEG && same(n->kids[1]->kids[0], n->kids[0]))
        return reduced(n->kids[0], mu, w);
    }
    return false;
  }
  bool one_plus_squared(const NodeP& n, std::string* mu, std::string* w) const {
    if (n->kind != Node::ADD) return false;
    if (is_num(n->kids[0], 1.0)) return squared(n->kids[1], mu, w);
    if (is_num(n->kids[1], 1.0)) return squared(n->kids[0], mu, w);
    return false;
  }
  // n if f is x^n written as x, ipow(x, n) or x * x ..., else 0
  static int x_power(const NodeP& n) {
    if (is_x(n)) return 1;
    if (is_call(n, "ipow", 2) && is_x(n->kids[0]) && n->kids[1]->kind == Node::NUM) {
      const double v = n->kids[1]->num;
      if (v >= 1.0 && v <= 9.0 && v == (double)(int)v) return (int)v;
    }
    return 0;
  }
};

void flatten(const NodeP& n, Node::Kind k, std::vector<NodeP>* out) {
  if (n->kind == k) {
    flatten(n->kids[0], k, out);
    flatten(n->kids[1], k, out);
  } else {
    out->push_back(n);
  }
}

}  // namespace

bool rtc_recognise(const std::string& expr, const std::vector<std::string>& names,
                   RecognisedModel* out) {
  out->model = -1;
  out->order.clear();
  out->shape[0] = out->shape[1] = 0;
  Parser ps(expr);
  NodeP root = ps.ternary();
  ps.ws();
  if (ps.bad || ps.i != expr.size()) return false;
  const std::set<std::string> keys(names.begin(), names.end());
  if (keys.size() != names.size()) return false;
  const Shape sh(keys);
  struct Peak {
    std::string a, mu, w;
  };
  std::map<int, std::string> bg;
  std::vector<Peak> gauss, lorentz;
  std::vector<NodeP> terms;
  flatten(root, Node::ADD, &terms);
  for (const NodeP& t : terms) {
    int deg = -1;
    std::string key;
    Peak pk;
    if (sh.is_key(t)) {
      deg = 0;
      key = t->name;
    } else if (t->kind == Node::MUL) {
      std::vector<NodeP> facs, rest;
      flatten(t, Node::MUL, &facs);
      int nkeys = 0;
      for (const NodeP& f : facs) {
        if (sh.is_key(f)) {
          ++nkeys;
          key = f->name;
        } else {
          rest.push_back(f);
        }
      }
      if (nkeys != 1 || rest.empty()) return false;
      if (rest.size() == 1) {
        const NodeP& g = rest[0];
        if (Shape::is_call(g, "exp", 1) && sh.neg_squared(g->kids[0], &pk.mu, &pk.w)) {
          pk.a = key;
          gauss.push_back(pk);
          continue;
        }
        if (g->kind == Node::DIV && Shape::is_num(g->kids[0], 1.0) &&
            sh.one_plus_squared(g->kids[1], &pk.mu, &pk.w)) {
          pk.a = key;
          lorentz.push_back(pk);
          continue;
        }
        deg = Shape::x_power(g);
        if (deg == 0) return false;
      } else {
        for (const NodeP& f : rest)
          if (!Shape::is_x(f)) return false;
        deg = (int)rest.size();
      }
    } else if (t->kind == Node::DIV && sh.is_key(t->kids[0]) &&
               sh.one_plus_squared(t->kids[1], &pk.mu, &pk.w)) {
      pk.a = t->kids[0]->name;
      lorentz.push_back(pk);
      continue;
    } else {
      return false;
    }
    if (bg.count(deg)) return false;
    bg[deg] = key;
  }
  const int nbg = (int)bg.size();
  for (int dgr = 0; dgr < nbg; ++dgr)
    if (!bg.count(dgr)) return false;  // degrees 0 .. nbg-1, each once
  std::vector<std::string> used;
  for (int dgr = 0; dgr < nbg; ++dgr) used.push_back(bg[dgr]);
  const std::vector<Peak>& peaks = gauss.empty() ? lorentz : gauss;
  if (!gauss.empty() && !lorentz.empty()) return false;
  if (peaks.empty()) {
    if (nbg < 1 || nbg > 16) return false;
    out->model = MHX_MODEL_POLY;
  } else {
    if (peaks.size() > 6 || nbg > 4) return false;
    for (const Peak& p : peaks) {
      used.push_back(p.a);
      used.push_back(p.mu);
      used.push_back(p.w);
    }
    out->model = gauss.empty() ? MHX_MODEL_LORENTZ_PEAKS : MHX_MODEL_GAUSS_PEAKS;
    out->shape[0] = nbg;
    out->shape[1] = (int)peaks.size();
  }
  // every parameter exactly once (names the closure declares and never reads are simply not
  // part of the enumerated model's gather map)
  if (std::set<std::string>(used.begin(), used.end()).size() != used.size()) {
    out->model = -1;
    return false;
  }
  for (const std::string& u : used)
    for (size_t j = 0; j < names.size(); ++j)
      if (names[j] == u) out->order.push_back((int)j);
  return true;
}

}  // namespace mhx
