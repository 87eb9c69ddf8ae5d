// mhx_rtc.cpp -- run-time compilation of user-expression models / prior bodies with hiprtc.
#include "mhx_rtc.hpp"

#include <dlfcn.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <sys/stat.h>
#include <unistd.h>

#include "mhx_launch.hpp"

// the device sources, embedded at build time (Makefile -> tools/embed_src.py)
#include "mhx_embedded_src.inc"

namespace mhx {

// ---- hiprtc, loaded lazily so that libmhx.so itself has no link-time dependency on it ------
namespace {
typedef struct _hiprtcProgram* hiprtcProgram;
struct Hiprtc {
  void* h = nullptr;
  int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**);
  int (*CompileProgram)(hiprtcProgram, int, const char**);
  int (*GetProgramLogSize)(hiprtcProgram, size_t*);
  int (*GetProgramLog)(hiprtcProgram, char*);
  int (*GetCodeSize)(hiprtcProgram, size_t*);
  int (*GetCode)(hiprtcProgram, char*);
  int (*DestroyProgram)(hiprtcProgram*);
  const char* (*GetErrorString)(int);
  int (*Version)(int*, int*);
  bool ok = false;
};
Hiprtc& rtc() {
  static Hiprtc r;
  if (r.h) return r;
  const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  for (const char* n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h) return r;
#define SYM(field, name) *(void**)(&r.field) = dlsym(r.h, name)
  SYM(CreateProgram, "hiprtcCreateProgram");
  SYM(CompileProgram, "hiprtcCompileProgram");
  SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
  SYM(GetProgramLog, "hiprtcGetProgramLog");
  SYM(GetCodeSize, "hiprtcGetCodeSize");
  SYM(GetCode, "hiprtcGetCode");
  SYM(DestroyProgram, "hiprtcDestroyProgram");
  SYM(GetErrorString, "hiprtcGetErrorString");
  SYM(Version, "hiprtcVersion");
#undef SYM
  r.ok = r.CreateProgram && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog &&
         r.GetCodeSize && r.GetCode && r.DestroyProgram;
  return r;
}
}  // namespace

UserProgram::~UserProgram() {
  if (module) (void)hipModuleUnload(module);
}

// ---- expression grammar ----------------------------------------------------------------------
static const char* kFunctions[] = {"exp", "log", "sqrt", "sin", "cos", "tan", "atan", "tanh",
                                   "abs", "pow", "min",  "max", "floor", "ipow", nullptr};

int rtc_prepare_expr(const std::string& expr, const std::vector<std::string>& names,
                     const char* extra, std::string* out, std::string* err) {
  std::set<std::string> params(names.begin(), names.end());
  std::set<std::string> extras;
  if (extra) {
    std::istringstream is(extra);
    std::string w;
    while (is >> w) extras.insert(w);
  }
  std::set<std::string> funcs;
  for (int i = 0; kFunctions[i]; ++i) funcs.insert(kFunctions[i]);
  std::string o;
  size_t i = 0, n = expr.size();
  int depth = 0;
  if (n == 0 || n > 16384) {
    *err = "expression is empty or longer than 16384 characters";
    return -1;
  }
  while (i < n) {
    const unsigned char c = (unsigned char)expr[i];
    if (isspace(c)) {
      o += ' ';
      ++i;
    } else if (isalpha(c) || c == '_') {
      size_t j = i;
      while (j < n && (isalnum((unsigned char)expr[j]) || expr[j] == '_')) ++j;
      const std::string id = expr.substr(i, j - i);
      if (extras.count(id)) {
        o += id;
      } else if (params.count(id)) {
        o += "p_" + id;
      } else if (funcs.count(id)) {
        if (id == "abs") o += "fabs";
        else if (id == "min") o += "mhx_ux_min";
        else if (id == "max") o += "mhx_ux_max";
        else if (id == "ipow") o += "mhx_ux_ipow";
        else if (id == "exp") o += "mhx_ux_exp";
        else if (id == "log") o += "mhx_ux_log";
        else o += id;
      } else {
        *err = "unknown identifier '" + id + "' in expression";
        return -1;
      }
      i = j;
    } else if (isdigit(c) || (c == '.' && i + 1 < n && isdigit((unsigned char)expr[i + 1]))) {
      size_t j = i;
      bool is_float = false;
      while (j < n && isdigit((unsigned char)expr[j])) ++j;
      if (j < n && expr[j] == '.') {
        is_float = true;
        ++j;
        while (j < n && isdigit((unsigned char)expr[j])) ++j;
      }
      if (j < n && (expr[j] == 'e' || expr[j] == 'E')) {
        size_t k = j + 1;
        if (k < n && (expr[k] == '+' || expr[k] == '-')) ++k;
        if (k < n && isdigit((unsigned char)expr[k])) {
          is_float = true;
          j = k;
          while (j < n && isdigit((unsigned char)expr[j])) ++j;
        }
      }
      o += expr.substr(i, j - i);
      if (!is_float) o += ".0";  // (/ 1 2) must not become an integer division
      if (j < n && (isalpha((unsigned char)expr[j]) || expr[j] == '_')) {
        *err = "malformed number in expression";
        return -1;
      }
      i = j;
    } else if (strchr("+-*/(),?:<>=!&|", c)) {
      if (c == '(') ++depth;
      if (c == ')' && --depth < 0) {
        *err = "unbalanced ')' in expression";
        return -1;
      }
      o += (char)c;
      ++i;
    } else {
      *err = std::string("character '") + (char)c + "' is not allowed in an expression";
      return -1;
    }
  }
  if (depth != 0) {
    *err = "unbalanced '(' in expression";
    return -1;
  }
  *out = o;
  return 0;
}

// ---- source generation --------------------------------------------------------------------------
static std::string generate(const std::vector<UserExpr>& models,
                            const std::vector<UserExpr>& priors, bool builtin_fallback,
                            int min_waves, int threads, bool with_split = false) {
  std::ostringstream s;
  bool early_reject = false;
  for (const UserExpr& m : models) early_reject = early_reject || m.early_reject;
  if (early_reject) s << "#define MHX_EARLY_REJECT 1\n";
  s << "#define MHX_USER_THREADS " << threads << "\n"
    << "#include \"mhx_kernels.hpp\"\n"
       "namespace mhx {\n"
       "__device__ __forceinline__ double mhx_ux_min(double a, double b) { return a < b ? a : b; }\n"
       "__device__ __forceinline__ double mhx_ux_max(double a, double b) { return a > b ? a : b; }\n"
       // (expt base n) with an integer n: SBCL's intexp order of multiplications
       "__device__ __forceinline__ double mhx_ux_ipow(double base, double pw) {\n"
       "  int power = (int)pw; const bool neg = power < 0; if (neg) power = -power;\n"
       "  int nextn = power >> 1; double total = (power & 1) ? base : 1.0;\n"
       "  while (nextn != 0) { base = base * base; if (nextn & 1) total = base * total; nextn >>= 1; }\n"
       "  return neg ? 1.0 / total : total;\n}\n";
  // exp / log: the engine's own < 1 ulp routines (15 and 22 VALU instructions - both read a
  // table from LDS - against ocml's 37 and 93); MHX_EXPR_OCML_MATH=1 selects ocml's.  (log x) of x <= 0 is an error in the
  // reference; here it is a NaN, which marks the chain as trapped.
  const bool ocml = getenv("MHX_EXPR_OCML_MATH") && atoi(getenv("MHX_EXPR_OCML_MATH")) != 0;
  s << "__device__ __forceinline__ double mhx_ux_exp(double a) { return "
    << (ocml ? "exp(a)" : "gexp(a)") << "; }\n"
    << "__device__ __forceinline__ double mhx_ux_log(double a) { return "
    << (ocml ? "(a > 0.0 ? log(a) : __builtin_nan(\"\"))"
             : "tlog(a)")
    << "; }\n";
  // Divisions by expressions that do not depend on x (1/w, 1/tau ...) are loop invariant; with
  // reciprocal math the compiler forms the reciprocal once per step instead of dividing per
  // data point (<= 1 ulp per quotient, inside the stated tolerance).  MHX_EXPR_EXACT_DIV=1
  // keeps IEEE divisions.
  const bool recip = !(getenv("MHX_EXPR_EXACT_DIV") && atoi(getenv("MHX_EXPR_EXACT_DIV")) != 0);
  for (size_t m = 0; m < models.size(); ++m) {
    const UserExpr& u = models[m];
    const int np = (int)u.names.size();
    if (!u.builtin.empty()) {  // an ahead-of-time model struct: only named, nothing to define
      s << "using UserModel" << m << " = " << u.builtin << ";\n";
      continue;
    }
    s << "struct UserModel" << m << " {\n"
      << "  struct Prep { double p[" << (np > 0 ? np : 1) << "]; };\n"
      << "  template <class PF>\n"
      << "  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {\n"
      << "    Prep q;\n";
    for (int j = 0; j < np; ++j) s << "    q.p[" << j << "] = uniform_f64(pf(" << j << "));\n";
    s << "    return q;\n  }\n"
      // xcol0 is x; xcol1 the dataset's second column (mhx_set_dataset_cols), which the sweep hands
      // over only to models that say they read it (kXCols)
      << "  static __device__ __forceinline__ double eval2(const Prep& q, double x, double xcol1) {\n"
      << (recip ? "#pragma clang fp reciprocal(on)\n" : "")
      << "    const double xcol0 = x; (void)xcol0; (void)xcol1;\n";
    for (int j = 0; j < np; ++j)
      s << "    const double p_" << u.names[j] << " = q.p[" << j << "]; (void)p_" << u.names[j]
        << ";\n";
    s << "    return (double)(" << u.expr << ");\n  }\n"
      << "  static __device__ __forceinline__ double eval(const Prep& q, double x) {"
         " return eval2(q, x, 0.0); }\n";
    if (u.xcols > 1) s << "  static constexpr int kXCols = " << u.xcols << ";\n";
    if (!u.lik_expr.empty())
      s << "  static __device__ __forceinline__ double lik_term(double y, double model, double error) {\n"
        << "    (void)y; (void)model; (void)error;\n"
        << "    return (double)(" << u.lik_expr << ");\n  }\n";
    s << "};\n";
  }
  static const char* kLikName[] = {"MHX_LIK_NORMAL", "MHX_LIK_NORMAL_CUTOFF", "MHX_LIK_POISSON",
                                   "MHX_LIK_EXPR"};
  // dealing the workgroup's proposals by cost (group_logpost): for the functions whose model is
  // an ahead-of-time struct with per-window peak masks
  std::ostringstream deal_any, deal_cases;
  deal_any << "false";
  for (size_t m = 0; m < models.size(); ++m) {
    const int lik = models[m].lik;
    if (!models[m].builtin.empty() && lik >= 0 && lik <= 2) {
      const char* wg = models[m].wgrid ? ", true" : "";
      deal_any << " || FixedSpec<UserModel" << m << ", " << kLikName[lik] << wg << ">::kDeal";
      deal_cases << "      case " << m << ": return FixedSpec<UserModel" << m << ", " << kLikName[lik]
                 << wg << ">::cost(f, pf, scratch);\n";
    }
  }
  s << "struct UserSpec {\n"
       "  static constexpr bool kDeal = " << deal_any.str() << ";\n"
       "  template <class PF>\n"
       "  static __device__ __forceinline__ int cost(const FnDesc& f, PF pf, double* scratch) {\n"
       "    switch (f.user_slot) {\n" << deal_cases.str() <<
       "      default: (void)pf; (void)scratch; return 0;\n    }\n  }\n"
       "  template <class PF>\n"
       "  static __device__ __forceinline__ double loglik(const FnDesc& f, PF pf, bool active,\n"
       "                                                  GroupLds& lds, double* scratch) {\n"
       "    switch (f.user_slot) {\n";
  // a slot belongs to one function, whose likelihood is known now: only that sweep is compiled
  for (size_t m = 0; m < models.size(); ++m) {
    const int lik = models[m].lik;
    if (!models[m].builtin.empty() && lik >= 0 && lik <= 2)
      // the whole FixedSpec: fast-path vote, tile-level peak skipping, parameters in SGPRs
      s << "      case " << m << ": return FixedSpec<UserModel" << m << ", " << kLikName[lik]
        << (models[m].wgrid ? ", true" : "") << ">::loglik(f, pf, active, lds, scratch);\n";
    else if (lik >= 0 && lik <= 3)
      s << "      case " << m << ": return GenericSpec::one_lik<UserModel" << m << ", "
        << kLikName[lik] << ">(f, pf, active, lds);\n";
    else
      s << "      case " << m << ": return GenericSpec::by_lik<UserModel" << m
        << ">(f, pf, active, lds);\n";
  }
  s << (builtin_fallback
            ? "      default: return GenericSpec::loglik(f, pf, active, lds, scratch);\n"
            : "      default: (void)scratch; return 0.0;  // every function has a slot\n")
    << "    }\n  }\n"
       "  static constexpr bool kSplit = true;\n"
       "  template <class PF>\n"
       "  static __device__ __forceinline__ double loglik_part(const FnDesc& f, PF pf, int64_t p0,\n"
       "                                                       int64_t p1, double* scratch) {\n"
       "    switch (f.user_slot) {\n";
  for (size_t m = 0; m < models.size(); ++m) {
    const int lik = models[m].lik;
    if (with_split && lik >= 0 && lik <= 3)
      s << "      case " << m << ": return FixedSpec<UserModel" << m << ", " << kLikName[lik]
        << ">::loglik_part(f, pf, p0, p1, scratch);\n";
  }
  s << "      default: (void)scratch; return 0.0;\n    }\n  }\n"
       "  static __device__ __forceinline__ double logprior(const FnDesc& f, const double* th,\n"
       "                                                    double bounds_total) {\n"
       "    switch (f.prior_slot) {\n";
  for (size_t m = 0; m < priors.size(); ++m) {
    const UserExpr& u = priors[m];
    s << "      case " << m << ": {\n";
    for (size_t j = 0; j < u.names.size(); ++j) {
      if (u.index[j] >= 0)
        s << "        const double p_" << u.names[j] << " = th[" << u.index[j] << "];";
      else  // <key>-bound: the penalty of that key in this function's bounds block
        s << "        const double p_" << u.names[j] << " = bound_of(f, th, "
          << (-u.index[j] - 1) << ");";
      s << " (void)p_" << u.names[j] << ";\n";
    }
    s << "        return (double)(" << u.expr << ");\n      }\n";
  }
  s << "      default: return bounds_total;\n    }\n  }\n};\n}  // namespace mhx\n"
       "using namespace mhx;\n"
       "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_logpost(\n"
       "    const ProblemDesc* P, const double* theta, int64_t n, double* out, double* parts) {\n"
       "  k_logpost_body<UserSpec>(P, theta, n, out, parts);\n}\n"
       "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_init(const ProblemDesc* P,\n"
       "                                                              ChainState S) {\n"
       "  k_init_body<UserSpec>(P, S);\n}\n"
       "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_step(\n"
       "    const ProblemDesc* P, ChainState S, const double* L, int per_chain_l, const double* z,\n"
       "    const double* u, const double* T, unsigned char* accepted) {\n"
       "  k_step_injected_body<UserSpec>(P, S, L, per_chain_l, z, u, T, accepted);\n}\n"
       "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS, "
    << min_waves
    << ") void mhx_user_adaptive(\n"
       "    const ProblemDesc* P, ChainState S, RunDesc R, int64_t max_iters, int plain) {\n"
       "  k_adaptive_body<UserSpec>(P, S, R, max_iters, plain);\n}\n";
  if (with_split)
    s << "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_split_sweep(\n"
         "    const ProblemDesc* P, ChainState S) {\n"
         "  k_split_sweep_body<UserSpec>(P, S);\n}\n"
         "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_split_tsweep(\n"
         "    const ProblemDesc* P, const FnDesc* slices, ChainState S, int n_slices) {\n"
         "  k_split_tsweep_body<UserSpec>(P, slices, S, n_slices);\n}\n"
         "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_split_step(\n"
         "    const ProblemDesc* P, ChainState S, RunDesc R, int mode, int plain) {\n"
         "  k_adaptive_body<UserSpec, true>(P, S, R, 1, plain, mode);\n}\n"
         "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS) void mhx_user_persist(\n"
         "    const ProblemDesc* P, ChainState S, RunDesc R, int64_t max_iters, int plain) {\n"
         "  k_persist_body<UserSpec, false>(P, nullptr, S, R, 0, max_iters, plain);\n}\n"
         "extern \"C\" __global__ __launch_bounds__(MHX_USER_THREADS, 4) void mhx_user_persist_ts(\n"
         "    const ProblemDesc* P, const FnDesc* slices, ChainState S, RunDesc R, int n_slices,\n"
         "    int64_t max_iters, int plain) {\n"
         "  k_persist_body<UserSpec, true>(P, slices, S, R, n_slices, max_iters, plain);\n}\n";
  return s.str();
}

// ---- on-disk cache of compiled code objects ---------------------------------------------------
// A compile costs 1-2 s; the same problem in the next process should not pay it again.  Files live
// in $MHX_RTC_CACHE_DIR, else $XDG_CACHE_HOME/mhx, else ~/.cache/mhx (MHX_RTC_CACHE_DIR=off
// disables); a file is trusted only if the full key stored inside it - hiprtc version, build
// options, embedded headers, generated source - equals the one being asked for.
static uint64_t fnv1a(const std::string& s, uint64_t h) {
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ULL;
  }
  return h;
}
static std::string cache_dir() {
  const char* d = getenv("MHX_RTC_CACHE_DIR");
  if (d && (!strcmp(d, "off") || !strcmp(d, "0") || !*d)) return "";
  std::string dir;
  if (d) {
    dir = d;
  } else if (const char* x = getenv("XDG_CACHE_HOME")) {
    dir = std::string(x) + "/mhx";
  } else if (const char* h = getenv("HOME")) {
    dir = std::string(h) + "/.cache";
    (void)mkdir(dir.c_str(), 0700);
    dir += "/mhx";
  } else {
    return "";
  }
  (void)mkdir(dir.c_str(), 0700);
  return dir;
}
static std::string cache_path(const std::string& key) {
  const std::string dir = cache_dir();
  if (dir.empty()) return "";
  char name[64];
  snprintf(name, sizeof name, "/%016llx%016llx.co",
           (unsigned long long)fnv1a(key, 14695981039346656037ULL),
           (unsigned long long)fnv1a(key, 0x9E3779B97F4A7C15ULL));
  return dir + name;
}
static bool cache_load(const std::string& key, std::vector<char>* code) {
  const std::string path = cache_path(key);
  if (path.empty()) return false;
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) return false;
  bool ok = false;
  char magic[6] = {0};
  uint64_t kl = 0, cl = 0;
  if (fread(magic, 1, 6, fp) == 6 && !memcmp(magic, "MHXC1\n", 6) && fread(&kl, 8, 1, fp) == 1 &&
      kl == key.size()) {
    std::string k(kl, '\0');
    if (fread(&k[0], 1, kl, fp) == kl && k == key && fread(&cl, 8, 1, fp) == 1 && cl > 0 &&
        cl < (1ull << 30)) {
      code->resize(cl);
      ok = fread(code->data(), 1, cl, fp) == cl;
    }
  }
  fclose(fp);
  return ok;
}
static void cache_store(const std::string& key, const std::vector<char>& code) {
  const std::string path = cache_path(key);
  if (path.empty()) return;
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* fp = fopen(tmp.c_str(), "wb");
  if (!fp) return;
  const uint64_t kl = key.size(), cl = code.size();
  const bool ok = fwrite("MHXC1\n", 1, 6, fp) == 6 && fwrite(&kl, 8, 1, fp) == 1 &&
                  fwrite(key.data(), 1, kl, fp) == kl && fwrite(&cl, 8, 1, fp) == 1 &&
                  fwrite(code.data(), 1, cl, fp) == cl;
  fclose(fp);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

static int build_once(const std::vector<UserExpr>& models, const std::vector<UserExpr>& priors,
                      bool builtin_fallback, bool with_split, const Family& fam, int min_waves,
                      UserProgram* prog, std::string* err) {
  Hiprtc& r = rtc();
  if (!r.ok) {
    *err = "libhiprtc.so could not be loaded: expression models need ROCm's hiprtc";
    return -1;
  }
  prog->source = generate(models, priors, builtin_fallback, min_waves, fam.threads, with_split);
  if (const char* dump = getenv("MHX_RTC_DUMP")) {  // the generated translation unit, for study
    if (FILE* fp = fopen(dump, "w")) {
      fputs(prog->source.c_str(), fp);
      fclose(fp);
    }
  }
  const char* hdr_src[] = {kSrc_mhx_kernels_hpp, kSrc_mhx_device_hpp, kSrc_mhx_types_hpp,
                           kSrc_mhx_h, kSrc_mhx_exp2_table_inc, kSrc_mhx_log_table_inc};
  const char* hdr_name[] = {"mhx_kernels.hpp", "mhx_device.hpp", "mhx_types.hpp",
                            "../../include/mhx.h", "mhx_exp2_table.inc", "mhx_log_table.inc"};
  // the same family defines the ahead-of-time build of this workgroup shape gets (Makefile)
  const std::string wpg = "-DMHX_WPG=" + std::to_string(fam.waves_per_group);
  const std::string famns = "-DMHX_FAMILY=w" + std::to_string(fam.waves_per_group);
  // MHX_RTC_FLAGS: extra compiler options, blank separated (tuning experiments: -DMHX_PPI_MASKED=2 ...)
  std::vector<std::string> extra;
  if (const char* xf = getenv("MHX_RTC_FLAGS")) {
    std::istringstream is(xf);
    std::string w;
    while (is >> w) extra.push_back(w);
  }
  // -DMHX_PPI_MASKED=2: the run-time-masked tile loops (more than two peaks) with 2 points per
  // iteration instead of the ahead-of-time build's 4.  Half the loop body: kernels loaded with
  // hipModuleLoadData lose far more than ahead-of-time ones once their hot loop outgrows the
  // instruction cache (bench.py --workload g23: 4.5e6 chain-steps/s with 4 points, 1.28e7 with 2;
  // the same kernel built ahead of time: 1.48e7 / 1.41e7).
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   wpg.c_str(), famns.c_str(), "-DMHX_PPI_MASKED=2"};
  for (const std::string& x : extra) opts.push_back(x.c_str());
  int vmaj = 0, vmin = 0;
  if (r.Version) (void)r.Version(&vmaj, &vmin);
  std::string key = "mhx-rtc-1|hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin) + "|";
  for (const char* o : opts) key += std::string(o) + " ";
  for (const char* h : hdr_src)
    key += "|" + std::to_string((unsigned long long)fnv1a(h, 14695981039346656037ULL));
  key += "|" + prog->source;
  std::vector<char> code;
  bool from_cache = cache_load(key, &code);
  if (from_cache) {  // a file that does not load (truncated, other driver) is dropped and rebuilt
    if (prog->module) {
      (void)hipModuleUnload(prog->module);
      prog->module = nullptr;
    }
    if (hipModuleLoadData(&prog->module, code.data()) != hipSuccess) {
      prog->module = nullptr;
      (void)remove(cache_path(key).c_str());
      from_cache = false;
    }
  }
  if (!from_cache) {
    hiprtcProgram p = nullptr;
    int rc = r.CreateProgram(&p, prog->source.c_str(), "mhx_user.hip", 6, hdr_src, hdr_name);
    if (rc != 0) {
      *err = std::string("hiprtcCreateProgram: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?");
      return -1;
    }
    rc = r.CompileProgram(p, (int)opts.size(), opts.data());
    size_t ls = 0;
    if (r.GetProgramLogSize(p, &ls) == 0 && ls > 1) {
      prog->log.resize(ls);
      r.GetProgramLog(p, &prog->log[0]);
    }
    if (rc != 0) {
      *err = "hiprtc compilation of the expression failed:\n" + prog->log.substr(0, 3000);
      r.DestroyProgram(&p);
      return -1;
    }
    size_t cs = 0;
    r.GetCodeSize(p, &cs);
    code.resize(cs);
    r.GetCode(p, code.data());
    r.DestroyProgram(&p);
    if (prog->module) {
      (void)hipModuleUnload(prog->module);
      prog->module = nullptr;
    }
    const hipError_t le = hipModuleLoadData(&prog->module, code.data());
    if (le != hipSuccess) {
      *err = std::string("hipModuleLoadData: ") + hipGetErrorString(le);
      return -1;
    }
    cache_store(key, code);
  }
  hipError_t he = hipSuccess;
  struct { hipFunction_t* f; const char* n; } fs[] = {{&prog->f_logpost, "mhx_user_logpost"},
                                                      {&prog->f_init, "mhx_user_init"},
                                                      {&prog->f_step, "mhx_user_step"},
                                                      {&prog->f_adaptive, "mhx_user_adaptive"}};
  for (auto& x : fs) {
    he = hipModuleGetFunction(x.f, prog->module, x.n);
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(*x.f),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)fam.lds_bytes);
    if (he != hipSuccess) {
      *err = std::string("module function ") + x.n + ": " + hipGetErrorString(he);
      return -1;
    }
  }
  // (the device math reads its tables at absolute LDS addresses: no static LDS in these kernels)
  auto static_lds = [](hipFunction_t f) {
    int v = 0;
    return hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f) == hipSuccess ? v : -1;
  };
  for (auto& x : fs)
    if (static_lds(*x.f) != 0) {
      *err = std::string("module function ") + x.n + " has static LDS";
      return -1;
    }
  prog->has_split = false;
  if (with_split) {
    he = hipModuleGetFunction(&prog->f_split_sweep, prog->module, "mhx_user_split_sweep");
    if (he == hipSuccess)
      he = hipModuleGetFunction(&prog->f_split_step, prog->module, "mhx_user_split_step");
    if (he == hipSuccess)
      he = hipModuleGetFunction(&prog->f_split_tsweep, prog->module, "mhx_user_split_tsweep");
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(prog->f_split_tsweep),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam.lds_bytes);
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(prog->f_split_step),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam.lds_bytes);
    if (he == hipSuccess)
      he = hipModuleGetFunction(&prog->f_persist, prog->module, "mhx_user_persist");
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(prog->f_persist),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam.lds_bytes);
    if (he == hipSuccess)
      he = hipModuleGetFunction(&prog->f_persist_ts, prog->module, "mhx_user_persist_ts");
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(prog->f_persist_ts),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam.lds_bytes);
    if (he != hipSuccess) {
      *err = std::string("split-mode module functions: ") + hipGetErrorString(he);
      return -1;
    }
    if (static_lds(prog->f_split_sweep) != 0 || static_lds(prog->f_split_step) != 0 ||
        static_lds(prog->f_split_tsweep) != 0 || static_lds(prog->f_persist) != 0 ||
        static_lds(prog->f_persist_ts) != 0) {
      *err = "split-mode module functions have static LDS";
      return -1;
    }
    prog->has_split = true;
  }
  prog->fam = &fam;
  return 0;
}

// The stepping kernel is first built for 4 waves per SIMD (<= 128 VGPRs: two workgroups per
// CU, which is what hides the latency of the dependent fp64 chains).  If the expression is too
// big for that register budget - the kernel spills into scratch beyond the few bytes the
// controller's own code uses - it is rebuilt for 2 waves per SIMD (256 VGPRs).
// MHX_RTC_MIN_WAVES=2|4 pins the choice.
int rtc_build(const std::vector<UserExpr>& models, const std::vector<UserExpr>& priors,
              bool builtin_fallback, bool with_split, const Family& fam, UserProgram* prog,
              std::string* err) {
  int pinned = 0;
  if (const char* s = getenv("MHX_RTC_MIN_WAVES")) pinned = atoi(s);
  if (pinned == 2 || pinned == 4)
    return build_once(models, priors, builtin_fallback, with_split, fam, pinned, prog, err);
  int rc = build_once(models, priors, builtin_fallback, with_split, fam, 4, prog, err);
  if (rc != 0) return rc;
  // Only the 8-wave family can trade occupancy for registers: a 1024-thread workgroup puts 4
  // waves on every SIMD whatever the bound says, and asking the compiler for "2" there only
  // changes its scheduling for the worse (measured: bench.py --workload g23, 1.5e7 -> 4.5e6
  // chain-steps/s).
  if (fam.threads > 512) return 0;
  int scratch = 0;
  if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, prog->f_adaptive) !=
          hipSuccess ||
      scratch <= 128)
    return 0;
  return build_once(models, priors, builtin_fallback, with_split, fam, 2, prog, err);
}

std::shared_ptr<UserProgram> rtc_get(const std::vector<UserExpr>& models,
                                     const std::vector<UserExpr>& priors, bool builtin_fallback,
                                     bool with_split, const Family& fam, std::string* err) {
  // The newest kCap modules stay loaded for the life of the process (an engine that is created
  // again for the same problem - a loop over datasets, a test suite - does not compile again).
  // The containers are leaked on purpose: unloading modules from a static destructor would race
  // the HIP runtime's own shutdown.
  constexpr size_t kCap = 48;
  static std::mutex& mu = *new std::mutex;
  static auto& cache = *new std::map<std::string, std::shared_ptr<UserProgram>>;
  static auto& order = *new std::vector<std::string>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::string key = "dev" + std::to_string(dev) + "|" +
                          generate(models, priors, builtin_fallback, 4, fam.threads, with_split);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  std::shared_ptr<UserProgram> prog(new UserProgram());
  if (rtc_build(models, priors, builtin_fallback, with_split, fam, prog.get(), err) != 0)
    return nullptr;
  if (order.size() >= kCap) {  // engines still using the oldest one keep it alive themselves
    cache.erase(order.front());
    order.erase(order.begin());
  }
  cache[key] = prog;
  order.push_back(key);
  return prog;
}

static inline unsigned grid_for(const UserProgram& p, int64_t n) {
  return (unsigned)((n + p.fam->waves_per_group - 1) / p.fam->waves_per_group);
}

hipError_t rtc_launch_logpost(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                              const double* theta, int64_t n, double* out, double* parts) {
  if (n <= 0) return hipSuccess;
  void* args[] = {(void*)&P, (void*)&theta, (void*)&n, (void*)&out, (void*)&parts};
  return hipModuleLaunchKernel(p.f_logpost, grid_for(p, n), 1, 1, (unsigned)p.fam->threads, 1, 1,
                               (unsigned)p.fam->lds_bytes, st, args, nullptr);
}
hipError_t rtc_launch_init(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                           const ChainState& S) {
  ChainState s = S;
  void* args[] = {(void*)&P, (void*)&s};
  return hipModuleLaunchKernel(p.f_init, grid_for(p, S.n_chains), 1, 1, (unsigned)p.fam->threads, 1, 1,
                               (unsigned)p.fam->lds_bytes, st, args, nullptr);
}
hipError_t rtc_launch_step_injected(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                    const ChainState& S, const double* L, int per_chain_l,
                                    const double* z, const double* u, const double* T,
                                    unsigned char* accepted) {
  ChainState s = S;
  void* args[] = {(void*)&P, (void*)&s, (void*)&L, (void*)&per_chain_l, (void*)&z,
                  (void*)&u, (void*)&T, (void*)&accepted};
  return hipModuleLaunchKernel(p.f_step, grid_for(p, S.n_chains), 1, 1, (unsigned)p.fam->threads, 1, 1,
                               (unsigned)p.fam->lds_bytes, st, args, nullptr);
}
hipError_t rtc_launch_split_sweep(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                  const ChainState& S, int slices) {
  ChainState s = S;
  void* args[] = {(void*)&P, (void*)&s};
  return hipModuleLaunchKernel(p.f_split_sweep, (unsigned)slices, (unsigned)S.n_chains, 1,
                               (unsigned)p.fam->threads, 1, 1, (unsigned)p.fam->sweep_lds_bytes, st,
                               args, nullptr);
}
hipError_t rtc_launch_split_tsweep(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                   const FnDesc* slices, const ChainState& S, int n_slices) {
  ChainState s = S;
  void* args[] = {(void*)&P, (void*)&slices, (void*)&s, (void*)&n_slices};
  return hipModuleLaunchKernel(p.f_split_tsweep, (unsigned)n_slices,
                               grid_for(p, S.slot_chain ? S.n_slots : S.n_chains), 1,
                               (unsigned)p.fam->threads, 1, 1, (unsigned)p.fam->lds_bytes, st, args,
                               nullptr);
}
hipError_t rtc_launch_split_step(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                 const ChainState& S, const RunDesc& R, int mode, int plain) {
  ChainState s = S;
  RunDesc r = R;
  void* args[] = {(void*)&P, (void*)&s, (void*)&r, (void*)&mode, (void*)&plain};
  return hipModuleLaunchKernel(p.f_split_step, grid_for(p, S.slot_chain ? S.n_slots : S.n_chains), 1, 1,
                               (unsigned)p.fam->threads, 1, 1, (unsigned)p.fam->lds_bytes, st, args,
                               nullptr);
}
hipError_t rtc_launch_persist(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                              const ChainState& S, const RunDesc& R, int slices, int64_t max_iters,
                              int plain) {
  ChainState s = S;
  RunDesc r = R;
  void* args[] = {(void*)&P, (void*)&s, (void*)&r, (void*)&max_iters, (void*)&plain};
  return hipModuleLaunchKernel(p.f_persist, 1u + (unsigned)slices, (unsigned)S.n_chains, 1,
                               (unsigned)p.fam->threads, 1, 1, (unsigned)p.fam->lds_bytes, st, args,
                               nullptr);
}
int rtc_persist_per_cu(const UserProgram& p, int ts) {
  int n = 0;
  hipFunction_t f = ts ? p.f_persist_ts : p.f_persist;
  if (!f || hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, p.fam->threads, p.fam->lds_bytes) != hipSuccess)
    return 0;
  return n;
}
hipError_t rtc_launch_persist_ts(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                 const FnDesc* slices, const ChainState& S, const RunDesc& R,
                                 int n_slices, int64_t max_iters, int plain) {
  ChainState s = S;
  RunDesc r = R;
  void* args[] = {(void*)&P, (void*)&slices, (void*)&s, (void*)&r, (void*)&n_slices,
                  (void*)&max_iters, (void*)&plain};
  return hipModuleLaunchKernel(p.f_persist_ts, 1u + (unsigned)n_slices,
                               grid_for(p, S.slot_chain ? S.n_slots : S.n_chains), 1,
                               (unsigned)p.fam->threads, 1, 1, (unsigned)p.fam->lds_bytes, st, args,
                               nullptr);
}
hipError_t rtc_launch_adaptive(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                               const ChainState& S, const RunDesc& R, int64_t max_iters,
                               int plain) {
  ChainState s = S;
  RunDesc r = R;
  void* args[] = {(void*)&P, (void*)&s, (void*)&r, (void*)&max_iters, (void*)&plain};
  return hipModuleLaunchKernel(p.f_adaptive, grid_for(p, S.slot_chain ? S.n_slots : S.n_chains), 1, 1,
                               (unsigned)p.fam->threads, 1, 1,
                               (unsigned)p.fam->lds_bytes, st, args, nullptr);
}

}  // namespace mhx
