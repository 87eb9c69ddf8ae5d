// mhx_rtc.hpp -- user-expression models and prior bodies: the reference's
// :function is an arbitrary closure (M:1134-1137) and its priors arbitrary prior-bounds-let
// bodies (M:346-369, nv-specific.lisp:25-34).  A host shim hands their bodies over as
// C-syntax expressions; they are spliced into a model / prior struct next to the SAME kernel
// bodies (mhx_kernels.hpp, embedded in the library) and compiled for gfx950 with hiprtc.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "mhx_launch.hpp"
#include "mhx_types.hpp"

namespace mhx {

constexpr int SPEC_USER = 100;

struct UserExpr {
  std::string expr;                // validated, identifiers already rewritten
  std::vector<std::string> names;  // as given by the caller
  std::vector<int> index;
  std::string builtin;   // models: non-empty = the C++ type of an ahead-of-time model struct
                         // ("PeaksModel<1, 3, false>"): no expression, the function is only
                         // given its own compile-time specialisation
  int xcols = 1;         // expression models: 2 when the text names xcol1 (a second column of x)
  bool wgrid = false;    // builtin peaks models: the function has a per-window grid table
                         // (FnDesc::tgh) - its kernel is FixedSpec<Model, LIK, true>
  bool early_reject = false;  // builtin models: compile the program with sweep()'s exact early
                              // rejection (-DMHX_EARLY_REJECT; mhx_kernels.hpp)
  int lik = -1;          // models: the function's likelihood kind (-1: dispatch at run time)
  std::string lik_expr;  // models with MHX_LIK_EXPR: the per-point term over y, model, error
};

struct UserProgram {
  hipModule_t module = nullptr;
  hipFunction_t f_logpost = nullptr, f_init = nullptr, f_step = nullptr, f_adaptive = nullptr;
  hipFunction_t f_split_sweep = nullptr, f_split_step = nullptr;  // only with has_split
  hipFunction_t f_split_tsweep = nullptr;                         // (the tile-sliced sweep)
  hipFunction_t f_persist = nullptr, f_persist_ts = nullptr;      // (the persistent split kernels)
  bool has_split = false;
  const Family* fam = nullptr;  // the kernel family (workgroup shape) the module was built for
  std::string source, log;
  ~UserProgram();
};

// Checks `expr` against the expression grammar of include/mhx.h and rewrites it for splicing:
// parameter identifiers -> p_<name>, integer literals -> doubles, abs/min/max -> device forms.
// extra: additional identifiers allowed as they are, separated by blanks ("x", "bounds_total",
// "y model error").
int rtc_prepare_expr(const std::string& expr, const std::vector<std::string>& names,
                     const char* extra, std::string* out, std::string* err);

// What an expression IS (mhx_expr.cpp): a polynomial background plus Gaussian or Lorentzian peaks
// over distinct parameters is served by the enumerated model's kernels.  `expr` is the text as
// the caller gave it, `names` its parameter names.  order[j] = which of `names` is local
// parameter j of the enumerated model (bg_0 .., then A, mu, w per peak).
struct RecognisedModel {
  int model = -1;  // MHX_MODEL_POLY / _GAUSS_PEAKS / _LORENTZ_PEAKS, -1: not of the shape
  int shape[2] = {0, 0};
  std::vector<int> order;
};
bool rtc_recognise(const std::string& expr, const std::vector<std::string>& names,
                   RecognisedModel* out);

// models[slot] / priors[slot] -> compiled module.  builtin_fallback: the problem also has
// functions with ahead-of-time models, so the generic dispatcher must be part of the kernels.
// Returns 0 or fills *err.
int rtc_build(const std::vector<UserExpr>& models, const std::vector<UserExpr>& priors,
              bool builtin_fallback, bool with_split, const Family& fam, UserProgram* prog,
              std::string* err);
// The same through a process-wide cache (keyed by device and generated source): engines that
// describe the same problem share one compiled module.  Returns nullptr and fills *err on error.
std::shared_ptr<UserProgram> rtc_get(const std::vector<UserExpr>& models,
                                     const std::vector<UserExpr>& priors, bool builtin_fallback,
                                     bool with_split, const Family& fam, std::string* err);

hipError_t rtc_launch_logpost(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                              const double* theta, int64_t n, double* out, double* parts);
hipError_t rtc_launch_init(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                           const ChainState& S);
hipError_t rtc_launch_step_injected(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                    const ChainState& S, const double* L, int per_chain_l,
                                    const double* z, const double* u, const double* T,
                                    unsigned char* accepted);
hipError_t rtc_launch_split_sweep(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                  const ChainState& S, int slices);
hipError_t rtc_launch_split_tsweep(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                   const FnDesc* slices, const ChainState& S, int n_slices);
hipError_t rtc_launch_split_step(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                 const ChainState& S, const RunDesc& R, int mode, int plain);
hipError_t rtc_launch_adaptive(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                               const ChainState& S, const RunDesc& R, int64_t max_iters,
                               int plain);
hipError_t rtc_launch_persist(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                              const ChainState& S, const RunDesc& R, int slices, int64_t max_iters,
                              int plain);
int rtc_persist_per_cu(const UserProgram& p, int ts);
hipError_t rtc_launch_persist_ts(const UserProgram& p, hipStream_t st, const ProblemDesc* P,
                                 const FnDesc* slices, const ChainState& S, const RunDesc& R,
                                 int n_slices, int64_t max_iters, int plain);

}  // namespace mhx
