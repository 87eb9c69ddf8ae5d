// mhx_launch.hpp -- host-callable launchers of the gfx950 kernels (defined in mhx_kernels.hip,
// once per kernel family: see mhx_types.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "mhx_types.hpp"

namespace mhx {

// Compiled problem specialisations.  SPEC_GENERIC dispatches on (model, shape, likelihood) at
// run time; the others fix one model + likelihood for all K functions so that their
// parameters live in scalar registers and the kernel's register budget is its own.
enum SpecId {
  SPEC_GENERIC = 0,
  SPEC_GAUSS22_NORMAL = 1,   // 2 background terms + 2 Gaussian peaks, weighted normal (8 params)
  SPEC_GAUSS15_POISSON = 2,  // constant background + 5 Gaussian peaks, Poisson (16 params)
  SPEC_PVOIGT2_NORMAL = 3,   // 11-parameter two-peak pseudo-Voigt (global fits)
  SPEC_POLY2_NORMAL = 4,     // b + m x
  SPEC_POLY8_NORMAL = 5,     // degree-7 polynomial
  SPEC_LORDER_NORMAL = 6,    // test.lisp's 6-parameter lineshape
  SPEC_GAUSS22_CUTOFF = 7,
#ifdef MHX_AOT_G23
  SPEC__COUNT = 9
#else
  SPEC__COUNT = 8
#endif
};

int select_spec(const ProblemDesc& P);
const char* spec_name(int spec);

// One family of kernels: everything that depends on the workgroup shape goes through this table.
struct Family {
  int waves_per_group;  // chains per workgroup
  int threads;          // 64 * waves_per_group
  int tile_points;      // data points per LDS tile and array
  size_t lds_bytes;     // dynamic LDS of the stepping kernels
  size_t sweep_lds_bytes;  // dynamic LDS of the split-mode sweep kernel (the math tables)
  hipError_t (*configure)();
  hipError_t (*logpost)(int spec, hipStream_t st, const ProblemDesc* P, const double* theta,
                        int64_t n, double* out, double* parts);
  hipError_t (*init)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S);
  hipError_t (*step_injected)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S,
                              const double* L, int per_chain_l, const double* z, const double* u,
                              const double* T, unsigned char* accepted);
  hipError_t (*adaptive)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S,
                         const RunDesc& R, int64_t max_iters, int plain);
  hipError_t (*initial_l)(hipStream_t st, const ChainState& S, const RunDesc& R, int have_l,
                          double T0);
  hipError_t (*l_matrix)(hipStream_t st, const ChainState& S, int64_t chain, int take, int* fwd,
                         double* cov, double* out, int* info);
  hipError_t (*acceptance)(hipStream_t st, const ChainState& S, int take, double* out);
  hipError_t (*pool_stats)(hipStream_t st, const ChainState& S, const RunDesc& R);
  hipError_t (*pool_reduce)(hipStream_t st, const ChainState& S);
  hipError_t (*pool_factor)(hipStream_t st, const ChainState& S);
  hipError_t (*modify)(hipStream_t st, const ChainState& S, int action, int64_t n);
  // split mode (mhx_kernels.hpp): the partial sums of every pending proposal over `slices`
  // workgroups per chain, and the two halves of the loop iteration around them
  bool (*split_capable)(int spec);
  hipError_t (*split_sweep)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S,
                            int slices);
  // ... or, tile-sliced (k_split_tsweep): groups of chains on slices of whole windows, `slices` =
  // the device copy of the slice table [K][n_slices]
  hipError_t (*split_tsweep)(int spec, hipStream_t st, const ProblemDesc* P, const FnDesc* slices,
                             const ChainState& S, int n_slices);
  hipError_t (*split_step)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S,
                           const RunDesc& R, int mode, int plain);
  // ... or ONE launch for many iterations of a handful of chains (k_persist): grid (1 + slices,
  // chains), every workgroup resident at once (the caller checks)
  hipError_t (*persist)(int spec, hipStream_t st, const ProblemDesc* P, const ChainState& S,
                        const RunDesc& R, int slices, int64_t max_iters, int plain);
  // workgroups of the persistent kernel (ts: the tile-sliced one) a CU holds at once, as the
  // runtime's occupancy calculator sees the compiled kernel (0: not compiled for this spec)
  int (*persist_per_cu)(int spec, int ts);
  // ... the same for the tile-sliced mode (k_persist_ts): grid (1 + n_slices, chain groups)
  hipError_t (*persist_ts)(int spec, hipStream_t st, const ProblemDesc* P, const FnDesc* slices,
                           const ChainState& S, const RunDesc& R, int n_slices, int64_t max_iters,
                           int plain);
};

const Family& family_w8();
const Family& family_w16();

}  // namespace mhx
