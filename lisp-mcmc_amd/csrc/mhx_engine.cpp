// mhx_engine.cpp -- host side of libmhx: the C ABI of include/mhx.h over the gfx950 kernels.
//
// No CPU fallback exists: every numeric entry point launches HIP kernels on the configured
// device and fails with MHX_EDEVICE when that is impossible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mhx.h"
#include <dlfcn.h>

#include <memory>

#include "mhx_launch.hpp"
#include "mhx_rtc.hpp"
#include "mhx_types.hpp"

using namespace mhx;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return fail(MHX_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                  __FILE__, __LINE__);                                                  \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t alloc(size_t count, bool zero = true) {
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      return e;
    }
    n = count;
    if (zero) e = hipMemset(p, 0, count * sizeof(T));
    return e;
  }
};

struct Dataset {
  DevBuf<double> x, y, w, c, txlo, txhi, tgh;
  std::vector<double> hx;  // host copy of the padded x: tile ranges are formed per kernel family
  size_t n = 0;            // data points (hx holds the pads too)
  bool no_rec = false;     // MHX_NO_RECURRENCE=1 when the dataset was set
  bool set = false;
};

int64_t steps_to_settle_of(int d) { return 10 * (int64_t)std::max(50, d); }  // M:873

// ---- RCCL, loaded lazily (libmhx.so has no link-time dependency on it: a single-GPU host needs
// no collective library).  Only what the pooled-covariance tick uses; the types are restated
// from rccl.h (NCCL ABI: opaque communicator, 128-byte id, ncclDouble = 8, ncclSum = 0).
typedef struct mhxNcclComm* nccl_comm_t;
struct nccl_unique_id { char internal[128]; };
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_unique_id*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
// The library is opened RTLD_LOCAL, and that is load-bearing.  What is known (DESIGN.md
// section 6): four full-suite test processes of round 2 that had used RCCL and hiprtc ended in
// `double free or corruption (!prev)` / `free(): invalid pointer` AFTER their work was done,
// while this dlopen said RTLD_GLOBAL (later RTLD_GLOBAL | RTLD_NODELETE) - with the ROCm
// installation's librccl as well as with the one PyTorch bundles; none of the dozens of runs
// since it says RTLD_LOCAL has.  Both builds of librccl export, next to the nccl* entry points,
// several hundred C++ symbols of default visibility - weak std:: template instantiations and
// fmt::v7's STB_GNU_UNIQUE data objects (`nm -D`) - which RTLD_GLOBAL adds to the scope every
// LATER-loaded library's own weak symbols are resolved in; two libraries tearing down what each
// takes for its own object at exit is the mechanism - reproduced in round 4
// (tools/debug/rccl_exit_abort.py): old flags, RCCL through libmhx, a hiprtc model, THEN a torch
// operation, everything left to the interpreter's exit -> `double free or corruption (!prev)`,
// twice in two runs; RTLD_LOCAL, same process: exit status 0; LD_DEBUG=bindings shows 67 weak
// C++ symbols of libraries that do not depend on RCCL bound into it under the old flags, none
// under the new.  So the flag is kept local, nothing of RCCL's is
// ever visible to anybody but this file's dlsym calls, and one test keeps a process with RCCL,
// hiprtc and torch at exit status 0 (tests/test_gpu_rccl_stub.py).
// MHX_RCCL_DLOPEN_GLOBAL=1 restores the old flags - for that debug script only, and only in a
// library built with -DMHX_DEBUG_HOOKS (tests/hooks/libmhx_hooks.so; the release library has
// neither of the two test switches).
std::string g_rccl_why;  // why rccl().ok is false
Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  // MHX_RCCL_LIBRARY: the copy that belongs to the HIP runtime this process uses (the Python
  // binding points it at PyTorch's bundled librccl.so when it shares PyTorch's libamdhip64; the
  // tests point it at their stub); then the soname: a host that already has RCCL loaded shares
  // that instance
  const char* names[] = {getenv("MHX_RCCL_LIBRARY"), "librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so"};
#ifdef MHX_DEBUG_HOOKS
  const char* dg = getenv("MHX_RCCL_DLOPEN_GLOBAL");
  const int flags = (dg && atoi(dg) != 0) ? (RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE)
                                           : (RTLD_NOW | RTLD_LOCAL);
#else
  const int flags = RTLD_NOW | RTLD_LOCAL;
#endif
  const char* opened = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    r.h = dlopen(n, flags);
    if (r.h) {
      opened = n;
      break;
    }
    g_rccl_why = dlerror();
  }
  if (!r.h) {
    if (g_rccl_why.empty()) g_rccl_why = "librccl.so not found";
    return r;
  }
  // One HIP runtime per process: the collective is enqueued on OUR stream, so the library must be
  // bound to the very libamdhip64 libmhx uses.  (A librccl that needs another soname of the
  // runtime - a different ROCm major - would bring a second runtime into the process: its
  // hipStream_t handles are not ours.)  Compared by the address both sides resolve a HIP entry
  // point to.
  {
    void* theirs = dlsym(r.h, "hipStreamSynchronize");
    void* ours = reinterpret_cast<void*>(&hipStreamSynchronize);
    if (theirs && theirs != ours) {
      Dl_info a{}, b{};
      (void)dladdr(theirs, &a);
      (void)dladdr(ours, &b);
      char buf[768];
      snprintf(buf, sizeof buf, "%s is bound to HIP runtime %s, libmhx to %s: refused (one HIP "
               "runtime per process; set MHX_RCCL_LIBRARY to the librccl of the runtime in use)",
               opened, a.dli_fname ? a.dli_fname : "?", b.dli_fname ? b.dli_fname : "?");
      g_rccl_why = buf;
      dlclose(r.h);
      r.h = nullptr;
      return r;
    }
  }
#define SYM(field, name) *(void**)(&r.field) = dlsym(r.h, name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce &&
         r.GroupStart && r.GroupEnd;
  if (!r.ok) g_rccl_why = std::string(opened) + " lacks one of the eight nccl* entry points libmhx binds";
  return r;
}
const char* rccl_err(int rc) {
  return rccl().GetErrorString ? rccl().GetErrorString(rc) : "error";
}
constexpr int kNcclDouble = 8, kNcclSum = 0;

int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

struct mhx_engine {
  mhx_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stop_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  ProblemDesc P{};  // host copy; device pointers inside
  bool fn_set[MHX_MAX_FUNCTIONS] = {};
  Dataset data[MHX_MAX_FUNCTIONS];
  DevBuf<ProblemDesc> dP;
  bool problem_dirty = true;
  UserExpr fn_expr[MHX_MAX_FUNCTIONS];     // MHX_MODEL_EXPR bodies
  RecognisedModel fn_recog[MHX_MAX_FUNCTIONS];  // ... and what they are (mhx_expr.cpp)
  bool recognise = true;                   // mhx_set_expr_recognition
  UserExpr prior_expr[MHX_MAX_FUNCTIONS];  // prior-bounds-let bodies
  std::string lik_expr[MHX_MAX_FUNCTIONS]; // create-log-liklihood-function bodies (validated)
  std::shared_ptr<UserProgram> user_prog;  // run-time compiled kernels (shared, rtc_get)
  int spec = SPEC_GENERIC;
  std::string kernel_name;  // mhx_kernel_name
  std::string rtc_note;     // why run-time specialisation was not possible (first line)
  const Family* fam = &family_w8();  // kernel family (workgroup shape) of the current problem

  ChainState S{};
  DevBuf<double> theta, prob, best_theta, best_prob, hist_prob, hist_theta, L, temperature,
      mat_tmp, pool_stats, pool_vec, L_pool;
  DevBuf<int32_t> pool_valid;
  int64_t global_iter = 0;  // loop iterations launched since mhx_adaptive_begin (pooled cadence)
  uint64_t pool_refreshes = 0;
  DevBuf<int64_t> length, age, n_hist, loop_i, reset_index;
  DevBuf<uint64_t> draw;
  DevBuf<int32_t> shutting, status, fwd_idx, stop_flag;
  DevBuf<unsigned long long> step_counter;
  // split mode (few chains, long datasets): 0 = batch kernels, else workgroups per chain
  int split_slices = 0;
  int64_t split_portion = 512;  // iterations queued between two looks at the chain states
  // one portion of split-mode launches captured as a HIP graph (replayed; the kernel arguments -
  // chain state pointers, run description - are frozen in it, so it is dropped whenever the run or
  // the problem changes): fewer microseconds per launch than issuing 2 N + 1 kernels one by one
  hipGraphExec_t split_graph = nullptr;
  int64_t split_graph_iters = 0;
  int split_graph_plain = -1;
  DevBuf<double> split_prop, split_u, split_part;
  DevBuf<int32_t> split_pending;
  // tile-sliced split mode (k_split_tsweep): split_slices = slices of whole windows per function,
  // ts_table = their FnDescs [K][split_slices] on the device
  // persistent split mode (k_persist): the per-chain split mode's slots, ONE launch for many
  // iterations - when every workgroup of the launch is resident at once (finalize_problem)
  bool persist = false;
  DevBuf<unsigned long long> persist_msg;  // [C][64]: the chains' proposal blocks (mhx_types.hpp)
  DevBuf<unsigned char> persist_part;      // [C][K][slots][16]: {partial sum, generation} pairs
  DevBuf<int32_t> persist_error;           // [1]: a master gave up waiting (ChainState::persist_error)
  bool persist_launched = false;           // the open launch is a persistent one: look at the flag
  bool persist_off = false;                // a persistent launch failed on this engine: never again
  bool tsplit = false;
  int ts_initial = 0;  // slices a run starts with (compact_tsplit may cut finer as chains finish)
  // iterations the split modes have queued since the run began.  The tile-sliced mode repacks
  // only where this count is a multiple of the portion length: when and how the sums are
  // regrouped is then a function of the walk, not of how the host chunks its advance calls
  int64_t split_iter = 0;
  bool ts_repack_due = false;
  DevBuf<FnDesc> ts_table;
  bool chains_ready = false;

  RunDesc R{};
  DevBuf<double> temps;          // a window of the temperature schedule (ensure_temps)
  int64_t temps_count = 0;       // entries held on the device, from R.temps_first
  double run_temperature = 1.0;  // :temperature of the current run
  bool run_ready = false;
  int64_t chunk_iters = 64;

  mhx_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  int allreduce_device = 0;
  // native collective of the pooled tick: an RCCL communicator of this engine's own
  // (mhx_comm_init_rank: one process per GPU) or its group's (mhx_group_create)
  nccl_comm_t comm = nullptr;
  bool comm_owned = false;
  // slot -> chain map of the stepping kernel (ChainState::slot_chain): rebuilt from the chain
  // states whenever a quarter of the slots in use have finished (compact_slots)
  DevBuf<int32_t> slot_map;
  int64_t slots_mapped = 0;  // chains dealt at the last compact_slots (0: identity map)
  bool launch_open = false;  // steps enqueued by launch_steps_enqueue(), not yet finished
  int64_t launch_iters = 0;

  // device staging of the host-facing calls (mhx_logpost, mhx_step_injected, read-backs): grown
  // on demand and kept, instead of a hipMalloc / hipFree pair per call
  DevBuf<unsigned char> stage;

  // accounting
  uint64_t launches = 0;
  double kernel_ms = 0.0;
  uint64_t timed_launches = 0;
};

namespace {

void drop_split_graph(mhx_engine* e);

int use_device(mhx_engine* e) {
  HIP_TRY(hipSetDevice(e->device));
  return MHX_OK;
}

// wait for whatever an engine has in its queue (error paths: nothing is reported from here)
void drain(mhx_engine* e) {
  if (hipSetDevice(e->device) == hipSuccess && e->stream) (void)hipStreamSynchronize(e->stream);
}

// The C++ type (csrc/mhx_device.hpp) that evaluates function f with everything about its shape
// known at compile time, or "" when the shape is too big for scalar registers.
std::string builtin_model_type(const FnDesc& f) {
  switch (f.model) {
    case MHX_MODEL_POLY:
      if (f.n_idx >= 1 && f.n_idx <= 16) return "PolyModel<" + std::to_string(f.n_idx) + ">";
      return "";
    case MHX_MODEL_GAUSS_PEAKS:
    case MHX_MODEL_LORENTZ_PEAKS: {
      const int nbg = f.shape[0], npk = f.shape[1];
      if (npk < 1 || npk > 6 || nbg < 0 || nbg > 4) return "";
      return "PeaksModel<" + std::to_string(nbg) + ", " + std::to_string(npk) + ", " +
             (f.model == MHX_MODEL_LORENTZ_PEAKS ? "true" : "false") + ">";
    }
    case MHX_MODEL_LORDER_MIXED: return "LorderModel";
    case MHX_MODEL_EXP_DECAY: return "ExpDecayModel";
    case MHX_MODEL_SINUSOID: return "SinusoidModel";
    case MHX_MODEL_PVOIGT2: return "PVoigt2Model";
    default: return "";
  }
}

// May this engine use the persistent kernels at all (MHX_NO_PERSIST=1: never; persist_off: a
// launch of theirs once lost its sweep workgroups; their handshake blocks carry 60 parameters)
static bool persist_allowed(const mhx_engine* e) {
  const char* np_ = getenv("MHX_NO_PERSIST");
  return !e->persist_off && !(np_ && atoi(np_) != 0) && e->P.d <= 60;
}
// MHX_PERSIST_TS: 1 = the tile-sliced persistent form wherever two slices of it fit, 0 = never,
// unset (-1) = where it fits with at least three quarters of the default slicing
static int persist_ts_wanted() {
  const char* s = getenv("MHX_PERSIST_TS");
  return s ? (atoi(s) != 0 ? 1 : 0) : -1;
}
int64_t persist_capacity(const mhx_engine* e, bool ts);
// nwin windows cut into ts slices are ceil(nwin / ts) windows per slice - which may fill fewer
// slices than ts: 49 windows in 16 slices are 13 slices of 4 (the last one of 1) and three empty
// ones, whose workgroups would be launched (or, persistent, sit on a CU and poll) for nothing.
// The sums are the same bits either way: an empty slice's partial sum is +0.
static int64_t trim_slices(int64_t nwin, int64_t ts) {
  if (ts <= 0 || nwin <= 0) return ts;
  const int64_t per = (nwin + ts - 1) / ts;
  return (nwin + per - 1) / per;
}

// Split mode (mhx_kernels.hpp): how many workgroups share one chain's likelihood sums, or 0 for
// the batch kernels.  Worth it when the batch launch would leave most CUs without a workgroup
// and every (slice, wave) slot still gets at least 512 points of the longest dataset.
// MHX_SPLIT=0 switches it off, MHX_SPLIT=<n> forces n slices.
// cap_pc > 0: the per-chain persistent form (k_persist: one launch per portion of iterations) is
// allowed and the GPU holds that many of its workgroups at once - where C (1 + slices) of them
// fit, split mode costs 7.6 us per iteration instead of the two launches' 14 and pays off on
// shorter datasets (4096 points: 13.4 us in the batch kernel).
int choose_split(const mhx_engine* e, const Family& fam, bool capable, int64_t cap_pc) {
  if (!capable || e->cfg.adapt_mode == MHX_ADAPT_POOLED) return 0;
  int64_t longest = 0;
  for (int k = 0; k < e->P.K; ++k) longest = std::max<int64_t>(longest, e->P.fn[k].n);
  const int64_t C = e->cfg.n_chains;
  const int W = fam.waves_per_group;
  const int64_t by_data = longest / (512 * (int64_t)W);
  if (const char* s = getenv("MHX_SPLIT")) {
    const int v = atoi(s);
    return v <= 0 ? 0 : (int)std::min<int64_t>(std::max<int64_t>(by_data, 1), v);
  }
  // measured (tools/debug/split_sweep.sh, round 2; chain-steps/s batch | best split):
  //   config 2's problem   64 chains 6.6e5 | 2.4e6 (x4)    256: 2.6e6 | 4.5e6 (x4)
  //                        512: 5.3e6 | 5.4e6 (x2)         1024: 1.05e7 | 5.9e6    2048: 2.1e7 | 6.3e6
  //   config 3's problem   16: 4.2e3 | 1.5e5 (x24)    256: 6.7e4 | 2.9e5 (x8)    1024: 2.7e5 | 3.1e5
  // (round 2, with the recurrence in the split sweep as well:  config 3  16: 2.3e5 (x24)
  //  256: 4.6e5 (x4)    1024: 2.7e5 | 5.4e5 (x4);  config 2 unchanged: its split sweep is bound by
  //  the L2, every chain reading the dataset for itself)
  // The batch kernels (peak skipping, LDS tiles shared by 8 chains) win from about 128
  // workgroups on when a point is cheap; a point that costs 40 instructions and more (a log per
  // point, a pseudo-Voigt, an expression compiled as written) keeps the whole GPU busy in split
  // mode until the batch kernels have a workgroup for every CU.  Below that about 1024
  // workgroups in the sweep launch are best.
  bool heavy = false;
  for (int k = 0; k < e->P.K; ++k) {
    const FnDesc& fd = e->P.fn[k];
    heavy = heavy || fd.lik == MHX_LIK_POISSON || fd.lik == MHX_LIK_EXPR ||
            fd.model == MHX_MODEL_PVOIGT2 || fd.model == MHX_MODEL_EXPR;
  }
  const int64_t batch_groups = (C + W - 1) / W;
  if (batch_groups >= (heavy ? 256 : 128)) return 0;
  // two launches cost about 14 us per iteration: the fused batch kernel is quicker than that up
  // to roughly a dozen 1024-point tiles; one persistent launch is quicker from one slice's worth
  // of points on (the batch kernel: 13.4 us on 4096 points)
  if (by_data < 4) {
    if (by_data >= 1 && !heavy && C * (1 + by_data) <= cap_pc) return (int)by_data;
    return 0;
  }
  // (cheap points: beyond 8 slices the partial sums and the extra blocks cost more than they
  // bring once there are 32 chains and more - 64 chains: x16 2.2e6, x8 2.4e6, x4 2.4e6)
  const int64_t want = heavy ? std::max<int64_t>(4, 2048 / C)
                             : std::max<int64_t>(2, std::min<int64_t>(1024 / C, C < 32 ? 24 : 8));
  const int64_t slices = std::min<int64_t>(std::min<int64_t>(want, 24), by_data);
  return slices >= 2 ? (int)slices : 0;
}

// Tile-sliced split mode (k_split_tsweep): into how many slices of whole windows every function is
// cut, each walked by the workgroups of ALL chain groups, or 0.  For batches too small to give
// every CU a workgroup of the batch kernels and big enough to fill workgroups of their own:
// about two workgroups per CU in the sweep launch.  MHX_TSPLIT=0 switches it off (the per-chain
// split mode or the batch kernels then), MHX_TSPLIT=<n> asks for n slices; MHX_SPLIT=0 means the
// batch kernels here too, MHX_SPLIT=<n> alone the per-chain split mode.
// cap_pc, cap_ts > 0: the persistent forms are allowed (workgroups of k_persist / k_persist_ts the
// GPU holds at once).  One persistent launch costs about 10.3 us per iteration where the two
// launches cost 20 (measured round 4, two-peak problem, us per iteration, persistent | default
// of round 3):  8192 points  32 chains 12.4 | 18.2 (batch kernel)    256: 12.4 | 18.5
//   20000 points  128: 11.1 | 22.1 (per-chain split)    256: 13.2 | 28.9
//   50000 points  8: 10.4 | 20.4 (two launches)    128: 11.9 | 23.5
// so it serves from 4 windows on - unless the per-chain persistent form fits, which is quicker
// still on short datasets (20000 points, 8 ... 64 chains: 7.7 ... 8.7 us).
int choose_tsplit(const mhx_engine* e, const Family& fam, bool capable, int64_t cap_pc, int64_t cap_ts) {
  if (!capable || e->cfg.adapt_mode == MHX_ADAPT_POOLED) return 0;
  int64_t longest = 0;
  for (int k = 0; k < e->P.K; ++k) longest = std::max<int64_t>(longest, e->P.fn[k].n);
  const int64_t nwin = (longest + kPadPoints - 1) / kPadPoints;
  const int64_t C = e->cfg.n_chains;
  const int W = fam.waves_per_group;
  const int64_t groups = (C + W - 1) / W;
  int64_t want = 0;
  // (MHX_SPLIT set: the caller has decided - the batch kernels, or the per-chain split mode)
  if (getenv("MHX_SPLIT") && !getenv("MHX_TSPLIT")) return 0;
  if (const char* s0 = getenv("MHX_SPLIT"))
    if (atoi(s0) <= 0) return 0;
  if (const char* s = getenv("MHX_TSPLIT")) {
    want = atoi(s);
    if (want <= 0) return 0;
  } else {
    // two launches and the step kernel cost about 17 us per iteration (64 chains of config 2's
    // problem: 20.7 us with one window per workgroup): the fused batch kernel, 0.6-0.8 us per
    // 1024-point tile when points are cheap, is quicker than that up to about two dozen tiles
    bool heavy = false;
    for (int k = 0; k < e->P.K; ++k) {
      const FnDesc& fd = e->P.fn[k];
      heavy = heavy || fd.lik == MHX_LIK_POISSON || fd.lik == MHX_LIK_EXPR ||
              fd.model == MHX_MODEL_PVOIGT2 || fd.model == MHX_MODEL_EXPR;
    }
    if (groups >= 256) return 0;
    // fewer walkers than a workgroup has waves: the per-chain split mode - unless the persistent
    // form is allowed, whose sweep workgroups walk LDS tiles with peak skipping and the
    // recurrence, one window per wave however long the dataset (a single walker on 1e5 points
    // 7.6 -> 6.4 us per step, on 1e6 points 25.4 -> 11.1; 20000 points: 6.2 against 6.7, left alone)
    if (C < W && (cap_ts <= 0 || nwin < 12)) return 0;
    if (nwin < (heavy ? 4 : 12)) {
      // too short for two launches per iteration: as one persistent launch, or not at all
      // (a window per slice, or fewer slices of up to 4 windows where the GPU does not hold that
      // many workgroups at once: 20000 points, 512 walkers x7 instead of the per-chain split
      // mode's two launches)
      const int64_t fit = std::min<int64_t>(nwin, cap_ts / groups - 1);
      if (nwin < (heavy ? 2 : 4) || fit < 2 || (nwin + fit - 1) / fit > 4) return 0;
      const int pc = choose_split(e, fam, capable, cap_pc);
      if (pc > 0 && C * (1 + pc) <= cap_pc) return 0;  // (the per-chain persistent form)
      return (int)trim_slices(nwin, fit);
    }
    // measured (config 2's problem, chain-steps/s; slices 4 | 8 | 16 | 32 | 49):
    //   64 chains 1.5e6 | 2.1e6 | 2.6e6 | 3.1e6 | 3.1e6     256: 6.0e6 | 7.9e6 | 8.0e6 | 7.0e6 | 6.3e6
    //   1024: 1.47e7 | 1.33e7 | 1.13e7 | 9.2e6 | 8.8e6       (per-chain split mode: 2.4e6, 4.7e6; batch
    //   kernels at 1024: 1.31e7) - about 512 workgroups in the sweep launch
    want = 512 / groups;
  }
  want = std::min<int64_t>(want, nwin);
  if (!getenv("MHX_TSPLIT")) {
    want = trim_slices(nwin, want);
    // two or three slices as TWO LAUNCHES per iteration lose to the batch kernels on datasets
    // that are not long (measured round 4, us per iteration, two launches | batch kernels:
    //   1536 walkers x2: 5e4 points 76.5 | 56.7, 1e5: 102 | 90.8, 1e6: 690 | 846;
    //   1100 walkers x3: 5e4 points 58.6 | 55.2, 1e5: 75.9 | 89.9) - unless the persistent
    // form will take them (1024 walkers x3, 5e4 points: 35.5 | 49.1)
    const int64_t pfit = cap_ts > 0 ? std::min<int64_t>(want, cap_ts / groups - 1) : 0;
    const bool persistable = pfit >= std::max<int64_t>(2, (3 * want + 3) / 4) && persist_ts_wanted() != 0;
    if (!persistable && ((want == 2 && nwin < 128) || (want == 3 && nwin < 32))) return 0;
  }
  return want >= 2 ? (int)want : 0;
}

// The slice table of the tile-sliced split mode: function k, slice s = windows [s per_k,
// (s + 1) per_k) of its dataset, as a FnDesc of its own (what k_split_tsweep hands to sweep()).
int build_ts_table(mhx_engine* e, int ts) {
  std::vector<FnDesc> tab((size_t)e->P.K * ts);
  // one function, one window per slice, one persistent launch: a sweep workgroup's window stays
  // in its LDS from its first round on (sweep(): FnDesc::solo; no DMA, no tile barriers and no
  // layout vote in the rounds after it)
  bool resident = e->persist && e->P.K == 1 && kPadPoints <= 2 * e->fam->tile_points;
  for (int k = 0; k < e->P.K; ++k) {
    const FnDesc& f = e->P.fn[k];
    const int64_t nwin = (f.n + kPadPoints - 1) / kPadPoints;
    const int64_t per = (nwin + ts - 1) / ts;
    for (int sl = 0; sl < ts; ++sl) {
      FnDesc g = f;
      const int64_t off = (int64_t)sl * per * kPadPoints;
      g.lik_const = 0.0;
      g.solo = 0;
      resident = resident && per == 1;
      if (off >= f.n) {
        g.n = 0;
        g.n_tiles = 0;
      } else {
        g.n = std::min<int64_t>(per * kPadPoints, f.n - off);
        g.n_tiles = (g.n + e->fam->tile_points - 1) / e->fam->tile_points;
        g.x = f.x + off;
        g.y = f.y + off;
        if (f.w) g.w = f.w + off;
        if (f.c) g.c = f.c + off;
        if (f.txlo) g.txlo = f.txlo + off / kPadPoints;
        if (f.txhi) g.txhi = f.txhi + off / kPadPoints;
        if (f.tgh) g.tgh = f.tgh + off / kPadPoints;
      }
      tab[(size_t)k * ts + sl] = g;
    }
  }
  const char* nr = getenv("MHX_NO_RESIDENT_SLICES");
  if (resident && !(nr && atoi(nr) != 0))
    for (FnDesc& g : tab) g.solo = g.n_tiles > 0 ? 1 : 0;
  if (e->ts_table.n < tab.size() && e->ts_table.alloc(tab.size(), false) != hipSuccess)
    return fail(MHX_ENOMEM, "hipMalloc of the slice table failed");
  HIP_TRY(hipMemcpy(e->ts_table.p, tab.data(), tab.size() * sizeof(FnDesc), hipMemcpyHostToDevice));
  return MHX_OK;
}

// Which workgroup shape serves this problem (mhx_types.hpp).  16 chains per workgroup and
// 2048-point tiles pay off when the datasets are long (>= 4 such tiles) and there are enough
// chains to give every CU its one workgroup; otherwise 8 chains per workgroup (more, smaller
// workgroups; less barrier and pad overhead on short datasets).  MHX_FAMILY_WPG=8|16 pins it.
const Family& choose_family(const mhx_engine* e) {
  if (const char* s = getenv("MHX_FAMILY_WPG")) {
    if (atoi(s) == 16) return family_w16();
    if (atoi(s) == 8) return family_w8();
  }
  int64_t longest = 0;
  for (int k = 0; k < e->P.K; ++k) longest = std::max<int64_t>(longest, e->P.fn[k].n);
  const bool big = longest >= 4 * (int64_t)family_w16().tile_points &&
                   e->cfg.n_chains >= 16 * 256;
  return big ? family_w16() : family_w8();
}

// Workgroups of a persistent launch (k_persist, k_persist_ts: they wait for one another) the GPU
// holds at once: CUs times what the occupancy calculator gives the COMPILED kernel (registers,
// LDS and waves - an assumed "two per CU" was wrong for a kernel of 132 VGPRs and cost half the
// speed, see k_persist_ts).  Every slot: what the calculator promises is what the dispatcher
// gives - 128 groups x (1 + 3) = 512 workgroups on 256 CUs ran in one shift, 55.9 us per
// iteration against 72.0 of two launches - and a workgroup that finds its slot taken for a
// while by another kernel of the process arrives late, within the masters' patience
// (MHX_PERSIST_FILL=<per cent>: measurements).
int64_t persist_capacity(const mhx_engine* e, bool ts) {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0)
    cus = 256;
  const int per_cu = e->spec == SPEC_USER ? rtc_persist_per_cu(*e->user_prog, ts ? 1 : 0)
                                          : e->fam->persist_per_cu(e->spec, ts ? 1 : 0);
  int fill = 100;
  if (const char* f = getenv("MHX_PERSIST_FILL")) fill = std::max(10, std::min(100, atoi(f)));
  return (int64_t)cus * per_cu * fill / 100;
}

int finalize_problem(mhx_engine* e) {
  if (!e->problem_dirty) return MHX_OK;
  drop_split_graph(e);
  // a new problem may run in another mode and family: slot s is chain s until the next deal
  e->S.slot_chain = nullptr;
  e->S.n_slots = e->cfg.n_chains;
  e->slots_mapped = 0;
  for (int k = 0; k < e->P.K; ++k) {
    if (!e->fn_set[k]) return fail(MHX_ESTATE, "function %d was never set (mhx_set_function)", k);
    if (!e->data[k].set) return fail(MHX_ESTATE, "dataset %d was never set (mhx_set_dataset)", k);
  }
  // the kernel family fixes the tile size and with it the tiles per dataset; the x ranges that
  // tile-level peak skipping tests against (PeaksModel::tile_mask) are those of WINDOWS of
  // kPadPoints points, pads included, whatever the family (sweep: one mask and one seeding of the
  // Gaussian recurrence per window)
  e->fam = &choose_family(e);
  const size_t tp = (size_t)e->fam->tile_points;
  const size_t wp = (size_t)kPadPoints;
  for (int k = 0; k < e->P.K; ++k) {
    FnDesc& f = e->P.fn[k];
    Dataset& D = e->data[k];
    const size_t nt = ((size_t)f.n + tp - 1) / tp;
    const size_t ntp = std::max<size_t>(((size_t)f.n + wp - 1) / wp, 1);
    std::vector<double> tlo(ntp), thi(ntp);
    for (size_t t = 0; t < ntp; ++t) {
      double lo = INFINITY, hi = -INFINITY;
      bool ok = true;
      for (size_t i = t * wp; i < (t + 1) * wp; ++i) {
        ok = ok && std::isfinite(D.hx[i]);
        lo = std::min(lo, D.hx[i]);
        hi = std::max(hi, D.hx[i]);
      }
      tlo[t] = ok ? lo : -INFINITY;
      thi[t] = ok ? hi : INFINITY;
    }
    if (D.txlo.alloc(ntp, false) != hipSuccess || D.txhi.alloc(ntp, false) != hipSuccess)
      return fail(MHX_ENOMEM, "hipMalloc of dataset %d's tile ranges failed", k);
    HIP_TRY(hipMemcpy(D.txlo.p, tlo.data(), ntp * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.txhi.p, thi.data(), ntp * sizeof(double), hipMemcpyHostToDevice));
    f.txlo = D.txlo.p;
    f.txhi = D.txhi.p;
    f.n_tiles = (int64_t)nt;
    // Per-window grids.  A dataset that is not ONE grid (grid_H == 0) may still be a grid window
    // by window - concatenated scans, a re-gridded stretch, one jittered region: window w gets
    // H_w = 64 h when its data points are x_first + i h to 8 ulp of the window's max |x| - first
    // tried with the previous grid window's h, so that runs of windows on one grid carry the SAME
    // bits (the kernel re-derives its constants only where H changes: PeaksModel::regrid), then
    // with its own h = (x_last - x_first) / (points - 1) - and 0 otherwise (direct form there).
    // MHX_NO_WINDOW_GRIDS=1: off (round 3's rule: one grid or none).
    f.tgh = nullptr;
    {
      const char* nwg = getenv("MHX_NO_WINDOW_GRIDS");
      if (f.grid_H == 0.0 && !D.no_rec && !(nwg && atoi(nwg) != 0) && D.n >= 2) {
        std::vector<double> gh(ntp, 0.0);
        double h_prev = 0.0;
        bool any = false;
        for (size_t t = 0; t < ntp; ++t) {
          const size_t i0 = t * wp, cnt = std::min(wp, D.n > i0 ? D.n - i0 : 0);
          if (cnt < 2) continue;
          const double* xw = &D.hx[i0];
          if (!std::isfinite(xw[0]) || !std::isfinite(xw[cnt - 1])) continue;
          const double tol = 8.0 * 0x1p-52 * std::max(std::fabs(xw[0]), std::fabs(xw[cnt - 1]));
          auto fits = [&](double h) {
            if (h == 0.0 || !std::isfinite(h)) return false;
            for (size_t i = 0; i < cnt; ++i)
              if (!(std::fabs(xw[i] - (xw[0] + (double)i * h)) <= tol)) return false;
            return true;
          };
          double h = h_prev;
          if (!fits(h)) h = (xw[cnt - 1] - xw[0]) / (double)(cnt - 1);
          if (!fits(h)) continue;
          gh[t] = 64.0 * h;
          h_prev = h;
          any = true;
        }
        if (any) {
          if (D.tgh.alloc(ntp, false) != hipSuccess)
            return fail(MHX_ENOMEM, "hipMalloc of dataset %d's window grids failed", k);
          HIP_TRY(hipMemcpy(D.tgh.p, gh.data(), ntp * sizeof(double), hipMemcpyHostToDevice));
          f.tgh = D.tgh.p;
        }
      }
    }
    // the one tile stays in LDS (sweep).  Measured (tools/ab_tree.sh, test.lisp's shape): +9 %
    // with one chain, -5 % with 512 and -15 % with 2048 chains: only the single-walker case, the
    // reference's own way of working, gets it
    f.solo = (e->P.K == 1 && nt == 1 && e->cfg.n_chains <= e->fam->waves_per_group) ? 1 : 0;
  }
  // MHX_NO_TILE_SKIP=1: evaluate every Gaussian peak at every point (the skipping is exact, so
  // this only exists to show that the results do not change)
  const char* nts = getenv("MHX_NO_TILE_SKIP");
  const int tile_skip = (nts && atoi(nts) != 0) ? 0 : 1;
  bool any_expr = false;
  // MHX_NO_RECOGNISE=1: every expression compiled as written (A/B runs; bench.py --workload c2expr)
  const char* nrec = getenv("MHX_NO_RECOGNISE");
  const bool recognise = e->recognise && !(nrec && atoi(nrec) != 0);
  for (int k = 0; k < e->P.K; ++k) {
    FnDesc& f = e->P.fn[k];
    f.user_slot = f.prior_slot = -1;
    f.tile_skip = tile_skip;
    if (!e->fn_expr[k].expr.empty()) {
      // A function given as an expression: a body that IS polynomial background + Gaussian /
      // Lorentzian peaks runs as that enumerated model (same gather map, permuted into the
      // model's order), unless its likelihood is an expression too (which reads `model` from
      // an expression model) or recognition is off.  Decided here, not in
      // mhx_set_function_expr: the dataset's likelihood may be set before or after.
      const UserExpr& u = e->fn_expr[k];
      const RecognisedModel& r = e->fn_recog[k];
      if (recognise && r.model >= 0 && f.lik != MHX_LIK_EXPR) {
        f.model = r.model;
        f.n_idx = (int)r.order.size();
        for (int j = 0; j < f.n_idx; ++j) f.idx[j] = u.index[(size_t)r.order[(size_t)j]];
        f.shape[0] = r.shape[0];
        f.shape[1] = r.shape[1];
      } else {
        f.model = MHX_MODEL_EXPR;
        f.n_idx = (int)u.index.size();
        for (int j = 0; j < f.n_idx; ++j) f.idx[j] = u.index[(size_t)j];
        f.shape[0] = f.shape[1] = 0;
      }
    }
    {  // MHX_NO_YW=1: the shared-tile layout with x for every step (A/B runs; same bits either way)
      const char* ny = getenv("MHX_NO_YW");
      f.no_yw = (ny && atoi(ny) != 0) ? 1 : 0;
    }
    if (f.lik == MHX_LIK_EXPR) {
      if (e->lik_expr[k].empty())
        return fail(MHX_ESTATE, "dataset %d uses MHX_LIK_EXPR but mhx_set_likelihood_expr was "
                                "never called for it", k);
      if (f.model != MHX_MODEL_EXPR)
        return fail(MHX_EUNSUPPORTED, "function %d: an expression likelihood needs an expression "
                                      "model (mhx_set_function_expr)", k);
    }
    if (f.model == MHX_MODEL_EXPR && e->fn_expr[k].xcols > f.n_xcols)
      return fail(MHX_ESTATE, "function %d reads xcol1 but dataset %d has one column of x "
                              "(mhx_set_dataset_cols)", k, k);
    any_expr = any_expr || f.model == MHX_MODEL_EXPR || !e->prior_expr[k].expr.empty();
  }
  // Which kernels: an ahead-of-time specialisation if the problem matches one; otherwise kernels
  // compiled at run time (hiprtc) in which EVERY function - expression or enumerated model - gets
  // its own compile-time specialisation (parameters in SGPRs, fast exp path, tile-level peak
  // skipping): 2.4-2.8x the run-time-dispatched generic kernels on configs 2 and 3.  The generic
  // kernels remain for MHX_FORCE_GENERIC=1 / MHX_NO_RTC_SPECIALISE=1 and for machines without
  // hiprtc.
  {  // MHX_NO_DEAL=1: every wave judges its own chain's proposal (A/B runs; same bits either way)
    const char* nd = getenv("MHX_NO_DEAL");
    e->P.no_deal = (nd && atoi(nd) != 0) ? 1 : 0;
    e->P.test_lose_sweepers = 0;
#ifdef MHX_DEBUG_HOOKS  // (tests/hooks/libmhx_hooks.so: a persistent launch whose sweepers never come)
    const char* tl = getenv("MHX_TEST_LOSE_SWEEPERS");
    e->P.test_lose_sweepers = (tl && atoi(tl) != 0) ? 1 : 0;
#endif
  }
  const char* fg = getenv("MHX_FORCE_GENERIC");
  const char* ns = getenv("MHX_NO_RTC_SPECIALISE");
  const bool force_generic = fg && atoi(fg) != 0;
  const bool specialise = !(ns && atoi(ns) != 0) && !force_generic;
  // a function with a per-window grid table needs the kernel instance that follows it
  // (FixedSpec<Model, LIK, true>, mhx_kernels.hpp): compiled at run time, like every problem
  // without an ahead-of-time kernel - the ahead-of-time ones stay what they were
  bool any_wgrid = false;
  for (int k = 0; k < e->P.K; ++k) {
    FnDesc& f = e->P.fn[k];
    const bool can = specialise && f.tgh != nullptr && f.model == MHX_MODEL_GAUSS_PEAKS &&
                     f.shape[0] >= 1 && f.shape[0] <= 2 && f.shape[1] >= 1 && f.shape[1] <= 6 &&
                     f.lik != MHX_LIK_EXPR;
    if (!can) f.tgh = nullptr;  // (nobody would read it: the direct form everywhere, as before)
    any_wgrid = any_wgrid || can;
  }
  // MHX_EARLY_REJECT=1: sweep()'s exact early rejection (mhx_kernels.hpp) for a problem of ONE
  // function of a bounded enumerated model (peaks, polynomial: no term of theirs overflows at
  // finite parameters) with the weighted normal likelihood and no prior body - compiled at run
  // time too, so that the ahead-of-time kernels do not carry the test (4-5 % in a walk's first
  // iterations even when it never fires)
  bool any_er = false;
  {
    const char* er = getenv("MHX_EARLY_REJECT");
    const FnDesc& f0 = e->P.fn[0];
    any_er = er && atoi(er) != 0 && specialise && e->P.K == 1 && f0.lik == MHX_LIK_NORMAL &&
             (f0.model == MHX_MODEL_GAUSS_PEAKS || f0.model == MHX_MODEL_LORENTZ_PEAKS ||
              f0.model == MHX_MODEL_POLY) &&
             e->prior_expr[0].expr.empty() && !builtin_model_type(f0).empty() &&
             e->cfg.adapt_mode != MHX_ADAPT_POOLED;
  }
  const int aot = (any_expr || any_wgrid || any_er) ? SPEC_GENERIC : select_spec(e->P);
  e->rtc_note.clear();
  HIP_TRY(hipMemcpy(e->dP.p, &e->P, sizeof(ProblemDesc), hipMemcpyHostToDevice));
  if (!any_expr && !any_wgrid && !any_er && (aot != SPEC_GENERIC || !specialise)) {
    e->spec = force_generic ? SPEC_GENERIC : aot;
    e->user_prog.reset();
  } else {
    std::vector<UserExpr> models, priors;
    bool builtin = false;
    for (int k = 0; k < e->P.K; ++k) {
      FnDesc& f = e->P.fn[k];
      if (f.model == MHX_MODEL_EXPR) {
        f.user_slot = (int)models.size();
        models.push_back(e->fn_expr[k]);
        models.back().lik = f.lik;
        if (f.lik == MHX_LIK_EXPR) models.back().lik_expr = e->lik_expr[k];
      } else {
        const std::string type = specialise ? builtin_model_type(f) : std::string();
        if (!type.empty()) {
          f.user_slot = (int)models.size();
          UserExpr u;
          u.builtin = type;
          u.lik = f.lik;
          u.wgrid = f.tgh != nullptr;
          u.early_reject = any_er;
          models.push_back(u);
        } else {
          builtin = true;  // stays with the generic dispatcher inside the compiled kernels
        }
      }
      if (!e->prior_expr[k].expr.empty()) {
        f.prior_slot = (int)priors.size();
        priors.push_back(e->prior_expr[k]);
      }
    }
    HIP_TRY(hipMemcpy(e->dP.p, &e->P, sizeof(ProblemDesc), hipMemcpyHostToDevice));
    std::string err;
    // (before the kernels exist their occupancy is not known: the most they could have)
    const int64_t cap_guess = persist_allowed(e) ? 2 * 256 : 0;
    const bool want_split = choose_split(e, *e->fam, !builtin, cap_guess) > 0 ||
                            choose_tsplit(e, *e->fam, !builtin, cap_guess, cap_guess) > 0;
    std::shared_ptr<UserProgram> prog =
        rtc_get(models, priors, builtin, want_split, *e->fam, &err);
    if (prog) {
      e->user_prog = prog;
      e->spec = SPEC_USER;
    } else if (any_expr) {
      return fail(MHX_EUNSUPPORTED, "%s", err.c_str());
    } else {  // enumerated models only: the generic kernels serve (no hiprtc on this machine?)
      e->rtc_note = err.substr(0, err.find('\n'));
      for (int k = 0; k < e->P.K; ++k) e->P.fn[k].user_slot = -1;
      HIP_TRY(hipMemcpy(e->dP.p, &e->P, sizeof(ProblemDesc), hipMemcpyHostToDevice));
      e->user_prog.reset();
      e->spec = SPEC_GENERIC;
    }
  }
  // split mode for small batches on long datasets
  {
    const bool capable = e->spec == SPEC_USER ? e->user_prog->has_split
                                              : e->fam->split_capable(e->spec);
    const bool pa = capable && persist_allowed(e);
    const int64_t cap_pc = pa ? persist_capacity(e, false) : 0;
    const int64_t cap_ts = pa && persist_ts_wanted() != 0 ? persist_capacity(e, true) : 0;
    const int ts = choose_tsplit(e, *e->fam, capable, cap_pc, cap_ts);
    e->tsplit = ts > 0;
    e->split_slices = e->tsplit ? ts : choose_split(e, *e->fam, capable, cap_pc);
    e->S.split_slots = e->tsplit ? e->split_slices : e->split_slices * e->fam->waves_per_group;
    e->S.split_part = nullptr;
    e->ts_initial = e->tsplit ? ts : 0;
    if (e->tsplit) {
      const int rc = build_ts_table(e, ts);
      if (rc != MHX_OK) return rc;
    }
    if (e->split_slices > 0) {
      // (tile-sliced: room for the most slices a later re-slicing may take - compact_slots)
      int64_t nwin_max = 1;
      for (int k = 0; k < e->P.K; ++k)
        nwin_max = std::max<int64_t>(nwin_max, (e->P.fn[k].n + kPadPoints - 1) / kPadPoints);
      const size_t slots_max =
          e->tsplit ? (size_t)std::max<int64_t>(e->S.split_slots, std::min<int64_t>(nwin_max, 512))
                    : (size_t)e->S.split_slots;
      const size_t np = (size_t)e->cfg.n_chains * e->P.K * slots_max;
      if (e->split_part.alloc(np) != hipSuccess)
        return fail(MHX_ENOMEM, "hipMalloc of the split-mode partial sums failed");
      e->S.split_part = e->split_part.p;
    }
    HIP_TRY(hipMemset(e->split_pending.p, 0, (size_t)e->cfg.n_chains * sizeof(int32_t)));
    // the per-chain split mode as ONE launch per portion of iterations (k_persist): the chain's
    // master wave and its sweep workgroups hand each other the proposal and the partial sums
    // through memory, which needs every workgroup of the launch on the GPU at once
    // (MHX_NO_PERSIST=1: the two launches per iteration of rounds 1-3)
    e->persist = false;
    if (e->split_slices > 0) {
      const int64_t W = e->fam->waves_per_group;
      // tile-sliced: (1 + slices) workgroups per chain GROUP - with fewer slices, down to 2,
      // where the default slicing would not fit the GPU at once (a run's repacking keeps to the
      // same bound: compact_tsplit)
      const int64_t units = e->tsplit ? (e->cfg.n_chains + W - 1) / W : e->cfg.n_chains;
      const int64_t cap = persist_capacity(e, e->tsplit);
      int64_t slices = e->split_slices;
      // The tile-sliced form runs with fewer slices where the default slicing does not fit the
      // GPU at once - down to three quarters of it (MHX_PERSIST_TS=1: down to 2; =0: never the
      // persistent form).  Measured round 4 (two-peak problem, 1e5 points, us per iteration,
      // persistent | two launches):  8 chains x49 10.1 | 20.0    64: x49 12.2 | 23.0
      //   128: x27 14.3 | x32 26.5    256: x13 19.6 | x16 33.6    512: x6 31.2 | x8 46.6
      //   1024: x2 68.9 | x4 71.7;   1e6 points  8: 15.4 | 23.8    64: 34.5 | 45.8
      //   256: 120 | 121    1024: x2 611 | x4 429 (half the slices: not taken).
      // (Round 3 had measured the persistent form no faster and left it behind a switch: its
      // kernel took 132 VGPRs, one workgroup fitted a CU, and the launch ran in two shifts.)
      const int wanted = persist_ts_wanted();
      if (e->tsplit && !getenv("MHX_TSPLIT")) {
        int64_t nwin_all = 1;
        for (int k = 0; k < e->P.K; ++k)
          nwin_all = std::max<int64_t>(nwin_all, (e->P.fn[k].n + kPadPoints - 1) / kPadPoints);
        const int64_t fit = trim_slices(nwin_all, std::min<int64_t>(slices, cap / units - 1));
        const int64_t least = wanted > 0 ? 2 : std::max<int64_t>(2, (3 * slices + 3) / 4);
        // (fewer slices only where an iteration is short enough for the saved launches to
        // matter: up to 48 windows per slice - 1e6 points, 1024 walkers: x3, 163 windows each,
        // 500 us against the two launches' x4 424; 512: x7, 70 each, 226 against x8 223;
        // 256: x15, 33 each, 106 against x16 121; 1e5 points, 1024 walkers: x3 55.9 against x4 72.0)
        const bool short_rounds = fit > 0 && (nwin_all + fit - 1) / fit <= 48;
        slices = fit >= least && (fit == slices || short_rounds || wanted > 0) ? fit : 0;
      }
      const bool want = e->tsplit ? wanted != 0 : true;
      e->persist = want && persist_allowed(e) && slices >= (e->tsplit ? 2 : 1) &&
                   units * (1 + slices) <= cap;
      if (e->persist && e->tsplit) {  // (the table of the persistent form: fewer slices, resident windows)
        const int rc = build_ts_table(e, (int)slices);
        if (rc != MHX_OK) return rc;
        e->split_slices = (int)slices;
        e->S.split_slots = (int)slices;
        e->ts_initial = (int)slices;
      }
      if (e->persist) {
        const size_t nm = 64 * (size_t)e->cfg.n_chains;
        int64_t nwin_all = 1;
        for (int k = 0; k < e->P.K; ++k)
          nwin_all = std::max<int64_t>(nwin_all, (e->P.fn[k].n + kPadPoints - 1) / kPadPoints);
        const size_t slots_cap = e->tsplit ? (size_t)std::max<int64_t>(e->S.split_slots, std::min<int64_t>(nwin_all, 512))
                                           : (size_t)e->S.split_slots;
        const size_t npb = 16 * (size_t)e->cfg.n_chains * e->P.K * slots_cap;
        if ((e->persist_msg.n < nm && e->persist_msg.alloc(nm) != hipSuccess) ||
            (e->persist_part.n < npb && e->persist_part.alloc(npb) != hipSuccess))
          return fail(MHX_ENOMEM, "hipMalloc of the persistent kernel's handshake buffers failed");
        if (!e->persist_error.p && e->persist_error.alloc(1) != hipSuccess)
          return fail(MHX_ENOMEM, "hipMalloc of the persistent kernel's error word failed");
        e->S.persist_msg = e->persist_msg.p;
        e->S.persist_part = e->persist_part.p;
        e->S.persist_error = e->persist_error.p;
      }
    }
  }
  // a name for what was chosen (mhx_kernel_name)
  static const char* kLik[] = {"normal", "normal_cutoff", "poisson", "expr"};
  e->kernel_name = "w" + std::to_string(e->fam->waves_per_group) + "/";
  if (e->spec == SPEC_USER) {
    e->kernel_name += "rtc[";
    for (int k = 0; k < e->P.K; ++k) {
      const FnDesc& f = e->P.fn[k];
      const std::string t = f.model == MHX_MODEL_EXPR ? std::string("expr")
                            : (f.user_slot >= 0 ? builtin_model_type(f) + (f.tgh ? "+wgrid" : "") +
                                                      (any_er ? "+early-reject" : "")
                                                : std::string("generic"));
      e->kernel_name += (k ? ", " : "") + t + ":" + kLik[f.lik & 3];
    }
    e->kernel_name += "]";
  } else {
    e->kernel_name += spec_name(e->spec);
    if (!e->rtc_note.empty()) e->kernel_name += " [not specialised: " + e->rtc_note + "]";
  }
  if (e->split_slices > 0)
    e->kernel_name += (e->tsplit ? (e->persist ? " persistent tsplit x" : " tsplit x")
                                 : (e->persist ? " persistent split x" : " split x")) +
                      std::to_string(e->split_slices);
  e->problem_dirty = false;
  return MHX_OK;
}

// launch through the ahead-of-time table or the run-time compiled module
hipError_t do_logpost(mhx_engine* e, const double* th, int64_t n, double* out, double* parts) {
  return e->spec == SPEC_USER
             ? rtc_launch_logpost(*e->user_prog, e->stream, e->dP.p, th, n, out, parts)
             : e->fam->logpost(e->spec, e->stream, e->dP.p, th, n, out, parts);
}
hipError_t do_init(mhx_engine* e) {
  return e->spec == SPEC_USER ? rtc_launch_init(*e->user_prog, e->stream, e->dP.p, e->S)
                              : e->fam->init(e->spec, e->stream, e->dP.p, e->S);
}
hipError_t do_step_injected(mhx_engine* e, const double* L, int pcl, const double* z,
                            const double* u, const double* T, unsigned char* acc) {
  return e->spec == SPEC_USER
             ? rtc_launch_step_injected(*e->user_prog, e->stream, e->dP.p, e->S, L, pcl, z, u, T, acc)
             : e->fam->step_injected(e->spec, e->stream, e->dP.p, e->S, L, pcl, z, u, T, acc);
}
hipError_t do_split_sweep(mhx_engine* e) {
  if (e->tsplit)
    return e->spec == SPEC_USER
               ? rtc_launch_split_tsweep(*e->user_prog, e->stream, e->dP.p, e->ts_table.p, e->S,
                                         e->split_slices)
               : e->fam->split_tsweep(e->spec, e->stream, e->dP.p, e->ts_table.p, e->S,
                                      e->split_slices);
  return e->spec == SPEC_USER
             ? rtc_launch_split_sweep(*e->user_prog, e->stream, e->dP.p, e->S, e->split_slices)
             : e->fam->split_sweep(e->spec, e->stream, e->dP.p, e->S, e->split_slices);
}
hipError_t do_split_step(mhx_engine* e, int mode, int plain) {
  return e->spec == SPEC_USER
             ? rtc_launch_split_step(*e->user_prog, e->stream, e->dP.p, e->S, e->R, mode, plain)
             : e->fam->split_step(e->spec, e->stream, e->dP.p, e->S, e->R, mode, plain);
}
hipError_t do_persist(mhx_engine* e, int64_t iters, int plain) {
  if (e->tsplit)
    return e->spec == SPEC_USER
               ? rtc_launch_persist_ts(*e->user_prog, e->stream, e->dP.p, e->ts_table.p, e->S, e->R,
                                       e->split_slices, iters, plain)
               : e->fam->persist_ts(e->spec, e->stream, e->dP.p, e->ts_table.p, e->S, e->R,
                                    e->split_slices, iters, plain);
  return e->spec == SPEC_USER
             ? rtc_launch_persist(*e->user_prog, e->stream, e->dP.p, e->S, e->R, e->split_slices, iters, plain)
             : e->fam->persist(e->spec, e->stream, e->dP.p, e->S, e->R, e->split_slices, iters, plain);
}
hipError_t do_adaptive(mhx_engine* e, int64_t iters, int plain) {
  return e->spec == SPEC_USER
             ? rtc_launch_adaptive(*e->user_prog, e->stream, e->dP.p, e->S, e->R, iters, plain)
             : e->fam->adaptive(e->spec, e->stream, e->dP.p, e->S, e->R, iters, plain);
}

int check_model(int model, const int32_t* shape, int n_shape, int n_index) {
  switch (model) {
    case MHX_MODEL_POLY:
      if (n_index < 1) return fail(MHX_EINVAL, "POLY needs >= 1 parameter");
      return MHX_OK;
    case MHX_MODEL_GAUSS_PEAKS:
    case MHX_MODEL_LORENTZ_PEAKS:
      if (n_shape < 2 || shape[0] < 0 || shape[1] < 0 || shape[0] + 3 * shape[1] != n_index)
        return fail(MHX_EINVAL, "peaks model: shape {nbg,npk} must satisfy nbg+3*npk == n_index");
      return MHX_OK;
    case MHX_MODEL_LORDER_MIXED:
      return n_index == 6 ? MHX_OK : fail(MHX_EINVAL, "LORDER_MIXED takes 6 parameters");
    case MHX_MODEL_EXP_DECAY:
      return n_index == 3 ? MHX_OK : fail(MHX_EINVAL, "EXP_DECAY takes 3 parameters");
    case MHX_MODEL_SINUSOID:
      return n_index == 4 ? MHX_OK : fail(MHX_EINVAL, "SINUSOID takes 4 parameters");
    case MHX_MODEL_PVOIGT2:
      return n_index == 11 ? MHX_OK : fail(MHX_EINVAL, "PVOIGT2 takes 11 parameters");
    default:
      return fail(MHX_EINVAL, "unknown model id %d", model);
  }
}

// M:379-380: (reduce (lambda (x y) (+ x (log y))) (up-to n)) sums SINGLE-float logs
double log_factorial_ref(long n, bool in_double, std::vector<float>& cache) {
  if (in_double) return std::lgamma((double)n + 1.0);
  if (n <= 0) return 0.0;
  if ((long)cache.size() <= n) {
    size_t old = cache.size();
    if (old == 0) {
      cache.push_back(0.0f);
      old = 1;
    }
    cache.resize((size_t)n + 1);
    for (size_t m = old; m <= (size_t)n; ++m) cache[m] = cache[m - 1] + (float)std::log((double)m);
  }
  return (double)cache[(size_t)n];
}

int alloc_state(mhx_engine* e) {
  const int64_t C = e->cfg.n_chains;
  const int d = e->cfg.n_params;
  const int64_t sts = steps_to_settle_of(d);
  int want = e->cfg.history_capacity > 0 ? e->cfg.history_capacity : 1024;
  want = std::max<int>(want, (int)std::max<int64_t>(1000, sts));
  const int Rcap = pow2_ceil(want);
  auto& S = e->S;
  S.n_chains = C;
  S.slot_chain = nullptr;
  S.n_slots = (int64_t)C;
  S.chain_offset = e->cfg.chain_offset;
  S.d = d;
  S.R = Rcap;
  S.seed = e->cfg.seed;
#define ALLOC(buf, count)                                             \
  if (e->buf.alloc((size_t)(count)) != hipSuccess)                    \
    return fail(MHX_ENOMEM, "hipMalloc of %s (%zu elements) failed", #buf, (size_t)(count));
  ALLOC(theta, C * d);
  ALLOC(prob, C);
  ALLOC(best_theta, C * d);
  ALLOC(best_prob, C);
  ALLOC(length, C);
  ALLOC(age, C);
  ALLOC(draw, C);
  ALLOC(n_hist, C);
  ALLOC(hist_prob, C * Rcap);
  ALLOC(hist_theta, C * Rcap * d);
  ALLOC(L, C * d * d);
  ALLOC(temperature, C);
  ALLOC(loop_i, C);
  ALLOC(reset_index, C);
  ALLOC(shutting, C);
  ALLOC(status, C);
  ALLOC(fwd_idx, C * sts);
  ALLOC(mat_tmp, C * 2 * d * d);
  ALLOC(pool_stats, C * (1 + d + d * d));
  ALLOC(stop_flag, 1);
  ALLOC(step_counter, 1);
  ALLOC(pool_vec, 1 + d + d * d);
  ALLOC(L_pool, d * d);
  ALLOC(pool_valid, 1);
  ALLOC(split_prop, C * d);
  ALLOC(split_u, C);
  ALLOC(split_pending, C);
#undef ALLOC
  S.theta = e->theta.p;
  S.prob = e->prob.p;
  S.best_theta = e->best_theta.p;
  S.best_prob = e->best_prob.p;
  S.length = e->length.p;
  S.age = e->age.p;
  S.draw = e->draw.p;
  S.n_hist = e->n_hist.p;
  S.hist_prob = e->hist_prob.p;
  S.hist_theta = e->hist_theta.p;
  S.L = e->L.p;
  S.temperature = e->temperature.p;
  S.loop_i = e->loop_i.p;
  S.reset_index = e->reset_index.p;
  S.shutting = e->shutting.p;
  S.status = e->status.p;
  S.fwd_idx = e->fwd_idx.p;
  S.mat_tmp = e->mat_tmp.p;
  S.pool_stats = e->pool_stats.p;
  S.step_counter = e->step_counter.p;
  S.pool_vec = e->pool_vec.p;
  S.L_pool = e->L_pool.p;
  S.pool_valid = e->pool_valid.p;
  S.split_prop = e->split_prop.p;
  S.split_u = e->split_u.p;
  S.split_pending = e->split_pending.p;
  S.split_part = nullptr;
  S.split_slots = 0;
  return MHX_OK;
}

void drop_split_graph(mhx_engine* e) {
  if (e->split_graph) (void)hipGraphExecDestroy(e->split_graph);
  e->split_graph = nullptr;
  e->split_graph_iters = 0;
  e->split_graph_plain = -1;
}

// `bytes` of staging memory, 256-byte aligned pieces carved by the caller
int ensure_stage(mhx_engine* e, size_t bytes) {
  if (e->stage.n >= bytes) return MHX_OK;
  const size_t want = std::max<size_t>(bytes, 2 * e->stage.n);
  if (e->stage.alloc(want, false) != hipSuccess)
    return fail(MHX_ENOMEM, "hipMalloc of %zu staging bytes failed", want);
  return MHX_OK;
}
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The slot -> chain map on the device (ChainState::slot_chain).  Grown whenever a deal needs more
// entries than it holds: which of compact_tsplit / compact_slots / deal_initial runs first depends
// on the problem the engine was finalised for, and the three need different lengths.
int put_slot_map(mhx_engine* e, const std::vector<int32_t>& map) {
  if (e->slot_map.n < map.size()) {
    HIP_TRY(hipStreamSynchronize(e->stream));  // (nobody reads the old map any more)
    drop_split_graph(e);                       // (its address is frozen into captured launches)
    e->S.slot_chain = nullptr;
    if (e->slot_map.alloc(map.size() + 64, false) != hipSuccess)
      return fail(MHX_ENOMEM, "hipMalloc of the slot map (%zu entries) failed", map.size());
  }
  HIP_TRY(hipMemcpy(e->slot_map.p, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  e->S.slot_chain = e->slot_map.p;
  e->S.n_slots = (int64_t)map.size();
  return MHX_OK;
}

// Chains that have finished give up their wave slots - where that pays.  In a long run of many
// chains the walks end at very different loop indices (:prob-settle), and a workgroup whose
// chains are mostly done carries idle waves through every tile of every sweep until its last
// chain ends.  But a wave that has its SIMD to itself runs at 39 % of the SIMD's issue rate where
// four sharing it get 25 % each, so as long as there is a CU for every workgroup the survivors
// are better left spread out (measured: packing 4096 chains' survivors made complete runs of
// config 2 8 % and of poly7 20 % SLOWER; dealing them evenly over the same 256 workgroups made
// them 20 % and 11 % faster).  So: the chains still walking are dealt round-robin over
// max(what they need, min(what the launch had, what the GPU holds)) workgroups, empty slots
// marked -1: every CU and SIMD gets its share of the waves that are left.  The chains' results do not depend on the slot they walk in (per-walker
// adaptation; Philox is keyed by the chain's global id; in the pooled mode the statistics kernels
// index chains, not slots, and the pooled factor is the same for every chain).  Batch kernels
// only (split mode has no idle waves).  MHX_NO_COMPACT=1: off.
// Tile-sliced split mode: the chains still walking are PACKED into as few groups as hold them
// (a group's workgroups walk their slices whether one of its chains is pending or all eight), and
// the functions are cut again so that the sweep launch keeps about 512 workgroups.  The partial
// sums are then grouped by other slices: results to rounding, like everything in the split modes.
int compact_tsplit(mhx_engine* e, const std::vector<int32_t>& st, int64_t running) {
  const int64_t W = e->fam->waves_per_group;
  const int64_t mapped = e->slots_mapped > 0 ? e->slots_mapped : e->cfg.n_chains;
  if (running <= 0 || running * 4 > mapped * 3) return MHX_OK;
  const int64_t groups = (running + W - 1) / W;
  std::vector<int32_t> map((size_t)(groups * W), -1);
  size_t j = 0;
  for (size_t c = 0; c < st.size(); ++c)
    if (st[c] == MHX_CHAIN_RUNNING) map[j++] = (int32_t)c;
  {
    const int rc = put_slot_map(e, map);
    if (rc != MHX_OK) return rc;
  }
  e->slots_mapped = running;
  int64_t nwin = 1;
  for (int k = 0; k < e->P.K; ++k)
    nwin = std::max<int64_t>(nwin, (e->P.fn[k].n + kPadPoints - 1) / kPadPoints);
  const char* forced = getenv("MHX_TSPLIT");
  int64_t ts = forced ? e->split_slices
                      : std::max<int64_t>(e->split_slices,
                                          std::min<int64_t>(std::min<int64_t>(512 / groups, nwin), 512));
  if (e->persist && !forced)  // (every workgroup of a persistent launch on the GPU at once)
    ts = std::max<int64_t>(2, std::min<int64_t>(ts, persist_capacity(e, true) / groups - 1));
  if (!forced) ts = std::max<int64_t>(2, trim_slices(nwin, ts));
  if (ts != e->split_slices) {
    const int rc = build_ts_table(e, (int)ts);
    if (rc != MHX_OK) return rc;
    e->split_slices = (int)ts;
    e->S.split_slots = (int)ts;
  }
  drop_split_graph(e);  // (the chain state is an argument frozen into the captured launches)
  return MHX_OK;
}

// a new run: every chain walks again, in the slices the problem was finalised with
int reset_tsplit(mhx_engine* e) {
  if (!e->tsplit || e->split_slices == e->ts_initial) return MHX_OK;
  const int rc = build_ts_table(e, e->ts_initial);
  if (rc != MHX_OK) return rc;
  e->split_slices = e->ts_initial;
  e->S.split_slots = e->ts_initial;
  return MHX_OK;
}

int compact_slots(mhx_engine* e, const std::vector<int32_t>& st, int64_t running) {
  const char* nc = getenv("MHX_NO_COMPACT");
  if ((nc && atoi(nc) != 0) || !e->fam) return MHX_OK;
  if (e->tsplit) {
    if (!e->ts_repack_due) return MHX_OK;
    e->ts_repack_due = false;
    return compact_tsplit(e, st, running);
  }
  if (e->split_slices > 0) return MHX_OK;
  const int64_t W = e->fam->waves_per_group;
  const int64_t in_use = e->S.slot_chain ? e->S.n_slots : e->cfg.n_chains;
  const int64_t groups = (in_use + W - 1) / W;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0)
    cus = 256;
  const int64_t resident = (int64_t)cus * (W <= 8 ? 2 : 1);  // workgroups the GPU holds at once
  const char* force = getenv("MHX_COMPACT_ALWAYS");          // (tests: repack small launches too)
  const int64_t floor_groups = (force && atoi(force) != 0) ? 1 : resident;
  // chains mapped at the last deal (all of them before the first): a new deal when a quarter of
  // them has finished since.  With no more workgroups than the GPU holds their number stays, and
  // the deal only evens out how many waves each CU and SIMD still has to run.
  const int64_t mapped = e->slots_mapped > 0 ? e->slots_mapped : e->cfg.n_chains;
  if (running <= 0 || running * 4 > mapped * 3) return MHX_OK;
  const int64_t target =
      std::max<int64_t>((running + W - 1) / W, std::min<int64_t>(groups, floor_groups));
  std::vector<int32_t> map((size_t)(target * W), -1);
  int64_t j = 0;
  for (size_t c = 0; c < st.size(); ++c)
    if (st[c] == MHX_CHAIN_RUNNING) {
      map[(size_t)((j % target) * W + j / target)] = (int32_t)c;
      ++j;
    }
  const int rc = put_slot_map(e, map);
  if (rc != MHX_OK) return rc;
  e->slots_mapped = running;
  return MHX_OK;
}

// ... and the same at the start of a run whose chains do not fill the GPU: 1024 chains are 128
// workgroups of 8 - two waves on every SIMD of half the CUs, the other half idle; dealt over 256
// workgroups of 4 every wave has a SIMD to itself.
int deal_initial(mhx_engine* e) {
  const char* nc = getenv("MHX_NO_COMPACT");
  if ((nc && atoi(nc) != 0) || e->split_slices > 0 || !e->fam) return MHX_OK;
  const int64_t W = e->fam->waves_per_group, C = e->cfg.n_chains;
  const int64_t groups = (C + W - 1) / W;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0)
    cus = 256;
  // up to one workgroup per CU: workgroups of 4; between one and two per CU (8-wave family):
  // two on EVERY CU instead of two on some and one on the others
  const int64_t resident = (int64_t)cus * (W <= 8 ? 2 : 1);
  const int64_t target = groups <= cus ? std::min<int64_t>(cus, std::max<int64_t>(groups, (C + 3) / 4))
                                       : (groups < resident ? resident : groups);
  if (target <= groups || C <= W) return MHX_OK;
  std::vector<int32_t> map((size_t)(target * W), -1);
  for (int64_t c = 0; c < C; ++c) map[(size_t)((c % target) * W + c / target)] = (int32_t)c;
  const int rc = put_slot_map(e, map);
  if (rc != MHX_OK) return rc;
  e->slots_mapped = C;
  return MHX_OK;
}

int count_running(mhx_engine* e, int64_t* n_running) {
  std::vector<int32_t> st((size_t)e->cfg.n_chains);
  HIP_TRY(hipMemcpy(st.data(), e->status.p, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t r = 0;
  for (int32_t s : st) r += s == MHX_CHAIN_RUNNING;
  *n_running = r;
  return compact_slots(e, st, r);
}

// The temperature schedule of M:878, entries [lo, hi): (max 1 (* (cos (* x pi (+ 1 (* 2 (floor
// temp-steps 5000))) (/ (* 2 temp-steps)))) temperature)), products left to right, the rational
// factor converted to double.  The device holds a WINDOW of it over the loop indices the next
// launch can reach (every running chain that still anneals has loop index 1 + iterations since
// begin), so a run with n = 1e9 "until stopped" costs 32 MB, not 8 GB.
const int64_t kTempsWindow = [] {  // MHX_TEMPS_WINDOW: a small window for tests of the sliding
  const char* s = getenv("MHX_TEMPS_WINDOW");
  const long long v = s ? atoll(s) : 0;
  return (int64_t)(v >= 16 ? v : (1 << 22));
}();
int ensure_temps(mhx_engine* e, int64_t lo, int64_t hi) {
  RunDesc& R = e->R;
  hi = std::min(hi, R.temp_steps);
  if (lo >= hi) return MHX_OK;  // beyond the schedule: nothing is read
  if (lo >= R.temps_first && hi <= R.temps_first + e->temps_count) return MHX_OK;
  const int64_t top = std::min(R.temp_steps, lo + std::max<int64_t>(hi - lo, kTempsWindow));
  std::vector<double> temps((size_t)(top - lo));
  const double kfac = (double)(1 + 2 * (R.temp_steps / 5000));
  const double inv = 1.0 / (double)(2 * R.temp_steps);
  for (int64_t x = lo; x < top; ++x) {
    const double arg = (((double)x * M_PI) * kfac) * inv;
    const double v = std::cos(arg) * e->run_temperature;
    temps[(size_t)(x - lo)] = v > 1.0 ? v : 1.0;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));  // (a launch may still be reading the old window)
  if ((int64_t)e->temps.n < top - lo) {
    // the new buffer first, the old one released only once it exists: a failed allocation
    // leaves R.temps and the window it describes as they were
    DevBuf<double> fresh;
    if (fresh.alloc(temps.size(), false) != hipSuccess)
      return fail(MHX_ENOMEM, "hipMalloc(temperature schedule, %zu) failed", temps.size());
    std::swap(e->temps.p, fresh.p);
    std::swap(e->temps.n, fresh.n);
    R.temps = nullptr;  // (the old window went with `fresh`; set again below)
    e->temps_count = 0;
  }
  if (hipMemcpy(e->temps.p, temps.data(), temps.size() * sizeof(double), hipMemcpyHostToDevice) !=
      hipSuccess) {
    R.temps = nullptr;
    e->temps_count = 0;
    e->run_ready = false;  // no schedule on the device: the run cannot go on
    return fail(MHX_EDEVICE, "copying the temperature schedule to the device failed");
  }
  R.temps = e->temps.p;
  R.temps_first = lo;
  e->temps_count = top - lo;
  drop_split_graph(e);  // the run description is frozen into captured launches
  return MHX_OK;
}

// one timed launch of the fused step kernel: enqueue (returns without waiting, so that a group
// keeps every GPU busy at once) and finish (waits for it and does the accounting)
int launch_steps_finish(mhx_engine* e);
int launch_steps_enqueue(mhx_engine* e, int64_t iters, int plain) {
  if (e->launch_open) {
    const int rc = launch_steps_finish(e);
    if (rc != MHX_OK) return rc;
  }
  if (!plain) {
    const int rc = ensure_temps(e, e->global_iter + 1, e->global_iter + 1 + iters);
    if (rc != MHX_OK) return rc;
  }
  HIP_TRY(hipEventRecord(e->ev0, e->stream));
  if (e->split_slices > 0 && e->persist && !e->tsplit) {
    // persistent split mode: one launch, `iters` iterations (the kernel leaves its loop when no
    // chain is running any more); the sync words start every launch at zero
    // (launches of at most 2^22 iterations: what mhx_adaptive_advance asks for at most anyway)
    for (int64_t left = iters; left > 0; left -= (int64_t)1 << 22) {
      HIP_TRY(hipMemsetAsync(e->persist_msg.p, 0, e->persist_msg.n * sizeof(unsigned long long), e->stream));
      HIP_TRY(hipMemsetAsync(e->persist_part.p, 0, e->persist_part.n, e->stream));
      HIP_TRY(do_persist(e, std::min<int64_t>(left, (int64_t)1 << 22), plain));
      e->persist_launched = true;
      e->launches += 1;
    }
  } else if (e->split_slices > 0) {
    // split mode: prime (first half of iteration 1), then per iteration the sweep over all
    // slices and the chain's own launch (second half + first half of the next; the last one
    // leaves nothing outstanding).  2 * iters + 1 small launches, queued without waiting.
    // The batch kernel leaves its loop when no chain of the workgroup is running; here the HOST
    // drives the iterations, so they go out in portions with a look at the chain states between
    // them (the caller may ask for 2^40 iterations meaning "until done").
    int64_t left = iters;
    const int64_t portion = std::max<int64_t>(e->split_portion, 16);
    if (e->ts_repack_due) {  // the last launch ended on a portion boundary: look at the chains now
      int64_t running = 0;
      HIP_TRY(hipStreamSynchronize(e->stream));
      const int rc = count_running(e, &running);
      if (rc != MHX_OK) return rc;
      HIP_TRY(hipEventRecord(e->ev0, e->stream));
    }
    while (left > 0) {
      // (portions end where split_iter is a multiple of their length, whatever the caller asked for)
      const int64_t now = std::min<int64_t>(left, portion - e->split_iter % portion);
      // (round 4 tried ONE launch per iteration - the last workgroup of a chain group, by an
      // atomic ticket, running the group's step at the end of the sweep's launch: bit-identical,
      // and 27.7 us per iteration of 64 walkers where these two launches take 25.1: between two
      // nodes of a graph lie 0.3 us; what an iteration costs are the dependent memory round
      // trips inside both kernels, which the fusion only lines up behind a ticket)
      auto issue = [&](int64_t count) -> hipError_t {
        hipError_t he = do_split_step(e, 0, plain);
        for (int64_t it = 0; it < count && he == hipSuccess; ++it) {
          he = do_split_sweep(e);
          if (he == hipSuccess) he = do_split_step(e, it + 1 < count ? 1 : 2, plain);
        }
        return he;
      };
      const char* ng = getenv("MHX_NO_GRAPH");
      // (only whole portions: a remainder of another length would mean capturing again)
      const bool use_graph = !(ng && atoi(ng) != 0) && now == e->split_portion && !e->persist;
      bool done = false;
      if (e->persist) {  // tile-sliced, persistent: the portion is ONE launch (k_persist_ts)
        HIP_TRY(hipMemsetAsync(e->persist_msg.p, 0, e->persist_msg.n * sizeof(unsigned long long), e->stream));
        HIP_TRY(hipMemsetAsync(e->persist_part.p, 0, e->persist_part.n, e->stream));
        HIP_TRY(do_persist(e, now, plain));
        e->persist_launched = true;
        done = true;
      }
      if (use_graph) {
        if (!e->split_graph || e->split_graph_iters != now || e->split_graph_plain != plain) {
          drop_split_graph(e);
          hipGraph_t g = nullptr;
          if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const hipError_t he = issue(now);
            const hipError_t ce = hipStreamEndCapture(e->stream, &g);
            if (he == hipSuccess && ce == hipSuccess && g &&
                hipGraphInstantiate(&e->split_graph, g, nullptr, nullptr, 0) == hipSuccess) {
              e->split_graph_iters = now;
              e->split_graph_plain = plain;
            } else {
              e->split_graph = nullptr;
            }
            if (g) (void)hipGraphDestroy(g);
          }
          (void)hipGetLastError();
        }
        if (e->split_graph) {
          HIP_TRY(hipGraphLaunch(e->split_graph, e->stream));
          done = true;
        }
      }
      if (!done) HIP_TRY(issue(now));
      e->launches += e->persist ? 1 : 2 * (uint64_t)now + 1;
      left -= now;
      e->split_iter += now;
      e->ts_repack_due = e->tsplit && e->split_iter % portion == 0;
      if (left > 0) {
        int64_t running = 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        const int rc = count_running(e, &running);
        if (rc != MHX_OK) return rc;
        if (running == 0) break;
      }
    }
  } else {
    HIP_TRY(do_adaptive(e, iters, plain));
  }
  HIP_TRY(hipEventRecord(e->ev1, e->stream));
  e->launch_open = true;
  e->launch_iters = iters;
  // counted when the launch is in the queue, not when somebody has waited for it: the chains'
  // loop indices move with the kernel, and the schedule window and the pooled cadence of the
  // next launch are derived from this count whatever happens on the host in between
  if (!plain) e->global_iter += iters;
  return MHX_OK;
}
int launch_steps_finish(mhx_engine* e) {
  if (!e->launch_open) return MHX_OK;
  e->launch_open = false;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipEventSynchronize(e->ev1));
  if (e->persist_launched) {
    // a persistent launch counts on all its workgroups being on the GPU at once; when another
    // kernel held it, a master gave up waiting for its sweep workgroups: the iteration was taken
    // back (no chain was touched), this run is over, and the engine returns to two launches
    e->persist_launched = false;
    int32_t err = 0;
    HIP_TRY(hipMemcpy(&err, e->persist_error.p, sizeof err, hipMemcpyDeviceToHost));
    if (err != 0) {
      HIP_TRY(hipMemset(e->persist_error.p, 0, sizeof err));
      e->persist = false;
      e->persist_off = true;
      e->problem_dirty = true;  // (modes and kernel name are derived again, without persistence)
      e->run_ready = false;
      return fail(MHX_EDEVICE, "a persistent launch did not get all its workgroups onto the GPU at once "
                               "(another kernel held it); no chain was touched - begin again "
                               "(the engine now uses two launches per iteration)");
    }
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  e->kernel_ms += ms;
  e->timed_launches++;
  e->launches++;
  // keep a launch near 50 ms so the stop flag and the host stay responsive
  if (ms > 0.f) {
    double per_iter = ms / (double)std::max<int64_t>(e->launch_iters, 1);
    int64_t want = (int64_t)(50.0 / std::max(per_iter, 1e-6));
    e->chunk_iters = std::min<int64_t>(std::max<int64_t>(want, 8), 1 << 16);
  }
  return MHX_OK;
}
int launch_steps(mhx_engine* e, int64_t iters, int plain) {
  const int rc = launch_steps_enqueue(e, iters, plain);
  return rc != MHX_OK ? rc : launch_steps_finish(e);
}

// Pooled adaptation tick (MHX_ADAPT_POOLED): per-chain displacement statistics -> sum over the
// chains of this rank (fixed order) -> all-reduce over ranks (caller's hook: RCCL through
// torch.distributed, or any MPI-like sum) -> covariance, Cholesky, 2.38^2/d on every rank.
// (with an RCCL communicator everything stays on the engine's stream: no host synchronisation
// between the statistics kernels, the all-reduce and the factorisation)
int pool_enqueue_stats(mhx_engine* e) {
  HIP_TRY(e->fam->pool_stats(e->stream, e->S, e->R));
  HIP_TRY(e->fam->pool_reduce(e->stream, e->S));
  return MHX_OK;
}
int pool_enqueue_allreduce(mhx_engine* e) {  // inside ncclGroupStart/End when a group drives it
  const size_t E = 1 + (size_t)e->P.d + (size_t)e->P.d * e->P.d;
  const int rc = rccl().AllReduce(e->pool_vec.p, e->pool_vec.p, E, kNcclDouble, kNcclSum, e->comm,
                                  e->stream);
  if (rc != 0)
    return fail(MHX_ECOMM, "ncclAllReduce: %s", rccl_err(rc));
  return MHX_OK;
}
int pool_enqueue_factor(mhx_engine* e) {
  HIP_TRY(e->fam->pool_factor(e->stream, e->S));
  e->launches += 3;
  e->pool_refreshes++;
  return MHX_OK;
}
int pool_refresh(mhx_engine* e) {
  const size_t E = 1 + (size_t)e->P.d + (size_t)e->P.d * e->P.d;
  if (e->comm) {
    int rc = pool_enqueue_stats(e);
    if (rc == MHX_OK) rc = pool_enqueue_allreduce(e);
    if (rc == MHX_OK) rc = pool_enqueue_factor(e);
    return rc;  // stream-ordered before the next step launch
  }
  HIP_TRY(e->fam->pool_stats(e->stream, e->S, e->R));
  HIP_TRY(e->fam->pool_reduce(e->stream, e->S));
  if (e->allreduce) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    int rc;
    if (e->allreduce_device) {
      rc = e->allreduce(e->allreduce_ctx, e->pool_vec.p, E, 1);
    } else {
      std::vector<double> h(E);
      HIP_TRY(hipMemcpy(h.data(), e->pool_vec.p, E * sizeof(double), hipMemcpyDeviceToHost));
      rc = e->allreduce(e->allreduce_ctx, h.data(), E, 0);
      if (rc == 0)
        HIP_TRY(hipMemcpy(e->pool_vec.p, h.data(), E * sizeof(double), hipMemcpyHostToDevice));
    }
    if (rc != 0) return fail(MHX_ECOMM, "all-reduce hook returned %d", rc);
  }
  HIP_TRY(e->fam->pool_factor(e->stream, e->S));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches += 3;
  e->pool_refreshes++;
  return MHX_OK;
}

}  // namespace

extern "C" {

int mhx_version(void) { return MHX_VERSION; }
#ifndef MHX_SOURCE_ID
#define MHX_SOURCE_ID "unknown"
#endif
#ifdef MHX_DEBUG_HOOKS
const char* mhx_build_id(void) { return "csrc:" MHX_SOURCE_ID "+hooks"; }
#else
const char* mhx_build_id(void) { return "csrc:" MHX_SOURCE_ID; }
#endif
const char* mhx_last_error(void) { return g_err.c_str(); }

int mhx_device_count(int* count) {
  if (!count) return fail(MHX_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(MHX_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return MHX_OK;
}

int mhx_create(const mhx_config* cfg, mhx_engine** out) {
  if (!cfg || !out) return fail(MHX_EINVAL, "cfg/out is NULL");
  *out = nullptr;
  if (cfg->n_chains < 1) return fail(MHX_EINVAL, "n_chains must be >= 1");
  if (cfg->n_params < 1 || cfg->n_params > MHX_MAX_PARAMS)
    return fail(MHX_EINVAL, "n_params must be in [1,%d]", MHX_MAX_PARAMS);
  if (cfg->n_functions < 1 || cfg->n_functions > MHX_MAX_FUNCTIONS)
    return fail(MHX_EINVAL, "n_functions must be in [1,%d]", MHX_MAX_FUNCTIONS);
  if (cfg->adapt_mode != MHX_ADAPT_FAITHFUL && cfg->adapt_mode != MHX_ADAPT_POOLED)
    return fail(MHX_EINVAL, "adapt_mode");
  if ((uint64_t)(cfg->chain_offset + cfg->n_chains) > 0xFFFFFFFFull)
    return fail(MHX_EINVAL, "global chain ids must fit 32 bits (Philox counter word)");
  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev <= 0)
    return fail(MHX_EDEVICE, "no HIP device available (%s): libmhx has no CPU path",
                he != hipSuccess ? hipGetErrorString(he) : "0 devices");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(MHX_EINVAL, "device %d out of range (have %d)", cfg->device, ndev);
  mhx_engine* e = new (std::nothrow) mhx_engine();
  if (!e) return fail(MHX_ENOMEM, "host allocation failed");
  e->cfg = *cfg;
  e->device = cfg->device;
  int rc = MHX_OK;
  do {
    if ((rc = use_device(e)) != MHX_OK) break;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) != hipSuccess) {
      rc = fail(MHX_EDEVICE, "hipGetDeviceProperties failed");
      break;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("MHX_ALLOW_ANY_ARCH")) {
      rc = fail(MHX_EDEVICE, "device %d is %s; libmhx is built for gfx950 (MI355X) only",
                e->device, prop.gcnArchName);
      break;
    }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&e->stop_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
      rc = fail(MHX_EDEVICE, "stream/event creation failed");
      break;
    }
    if (family_w8().configure() != hipSuccess || family_w16().configure() != hipSuccess) {
      rc = fail(MHX_EDEVICE, "hipFuncSetAttribute(max dynamic LDS = %zu / %zu) failed",
                family_w8().lds_bytes, family_w16().lds_bytes);
      break;
    }
    if (e->dP.alloc(1) != hipSuccess) {
      rc = fail(MHX_ENOMEM, "hipMalloc(ProblemDesc) failed");
      break;
    }
    e->P.d = cfg->n_params;
    e->P.K = cfg->n_functions;
    if ((rc = alloc_state(e)) != MHX_OK) break;
    e->R.stop_flag = e->stop_flag.p;
  } while (0);
  if (rc != MHX_OK) {
    std::string keep = g_err;
    mhx_destroy(e);
    g_err = keep;
    return rc;
  }
  *out = e;
  return MHX_OK;
}

void mhx_destroy(mhx_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->comm && e->comm_owned && rccl().ok) (void)rccl().CommDestroy(e->comm);
  e->comm = nullptr;
  drop_split_graph(e);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->stop_stream) (void)hipStreamDestroy(e->stop_stream);
  delete e;
}

int mhx_set_function(mhx_engine* e, int k, int model_id, const int32_t* shape, int n_shape,
                     const int32_t* param_index, int n_index) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "function index %d out of range", k);
  if (n_index < 0 || n_index > MHX_MAX_FN_PARAMS || (n_index > 0 && !param_index))
    return fail(MHX_EINVAL, "n_index must be in [0,%d]", MHX_MAX_FN_PARAMS);
  if (n_shape < 0 || n_shape > 4 || (n_shape > 0 && !shape)) return fail(MHX_EINVAL, "n_shape");
  int rc = check_model(model_id, shape, n_shape, n_index);
  if (rc != MHX_OK) return rc;
  FnDesc& f = e->P.fn[k];
  for (int j = 0; j < n_index; ++j) {
    if (param_index[j] < 0 || param_index[j] >= e->P.d)
      return fail(MHX_EINVAL, "param_index[%d] = %d outside [0,%d)", j, param_index[j], e->P.d);
    f.idx[j] = param_index[j];
  }
  f.model = model_id;
  e->fn_expr[k] = UserExpr();
  e->fn_recog[k] = RecognisedModel();
  f.n_idx = n_index;
  for (int i = 0; i < 4; ++i) f.shape[i] = i < n_shape ? shape[i] : 0;
  e->fn_set[k] = true;
  e->problem_dirty = true;
  return MHX_OK;
}

// x1: the second column of a vector-valued x (mhx_set_dataset_cols), or NULL
static int set_dataset_impl(mhx_engine* e, int k, const double* x, const double* x1, const double* y,
                            const double* sigma, size_t n, int likelihood) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "dataset index %d out of range", k);
  if (n > 0 && (!x || !y)) return fail(MHX_EINVAL, "x/y is NULL");
  if (likelihood < MHX_LIK_NORMAL || likelihood > MHX_LIK_EXPR)
    return fail(MHX_EINVAL, "unknown likelihood %d", likelihood);
  if (x1 && likelihood == MHX_LIK_NORMAL_CUTOFF)
    return fail(MHX_EUNSUPPORTED, "a second column of x with log-liklihood-normal-cutoff: the tiles' "
                                  "fourth array is taken by the clamp's per-point constants");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  // padded to a whole number of tiles of either kernel family
  const size_t np = std::max<size_t>((n + kPadPoints - 1) / kPadPoints, 1) * kPadPoints;
  std::vector<double> hx(np), hy(np), hw(np), hc(np);
  long double csum = 0.0L;
  const double half_log_2pi = -0.5 * std::log(2.0 * M_PI);  // (* -1/2 (log (* 2 pi))) M:377
  std::vector<float> lf_cache;
  for (size_t i = 0; i < n; ++i) {
    hx[i] = x[i];
    hy[i] = y[i];
    if (likelihood == MHX_LIK_POISSON) {
      if (!(y[i] >= 0.0) || y[i] != std::floor(y[i]) || y[i] > 1e9)
        return fail(MHX_EINVAL, "poisson count y[%zu] = %g is not a non-negative integer", i, y[i]);
      hw[i] = 0.0;
      hc[i] = 0.0;
      csum -= (long double)log_factorial_ref((long)y[i], e->cfg.poisson_logfact_double != 0,
                                             lf_cache);
    } else if (likelihood == MHX_LIK_EXPR) {
      hw[i] = sigma ? sigma[i] : 1.0;  // handed to the expression as `error`, untouched
      hc[i] = 0.0;
    } else {
      const double s = sigma ? sigma[i] : 1.0;  // (if data-error data-error 1) M:1144
      if (!(s > 0.0) || !std::isfinite(s))
        return fail(MHX_EINVAL, "sigma[%zu] = %g must be finite and > 0 (M:376)", i, s);
      hw[i] = 1.0 / s;
      hy[i] = y[i] * hw[i];  // the kernel forms r = y/sigma - m/sigma with one fma
      hc[i] = half_log_2pi + (-1.0 * std::log(s));  // first two terms of M:377
      csum += (long double)hc[i];
    }
  }
  for (size_t i = n; i < np; ++i) {  // neutral pads: r = (0 - m) * 0 = 0
    hx[i] = n ? x[n - 1] : 0.0;
    hy[i] = 0.0;
    hw[i] = 0.0;
    hc[i] = 0.0;
  }
  if (x1)  // (the c array is read by the cutoff likelihood only, which was refused above)
    for (size_t i = 0; i < np; ++i) hc[i] = n ? x1[i < n ? i : n - 1] : 0.0;
  Dataset& D = e->data[k];
  if (D.x.alloc(np, false) != hipSuccess || D.y.alloc(np, false) != hipSuccess ||
      D.w.alloc(np, false) != hipSuccess || D.c.alloc(np, false) != hipSuccess)
    return fail(MHX_ENOMEM, "hipMalloc of dataset %d (%zu points) failed", k, np);
  HIP_TRY(hipMemcpy(D.x.p, hx.data(), np * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(D.y.p, hy.data(), np * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(D.w.p, hw.data(), np * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(D.c.p, hc.data(), np * sizeof(double), hipMemcpyHostToDevice));
  FnDesc& f = e->P.fn[k];
  f.x = D.x.p;
  f.y = D.y.p;
  f.w = D.w.p;
  f.c = D.c.p;
  f.txlo = f.txhi = nullptr;  // per-tile x ranges and n_tiles: finalize_problem, per family
  f.n = (int64_t)n;
  f.n_tiles = 0;
  D.hx.swap(hx);
  f.lik = likelihood;
  f.lik_const = (double)csum;
  f.xmin = f.xmax = n ? x[0] : 0.0;
  for (size_t i = 1; i < n; ++i) {
    f.xmin = std::min(f.xmin, x[i]);
    f.xmax = std::max(f.xmax, x[i]);
  }
  // A uniformly spaced x grid (spectra, histograms, time series: x_i = x_0 + i h) lets the
  // Gaussian peaks advance by a two-multiply recurrence instead of an exp per point
  // (PeaksModel, csrc/mhx_device.hpp).  Accepted when every x_i is within 8 ulp of max |x| of
  // x_0 + i h with h = (x_(n-1) - x_0) / (n - 1); MHX_NO_RECURRENCE=1 keeps the direct form.
  f.grid_H = 0.0;
  f.tgh = nullptr;  // (per-window grids: finalize_problem)
  f.n_xcols = x1 ? 2 : 1;
  D.n = n;
  {
    const char* nr = getenv("MHX_NO_RECURRENCE");
    D.no_rec = nr && atoi(nr) != 0;
    if (!(nr && atoi(nr) != 0) && n >= 2 && std::isfinite(x[0]) && std::isfinite(x[n - 1])) {
      const double h = (x[n - 1] - x[0]) / (double)(n - 1);
      const double tol = 8.0 * 0x1p-52 * std::max(std::fabs(x[0]), std::fabs(x[n - 1]));
      bool grid = h != 0.0 && std::isfinite(h);
      for (size_t i = 0; i < n && grid; ++i)
        grid = std::fabs(x[i] - (x[0] + (double)i * h)) <= tol;
      if (grid) f.grid_H = 64.0 * h;
    }
  }
  D.set = true;
  e->problem_dirty = true;
  return MHX_OK;
}

int mhx_set_dataset(mhx_engine* e, int k, const double* x, const double* y, const double* sigma,
                    size_t n, int likelihood) {
  return set_dataset_impl(e, k, x, nullptr, y, sigma, n, likelihood);
}

int mhx_set_dataset_cols(mhx_engine* e, int k, const double* const* xcols, int n_cols,
                         const double* y, const double* sigma, size_t n, int likelihood) {
  if (!xcols || n_cols < 1 || n_cols > 2 || !xcols[0] || (n_cols == 2 && !xcols[1]))
    return fail(MHX_EINVAL, "x must come as 1 or 2 columns (%d given)", n_cols);
  return set_dataset_impl(e, k, xcols[0], n_cols == 2 ? xcols[1] : nullptr, y, sigma, n, likelihood);
}

int mhx_set_bounds(mhx_engine* e, int k, const int32_t* idx, const double* lo, const double* hi,
                   int n) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "function index %d out of range", k);
  if (n < 0 || n > MHX_MAX_BOUNDS || (n > 0 && (!idx || !lo || !hi)))
    return fail(MHX_EINVAL, "n bounds must be in [0,%d]", MHX_MAX_BOUNDS);
  FnDesc& f = e->P.fn[k];
  for (int i = 0; i < n; ++i) {
    if (idx[i] >= e->P.d) return fail(MHX_EINVAL, "bounds idx[%d] = %d outside the vector", i, idx[i]);
    f.bidx[i] = idx[i] < 0 ? -1 : idx[i];
    f.blo[i] = lo[i];
    f.bhi[i] = hi[i];
  }
  f.n_bounds = n;
  e->problem_dirty = true;
  return MHX_OK;
}

// does `expr` mention the identifier `id` (as a whole word)?
static bool expr_names(const char* expr, const char* id) {
  const size_t n = strlen(id);
  for (const char* p = expr; (p = strstr(p, id)) != nullptr; p += n) {
    const bool left = p == expr || !(isalnum((unsigned char)p[-1]) || p[-1] == '_');
    const bool right = !(isalnum((unsigned char)p[n]) || p[n] == '_');
    if (left && right) return true;
  }
  return false;
}
static bool valid_ident(const char* s) {
  if (!s || !(isalpha((unsigned char)s[0]) || s[0] == '_')) return false;
  for (const char* p = s; *p; ++p)
    if (!(isalnum((unsigned char)*p) || *p == '_')) return false;
  return strcmp(s, "x") != 0 && strcmp(s, "xcol0") != 0 && strcmp(s, "xcol1") != 0 &&
         strcmp(s, "bounds_total") != 0;
}

int mhx_set_function_expr(mhx_engine* e, int k, const char* expr, const char* const* param_names,
                          const int32_t* param_index, int n_index) {
  if (!e || !expr) return fail(MHX_EINVAL, "engine/expr is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "function index %d out of range", k);
  if (n_index < 0 || n_index > MHX_MAX_FN_PARAMS || (n_index > 0 && (!param_names || !param_index)))
    return fail(MHX_EINVAL, "n_index must be in [0,%d]", MHX_MAX_FN_PARAMS);
  UserExpr u;
  for (int j = 0; j < n_index; ++j) {
    if (!valid_ident(param_names[j]))
      return fail(MHX_EINVAL, "parameter name %d is not an identifier (or is x / bounds_total)", j);
    if (param_index[j] < 0 || param_index[j] >= e->P.d)
      return fail(MHX_EINVAL, "param_index[%d] = %d outside [0,%d)", j, param_index[j], e->P.d);
    u.names.push_back(param_names[j]);
    u.index.push_back(param_index[j]);
  }
  std::string err;
  if (rtc_prepare_expr(expr, u.names, "x xcol0 xcol1", &u.expr, &err) != 0)
    return fail(MHX_EINVAL, "%s", err.c_str());
  u.xcols = expr_names(expr, "xcol1") ? 2 : 1;
  FnDesc& f = e->P.fn[k];
  f.model = MHX_MODEL_EXPR;
  f.n_idx = n_index;
  for (int j = 0; j < n_index; ++j) f.idx[j] = param_index[j];
  for (int i = 0; i < 4; ++i) f.shape[i] = 0;
  e->fn_expr[k] = u;
  (void)rtc_recognise(expr, u.names, &e->fn_recog[k]);  // which kernels: finalize_problem
  e->fn_set[k] = true;
  e->problem_dirty = true;
  return MHX_OK;
}

int mhx_expr_classify(const char* expr, const char* const* param_names, int n_names,
                      int32_t* model, int32_t* shape, int32_t* order, int32_t* n_order) {
  if (!expr || !model) return fail(MHX_EINVAL, "expr/model is NULL");
  if (n_names < 0 || n_names > MHX_MAX_FN_PARAMS || (n_names > 0 && !param_names))
    return fail(MHX_EINVAL, "n_names must be in [0,%d]", MHX_MAX_FN_PARAMS);
  std::vector<std::string> names;
  for (int j = 0; j < n_names; ++j) {
    if (!valid_ident(param_names[j]))
      return fail(MHX_EINVAL, "parameter name %d is not an identifier (or is x / bounds_total)", j);
    names.push_back(param_names[j]);
  }
  std::string out, err;
  if (rtc_prepare_expr(expr, names, "x xcol0 xcol1", &out, &err) != 0) return fail(MHX_EINVAL, "%s", err.c_str());
  RecognisedModel r;
  (void)rtc_recognise(expr, names, &r);
  *model = r.model >= 0 ? r.model : MHX_MODEL_EXPR;
  if (shape) {
    shape[0] = r.shape[0];
    shape[1] = r.shape[1];
  }
  if (n_order) *n_order = (int32_t)r.order.size();
  if (order)
    for (size_t j = 0; j < r.order.size(); ++j) order[j] = r.order[j];
  return MHX_OK;
}

int mhx_set_expr_recognition(mhx_engine* e, int on) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  e->recognise = on != 0;
  e->problem_dirty = true;
  return MHX_OK;
}

int mhx_set_prior_expr(mhx_engine* e, int k, const char* expr, const char* const* names,
                       const int32_t* index, int n) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "function index %d out of range", k);
  if (!expr || !*expr) {  // back to the plain bounds-total body
    e->prior_expr[k] = UserExpr();
    e->problem_dirty = true;
    return MHX_OK;
  }
  if (n < 0 || n > MHX_MAX_PARAMS || (n > 0 && (!names || !index)))
    return fail(MHX_EINVAL, "n must be in [0,%d]", MHX_MAX_PARAMS);
  UserExpr u;
  for (int j = 0; j < n; ++j) {
    if (!valid_ident(names[j])) return fail(MHX_EINVAL, "name %d is not an identifier", j);
    if (index[j] < 0 || index[j] >= e->P.d)
      return fail(MHX_EINVAL, "index[%d] = %d outside [0,%d)", j, index[j], e->P.d);
    u.names.push_back(names[j]);
    u.index.push_back(index[j]);
  }
  // the macro also binds <key>-bound, the penalty of that one key (M:356-360, docstring M:348):
  // identifier <name>_bound, encoded as index -(g+1)
  for (int j = 0; j < n; ++j) {
    u.names.push_back(std::string(names[j]) + "_bound");
    u.index.push_back(-(index[j] + 1));
  }
  std::string err;
  if (rtc_prepare_expr(expr, u.names, "bounds_total", &u.expr, &err) != 0)
    return fail(MHX_EINVAL, "%s", err.c_str());
  e->prior_expr[k] = u;
  e->problem_dirty = true;
  return MHX_OK;
}

int mhx_set_likelihood_expr(mhx_engine* e, int k, const char* expr) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (k < 0 || k >= e->P.K) return fail(MHX_EINVAL, "function index %d out of range", k);
  if (!expr || !*expr) return fail(MHX_EINVAL, "likelihood expression is empty");
  std::string out, err;
  if (rtc_prepare_expr(expr, std::vector<std::string>(), "y model error", &out, &err) != 0)
    return fail(MHX_EINVAL, "%s", err.c_str());
  e->lik_expr[k] = out;
  e->problem_dirty = true;
  return MHX_OK;
}

// in two halves, so that a group has every device working before it waits for the first
static int init_chains_enqueue(mhx_engine* e, const double* theta0, int broadcast) {
  if (!e || !theta0) return fail(MHX_EINVAL, "engine/theta0 is NULL");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if ((rc = finalize_problem(e)) != MHX_OK) return rc;
  const int64_t C = e->cfg.n_chains;
  const int d = e->P.d;
  if (broadcast) {
    std::vector<double> full((size_t)C * d);
    for (int64_t c = 0; c < C; ++c) memcpy(&full[(size_t)c * d], theta0, sizeof(double) * d);
    HIP_TRY(hipMemcpy(e->theta.p, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(e->theta.p, theta0, (size_t)C * d * sizeof(double), hipMemcpyHostToDevice));
  }
  e->chains_ready = false;
  e->run_ready = false;
  HIP_TRY(do_init(e));
  return MHX_OK;
}
static int init_chains_finish(mhx_engine* e) {
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches++;
  e->chains_ready = true;
  return MHX_OK;
}
int mhx_init_chains(mhx_engine* e, const double* theta0, int broadcast) {
  const int rc = init_chains_enqueue(e, theta0, broadcast);
  return rc != MHX_OK ? rc : init_chains_finish(e);
}

int mhx_logpost(mhx_engine* e, const double* theta, size_t n, double* out, double* parts) {
  if (!e || (n > 0 && (!theta || !out))) return fail(MHX_EINVAL, "NULL argument");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if ((rc = finalize_problem(e)) != MHX_OK) return rc;
  if (n == 0) return MHX_OK;
  const int d = e->P.d;
  const size_t o_th = 0, o_out = align256(n * d * sizeof(double)),
               o_parts = o_out + align256(n * sizeof(double));
  if ((rc = ensure_stage(e, o_parts + 2 * n * sizeof(double))) != MHX_OK) return rc;
  double* dth = reinterpret_cast<double*>(e->stage.p + o_th);
  double* dout = reinterpret_cast<double*>(e->stage.p + o_out);
  double* dparts = reinterpret_cast<double*>(e->stage.p + o_parts);
  HIP_TRY(hipMemcpy(dth, theta, n * d * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(do_logpost(e, dth, (int64_t)n, dout, dparts));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches++;
  HIP_TRY(hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
  if (parts) HIP_TRY(hipMemcpy(parts, dparts, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_step_injected(mhx_engine* e, const double* L, int per_chain_l, const double* z,
                      const double* u, const double* T, uint8_t* accepted_out) {
  if (!e || !L || !z || !u || !T) return fail(MHX_EINVAL, "NULL argument");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if ((rc = finalize_problem(e)) != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains, d = (size_t)e->P.d;
  const size_t nL = (per_chain_l ? C : 1) * d * d;
  const size_t o_z = align256(nL * sizeof(double)), o_u = o_z + align256(C * d * sizeof(double)),
               o_T = o_u + align256(C * sizeof(double)), o_acc = o_T + align256(C * sizeof(double));
  if ((rc = ensure_stage(e, o_acc + C)) != MHX_OK) return rc;
  double* dL = reinterpret_cast<double*>(e->stage.p);
  double* dz = reinterpret_cast<double*>(e->stage.p + o_z);
  double* du = reinterpret_cast<double*>(e->stage.p + o_u);
  double* dT = reinterpret_cast<double*>(e->stage.p + o_T);
  unsigned char* dacc = e->stage.p + o_acc;
  HIP_TRY(hipMemcpy(dL, L, nL * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dz, z, C * d * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(du, u, C * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dT, T, C * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dacc, 0, C));
  HIP_TRY(do_step_injected(e, dL, per_chain_l, dz, du, dT, dacc));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches++;
  if (accepted_out) HIP_TRY(hipMemcpy(accepted_out, dacc, C, hipMemcpyDeviceToHost));
  return MHX_OK;
}

void mhx_run_opts_default(mhx_run_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->n = 100000;         // (n 100000)        M:862
  o->temperature = 1e3;  // (temperature 1d3) M:862
  o->auto_mode = 1;      // (auto (or :prob-settle :slope-settle nil)) evaluates to :prob-settle
}

static int adaptive_begin_enqueue(mhx_engine* e, const mhx_run_opts* o) {
  if (!e || !o) return fail(MHX_EINVAL, "NULL argument");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (o->n < 0) return fail(MHX_EINVAL, "n must be >= 0");
  if (o->auto_mode != 0 && o->auto_mode != 1)
    return fail(MHX_EUNSUPPORTED, ":slope-settle is outside the accelerated path");
  if (!(o->temperature > 0.0) || !std::isfinite(o->temperature))
    return fail(MHX_EINVAL, "temperature must be finite and > 0");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if ((rc = finalize_problem(e)) != MHX_OK) return rc;
  drop_split_graph(e);  // the run description is an argument frozen into the captured launches
  e->S.slot_chain = nullptr;  // every chain walks again: slot s is chain s (compact_slots)
  e->S.n_slots = e->cfg.n_chains;
  e->slots_mapped = 0;
  e->split_iter = 0;
  e->ts_repack_due = false;
  if ((rc = reset_tsplit(e)) != MHX_OK) return rc;
  if ((rc = deal_initial(e)) != MHX_OK) return rc;
  const int d = e->P.d;
  RunDesc& R = e->R;
  R.n = o->n;                                       // (floor n) M:866
  R.sts = steps_to_settle_of(d);                    // M:873
  R.temp_steps = std::max<int64_t>(R.n, 10 * R.sts);  // M:875
  R.tail = std::max<int64_t>(2000, R.sts);          // (max 2000 steps-to-settle) M:906, M:917
  R.auto_mode = o->auto_mode;
  R.has_mwl = o->max_walker_length > 0;
  R.mwl = o->max_walker_length / 2;  // (floor max-walker-length 2) M:868
  R.adapt_mode = e->cfg.adapt_mode;
  R.stop_flag = e->stop_flag.p;
  if (R.has_mwl && R.mwl < 1) return fail(MHX_EINVAL, "max_walker_length must be >= 2");
  // temps M:878: the first window now, the rest as the loop gets there (ensure_temps)
  e->run_temperature = o->temperature;
  e->temps_count = 0;
  R.temps_first = 0;
  e->global_iter = 0;
  if ((rc = ensure_temps(e, 0, std::min<int64_t>(R.temp_steps, kTempsWindow))) != MHX_OK) return rc;
  HIP_TRY(hipMemset(e->stop_flag.p, 0, sizeof(int32_t)));  // (setf mfit-walker-estop nil) M:865
  const size_t C = (size_t)e->cfg.n_chains, dd = (size_t)d * d;
  if (o->l_matrix) {
    if (o->l_matrix_per_chain) {
      HIP_TRY(hipMemcpy(e->L.p, o->l_matrix, C * dd * sizeof(double), hipMemcpyHostToDevice));
    } else {
      std::vector<double> full(C * dd);
      for (size_t c = 0; c < C; ++c) memcpy(&full[c * dd], o->l_matrix, dd * sizeof(double));
      HIP_TRY(hipMemcpy(e->L.p, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  e->run_ready = false;
  HIP_TRY(hipMemsetAsync(e->pool_valid.p, 0, sizeof(int32_t), e->stream));
  HIP_TRY(e->fam->initial_l(e->stream, e->S, R, o->l_matrix ? 1 : 0, o->temperature));
  return MHX_OK;
}
static int adaptive_begin_finish(mhx_engine* e) {
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches++;
  e->run_ready = true;
  e->global_iter = 0;
  return MHX_OK;
}
int mhx_adaptive_begin(mhx_engine* e, const mhx_run_opts* o) {
  const int rc = adaptive_begin_enqueue(e, o);
  return rc != MHX_OK ? rc : adaptive_begin_finish(e);
}

// A failure between begin and the end of a run leaves host and device bookkeeping apart (some
// launches went out, others did not): the run is over, a new mhx_adaptive_begin starts the next.
static int run_failed(mhx_engine* e, int rc) {
  e->run_ready = false;
  return rc;
}

int mhx_adaptive_advance(mhx_engine* e, int64_t max_iters, int64_t* n_running) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->run_ready) return fail(MHX_ESTATE, "mhx_adaptive_begin has not been called");
  if (e->problem_dirty) {
    e->run_ready = false;
    return fail(MHX_ESTATE, "the problem was changed after mhx_adaptive_begin: begin again");
  }
  if (max_iters < 0) return fail(MHX_EINVAL, "max_iters < 0");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if (e->cfg.adapt_mode == MHX_ADAPT_POOLED) {
    // launches end on the pooled cadence (every 200 iterations since begin), where the
    // displacement statistics of all chains and ranks are combined
    int64_t left = max_iters;
    while (left > 0) {
      const int64_t to_tick = 200 - (e->global_iter % 200);
      const int64_t now = std::min(left, to_tick);
      if ((rc = launch_steps(e, now, 0)) != MHX_OK) return run_failed(e, rc);
      left -= now;
      if (e->global_iter % 200 == 0 && (rc = pool_refresh(e)) != MHX_OK) return run_failed(e, rc);
    }
  } else {
    // (one launch per call unless it would outrun the schedule window: then window by window,
    // looking at the chains in between - "until done" is asked for as 2^40 iterations)
    int64_t left = max_iters;
    while (left > 0) {
      const int64_t now = std::min(left, kTempsWindow);
      if ((rc = launch_steps(e, now, 0)) != MHX_OK) return run_failed(e, rc);
      left -= now;
      if (left > 0) {
        int64_t running = 0;
        if ((rc = count_running(e, &running)) != MHX_OK) return run_failed(e, rc);
        if (running == 0) break;
      }
    }
  }
  if (n_running && (rc = count_running(e, n_running)) != MHX_OK) return run_failed(e, rc);
  return MHX_OK;
}

int mhx_adaptive_steps_full(mhx_engine* e, const mhx_run_opts* o) {
  int rc = mhx_adaptive_begin(e, o);
  if (rc != MHX_OK) return rc;
  int64_t running = 1;
  while (running > 0) {
    if ((rc = mhx_adaptive_advance(e, e->chunk_iters, &running)) != MHX_OK) return rc;
  }
  return MHX_OK;
}

int mhx_adaptive_steps(mhx_engine* e, int64_t n) {
  mhx_run_opts o;
  mhx_run_opts_default(&o);
  o.n = n;               // (walker-adaptive-steps-full walker :n n :temperature 10
  o.temperature = 10.0;  //                              :auto :prob-settle) M:947
  o.auto_mode = 1;
  return mhx_adaptive_steps_full(e, &o);
}

static int plain_steps(mhx_engine* e, int64_t n, const double* L, int per_chain_l, double T);

int mhx_many_steps(mhx_engine* e, int64_t n, const double* L, int per_chain_l) {
  return plain_steps(e, n, L, per_chain_l, 1.0);
}

int mhx_take_step(mhx_engine* e, const double* L, int per_chain_l, double temperature) {
  if (!(temperature > 0.0) || !std::isfinite(temperature))
    return fail(MHX_EINVAL, "temperature must be finite and > 0");
  return plain_steps(e, 1, L, per_chain_l, temperature);
}

// (dotimes (i n) (walker-take-step w :l-matrix L :temperature T)) M:852-853, M:1072-1095
static int plain_steps(mhx_engine* e, int64_t n, const double* L, int per_chain_l, double T) {
  if (!e || !L)
    return fail(MHX_EINVAL, "NULL argument (the nil :l-matrix defaults of M:851 / M:1074 are "
                            "formed by the host shims from mhx_get_trace)");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (n < 0) return fail(MHX_EINVAL, "n < 0");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if ((rc = finalize_problem(e)) != MHX_OK) return rc;
  drop_split_graph(e);
  e->S.slot_chain = nullptr;
  e->S.n_slots = e->cfg.n_chains;
  e->slots_mapped = 0;
  e->split_iter = 0;
  e->ts_repack_due = false;
  if ((rc = reset_tsplit(e)) != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains, dd = (size_t)e->P.d * e->P.d;
  if (per_chain_l) {
    HIP_TRY(hipMemcpy(e->L.p, L, C * dd * sizeof(double), hipMemcpyHostToDevice));
  } else {
    std::vector<double> full(C * dd);
    for (size_t c = 0; c < C; ++c) memcpy(&full[c * dd], L, dd * sizeof(double));
    HIP_TRY(hipMemcpy(e->L.p, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // (dotimes (i n) (walker-take-step w :l-matrix L)) M:852-853: loop index 0..n-1, T = 1
  RunDesc& R = e->R;
  R.n = n;
  R.sts = steps_to_settle_of(e->P.d);
  R.temp_steps = 0;
  R.tail = 0;
  R.auto_mode = 0;
  R.has_mwl = 0;
  R.temps = nullptr;
  R.stop_flag = e->stop_flag.p;
  HIP_TRY(hipMemset(e->stop_flag.p, 0, sizeof(int32_t)));
  {
    std::vector<int64_t> zero(C, 0);
    std::vector<int32_t> st(C), run(C, MHX_CHAIN_RUNNING);
    std::vector<double> one(C, T);
    HIP_TRY(hipMemcpy(st.data(), e->status.p, C * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t c = 0; c < C; ++c)
      if (st[c] == MHX_CHAIN_FP_TRAP) run[c] = MHX_CHAIN_FP_TRAP;
    HIP_TRY(hipMemcpy(e->loop_i.p, zero.data(), C * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->status.p, run.data(), C * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->temperature.p, one.data(), C * sizeof(double), hipMemcpyHostToDevice));
  }
  int64_t running = 1;
  while (running > 0) {
    if ((rc = launch_steps(e, e->chunk_iters, 1)) != MHX_OK) return rc;
    if ((rc = count_running(e, &running)) != MHX_OK) return rc;
  }
  e->run_ready = false;
  return MHX_OK;
}

int mhx_request_stop(mhx_engine* e) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  static const int32_t one = 1;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->stop_flag.p, &one, sizeof one, hipMemcpyHostToDevice, e->stop_stream));
  HIP_TRY(hipStreamSynchronize(e->stop_stream));
  return MHX_OK;
}

int mhx_set_allreduce(mhx_engine* e, mhx_allreduce_fn fn, void* ctx, int wants_device_buffer) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  // a hook set after mhx_comm_init_rank replaces the library's own communicator (which would
  // otherwise take precedence in pool_refresh): the caller has decided that every rank exchanges
  // through the hook
  if (fn && e->comm) {
    if (e->comm_owned && rccl().ok) (void)rccl().CommDestroy(e->comm);
    e->comm = nullptr;
    e->comm_owned = false;
  }
  e->allreduce = fn;
  e->allreduce_ctx = ctx;
  e->allreduce_device = wants_device_buffer;
  return MHX_OK;
}

int mhx_get_state(mhx_engine* e, double* theta, double* logpost, double* best_theta,
                  double* best_logpost, int64_t* length, int64_t* age) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains, d = (size_t)e->P.d;
  if (theta) HIP_TRY(hipMemcpy(theta, e->theta.p, C * d * sizeof(double), hipMemcpyDeviceToHost));
  if (logpost) HIP_TRY(hipMemcpy(logpost, e->prob.p, C * sizeof(double), hipMemcpyDeviceToHost));
  if (best_theta)
    HIP_TRY(hipMemcpy(best_theta, e->best_theta.p, C * d * sizeof(double), hipMemcpyDeviceToHost));
  if (best_logpost)
    HIP_TRY(hipMemcpy(best_logpost, e->best_prob.p, C * sizeof(double), hipMemcpyDeviceToHost));
  if (length) HIP_TRY(hipMemcpy(length, e->length.p, C * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (age) HIP_TRY(hipMemcpy(age, e->age.p, C * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_chain(mhx_engine* e, int64_t chain, double* theta, double* logpost,
                  double* best_theta, double* best_logpost, int64_t* length, int64_t* age) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (chain < 0 || chain >= e->cfg.n_chains) return fail(MHX_EINVAL, "chain out of range");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t d = (size_t)e->P.d, c = (size_t)chain;
  if (theta) HIP_TRY(hipMemcpy(theta, e->theta.p + c * d, d * sizeof(double), hipMemcpyDeviceToHost));
  if (logpost) HIP_TRY(hipMemcpy(logpost, e->prob.p + c, sizeof(double), hipMemcpyDeviceToHost));
  if (best_theta)
    HIP_TRY(hipMemcpy(best_theta, e->best_theta.p + c * d, d * sizeof(double), hipMemcpyDeviceToHost));
  if (best_logpost)
    HIP_TRY(hipMemcpy(best_logpost, e->best_prob.p + c, sizeof(double), hipMemcpyDeviceToHost));
  if (length) HIP_TRY(hipMemcpy(length, e->length.p + c, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (age) HIP_TRY(hipMemcpy(age, e->age.p + c, sizeof(int64_t), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_chain_status(mhx_engine* e, int32_t* status, int64_t* loop_index) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains;
  if (status) HIP_TRY(hipMemcpy(status, e->status.p, C * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (loop_index)
    HIP_TRY(hipMemcpy(loop_index, e->loop_i.p, C * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_lmatrix(mhx_engine* e, double* L) {
  if (!e || !L) return fail(MHX_EINVAL, "NULL argument");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains, dd = (size_t)e->P.d * e->P.d;
  HIP_TRY(hipMemcpy(L, e->L.p, C * dd * sizeof(double), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_temperature(mhx_engine* e, double* T) {
  if (!e || !T) return fail(MHX_EINVAL, "NULL argument");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  HIP_TRY(hipMemcpy(T, e->temperature.p, (size_t)e->cfg.n_chains * sizeof(double),
                    hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_acceptance(mhx_engine* e, int take, double* out) {
  if (!e || !out) return fail(MHX_EINVAL, "NULL argument");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (take < 1) return fail(MHX_EINVAL, "take must be >= 1");
  if (take > e->S.R) return fail(MHX_EINVAL, "take %d exceeds history_capacity %d", take, e->S.R);
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t C = (size_t)e->cfg.n_chains;
  if ((rc = ensure_stage(e, C * sizeof(double))) != MHX_OK) return rc;
  double* d = reinterpret_cast<double*>(e->stage.p);
  HIP_TRY(e->fam->acceptance(e->stream, e->S, take, d));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(out, d, C * sizeof(double), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_get_trace(mhx_engine* e, int64_t chain, int take, double* prob, double* theta,
                  int* n_out) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (chain < 0 || chain >= e->cfg.n_chains) return fail(MHX_EINVAL, "chain out of range");
  if (take < 0) return fail(MHX_EINVAL, "take < 0");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const int Rcap = e->S.R, d = e->P.d;
  int64_t nh = 0, len = 0;
  HIP_TRY(hipMemcpy(&nh, e->n_hist.p + chain, sizeof nh, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&len, e->length.p + chain, sizeof len, hipMemcpyDeviceToHost));
  int64_t avail = std::min<int64_t>(std::min<int64_t>(len, nh), Rcap);
  int t = (int)std::min<int64_t>(avail, take);
  if (n_out) *n_out = t;
  if (t == 0) return MHX_OK;
  // the newest t entries are ring slots (nh - t) ... (nh - 1) modulo the capacity: one or two
  // contiguous runs - only those are copied, oldest first, and turned round on the host
  const int64_t lo = (nh - t) & (int64_t)(Rcap - 1);
  const int64_t run1 = std::min<int64_t>(t, Rcap - lo), run2 = t - run1;
  std::vector<double> hp((size_t)t), ht;
  const double* rp = e->hist_prob.p + chain * Rcap;
  HIP_TRY(hipMemcpy(hp.data(), rp + lo, (size_t)run1 * sizeof(double), hipMemcpyDeviceToHost));
  if (run2 > 0)
    HIP_TRY(hipMemcpy(hp.data() + run1, rp, (size_t)run2 * sizeof(double), hipMemcpyDeviceToHost));
  if (theta) {
    ht.resize((size_t)t * d);
    const double* rt = e->hist_theta.p + chain * Rcap * d;
    HIP_TRY(hipMemcpy(ht.data(), rt + lo * d, (size_t)run1 * d * sizeof(double), hipMemcpyDeviceToHost));
    if (run2 > 0)
      HIP_TRY(hipMemcpy(ht.data() + run1 * d, rt, (size_t)run2 * d * sizeof(double),
                        hipMemcpyDeviceToHost));
  }
  for (int s = 0; s < t; ++s) {  // s = 0 is the newest = the last one copied
    const size_t src = (size_t)(t - 1 - s);
    if (prob) prob[s] = hp[src];
    if (theta) memcpy(theta + (size_t)s * d, &ht[src * d], sizeof(double) * d);
  }
  return MHX_OK;
}

int mhx_get_proposal_factor(mhx_engine* e, int64_t chain, int take, double* L_out, int* status,
                            int* n_forward) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (chain < 0 || chain >= e->cfg.n_chains) return fail(MHX_EINVAL, "chain out of range");
  if (take < 1 || take > e->S.R)
    return fail(MHX_EINVAL, "take must be in [1, history_capacity = %d]", e->S.R);
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t dd = (size_t)e->P.d * e->P.d;
  const size_t o_out = align256(dd * sizeof(double)), o_fwd = o_out + align256(dd * sizeof(double)),
               o_info = o_fwd + align256((size_t)take * sizeof(int32_t));
  if ((rc = ensure_stage(e, o_info + 2 * sizeof(int32_t))) != MHX_OK) return rc;
  double* cov = reinterpret_cast<double*>(e->stage.p);
  double* out = reinterpret_cast<double*>(e->stage.p + o_out);
  int32_t* fwd = reinterpret_cast<int32_t*>(e->stage.p + o_fwd);
  int32_t* info = reinterpret_cast<int32_t*>(e->stage.p + o_info);
  HIP_TRY(hipMemsetAsync(e->stage.p, 0, o_info + 2 * sizeof(int32_t), e->stream));
  HIP_TRY(e->fam->l_matrix(e->stream, e->S, chain, take, fwd, cov, out, info));
  HIP_TRY(hipStreamSynchronize(e->stream));
  int32_t hinfo[2];
  HIP_TRY(hipMemcpy(hinfo, info, sizeof hinfo, hipMemcpyDeviceToHost));
  if (status) *status = hinfo[0];
  if (n_forward) *n_forward = hinfo[1];
  if (L_out) HIP_TRY(hipMemcpy(L_out, out, dd * sizeof(double), hipMemcpyDeviceToHost));
  return MHX_OK;
}

int mhx_set_history(mhx_engine* e, int64_t chain, const double* prob, const double* theta, int n) {
  if (!e || !prob || !theta) return fail(MHX_EINVAL, "NULL argument");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (chain < 0 || chain >= e->cfg.n_chains) return fail(MHX_EINVAL, "chain out of range");
  if (n < 1) return fail(MHX_EINVAL, "a walk has at least one step");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const int Rcap = e->S.R, d = e->P.d;
  const int keep = std::min(n, Rcap);  // the ring holds the newest `keep` steps
  std::vector<double> hp((size_t)Rcap, 0.0), ht((size_t)Rcap * d, 0.0);
  // chronological order: step s (0 = newest) of the input is entry keep-1-s of the ring
  for (int s = 0; s < keep; ++s) {
    const int slot = keep - 1 - s;
    hp[(size_t)slot] = prob[s];
    memcpy(&ht[(size_t)slot * d], theta + (size_t)s * d, sizeof(double) * d);
  }
  // most-likely-step as :add-step would have left it (strictly greater wins, oldest first)
  int best = n - 1;
  for (int s = n - 2; s >= 0; --s)
    if (prob[s] > prob[best]) best = s;
  const int64_t nh = keep, len = keep, age = n;
  HIP_TRY(hipMemcpy(e->hist_prob.p + chain * Rcap, hp.data(), hp.size() * sizeof(double),
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->hist_theta.p + chain * Rcap * d, ht.data(), ht.size() * sizeof(double),
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->n_hist.p + chain, &nh, sizeof nh, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->length.p + chain, &len, sizeof len, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->age.p + chain, &age, sizeof age, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->theta.p + chain * d, theta, sizeof(double) * d, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->prob.p + chain, prob, sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->best_theta.p + chain * d, theta + (size_t)best * d, sizeof(double) * d,
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->best_prob.p + chain, prob + best, sizeof(double), hipMemcpyHostToDevice));
  return MHX_OK;
}

int mhx_walker_modify(mhx_engine* e, int action, int64_t n) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (!e->chains_ready) return fail(MHX_ESTATE, "mhx_init_chains has not been called");
  if (action < MHX_MODIFY_BURN_WALKS || action > MHX_MODIFY_RESET_TO_MOST_LIKELY)
    return fail(MHX_EINVAL, "unknown walker-modify action %d", action);
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if (action <= MHX_MODIFY_KEEP_WALKS) {
    // (subseq walk 0 (- length burn-number)) / (subseq walk 0 keep-number), M:566-569: a bounding
    // index outside the list is an error in the reference; nothing is changed then
    std::vector<int64_t> len((size_t)e->cfg.n_chains);
    HIP_TRY(hipMemcpy(len.data(), e->length.p, len.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (int64_t v : len)
      if (n < 0 || n > v)
        return fail(MHX_EINVAL, "%s %lld is outside a walk of length %lld",
                    action == MHX_MODIFY_BURN_WALKS ? ":burn-number" : ":keep-number",
                    (long long)n, (long long)v);
  }
  HIP_TRY(e->fam->modify(e->stream, e->S, action, n));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->launches++;
  return MHX_OK;
}

int mhx_get_pooled(mhx_engine* e, double* stats, double* L_pool, int32_t* valid,
                   uint64_t* refreshes) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  const size_t d = (size_t)e->P.d, E = 1 + d + d * d;
  if (stats) HIP_TRY(hipMemcpy(stats, e->pool_vec.p, E * sizeof(double), hipMemcpyDeviceToHost));
  if (L_pool) HIP_TRY(hipMemcpy(L_pool, e->L_pool.p, d * d * sizeof(double), hipMemcpyDeviceToHost));
  if (valid) HIP_TRY(hipMemcpy(valid, e->pool_valid.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (refreshes) *refreshes = e->pool_refreshes;
  return MHX_OK;
}

int mhx_get_counters(mhx_engine* e, uint64_t* chain_steps, uint64_t* kernel_launches) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if (chain_steps) {
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, e->step_counter.p, sizeof v, hipMemcpyDeviceToHost));
    *chain_steps = v;
  }
  if (kernel_launches) *kernel_launches = e->launches;
  return MHX_OK;
}

const char* mhx_kernel_name(mhx_engine* e) {
  if (!e) {
    fail(MHX_EINVAL, "engine is NULL");
    return nullptr;
  }
  if (use_device(e) != MHX_OK || finalize_problem(e) != MHX_OK) return nullptr;
  return e->kernel_name.c_str();
}

// ---- native RCCL for one-process-per-GPU hosts -------------------------------------------------
int mhx_comm_get_unique_id(uint8_t id[128]) {
  if (!id) return fail(MHX_EINVAL, "id is NULL");
  Rccl& r = rccl();
  if (!r.ok) return fail(MHX_ECOMM, "RCCL unavailable: %s", g_rccl_why.c_str());
  nccl_unique_id u;
  const int rc = r.GetUniqueId(&u);
  if (rc != 0) return fail(MHX_ECOMM, "ncclGetUniqueId: %s", rccl_err(rc));
  memcpy(id, u.internal, 128);
  return MHX_OK;
}

int mhx_comm_init_rank(mhx_engine* e, const uint8_t id[128], int rank, int n_ranks) {
  if (!e || !id) return fail(MHX_EINVAL, "NULL argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MHX_EINVAL, "rank %d of %d", rank, n_ranks);
  Rccl& r = rccl();
  if (!r.ok) return fail(MHX_ECOMM, "RCCL unavailable: %s", g_rccl_why.c_str());
  int rc = use_device(e);
  if (rc != MHX_OK) return rc;
  if (e->comm && e->comm_owned) (void)r.CommDestroy(e->comm);
  e->comm = nullptr;
  nccl_unique_id u;
  memcpy(u.internal, id, 128);
  const int nr = r.CommInitRank(&e->comm, n_ranks, u, rank);
  if (nr != 0) {
    e->comm = nullptr;
    return fail(MHX_ECOMM, "ncclCommInitRank: %s", rccl_err(nr));
  }
  e->comm_owned = true;
  return MHX_OK;
}

int mhx_kernel_timing(mhx_engine* e, int reset, double* avg_ms, uint64_t* launches,
                      double* total_ms) {
  if (!e) return fail(MHX_EINVAL, "engine is NULL");
  if (avg_ms) *avg_ms = e->timed_launches ? e->kernel_ms / (double)e->timed_launches : 0.0;
  if (launches) *launches = e->timed_launches;
  if (total_ms) *total_ms = e->kernel_ms;
  if (reset) {
    e->kernel_ms = 0.0;
    e->timed_launches = 0;
  }
  return MHX_OK;
}


// ---- several GPUs under ONE host process (the reference's own way of running many walkers is a
// list mapped in one image, M:1029-1033): one engine + stream per device, contiguous global chain
// ranges, launches enqueued on every device before any is waited for, and for the pooled
// covariance ONE ncclAllReduce per tick issued on the engines' streams inside ncclGroupStart/End.
struct mhx_group {
  std::vector<mhx_engine*> eng;
  std::vector<int64_t> first, count;
  bool rccl_comms = false;    // distinct devices: communicators from ncclCommInitAll
  DevBuf<double> local_sum;   // same-device groups (tests on one GPU): device-side sum buffer
  int64_t global_iter = 0;
};

namespace {
int group_fail_cleanup(mhx_group* g, int rc) {
  const std::string keep = g_err;
  mhx_group_destroy(g);
  g_err = keep;
  return rc;
}
// the engines of a group that share ONE device cannot use RCCL (a communicator wants distinct
// devices): their pool vectors are summed by the host through pinned copies.  Only the
// single-GPU rehearsal of the group code path runs through here.
int group_pool_sum_local(mhx_group* g) {
  const mhx_engine* e0 = g->eng[0];
  const size_t E = 1 + (size_t)e0->P.d + (size_t)e0->P.d * e0->P.d;
  std::vector<double> sum(E, 0.0), h(E);
  for (mhx_engine* e : g->eng) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(h.data(), e->pool_vec.p, E * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < E; ++i) sum[i] += h[i];  // rank order: reproducible
  }
  for (mhx_engine* e : g->eng) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(e->pool_vec.p, sum.data(), E * sizeof(double), hipMemcpyHostToDevice));
  }
  return MHX_OK;
}
int group_pool_tick(mhx_group* g) {
  int rc = MHX_OK;
  for (mhx_engine* e : g->eng) {
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = pool_enqueue_stats(e)) != MHX_OK) return rc;
  }
  if (g->eng.size() > 1 || g->rccl_comms) {
    if (g->rccl_comms) {
      // one collective per device, issued by this one thread: inside ncclGroupStart/End, each
      // call with ITS device current.  Whatever fails in between, the group is closed again
      // (an open group would swallow every later nccl call of the process).
      Rccl& r = rccl();
      int nr = r.GroupStart();
      if (nr != 0) return fail(MHX_ECOMM, "ncclGroupStart: %s", rccl_err(nr));
      for (mhx_engine* e : g->eng) {
        if (hipSetDevice(e->device) != hipSuccess) {
          rc = fail(MHX_EDEVICE, "hipSetDevice(%d) failed inside the collective group", e->device);
          break;
        }
        if ((rc = pool_enqueue_allreduce(e)) != MHX_OK) break;
      }
      const std::string keep = g_err;
      nr = r.GroupEnd();
      if (rc != MHX_OK) {
        g_err = keep;  // the first failure is the one reported
        return rc;
      }
      if (nr != 0) return fail(MHX_ECOMM, "ncclGroupEnd: %s", rccl_err(nr));
    } else if ((rc = group_pool_sum_local(g)) != MHX_OK) {
      return rc;
    }
  }
  for (mhx_engine* e : g->eng) {
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = pool_enqueue_factor(e)) != MHX_OK) return rc;
  }
  return MHX_OK;
}
// (see run_failed: after a failure inside a run no engine of the group may be advanced again)
int group_run_failed(mhx_group* g, int rc) {
  const std::string keep = g_err;
  for (mhx_engine* e : g->eng) {
    if (e->launch_open) (void)launch_steps_finish(e);
    drain(e);
    e->run_ready = false;
  }
  g_err = keep;
  return rc;
}
}  // namespace

int mhx_group_partition(int64_t n_chains, int n_parts, int part, int64_t* first, int64_t* count) {
  if (n_chains < 0 || n_parts < 1 || part < 0 || part >= n_parts || !first || !count)
    return fail(MHX_EINVAL, "mhx_group_partition: bad argument");
  const int64_t base = n_chains / n_parts, rem = n_chains % n_parts;
  *count = base + (part < rem ? 1 : 0);
  *first = base * part + std::min<int64_t>(part, rem);
  return MHX_OK;
}

int mhx_group_create(const mhx_config* cfg, const int32_t* devices, int n_devices,
                     mhx_group** out) {
  if (!cfg || !devices || !out) return fail(MHX_EINVAL, "NULL argument");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return fail(MHX_EINVAL, "n_devices must be in [1,64]");
  if (cfg->n_chains < n_devices) return fail(MHX_EINVAL, "fewer chains than devices");
  mhx_group* g = new (std::nothrow) mhx_group();
  if (!g) return fail(MHX_ENOMEM, "host allocation failed");
  bool distinct = true;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  for (int i = 0; i < n_devices; ++i) {
    mhx_config c = *cfg;
    int64_t first = 0, count = 0;
    (void)mhx_group_partition(cfg->n_chains, n_devices, i, &first, &count);
    c.n_chains = count;
    c.chain_offset = cfg->chain_offset + first;
    c.device = devices[i];
    mhx_engine* e = nullptr;
    const int rc = mhx_create(&c, &e);
    if (rc != MHX_OK) return group_fail_cleanup(g, rc);
    g->eng.push_back(e);
    g->first.push_back(first);
    g->count.push_back(count);
  }
  // MHX_GROUP_FORCE_RCCL=1 (tests, -DMHX_DEBUG_HOOKS builds only): engines that share a device
  // take the communicator branch too - only a stand-in librccl accepts that (tests/stub_rccl);
  // the real one wants one device per rank, which is why such groups otherwise sum through the
  // host (group_pool_sum_local) - and so does a group of ONE device, which the real librccl
  // does accept: ncclCommInitAll of one communicator, ncclGroupStart / ncclAllReduce /
  // ncclGroupEnd per tick, the calls a node with several GPUs makes, on the one GPU of a test box
#ifdef MHX_DEBUG_HOOKS
  const char* fr = getenv("MHX_GROUP_FORCE_RCCL");
  const bool force_rccl = fr && atoi(fr) != 0;
#else
  const bool force_rccl = false;
#endif
  if ((n_devices > 1 || force_rccl) && cfg->adapt_mode == MHX_ADAPT_POOLED &&
      (distinct || force_rccl)) {
    Rccl& r = rccl();
    if (!r.ok) return group_fail_cleanup(g, fail(MHX_ECOMM, "RCCL unavailable: %s", g_rccl_why.c_str()));
    std::vector<nccl_comm_t> comms((size_t)n_devices, nullptr);
    std::vector<int> devs(devices, devices + n_devices);
    const int nr = r.CommInitAll(comms.data(), n_devices, devs.data());
    if (nr != 0)
      return group_fail_cleanup(g, fail(MHX_ECOMM, "ncclCommInitAll: %s",
                                        rccl_err(nr)));
    for (int i = 0; i < n_devices; ++i) {
      g->eng[(size_t)i]->comm = comms[(size_t)i];
      g->eng[(size_t)i]->comm_owned = true;
    }
    g->rccl_comms = true;
  }
  *out = g;
  return MHX_OK;
}

void mhx_group_destroy(mhx_group* g) {
  if (!g) return;
  for (mhx_engine* e : g->eng) mhx_destroy(e);
  delete g;
}

int mhx_group_size(const mhx_group* g) { return g ? (int)g->eng.size() : 0; }

mhx_engine* mhx_group_engine(mhx_group* g, int i) {
  if (!g || i < 0 || i >= (int)g->eng.size()) {
    fail(MHX_EINVAL, "group / index");
    return nullptr;
  }
  return g->eng[(size_t)i];
}

int mhx_group_chain_range(const mhx_group* g, int i, int64_t* first, int64_t* count) {
  if (!g || i < 0 || i >= (int)g->eng.size()) return fail(MHX_EINVAL, "group / index");
  if (first) *first = g->first[(size_t)i];
  if (count) *count = g->count[(size_t)i];
  return MHX_OK;
}

#define MHX_GROUP_EACH(call)                       \
  do {                                             \
    if (!g) return fail(MHX_EINVAL, "group is NULL"); \
    for (mhx_engine* e : g->eng) {                 \
      const int rc_ = (call);                      \
      if (rc_ != MHX_OK) return rc_;               \
    }                                              \
    return MHX_OK;                                 \
  } while (0)

int mhx_group_set_function(mhx_group* g, int k, int model_id, const int32_t* shape, int n_shape,
                           const int32_t* param_index, int n_index) {
  MHX_GROUP_EACH(mhx_set_function(e, k, model_id, shape, n_shape, param_index, n_index));
}
int mhx_group_set_dataset(mhx_group* g, int k, const double* x, const double* y,
                          const double* sigma, size_t n, int likelihood) {
  MHX_GROUP_EACH(mhx_set_dataset(e, k, x, y, sigma, n, likelihood));  // replicated on every GPU
}
int mhx_group_set_dataset_cols(mhx_group* g, int k, const double* const* xcols, int n_cols,
                               const double* y, const double* sigma, size_t n, int likelihood) {
  MHX_GROUP_EACH(mhx_set_dataset_cols(e, k, xcols, n_cols, y, sigma, n, likelihood));
}
int mhx_group_set_bounds(mhx_group* g, int k, const int32_t* idx, const double* lo,
                         const double* hi, int n) {
  MHX_GROUP_EACH(mhx_set_bounds(e, k, idx, lo, hi, n));
}
int mhx_group_set_function_expr(mhx_group* g, int k, const char* expr,
                                const char* const* param_names, const int32_t* param_index,
                                int n_index) {
  MHX_GROUP_EACH(mhx_set_function_expr(e, k, expr, param_names, param_index, n_index));
}
int mhx_group_set_expr_recognition(mhx_group* g, int on) {
  MHX_GROUP_EACH(mhx_set_expr_recognition(e, on));
}
int mhx_group_set_prior_expr(mhx_group* g, int k, const char* expr, const char* const* names,
                             const int32_t* index, int n) {
  MHX_GROUP_EACH(mhx_set_prior_expr(e, k, expr, names, index, n));
}
int mhx_group_set_likelihood_expr(mhx_group* g, int k, const char* expr) {
  MHX_GROUP_EACH(mhx_set_likelihood_expr(e, k, expr));
}
int mhx_group_request_stop(mhx_group* g) { MHX_GROUP_EACH(mhx_request_stop(e)); }

int mhx_group_init_chains(mhx_group* g, const double* theta0, int broadcast) {
  if (!g || !theta0) return fail(MHX_EINVAL, "NULL argument");
  // the first step of every chain (M:1148-1163) is a full log-posterior: enqueued on every
  // device before the first is waited for
  for (size_t i = 0; i < g->eng.size(); ++i) {
    mhx_engine* e = g->eng[i];
    const double* th = broadcast ? theta0 : theta0 + (size_t)g->first[i] * (size_t)e->P.d;
    const int rc = init_chains_enqueue(e, th, broadcast);
    if (rc != MHX_OK) {  // (what is already in the queues runs out; nobody is marked ready)
      for (size_t j = 0; j < i; ++j) drain(g->eng[j]);
      return rc;
    }
  }
  int rc = MHX_OK;
  for (mhx_engine* e : g->eng) {
    const int r = init_chains_finish(e);
    if (rc == MHX_OK) rc = r;
  }
  return rc;
}

int mhx_group_adaptive_begin(mhx_group* g, const mhx_run_opts* o) {
  if (!g || !o) return fail(MHX_EINVAL, "NULL argument");
  for (size_t i = 0; i < g->eng.size(); ++i) {
    mhx_run_opts oi = *o;
    if (o->l_matrix && o->l_matrix_per_chain)
      oi.l_matrix = o->l_matrix + (size_t)g->first[i] * (size_t)g->eng[i]->P.d * g->eng[i]->P.d;
    const int rc = adaptive_begin_enqueue(g->eng[i], &oi);
    if (rc != MHX_OK) {
      for (size_t j = 0; j < i; ++j) drain(g->eng[j]);
      return rc;
    }
  }
  int rc = MHX_OK;
  for (mhx_engine* e : g->eng) {
    const int r = adaptive_begin_finish(e);
    if (rc == MHX_OK) rc = r;
  }
  if (rc != MHX_OK)
    for (mhx_engine* e : g->eng) e->run_ready = false;
  g->global_iter = 0;
  return rc;
}

int mhx_group_adaptive_advance(mhx_group* g, int64_t max_iters, int64_t* n_running) {
  if (!g) return fail(MHX_EINVAL, "group is NULL");
  if (max_iters < 0) return fail(MHX_EINVAL, "max_iters < 0");
  for (mhx_engine* e : g->eng)
    if (!e->run_ready) return fail(MHX_ESTATE, "mhx_group_adaptive_begin has not been called");
  const bool pooled = g->eng[0]->cfg.adapt_mode == MHX_ADAPT_POOLED;
  int rc = MHX_OK;
  auto count_all = [&](int64_t* total) -> int {
    *total = 0;
    for (mhx_engine* e : g->eng) {
      int64_t r = 0;
      int rc2 = use_device(e);
      if (rc2 == MHX_OK) rc2 = count_running(e, &r);
      if (rc2 != MHX_OK) return rc2;
      *total += r;
    }
    return MHX_OK;
  };
  int64_t left = max_iters;
  while (left > 0) {
    const int64_t now = pooled ? std::min<int64_t>(left, 200 - (g->global_iter % 200))
                               : std::min<int64_t>(left, kTempsWindow);
    for (mhx_engine* e : g->eng) {  // every GPU gets its launch before any is waited for
      if ((rc = use_device(e)) != MHX_OK) return group_run_failed(g, rc);
      if ((rc = launch_steps_enqueue(e, now, 0)) != MHX_OK) return group_run_failed(g, rc);
    }
    for (mhx_engine* e : g->eng)
      if ((rc = launch_steps_finish(e)) != MHX_OK) return group_run_failed(g, rc);
    g->global_iter += now;
    left -= now;
    if (pooled && g->global_iter % 200 == 0 && (rc = group_pool_tick(g)) != MHX_OK)
      return group_run_failed(g, rc);
    if (left > 0 && (!pooled || g->global_iter % 200 == 0)) {  // "until done": look at the chains
      int64_t total = 0;
      if ((rc = count_all(&total)) != MHX_OK) return group_run_failed(g, rc);
      if (total == 0) break;
    }
  }
  if (n_running && (rc = count_all(n_running)) != MHX_OK) return group_run_failed(g, rc);
  return MHX_OK;
}

int mhx_group_adaptive_steps_full(mhx_group* g, const mhx_run_opts* o) {
  int rc = mhx_group_adaptive_begin(g, o);
  if (rc != MHX_OK) return rc;
  int64_t running = 1;
  while (running > 0) {
    int64_t chunk = 1 << 16;
    for (mhx_engine* e : g->eng) chunk = std::min(chunk, e->chunk_iters);
    if ((rc = mhx_group_adaptive_advance(g, chunk, &running)) != MHX_OK) return rc;
  }
  return MHX_OK;
}

int mhx_group_get_state(mhx_group* g, double* theta, double* logpost, double* best_theta,
                        double* best_logpost, int64_t* length, int64_t* age) {
  if (!g) return fail(MHX_EINVAL, "group is NULL");
  for (size_t i = 0; i < g->eng.size(); ++i) {
    mhx_engine* e = g->eng[i];
    const size_t f = (size_t)g->first[i], d = (size_t)e->P.d;
    const int rc = mhx_get_state(e, theta ? theta + f * d : nullptr, logpost ? logpost + f : nullptr,
                                 best_theta ? best_theta + f * d : nullptr,
                                 best_logpost ? best_logpost + f : nullptr,
                                 length ? length + f : nullptr, age ? age + f : nullptr);
    if (rc != MHX_OK) return rc;
  }
  return MHX_OK;
}

int mhx_group_get_counters(mhx_group* g, uint64_t* chain_steps, uint64_t* kernel_launches) {
  if (!g) return fail(MHX_EINVAL, "group is NULL");
  uint64_t cs = 0, kl = 0;
  for (mhx_engine* e : g->eng) {
    uint64_t a = 0, b = 0;
    const int rc = mhx_get_counters(e, &a, &b);
    if (rc != MHX_OK) return rc;
    cs += a;
    kl += b;
  }
  if (chain_steps) *chain_steps = cs;
  if (kernel_launches) *kernel_launches = kl;
  return MHX_OK;
}

}  // extern "C"
