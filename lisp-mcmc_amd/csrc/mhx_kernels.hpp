// mhx_kernels.hip -- hand-written gfx950 kernels of the walker-adaptive-steps path.
//
//   k_logpost        walker-make-step's prob for a batch of parameter vectors  (M:1067-1070)
//   k_init           first step of every chain                                 (M:1148-1163)
//   k_step_injected  one walker-take-step per chain with the caller's z, u, T  (M:1072-1095)
//   k_adaptive       the do loop of walker-adaptive-steps-full, fused          (M:902-942)
//   k_acceptance     (walker-get :get :acceptance :take n)                     (M:506-508)
//
// One wavefront = one chain.  The 64 lanes split the data points of the likelihood sum
// (M:400); the kWavesPerGroup chains of a workgroup stream the same dataset in lock step
// through double-buffered LDS tiles; the per-chain sum is a wavefront butterfly.  Per point the
// kernel does ~50 binary64 VALU operations on 24 B of data that is L2/MALL resident, so the
// path is fp64-VALU bound, not a contraction: no MFMA.
#pragma once
#ifndef __HIPCC_RTC__  // hiprtc pre-includes the runtime header
#include <hip/hip_runtime.h>
#endif

#include "mhx_device.hpp"

namespace mhx {
inline namespace MHX_FAMILY {

constexpr int kCurParams = 32;
struct GroupLds {
  // the tables of tlog() and mexp2_negsq(): 6 KiB.  FIRST member: the split-mode sweep kernel,
  // which stages no tiles, allocates only this much dynamic LDS (kSweepLdsBytes), and the device
  // math finds its tables at the same addresses in every kernel (mhx_device.hpp).
  LdsHead head;
  double tiles[2][kMaxArrays][kTilePoints];        // 64 KiB
  double prop[kWavesPerGroup][MHX_MAX_PARAMS];     // proposal theta' of each wave, 4 KiB
  double prm[kWavesPerGroup][MHX_MAX_FN_PARAMS + 4];
#ifdef MHX_X_TIMING  // (measurement build: cycles per phase of an iteration, see MHX_TIM)
  unsigned long long tim[kWavesPerGroup][8], tlast[kWavesPerGroup];
#endif
  int vote[4];  // "is any chain of the workgroup still running": three flags used in turn
  // dealing the proposals of a workgroup to its wave slots by estimated cost (group_logpost)
  int deal_cost[kWavesPerGroup];
  double deal_ll[kWavesPerGroup];
  int deal_skip;  // iterations the workgroup goes without looking at its costs (they were equal)
  double park[kWavesPerGroup][12];  // a ChainPark per wave: the chain's scalars during a sweep
  // the chain's position theta while a stepping kernel runs, for d <= kCurParams (beyond that it
  // is re-read from HBM): two L2 round trips less per iteration of a latency-bound single walker
  double cur[kWavesPerGroup][kCurParams];
  int resident;  // 1: tile 0 of the problem's only function sits in tiles[0] (FnDesc::solo)
};
// dynamic LDS of k_split_sweep: the tables and, per wave, the proposal it judges and the model's
// scratch (dynamic, not __shared__ arrays: static LDS would sit in front of the tables, which the
// device math reads at absolute addresses - mhx_device.hpp)
struct SweepLds {
  LdsHead head;
  double prop[kWavesPerGroup][MHX_MAX_PARAMS];
  double scr[kWavesPerGroup][MHX_MAX_FN_PARAMS + 4];  // (model_wants_scratch)
};
constexpr unsigned kSweepLdsBytes = sizeof(SweepLds);
static_assert(__builtin_offsetof(SweepLds, head) == 0, "the tables must lead the dynamic LDS");
// 160 KiB of LDS per CU: one workgroup of the 16-wave family, TWO of the 8-wave family
static_assert(sizeof(GroupLds) * (kWavesPerGroup <= 8 ? 2 : 1) <= 160 * 1024,
              "GroupLds no longer fits the workgroups per CU its kernel family counts on");
static_assert(__builtin_offsetof(GroupLds, head) == 0, "the tables must lead the dynamic LDS");

// every kernel that sweeps starts with this (LDS comes up uninitialised)
__device__ __forceinline__ void lds_tables_begin() {
  LdsHead& h = *reinterpret_cast<LdsHead*>(mhx_lds_raw);
  const int t = threadIdx.x;  // (every workgroup has >= 256 threads... or loops)
  // (slot s holds entry (s - 48) & 127: logtab_entry() indexes by the argument's own mantissa bits)
  for (int i = t; i < 256; i += blockDim.x) h.logtab[i] = kLogTab[((i >> 1) + 80) & 127][i & 1];
  for (int i = t; i < 512; i += blockDim.x) h.exp2tab[i >> 1][i & 1] = kExp2Tab[i >> 1][i & 1];
}
__device__ __forceinline__ void lds_begin(GroupLds& lds) {
  if (threadIdx.x == 0) lds.resident = 0;
  if (threadIdx.x == 0) lds.deal_skip = 0;
#ifdef MHX_EARLY_REJECT  // (a run-time compiled program only: see sweep())
  if (threadIdx.x < kWavesPerGroup) lds.deal_ll[threadIdx.x] = __builtin_inf();  // no threshold
  if (threadIdx.x < 2) lds.deal_cost[threadIdx.x] = 0;                           // "somebody is still in"
#endif
  if (threadIdx.x < 4) lds.vote[threadIdx.x] = 0;
  lds_tables_begin();
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// LDS tile pipeline + per-lane accumulation of one function's likelihood sum
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) double* lds_dptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// Measurement build (-DMHX_X_TIMING, tools/debug/phase_cycles.py): MHX_TIM(lds, k) books the
// cycles since the wave's previous mark on phase k; k_adaptive leaves the eight sums in the
// chain's most-likely-parameters row.  Not part of the product build.
#ifdef MHX_X_TIMING
#define MHX_TIM_(LDS, K)                                                     \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();            \
    if (lane_id() == 0) {                                                    \
      (LDS).tim[wave_in_group()][K] += now_ - (LDS).tlast[wave_in_group()];  \
      (LDS).tlast[wave_in_group()] = now_;                                   \
    }                                                                        \
  } while (0)
#endif
// -DMHX_X_TIMING=1: the phases of the sweep (MHX_TIM); =2: the steps of the controller (MHX_TIMC:
// 0 vote + shut-down test, 1 random numbers, 2 proposal, 3 the whole log-posterior, 4 accept,
// 5 :add-step, 6 annealing + adaptation test, 7 loop overhead)
#if defined(MHX_X_TIMING) && MHX_X_TIMING + 0 == 2
#define MHX_TIM(LDS, K) do { } while (0)
#define MHX_TIMC(LDS, K) MHX_TIM_(LDS, K)
#elif defined(MHX_X_TIMING)
#define MHX_TIM(LDS, K) MHX_TIM_(LDS, K)
#define MHX_TIMC(LDS, K) do { } while (0)
#else
#define MHX_TIM(LDS, K) do { } while (0)
#define MHX_TIMC(LDS, K) do { } while (0)
#endif

// Tile t of every array of function f -> LDS buffer `buf`, by LDS-DMA (global_load_lds_dwordx4:
// no VGPR staging; each wave instruction lands 64 x 16 B = 1 KiB contiguously at a
// wave-uniform LDS base).  Asynchronous: retired by the s_waitcnt vmcnt(0) the compiler puts in
// front of the next __syncthreads().
template <int NARR>
__device__ __forceinline__ void tile_dma(const FnDesc& f, int64_t t, GroupLds& lds, int buf,
                                         int w) {
  const int64_t base = t * kTilePoints + 2 * (int)threadIdx.x;
  const int wbase = 2 * kWave * w;  // first element this wave fills
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.x + base), (lds_ptr_t)&lds.tiles[buf][0][wbase],
                                   16, 0, 0);
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.y + base), (lds_ptr_t)&lds.tiles[buf][1][wbase],
                                   16, 0, 0);
  if constexpr (NARR > 2)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.w + base),
                                     (lds_ptr_t)&lds.tiles[buf][2][wbase], 16, 0, 0);
  if constexpr (NARR > 3)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.c + base),
                                     (lds_ptr_t)&lds.tiles[buf][3][wbase], 16, 0, 0);
}

// VALU issue on a SIMD goes to the wave of highest priority, then to the OLDEST: left alone, the
// four chains of a SIMD do not advance through a tile together - the oldest runs nearly
// unimpeded and reaches the tile's barrier after 45 % of the tile's time, the next ones after
// 56 / 70 / 85 % (measured, config 2: tools/debug/phase_cycles.py), so for a third of every tile
// only one or two waves are left to fill the SIMD's issue slots, which one or two dependent
// fp64 streams cannot.  Priority by PROGRESS evens that out: a wave enters a tile at priority
// 3 and drops one level per quarter of the tile, so whoever is behind goes first and the four
// reach the barrier within a quarter tile of one another.  (s_setprio takes an immediate: `it` is a constant
// after unrolling, `sec` is not.)
template <int NIT, int NIN>
__device__ __forceinline__ void tile_prio(int sec, int it) {
#ifndef MHX_NO_TILE_PRIO  // (build knob for A/B measurements)
#ifndef MHX_PRIO_LADDER
#define MHX_PRIO_LADDER 0
#endif
#if MHX_PRIO_LADDER
  // (levels after 1/2, 3/4 and 7/8 of the tile - a shorter last level: no gain on config 2 and
  // -5 % on config 3 against quarters, measured)
  constexpr int b1 = NIT / 2, b2 = NIT >= 4 ? 3 * NIT / 4 : NIT, b3 = NIT >= 8 ? 7 * NIT / 8 : NIT;
#else
  constexpr int b1 = NIT / 4, b2 = NIT / 2, b3 = 3 * NIT / 4;  // a level per quarter of the tile
#endif
  static_assert(NIT >= 2 && (NIN & (NIN - 1)) == 0 && NIT % NIN == 0, "sections of 2^k iterations");
  // iteration g = sec * NIN + it of the tile starts a level when g is one of 0, b1, b2, b3.  `it`
  // is a constant after unrolling and `sec` is not, so the test is written per SECTION: what is
  // left of it in the code is one scalar compare of `sec` in front of each s_setprio (the
  // arithmetic form, g == b, came out as a vector shift and a vector compare per boundary:
  // +1.3 % on config 2, same box).
#pragma unroll
  for (int s = 0; s < NIT / NIN; ++s) {
    const int g = s * NIN + it;
    if (g == 0) { if (sec == s) __builtin_amdgcn_s_setprio(3); }
    else if (g == b1) { if (sec == s) __builtin_amdgcn_s_setprio(2); }
    else if (g == b2) { if (sec == s) __builtin_amdgcn_s_setprio(1); }
    else if (g == b3) { if (sec == s) __builtin_amdgcn_s_setprio(0); }
  }
#endif
}

template <bool B>
struct BoolC {
  static constexpr bool value = B;
};
template <unsigned M>
struct CMask {
  constexpr operator unsigned() const { return M; }
};

// ---- "yw" tiles: steps in which nobody reads x ---------------------------------------------
// Once a walk has settled, every peak of every chain of a workgroup advances by the uniform-grid
// recurrence and the background with them (Prep::bgrec): such a sweep reads y/sigma and 1/sigma
// and, at the start of each 2048-point window, ONE x per lane (the seed).  The x tile - a third
// of the LDS-DMA traffic and of the tile buffers - is dead weight then.  When every running chain
// of the workgroup is in that state (one vote per sweep: sweep()), the same LDS holds two-array
// tiles of TWICE the points: half the tile barriers and pipeline restarts per sweep, a third less
// DMA; the seeds' x come straight from L2, fetched one tile ahead.  Per lane and window the
// operations and their order are those of sweep()'s all-recurrence variants (same seeds at the
// same points, same masks, even points to acc0 and odd to acc1): the SAME BITS whichever layout a
// step takes (tests/test_gpu_families.py::test_yw_tiles_give_the_same_bits; MHX_NO_YW=1 switches
// the layout off).
template <class Model, int LIK>
struct yw_capable {
  static constexpr bool value = false;
};
template <int NBG, int NPK, int LIK>
struct yw_capable<PeaksModel<NBG, NPK, false>, LIK> {
  typedef PeaksModel<NBG, NPK, false> M;
  static constexpr bool value = LIK == MHX_LIK_NORMAL && M::kHasSkip && M::kHasRec && NPK <= 2 &&
                                M::kSeedSteps == kPadPoints / kWave && MHX_PPI == 2;
};

template <class Model, int LIK>
__device__ __forceinline__ double sweep_yw(const FnDesc& f, const typename Model::Prep& prep,
                                           bool active, GroupLds& lds) {
  constexpr int P = 2;
  constexpr int T2 = 2 * kTilePoints;           // points of a yw tile: 4096 (w16), 2048 (w8)
  constexpr int WPT = T2 / kPadPoints;          // windows per yw tile: 2 (w16), 1 (w8)
  constexpr int NITW = kPadPoints / kWave / P;  // iterations of a lane per window
  constexpr int NIN = NITW > 8 ? 8 : NITW;      // ... in sections of at most 8 unrolled ones
  constexpr int NSEC = NITW / NIN;
  static_assert(WPT >= 1 && T2 % kPadPoints == 0 && NITW % NIN == 0, "whole windows per yw tile");
  static_assert(2 * 2 * T2 <= 2 * kMaxArrays * kTilePoints, "the yw tiles live in GroupLds::tiles");
  constexpr unsigned kAll = (1u << Model::kPeaks) - 1u;
  const int l = lane_id();
  const int w = wave_in_group();
  const int64_t nw = (f.n + kPadPoints - 1) / kPadPoints;  // windows of the dataset (>= 1)
  const int64_t nt2 = (nw + WPT - 1) / WPT;
  double* const base = &lds.tiles[0][0][0];  // buffer b: y at base + 2 b T2, 1/sigma at + T2
  // yw tile j -> buffer b: two LDS-DMA instructions per array and thread (16 B each); the arrays
  // are padded to whole windows, so the second half of the last tile may not exist
  auto dma = [&](int64_t j, int b) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t p0 = j * T2 + (int64_t)q * kTilePoints;
      if (p0 < nw * kPadPoints) {  // uniform
        const int64_t g = p0 + 2 * (int)threadIdx.x;
        const int dst = q * kTilePoints + 2 * kWave * w;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.y + g),
                                         (lds_ptr_t)(base + (2 * b) * T2 + dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(f.w + g),
                                         (lds_ptr_t)(base + (2 * b + 1) * T2 + dst), 16, 0, 0);
      }
    }
  };
  // the lane's x at the first point of each window of tile j (what the recurrence is seeded from)
  auto load_xs = [&](int64_t j, double (&xs)[WPT]) {
#pragma unroll
    for (int q = 0; q < WPT; ++q) {
      const int64_t wi = j * WPT + q;
      xs[q] = f.x[(wi < nw ? wi : nw - 1) * kPadPoints + l];
    }
  };
  double acc0 = 0.0, acc1 = 0.0;
  double xs[WPT], xs_next[WPT];
  unsigned tile_masks = ~0u;
  dma(0, 0);
  load_xs(0, xs);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) lds.vote[3] = 0;  // (the vote of sweep(): everybody has read it)
  for (int64_t j = 0; j < nt2; ++j) {
    const int b = (int)(j & 1);
    if (j + 1 < nt2) {
      dma(j + 1, b ^ 1);  // (that buffer was last read in tile j - 1: everybody left it through the barrier)
      load_xs(j + 1, xs_next);
    }
    if (active) {
#pragma unroll
      for (int q = 0; q < WPT; ++q) {
        const int64_t wi = j * WPT + q;
        if (wi >= nw) break;
        if ((wi & 63) == 0) {  // the masks of 64 windows at once, lane i taking window wi + i
          const int64_t ti = wi + l < nw ? wi + l : nw - 1;
          tile_masks = Model::tile_mask(prep, f.txlo[ti], f.txhi[ti]);
        }
        // (no guarded window can occur: the vote asked for Prep::fast, |t| < kFastT over the
        // whole x range, of which every window's range is a part)
        const unsigned tm =
            (unsigned)__builtin_amdgcn_readlane((int)tile_masks, (int)(wi & 63)) & kAll;
        lds_cdptr_t py[P], pw[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {  // (opaque: see tile_work in sweep())
          py[i] = opaque_lds((lds_cdptr_t)(base + (2 * b) * T2 + q * kPadPoints) + i * kWave + l);
          pw[i] = opaque_lds((lds_cdptr_t)(base + (2 * b + 1) * T2 + q * kPadPoints) + i * kWave + l);
        }
        const int64_t left = f.n - wi * (int64_t)kPadPoints;
        const int nv = (int)(left < (int64_t)kPadPoints ? left : (int64_t)kPadPoints);
        const double x0 = xs[q];
        auto window = [&](auto mk) {
          const unsigned mask = mk;
          typename Model::Rec rs;
#pragma unroll 1
          for (int sec = 0; sec < NSEC; ++sec) {
            const int sb = sec * NIN * P * kWave;  // first point of the section in the window
            if (sb >= nv) break;
            double y[P], wv[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
              y[i] = py[i][sb];
              wv[i] = pw[i][sb];
            }
#pragma unroll
            for (int it = 0; it < NIN; ++it) {
              if (sb + it * P * kWave >= nv) break;  // (uniform; it also keeps the iterations apart)
              tile_prio<WPT * NITW, NIN>(q * NSEC + sec, it);
              double yn[P], wn[P];
#pragma unroll
              for (int i = 0; i < P; ++i) {
                yn[i] = wn[i] = 0.0;
                if (it + 1 < NIN) {
                  yn[i] = py[i][sb + (it + 1) * P * kWave];
                  wn[i] = pw[i][sb + (it + 1) * P * kWave];
                }
              }
              __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the arithmetic below
              if (it == 0 && sec == 0) {  // seeded at the lane's first point of the window
                if (mask != 0u) Model::rec_seed(prep, x0, mask, rs);
                Model::rec_seed_bg(prep, x0, rs);
              }
              const double xu[P] = {x0, x0};  // (not read: background and peaks by recurrence)
              double m[P];
              Model::template eval_mixed<P, true>(prep, xu, mask, mask, rs, m);
#pragma unroll
              for (int i = 0; i < P; ++i) {
                double& acc = (i & 1) ? acc1 : acc0;
                const double r = __builtin_fma(-m[i], wv[i], y[i]);
                acc = __builtin_fma(r, r, acc);
              }
#pragma unroll
              for (int i = 0; i < P; ++i) {
                y[i] = yn[i];
                wv[i] = wn[i];
              }
            }
          }
        };
        if constexpr (Model::kPeaks == 1) {
          if (tm & 1u) window(CMask<1u>{}); else window(CMask<0u>{});
        } else {
          switch (tm) {
            case 0u: window(CMask<0u>{}); break;
            case 1u: window(CMask<1u>{}); break;
            case 2u: window(CMask<2u>{}); break;
            default: window(CMask<3u>{}); break;
          }
        }
      }
    }
#ifndef MHX_NO_TILE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile j + 1 has landed
    __syncthreads();                                   // ... and so has everybody else's
#pragma unroll
    for (int q = 0; q < WPT; ++q) xs[q] = xs_next[q];
  }
  return wave_sum(acc0 + acc1);
}

// Sum over the points of function f.  Collective over the workgroup (barriers inside);
// waves with active == false only help to move tiles.  fast (wave-uniform): this chain's
// parameters satisfy the model's fast-path precondition (model_has_fast) - decided per WAVE:
// the tile pipeline and its barriers are common to both paths, only the arithmetic of the
// wave's own points differs, so a chain's bits never depend on which chains share its workgroup.
// rmask (wave-uniform, 0 unless fast): the peaks that go by the uniform-grid recurrence this
// step (PeaksModel, model_has_rec).
template <bool B, class T, class F>
struct PickType {
  typedef T type;
};
template <class T, class F>
struct PickType<false, T, F> {
  typedef F type;
};

// WG: per-window grids (PeaksModel::regrid).  On a dataset that is not ONE grid every window
// brings its own H = 64 h (FnDesc::tgh; 0: not a grid - that window takes the direct form), and
// the grid constants of this sweep's copy of the chain's Prep follow it.  A compile-time choice
// of the KERNEL (FixedSpec<Model, LIK, true>, which only run-time compiled kernels use: the
// engine sends a problem with a per-window table to hiprtc, mhx_engine.cpp): with the constants
// loop-carried the register allocator does visibly worse on the models of many peaks - as a
// run-time flag of one instance config 3's kernel went from 36 to 266 scratch accesses in its
// hot loops, as a second instance inside the same kernel from 74.8 to 104.5 ms per 20
// iterations and config 2's by 3.5 % (same box) - and the datasets of one grid, or none, must
// not pay for it.
template <class Model, int LIK, bool WG = false>
__device__ __forceinline__ double sweep(const FnDesc& f, const typename Model::Prep& prep_in,
                                        bool active, GroupLds& lds, bool fast = false,
                                        unsigned rmask = 0u, bool bgrec = false) {
  // a second column of x (model_xcols: expression models that name xcol1) rides in the tiles' fourth
  // array, which only the cutoff likelihood uses otherwise (the host refuses that pair)
  constexpr bool kX2 = model_xcols<Model>::value > 1;
  static_assert(!(kX2 && LIK == MHX_LIK_NORMAL_CUTOFF), "the fourth tile array is taken");
  constexpr int NARR = kX2 ? 4 : (LIK == MHX_LIK_POISSON ? 2 : (LIK == MHX_LIK_NORMAL_CUTOFF ? 4 : 3));
  constexpr bool kWinGrid = WG;
  static_assert(!WG || (model_has_rec<Model>::value && model_has_skip<Model>::value), "regrid");
  typename PickType<WG, typename Model::Prep, const typename Model::Prep&>::type prep = prep_in;
  constexpr bool wgrid = WG;
  double win_h = 0.0;   // lane i: H of window (wi & ~63) + i
  double h_cur = 0.0;   // the H `prep` was last re-gridded for (0: prepare()'s, i.e. none)
  (void)wgrid; (void)win_h; (void)h_cur;
  static_assert(kTilePoints == 2 * kThreads, "one double2 per thread per array per tile");
  const int l = lane_id();
  const int w = wave_in_group();
  const int64_t nt = f.n_tiles;
  double acc0 = 0.0, acc1 = 0.0;
#ifdef MHX_EARLY_REJECT
  // EXACT EARLY REJECTION (programs compiled at run time with -DMHX_EARLY_REJECT: the engine asks
  // for it under MHX_EARLY_REJECT=1 for one function of a bounded enumerated model with the
  // weighted normal likelihood and no prior body).  The sum of squares only grows - every lane's
  // accumulator and the butterfly are monotone in floating point too - and the accept test is a
  // threshold on it that the controller knows before the sweep (lds.deal_ll[w]; +inf: none).
  // After windows 0, 1, 3, 7, 15, 31, 63 a wave whose partial sum is over it leaves the sweep
  // (it goes on as a wave without a running chain: barriers and DMA only) and says so
  // (deal_ll[w] = -inf); when no wave of the workgroup is left the sweep ends.  A walk's first
  // iterations (T = 10, diag(theta) steps) accept nothing, and lose within one to eight windows.
  bool er_lost = false;
  int er_i = 0;
#endif
  // (Poisson) a constant of tlog() pinned in a VGPR - and only there: the pin cannot be optimised
  // away, and the other kernels have better uses for the register
  const double log_a3 = LIK == MHX_LIK_POISSON ? tlog_a3() : kTlogA3;
  (void)log_a3;
  // Tile-level peak skipping and the Gaussian recurrence work on WINDOWS of kPadPoints = 2048
  // points - one tile of the 16-wave family, two of the 8-wave family: the mask of peaks is
  // formed per window and the recurrence seeded at its start, so both families evaluate the same
  // expressions in the same order and give the same bits.
  constexpr int kTPW = kPadPoints / kTilePoints;
  const int64_t nw = (nt + kTPW - 1) / kTPW;
  unsigned tile_masks = ~0u;  // lane i: which peaks window (wi & ~63) + i needs (PeaksModel::tile_mask)
  // (REC) the peaks' running g, r live from one seed to the next: across the sections of a
  // window, or within a section
  typedef typename model_rec_state<Model>::type RecState;
  constexpr int kSeedPts = model_seed_steps<Model>::value;  // points of a lane from seed to seed
  static_assert(kSeedPts % 16 == 0 && (kPadPoints / kWave) % kSeedPts == 0, "seeds per window");
  (void)nw;
  if (nt == 0) return 0.0;
  // (Also: a sweep workgroup of the tile-sliced persistent kernel whose slice is ONE window walks
  // the same two tiles round after round - buffer 0 and buffer 1 - and takes them from memory in
  // its first round only: build_ts_table.)
  // A problem with ONE function of ONE tile (test.lisp's 334 points) walked by a single
  // workgroup keeps that tile in LDS for the whole launch: after the first sweep there is no DMA
  // and no barrier left in here, which takes the L2 round trip out of every step of a
  // latency-bound walk (+9 % on a single chain; with hundreds of chains it measured slower, so
  // the engine sets FnDesc::solo only there).
  const bool solo = f.solo != 0;
  // Which tile layout this sweep takes is a decision of the WORKGROUP (the tiles are shared): the
  // yw tiles when no running chain needs x beyond its seeds.  A wave that needs x raises a flag,
  // one barrier, everybody reads it; the flag is lowered again behind the first tile barrier
  // (every reader is past it by then, and the next vote only comes after this sweep's last one).
  if constexpr (yw_capable<Model, LIK>::value) {
    if (!solo && f.no_yw == 0) {
      constexpr unsigned kAllPeaks = (1u << model_peaks<Model>::value) - 1u;
      bool mine = !active || (fast && bgrec && rmask == kAllPeaks);
      mine = mine && (!active || Model::seed16(prep) == 0u);  // (seeds inside a window need x)
      if (!mine && lane_id() == 0) lds.vote[3] = 1;
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(*(volatile int*)&lds.vote[3]) == 0)
        return sweep_yw<Model, LIK>(f, prep, active, lds);
    }
  }
  unsigned seed16 = 0u, seed8 = 0u;  // (REC) the peaks re-seeded inside a window
  if constexpr (model_has_rec<Model>::value) {
    seed16 = Model::seed16(prep);
    seed8 = Model::seed8(prep);
  }
  (void)seed16;
  (void)seed8;
  const bool have = solo && __builtin_amdgcn_readfirstlane(lds.resident) != 0;
  // (every wave must have looked at the flag before the first one through the block below sets
  // it: a wave that arrived late and read 1 would skip the block - its part of the DMA and the
  // barrier - and sum over a tile that is not there yet)
  if (solo && !have) __syncthreads();
  MHX_TIM(lds, 2);
  if (!have) {
    tile_dma<NARR>(f, 0, lds, 0, w);
    // An LDS-DMA is ordered for the readers only by the ISSUING wave's vmcnt wait followed by a
    // barrier: written out here, never left to the compiler's handling of __syncthreads().
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (solo && threadIdx.x == 0) lds.resident = 1;
    __syncthreads();
    if constexpr (yw_capable<Model, LIK>::value) {
      if (threadIdx.x == 0) lds.vote[3] = 0;  // (the layout vote above)
    }
  }
  MHX_TIM(lds, 3);
  // Between the two tiles of a window (8-wave family): everybody is through with tile `tcur`
  // and the next one has landed; the buffer just read takes the tile after that.
  auto mid_hook = [&](int64_t tcur, int bufc) {
#ifndef MHX_NO_TILE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (have) return;  // (a resident window: both its tiles are where an earlier sweep put them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tcur + 2 < nt) tile_dma<NARR>(f, tcur + 2, lds, bufc, w);
  };
  // One pass of this loop is one WINDOW: a tile of the 16-wave family, two tiles of the 8-wave
  // family (which then always finds the window's first tile in buffer 0, its second in buffer 1).
  for (int64_t t = 0; t < nt; t += kTPW) {
    const int buf = (int)(t & 1);
    const int64_t wi = t / kTPW;  // the window
    constexpr bool wstart = true;
    (void)wi;
    // buffer buf^1 was last read in tile t-1, which every wave left through the barrier; (two
    // tiles per window: tile t itself was sent for by the previous window's mid_hook)
    if (t + 1 < nt && !have) tile_dma<NARR>(f, t + 1, lds, buf ^ 1, w);
    // Which Gaussian peaks this window needs (those that cannot change any of its sums by even
    // one bit are left out: PeaksModel::tile_mask; exact, so the results do not depend on it),
    // and whether the window can take the table exp: decided per WINDOW for such models, so one
    // very narrow proposed peak costs the guarded form where it sits, not over the whole dataset.
    unsigned tm = ~0u;
    bool fastw = fast;
    if constexpr (model_has_skip<Model>::value && model_has_fast<Model>::value) {
      // the words of 64 consecutive windows are worked out at once, lane i taking window wi + i
      if (wstart && (wi & 63) == 0) {
        const int64_t ti = wi + l < nw ? wi + l : nw - 1;
        tile_masks = Model::tile_mask(prep, f.txlo[ti], f.txhi[ti]);
      }
      const unsigned tw = (unsigned)__builtin_amdgcn_readlane((int)tile_masks, (int)(wi & 63));
      tm = tw & ~Model::kGuardBit;
      fastw = (tw & Model::kGuardBit) == 0u;
    }
    (void)tm;
    if constexpr (kWinGrid) {
      if (wgrid) {
        if ((wi & 63) == 0) {
          const int64_t ti = wi + l < nw ? wi + l : nw - 1;
          win_h = f.tgh[ti];
        }
        const double hw = readlane_f64(win_h, (int)(wi & 63));
        if (hw != 0.0) {
          if (__double_as_longlong(hw) != __double_as_longlong(h_cur)) {  // (uniform; rare: runs of windows share their H)
            Model::regrid(prep, hw);
            h_cur = hw;
          }
          rmask = Model::rec_mask(prep);
          bgrec = __builtin_amdgcn_readfirstlane((int)Model::rec_bg(prep)) != 0;
          seed16 = Model::seed16(prep);
          seed8 = Model::seed8(prep);
        } else {  // not a grid: every evaluated peak directly
          rmask = 0u;
          bgrec = false;
          seed16 = seed8 = 0u;
        }
      }
    }
    auto tile_work = [&](auto fastc) {
      constexpr bool FAST = decltype(fastc)::value;
      // The lane's element of tile point 0 ... P-1 of each array, as LDS addresses the compiler
      // cannot relate to one another (opaque_lds): left to itself it fuses the reads of two
      // successive points into ds_read2st64_b64, which moves 1 KiB in 8 LDS cycles where two
      // ds_read_b64 take 4 (MI355X_MICROARCH.md, LDS table) - with the Gaussians down to three
      // instructions per point the kernel would be LDS-bound (measured: SQ_WAIT_INST_LDS 14 %
      // of the wave cycles, VALU 58 % busy).
      constexpr int PA = MHX_PPI;  // (the 4-point masked loops keep the fused reads)
      lds_cdptr_t ax[PA], ay[PA], aw[PA], ac[PA];
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        ax[i] = opaque_lds((lds_cdptr_t)lds.tiles[buf][0] + i * kWave + l);
        ay[i] = opaque_lds((lds_cdptr_t)lds.tiles[buf][1] + i * kWave + l);
        aw[i] = opaque_lds((lds_cdptr_t)lds.tiles[buf][2] + i * kWave + l);
        ac[i] = opaque_lds((lds_cdptr_t)lds.tiles[buf][3] + i * kWave + l);
      }
      const double* tx = lds.tiles[buf][0];
      const double* ty = lds.tiles[buf][1];
      const double* tw = lds.tiles[buf][2];
      const double* tc = lds.tiles[buf][3];
      const int64_t gbase = t * kTilePoints;
      // One tile.  `mk` is the set of Gaussian peaks to evaluate: a run-time value, or a CMask<M>
      // whose value is a compile-time constant once the lambda is inlined - the per-peak tests
      // in PeaksModel::eval then fold away and each variant is straight-line code.
      auto tile_body_m = [&](auto mk, auto rk, auto whole, auto bgc) {
        const unsigned mask = mk;
        const unsigned rm = rk;  // the peaks of `mask` that go by the recurrence
        constexpr bool BGREC = decltype(bgc)::value;  // ... and the background with them
        // whole tile = every point is data: the pad test of the Poisson / expression
        // likelihoods (compare + select per point) is only compiled into the ragged variant
        constexpr bool kWhole = decltype(whole)::value;
        // P points per lane and iteration: point i of iteration it is element (it*P + i)*64 + l
        // of the tile.  Software-pipelined: the LDS reads of the next P points are issued before
        // the dependent fp64 chains of the current ones, so no wave waits on lgkmcnt.  P = 2
        // everywhere (a short dataset then evaluates the fewest pads: test.lisp's 334 points
        // cost 6 evaluations per lane with P = 2, 8 with P = 4) except where every peak is a basic
        // block of its own (run-time mask, more than 2 peaks): there 4 points give the block four
        // independent fp64 chains (+5 % on BASELINE config 3).
        constexpr int P = (FAST && model_has_skip<Model>::value && model_peaks<Model>::value > 2)
                              ? MHX_PPI_MASKED : MHX_PPI;
        // a tile is worked through in sections of at most 8 fully unrolled iterations (more than
        // that is beyond what the unroller accepts, and a rolled loop would lose the pipeline):
        // one section per tile in the 8-wave family, one or two in the 16-wave family
        constexpr int NIT = kTilePoints / kWave / P;
#ifndef MHX_NIN_MASKED
// unrolled iterations per section of the 4-point (run-time mask) loops: 8 = a whole 2048-point
// tile.  (Round 2: 4, 8 measured 4 % slower on config 3 - code size; since the instruction-level
// pass of round 3 took a third off those loops' code, 8 is 1.9 % faster - 73.2 against 74.6 ms -
// and 2 is 4 % slower; g23, compiled at run time with 2-point loops: no difference.)
#define MHX_NIN_MASKED 8
#endif
        constexpr int kMaxIn = (P == MHX_PPI) ? 8 : MHX_NIN_MASKED;
        constexpr int NIN = NIT > kMaxIn ? kMaxIn : NIT;
        constexpr int NSEC = NIT / NIN;
        static_assert(kTilePoints % (kWave * P) == 0 && NIT % NIN == 0, "whole sections per tile");
        // points of this tile that are data (the rest are neutral pads): short datasets such as
        // test.lisp's 334 points leave most of their only tile unused
        constexpr int kWinPoints = kTPW * kTilePoints;
        const int nv = (int)((f.n - gbase) < (int64_t)kWinPoints ? (f.n - gbase) : (int64_t)kWinPoints);
        RecState rs_tile;
        (void)rs_tile;
        // the sections of the window: NSEC per tile; the second tile of a window sits one buffer on
        constexpr int kBufStride = kMaxArrays * kTilePoints;
#pragma unroll 1
        for (int ws = 0; ws < kTPW * NSEC; ++ws) {
          const int tt = ws / NSEC, sec = ws % NSEC;
          if (kTPW > 1 && tt > 0 && sec == 0) {
            if (t + tt >= nt) break;
            mid_hook(t + tt - 1, buf ^ (tt - 1));
          }
          const int wb = ws * NIN * P * kWave;  // first point of this section in the window
          const int sbase = tt * kBufStride + sec * NIN * P * kWave;  // ... and in the buffers
          if (wb >= nv) break;
          RecState rs_sec;
          (void)rs_sec;
          RecState& rs = *(kSeedPts > NIN * P ? &rs_tile : &rs_sec);
          // (each section starts its own read pipeline: carrying the registers of the next points
          // over the section loop's back edge - one uncovered LDS read per tile less - measured
          // 14 % SLOWER on config 2 and 23 % on config 3)
          double x[P], y[P], wv[P], cv[P];
#pragma unroll
          for (int i = 0; i < P; ++i) {
            wv[i] = 0.0;
            cv[i] = 0.0;
            if constexpr (P == PA) {
              x[i] = ax[i][sbase];
              y[i] = ay[i][sbase];
              if constexpr (NARR > 2) wv[i] = aw[i][sbase];
              if constexpr (NARR > 3) cv[i] = ac[i][sbase];
            } else {
              x[i] = tx[sbase + i * kWave + l];
              y[i] = ty[sbase + i * kWave + l];
              if constexpr (NARR > 2) wv[i] = tw[sbase + i * kWave + l];
              if constexpr (NARR > 3) cv[i] = tc[sbase + i * kWave + l];
            }
          }
#pragma unroll
          for (int it = 0; it < NIN; ++it) {
            // uniform: one scalar compare per P points.  (Round 3: compiled out for datasets of
            // whole windows - neutral / masked pads make it unnecessary there - the all-recurrence
            // variants ran 1.5 % SLOWER and the direct-form variants spilled 624 B per lane: the
            // branch is what keeps the unrolled iterations apart for the register allocator, and
            // with four waves per SIMD the scalar instructions ride in issue slots the vector
            // pipe leaves free.  It stays.)
            if (wb + it * P * kWave >= nv) break;
            tile_prio<NIT, NIN>(sec, it);
            double xn[P], yn[P], wn[P], cn[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
              xn[i] = yn[i] = wn[i] = cn[i] = 0.0;
              if (it + 1 < NIN) {
                if constexpr (P == PA) {
                  const int j = sbase + (it + 1) * P * kWave;
                  xn[i] = ax[i][j];
                  yn[i] = ay[i][j];
                  if constexpr (NARR > 2) wn[i] = aw[i][j];
                  if constexpr (NARR > 3) cn[i] = ac[i][j];
                } else {
                  const int j = sbase + ((it + 1) * P + i) * kWave + l;
                  xn[i] = tx[j];
                  yn[i] = ty[j];
                  if constexpr (NARR > 2) wn[i] = tw[j];
                  if constexpr (NARR > 3) cn[i] = tc[j];
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the arithmetic below
            double m[P];
            if constexpr (model_has_rec<Model>::value) {  // (tested on its own: generic lambda)
              if constexpr (FAST) {
                // seeded at the lane's first point of the window (PeaksModel::kSeedSteps = the
                // points of a lane per window)
                // (point of the lane within its window: a constant but for `sec` and `t`)
                const int gp = (ws * NIN + it) * P;
                if ((it * P) % 16 == 0 && gp % kSeedPts == 0) {
                  if ((rm & mask) != 0u) Model::rec_seed(prep, x[0], rm & mask, rs);
                  if constexpr (BGREC) Model::rec_seed_bg(prep, x[0], rs);
                } else if (Model::kMultiSeed && (it * P) % (kSeedPts / 4) == 0) {
                  // narrower peaks, seeded every kSeedPts / 2 or / 4 points of the lane
                  // (PeaksModel::prepare): run-time masks, wave-uniform, rarely non-zero
                  const unsigned ms = (gp % (kSeedPts / 2) == 0 ? seed16 : seed8) & rm & mask;
                  if (ms != 0u) Model::rec_seed_some(prep, x[0], ms, rs);
                }
                Model::template eval_mixed<P, BGREC>(prep, x, mask, rm, rs, m);
              } else {
                model_eval_n<Model, FAST, P>(prep, x, mask, m);
              }
            } else if constexpr (kX2) {
#pragma unroll
              for (int i = 0; i < P; ++i) m[i] = Model::eval2(prep, x[i], cv[i]);
            } else {
              model_eval_n<Model, FAST, P>(prep, x, mask, m);
            }
            // even points feed acc0, odd points acc1, each in increasing point order: the
            // summation order the oracle's mirror mode restates
#pragma unroll
            for (int i = 0; i < P; ++i) {
              double& acc = (i & 1) ? acc1 : acc0;
              const int64_t gi = gbase + wb + (it * P + i) * kWave + l;  // index in the dataset
              (void)gi;
              if constexpr (LIK == MHX_LIK_NORMAL) {
                // the y array holds y/sigma (host, once): r = y/sigma - m/sigma in one fma
                const double r = __builtin_fma(-m[i], wv[i], y[i]);
                acc = __builtin_fma(r, r, acc);
              } else if constexpr (LIK == MHX_LIK_NORMAL_CUTOFF) {
                const double r = __builtin_fma(-m[i], wv[i], y[i]);
                const double tt = __builtin_fma(-0.5 * r, r, cv[i]);
                acc = acc + (tt > -5000.0 ? tt : -5000.0);  // (max -5000d0 term) M:426
                // pads carry c = 0, w = 0 -> max(-5000, 0) = 0
              } else if constexpr (LIK == MHX_LIK_EXPR) {
                // (funcall log-liklihood-function y (apply fn x params) stddev) M:415: the tiles
                // hold y and sigma as given; pads masked
                const double tt = Model::lik_term(y[i], m[i], wv[i]);
                acc = acc + ((kWhole || gi < f.n) ? tt : 0.0);
              } else {
                // (- (* k (log lambda)) lambda ...) M:383; pads masked (no neutral pad exists)
                const double tt = __builtin_fma(y[i], tlog_rate(m[i], log_a3), -m[i]);
                acc = acc + ((kWhole || gi < f.n) ? tt : 0.0);
              }
            }
#pragma unroll
            for (int i = 0; i < P; ++i) {
              x[i] = xn[i];
              y[i] = yn[i];
              wv[i] = wn[i];
              cv[i] = cn[i];
            }
          }
        }
      };
      auto tile_body = [&](auto mk, auto rk, auto bgc) {
        if constexpr (LIK == MHX_LIK_POISSON || LIK == MHX_LIK_EXPR) {
          if (gbase + kTPW * kTilePoints <= f.n)
            tile_body_m(mk, rk, BoolC<true>{}, bgc);
          else
            tile_body_m(mk, rk, BoolC<false>{}, bgc);
        } else {
          tile_body_m(mk, rk, BoolC<true>{}, bgc);  // neutral pads: nothing to test
        }
      };
      constexpr BoolC<false> nobg{};
      // (the model test first, on its own: inside this generic lambda only a condition that does
      // not depend on FAST keeps the skipping code from being checked against other models)
      if constexpr (model_has_skip<Model>::value) {
        if constexpr (FAST) {
          // one straight-line variant per (peaks evaluated, peaks by recurrence) with <= 2 peaks
          // (bgrec: every peak of the function goes by the recurrence this step, and the
          // background with them - those variants never read x beyond their seeds)
          if constexpr (Model::kPeaks == 1) {
            if (bgrec) {
              if (tm & 1u) tile_body(CMask<1u>{}, CMask<1u>{}, BoolC<true>{});
              else tile_body(CMask<0u>{}, CMask<0u>{}, BoolC<true>{});
            } else {
              switch ((tm & 1u) | ((rmask & tm & 1u) << 1)) {
                case 0u: tile_body(CMask<0u>{}, CMask<0u>{}, nobg); break;
                case 1u: tile_body(CMask<1u>{}, CMask<0u>{}, nobg); break;
                default: tile_body(CMask<1u>{}, CMask<1u>{}, nobg); break;
              }
            }
          } else if constexpr (Model::kPeaks == 2) {
            if (bgrec) {
              switch (tm & 3u) {
                case 0u: tile_body(CMask<0u>{}, CMask<0u>{}, BoolC<true>{}); break;
                case 1u: tile_body(CMask<1u>{}, CMask<1u>{}, BoolC<true>{}); break;
                case 2u: tile_body(CMask<2u>{}, CMask<2u>{}, BoolC<true>{}); break;
                default: tile_body(CMask<3u>{}, CMask<3u>{}, BoolC<true>{}); break;
              }
            } else {
              switch ((tm & 3u) | ((rmask & tm & 3u) << 2)) {
                case 0u: tile_body(CMask<0u>{}, CMask<0u>{}, nobg); break;
                case 1u: tile_body(CMask<1u>{}, CMask<0u>{}, nobg); break;
                case 5u: tile_body(CMask<1u>{}, CMask<1u>{}, nobg); break;
                case 2u: tile_body(CMask<2u>{}, CMask<0u>{}, nobg); break;
                case 10u: tile_body(CMask<2u>{}, CMask<2u>{}, nobg); break;
                case 3u: tile_body(CMask<3u>{}, CMask<0u>{}, nobg); break;
                case 7u: tile_body(CMask<3u>{}, CMask<1u>{}, nobg); break;
                case 11u: tile_body(CMask<3u>{}, CMask<2u>{}, nobg); break;
                default: tile_body(CMask<3u>{}, CMask<3u>{}, nobg); break;  // (only one peak rec'able: n/a)
              }
            }
          } else {
            // more peaks: wave-uniform branches around each peak (never bgrec: PeaksModel::prepare)
            tile_body(tm, rmask & tm, nobg);
          }
        } else {
          tile_body(CMask<~0u>{}, CMask<0u>{}, nobg);
        }
      } else {
        tile_body(CMask<~0u>{}, CMask<0u>{}, nobg);
      }
    };
    if (active) {
      if constexpr (model_has_fast<Model>::value) {
        if (fastw) tile_work(BoolC<true>{}); else tile_work(BoolC<false>{});
      } else {
        tile_work(BoolC<false>{});
      }
    } else if (kTPW > 1 && t + 1 < nt) {
      mid_hook(t, buf);  // a wave without a running chain still moves its part of the tiles
    }
#ifndef MHX_NO_TILE_PRIO
    __builtin_amdgcn_s_setprio(0);  // (tile_prio; a short tile may leave its loop at any level)
#endif
    MHX_TIM(lds, 4);
#ifdef MHX_EARLY_REJECT
    bool er_chk = false;
    if constexpr (LIK == MHX_LIK_NORMAL) {
      er_chk = !solo && ((wi + 1) & wi) == 0 && wi < 64 && t + kTPW < nt;
      if (er_chk) {
        if (active) {
          const double s_now = wave_sum(acc0 + acc1);
          // (a non-finite partial sum is never "over": such a proposal is swept to the end and
          // traps the chain as it always did)
          if (s_now > *(volatile double*)&lds.deal_ll[w] && s_now < __builtin_inf()) {
            active = false;
            er_lost = true;
          }
        }
        if (active && l == 0) lds.deal_cost[er_i & 1] = 1;
      }
    }
#endif
    if (!solo) {  // (a resident tile is never overwritten: nothing to wait for)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile t+1 has landed
      __syncthreads();                                   // ... and so has everybody else's
    }
    MHX_TIM(lds, 5);
#ifdef MHX_EARLY_REJECT
    if (er_chk) {  // (two flags in turn: the one of the next check is lowered behind this barrier)
      const int alive = __builtin_amdgcn_readfirstlane(*(volatile int*)&lds.deal_cost[er_i & 1]);
      if (threadIdx.x == 0) lds.deal_cost[(er_i + 1) & 1] = 0;
      ++er_i;
      if (!alive) break;  // every proposal of the workgroup is lost: nothing left to sweep for
    }
#endif
  }
#ifdef MHX_EARLY_REJECT
  if (er_lost) {
    if (l == 0) lds.deal_ll[w] = -__builtin_inf();
    return __builtin_inf();
  }
#endif
  return wave_sum(acc0 + acc1);
}

template <int LIK>
__device__ __forceinline__ double finish_lik(const FnDesc& f, double s) {
  if (LIK == MHX_LIK_NORMAL) return __builtin_fma(-0.5, s, f.lik_const);
  if (LIK == MHX_LIK_NORMAL_CUTOFF || LIK == MHX_LIK_EXPR) return s;
  return s + f.lik_const;
}

__device__ __forceinline__ double finish_by_lik(const FnDesc& f, double s) {
  switch (f.lik) {
    case MHX_LIK_NORMAL: return finish_lik<MHX_LIK_NORMAL>(f, s);
    case MHX_LIK_NORMAL_CUTOFF: return finish_lik<MHX_LIK_NORMAL_CUTOFF>(f, s);
    case MHX_LIK_POISSON: return finish_lik<MHX_LIK_POISSON>(f, s);
    default: return finish_lik<MHX_LIK_EXPR>(f, s);
  }
}

// ---- split mode -----------------------------------------------------------------------------
// With few chains and long datasets one wavefront per chain leaves the GPU idle.  In split mode
// the likelihood sum of ONE chain is spread over many workgroups: wave `slot` of the launch
// takes points [p0, p1) of every function, straight from global memory (nothing is shared
// between the waves of a workgroup here, so no LDS tiles), and writes its partial sum; the
// chain's own wave then adds the partials in slot order (deterministic) and does everything
// else of the step.  Per point the arithmetic is sweep()'s; only the grouping of the sum differs,
// so results agree with the batch kernels to rounding, not bit for bit.
template <class Model, int LIK, bool FAST>
__device__ __forceinline__ double sweep_direct(const FnDesc& f, const typename Model::Prep& prep,
                                               int64_t p0, int64_t p1) {
  const int l = lane_id();
  double acc0 = 0.0, acc1 = 0.0;
  // two 64-point blocks per iteration, the loads of the next pair issued before the arithmetic
  // of the current one; p0 and p1 are multiples of 128 inside the padded arrays
  for (int64_t b = p0; b < p1; b += 2 * kWave) {
    const int64_t i0 = b + l, i1 = b + kWave + l;
    const double x0 = f.x[i0], x1 = f.x[i1], y0 = f.y[i0], y1 = f.y[i1];
    double w0 = 0.0, w1 = 0.0, c0 = 0.0, c1 = 0.0;
    if constexpr (LIK != MHX_LIK_POISSON) { w0 = f.w[i0]; w1 = f.w[i1]; }
    if constexpr (LIK == MHX_LIK_NORMAL_CUTOFF || model_xcols<Model>::value > 1) { c0 = f.c[i0]; c1 = f.c[i1]; }
    double m0, m1;
    if constexpr (model_xcols<Model>::value > 1) {  // (the second column of x sits in the c array)
      m0 = Model::eval2(prep, x0, c0);
      m1 = Model::eval2(prep, x1, c1);
    } else {
      m0 = model_eval<Model, FAST>(prep, x0);
      m1 = model_eval<Model, FAST>(prep, x1);
    }
    if constexpr (LIK == MHX_LIK_NORMAL) {
      const double r0 = __builtin_fma(-m0, w0, y0), r1 = __builtin_fma(-m1, w1, y1);
      acc0 = __builtin_fma(r0, r0, acc0);
      acc1 = __builtin_fma(r1, r1, acc1);
    } else if constexpr (LIK == MHX_LIK_NORMAL_CUTOFF) {
      const double r0 = __builtin_fma(-m0, w0, y0), r1 = __builtin_fma(-m1, w1, y1);
      const double t0 = __builtin_fma(-0.5 * r0, r0, c0), t1 = __builtin_fma(-0.5 * r1, r1, c1);
      acc0 = acc0 + (t0 > -5000.0 ? t0 : -5000.0);
      acc1 = acc1 + (t1 > -5000.0 ? t1 : -5000.0);
    } else if constexpr (LIK == MHX_LIK_EXPR) {
      const double t0 = Model::lik_term(y0, m0, w0), t1 = Model::lik_term(y1, m1, w1);
      acc0 = acc0 + (i0 < f.n ? t0 : 0.0);
      acc1 = acc1 + (i1 < f.n ? t1 : 0.0);
    } else {
      const double t0 = __builtin_fma(y0, tlog_rate(m0), -m0), t1 = __builtin_fma(y1, tlog_rate(m1), -m1);
      acc0 = acc0 + (i0 < f.n ? t0 : 0.0);
      acc1 = acc1 + (i1 < f.n ? t1 : 0.0);
    }
  }
  return wave_sum(acc0 + acc1);
}

// ... with the uniform-grid recurrence of the Gaussian peaks (PeaksModel, model_has_rec): the
// lane's points are 64 grid points apart here as well, so g, r advance as in sweep(); seeded
// every kSeedSteps points of the lane counted from the slice's start (no windows, no skipping in
// split mode), every peak of `rmask` by recurrence, the others directly.  Only for steps whose
// every |t| is inside the table exp's range (Prep::fast).
template <class Model, int LIK, bool BGREC>
__device__ __forceinline__ double sweep_direct_rec(const FnDesc& f,
                                                   const typename Model::Prep& prep, int64_t p0,
                                                   int64_t p1, unsigned rmask) {
  const int l = lane_id();
  double acc0 = 0.0, acc1 = 0.0;
  constexpr int S = model_seed_steps<Model>::value;
  constexpr unsigned all = (1u << model_peaks<Model>::value) - 1u;
  typename model_rec_state<Model>::type rs;
  for (int64_t bb = p0; bb < p1; bb += (int64_t)S * kWave) {
#pragma unroll 1
    for (int j = 0; j < S; j += 2) {
      const int64_t b = bb + (int64_t)j * kWave;
      if (b >= p1) break;
      const int64_t i0 = b + l, i1 = b + kWave + l;
      const double xs[2] = {f.x[i0], f.x[i1]};
      const double y0 = f.y[i0], y1 = f.y[i1];
      double w0 = 0.0, w1 = 0.0;
      if constexpr (LIK != MHX_LIK_POISSON) { w0 = f.w[i0]; w1 = f.w[i1]; }
      if (j == 0) {
        Model::rec_seed(prep, xs[0], rmask, rs);
        if constexpr (BGREC) Model::rec_seed_bg(prep, xs[0], rs);
      }
      double m[2];
      Model::template eval_mixed<2, BGREC>(prep, xs, all, rmask, rs, m);
      if constexpr (LIK == MHX_LIK_NORMAL) {
        const double r0 = __builtin_fma(-m[0], w0, y0), r1 = __builtin_fma(-m[1], w1, y1);
        acc0 = __builtin_fma(r0, r0, acc0);
        acc1 = __builtin_fma(r1, r1, acc1);
      } else {
        static_assert(LIK == MHX_LIK_NORMAL || LIK == MHX_LIK_POISSON, "likelihoods of the peaks kernels");
        const double t0 = __builtin_fma(y0, tlog_rate(m[0]), -m[0]);
        const double t1 = __builtin_fma(y1, tlog_rate(m[1]), -m[1]);
        acc0 = acc0 + (i0 < f.n ? t0 : 0.0);
        acc1 = acc1 + (i1 < f.n ? t1 : 0.0);
      }
    }
  }
  return wave_sum(acc0 + acc1);
}

// A problem whose K functions all use one compiled model and likelihood
template <class Model, int LIK, bool WG = false>
struct FixedSpec {
  static constexpr bool kSplit = true;  // has loglik_part
  // proposals are dealt to wave slots by cost when the model's cost depends on the proposal
  // (per-window peak masks and forms: group_logpost)
#ifdef MHX_DEAL  // (off in the product build: see group_logpost, "DEALING")
  static constexpr bool kDeal = model_has_skip<Model>::value;
#else
  static constexpr bool kDeal = false;
#endif
  template <class PF>
  static __device__ __forceinline__ int cost(const FnDesc& f, PF pf, double* scratch) {
    if constexpr (kDeal) {
      typename Model::Prep prep = model_prepare<Model>(pf, f, scratch);
      return Model::sweep_cost(prep, f);
    } else {
      return 0;
    }
  }
  // partial likelihood sum over points [p0, p1) (split mode; no barriers, one wave)
  template <class PF>
  static __device__ __forceinline__ double loglik_part(const FnDesc& f, PF pf, int64_t p0,
                                                       int64_t p1, double* scratch) {
    typename Model::Prep prep = model_prepare<Model>(pf, f, scratch);
    if constexpr (model_has_fast<Model>::value) {
      if (Model::fast_ok(prep)) {
        if constexpr (model_has_rec<Model>::value &&
                      (LIK == MHX_LIK_NORMAL || LIK == MHX_LIK_POISSON)) {
          const unsigned rmask = Model::rec_mask_long(prep);
          if (rmask != 0u) {
            if (__builtin_amdgcn_readfirstlane((int)Model::rec_bg_long(prep)) != 0)
              return sweep_direct_rec<Model, LIK, true>(f, prep, p0, p1, rmask);
            return sweep_direct_rec<Model, LIK, false>(f, prep, p0, p1, rmask);
          }
        }
        return sweep_direct<Model, LIK, true>(f, prep, p0, p1);
      }
    }
    return sweep_direct<Model, LIK, false>(f, prep, p0, p1);
  }
  template <class PF>
  static __device__ __forceinline__ double loglik(const FnDesc& f, PF pf, bool active,
                                                  GroupLds& lds, double* scratch) {
    typename Model::Prep prep = model_prepare<Model>(pf, f, scratch);
    bool fast = false;
    unsigned rmask = 0u;
    if constexpr (model_has_fast<Model>::value)
      fast = __builtin_amdgcn_readfirstlane((int)Model::fast_ok(prep)) != 0;
    bool bgrec = false;
    if constexpr (model_has_rec<Model>::value) {
      // (models that choose fast / guarded per window - model_has_skip - use these in their fast
      // windows whatever `fast` says about the dataset as a whole)
      const bool may = fast || model_has_skip<Model>::value;
      rmask = may ? Model::rec_mask(prep) : 0u;
      bgrec = may && __builtin_amdgcn_readfirstlane((int)Model::rec_bg(prep)) != 0;
    }
    return finish_lik<LIK>(f, sweep<Model, LIK, WG>(f, prep, active, lds, fast, rmask, bgrec));
  }
  static __device__ __forceinline__ double logprior(const FnDesc&, const double*, double bt) {
    return bt;
  }
};

// Anything else: wave-uniform dispatch on (model, shape, likelihood)
struct GenericSpec {
  static constexpr bool kSplit = false;  // models whose parameters live in LDS: batch kernels only
  static constexpr bool kDeal = false;
  template <class PF>
  static __device__ __forceinline__ int cost(const FnDesc&, PF, double*) { return 0; }
  template <class PF>
  static __device__ __forceinline__ double loglik_part(const FnDesc&, PF, int64_t, int64_t, double*) {
    return 0.0;
  }
  static __device__ __forceinline__ double logprior(const FnDesc&, const double*, double bt) {
    return bt;
  }
  template <class Model, class PF>
  static __device__ __forceinline__ double by_lik(const FnDesc& f, PF pf, bool active, GroupLds& lds) {
    typename Model::Prep prep = model_prepare<Model>(pf, f, lds.prm[wave_in_group()]);
    switch (f.lik) {
      case MHX_LIK_NORMAL:
        return finish_lik<MHX_LIK_NORMAL>(f, sweep<Model, MHX_LIK_NORMAL>(f, prep, active, lds));
      case MHX_LIK_NORMAL_CUTOFF:
        return finish_lik<MHX_LIK_NORMAL_CUTOFF>(
            f, sweep<Model, MHX_LIK_NORMAL_CUTOFF>(f, prep, active, lds));
      default:
        return finish_lik<MHX_LIK_POISSON>(f, sweep<Model, MHX_LIK_POISSON>(f, prep, active, lds));
    }
  }
  // one model with ONE likelihood: what run-time compiled problems use (mhx_rtc.cpp knows both)
  template <class Model, int LIK, class PF>
  static __device__ __forceinline__ double one_lik(const FnDesc& f, PF pf, bool active,
                                                   GroupLds& lds) {
    typename Model::Prep prep = model_prepare<Model>(pf, f, lds.prm[wave_in_group()]);
    return finish_lik<LIK>(f, sweep<Model, LIK>(f, prep, active, lds));
  }
  template <class Model, class PF>
  static __device__ __forceinline__ double by_lik_dyn(const FnDesc& f, PF pf, bool active,
                                                   GroupLds& lds, double* scratch) {
    typename Model::Prep prep = Model::prepare(pf, f, scratch);
    __builtin_amdgcn_wave_barrier();
    switch (f.lik) {
      case MHX_LIK_NORMAL:
        return finish_lik<MHX_LIK_NORMAL>(f, sweep<Model, MHX_LIK_NORMAL>(f, prep, active, lds));
      case MHX_LIK_NORMAL_CUTOFF:
        return finish_lik<MHX_LIK_NORMAL_CUTOFF>(
            f, sweep<Model, MHX_LIK_NORMAL_CUTOFF>(f, prep, active, lds));
      default:
        return finish_lik<MHX_LIK_POISSON>(f, sweep<Model, MHX_LIK_POISSON>(f, prep, active, lds));
    }
  }
  template <class PF>
  static __device__ __forceinline__ double loglik(const FnDesc& f, PF pf, bool active,
                                                  GroupLds& lds, double* scratch) {
    switch (f.model) {
      case MHX_MODEL_POLY:
        return by_lik_dyn<PolyModelDyn>(f, pf, active, lds, scratch);
      case MHX_MODEL_GAUSS_PEAKS:
        return by_lik_dyn<PeaksModelDyn<false>>(f, pf, active, lds, scratch);
      case MHX_MODEL_LORENTZ_PEAKS:
        return by_lik_dyn<PeaksModelDyn<true>>(f, pf, active, lds, scratch);
      case MHX_MODEL_LORDER_MIXED:
        return by_lik<LorderModel>(f, pf, active, lds);
      case MHX_MODEL_EXP_DECAY:
        return by_lik<ExpDecayModel>(f, pf, active, lds);
      case MHX_MODEL_SINUSOID:
        return by_lik<SinusoidModel>(f, pf, active, lds);
      default:
        return by_lik<PVoigt2Model>(f, pf, active, lds);
    }
  }
};

// ------------------------------------------------------------------------------------------
// prior-bounds-let (M:346-369) and the posterior sum of walker-make-step (M:1067-1070)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double bound_penalty(double p, double lo, double hi) {
  if (lo < p && p < hi) return 0.0;
  const double a = fabs(p - hi), b = fabs(p - lo);
  const double m = a < b ? a : b;
  return -1e10 * (dexp(m * 1e-5) - 1.0);
}
// bounds-total of one block (M:361-364): the terms in the block's order.  Lane i looks at bound
// i (one round trip to the descriptor for all of them instead of one per bound: 8 bounds cost
// 3 k cycles one after the other, and BASELINE config 4 lists 32 of them for each of its 8
// functions); when no lane finds its parameter outside, every term is +0 and so is the sum;
// otherwise the lanes' terms are added in index order, as the loop over the list did.
__device__ __forceinline__ double logprior_fn(const FnDesc& f, const double* theta) {
  const int l = lane_id();
  const int nb = __builtin_amdgcn_readfirstlane(f.n_bounds);
  static_assert(MHX_MAX_BOUNDS <= kWave, "one lane per bound");
  double p = 0.0, lo = 0.0, hi = 0.0;
  bool out = false;
  if (l < nb) {
    const int ix = f.bidx[l];
    p = ix >= 0 ? theta[ix] : 0.0;  // (getf params key 0d0) M:353
    lo = f.blo[l];
    hi = f.bhi[l];
    out = !(lo < p && p < hi);
  }
  if (__builtin_amdgcn_readfirstlane((int)(__ballot(out) != 0ull)) == 0) return 0.0;
  const double b = out ? bound_penalty(p, lo, hi) : 0.0;
  double acc = 0.0;
  for (int i = 0; i < nb; ++i) {
    const double bi = readlane_f64(b, i);
    acc = i == 0 ? bi : acc + bi;
  }
  return acc;
}

// the <key>-bound variable of prior-bounds-let (M:356-360) for global parameter g; 0 when the
// block has no bound on that key
__device__ __forceinline__ double bound_of(const FnDesc& f, const double* theta, int g) {
  for (int i = 0; i < f.n_bounds; ++i)
    if (f.bidx[i] == g) return bound_penalty(theta[g], f.blo[i], f.bhi[i]);
  return 0.0;
}

// theta' of this wave is in lds.prop[w]; collective over the workgroup
//
// DEALING.  The chains of a workgroup meet at a barrier every window of every sweep, and the four
// waves of a SIMD share its issue slots: a sweep costs the sum over windows of the slowest SIMD's
// load.  What a proposal costs is known before the sweep (Model::sweep_cost: which peaks each
// window evaluates, and in which form), so the workgroup's proposals are dealt to its wave slots
// by cost: ranked, then laid out in a snake over the SIMDs (rank 0-3 -> slots 0-3, rank 4-7 ->
// slots 7-4, ...; slots s and s + 4 share a SIMD, slots 4 q .. 4 q + 3 sit on four different ones
// - tools/microbench/wave_simd.hip: wave w of a workgroup runs on SIMD cyc[(k0 + w) % 4], cyc =
// 0 2 1 3), which gives every SIMD one proposal of each quartile.  Wave slot s then JUDGES the
// proposal of chain src(s) - likelihood sums only; the chain's identity, its scalars, its prior
// and its accept test stay with the chain's own wave - and hands the sum back through LDS.  A
// likelihood sum is a lane-strided accumulation and a butterfly inside ONE wave, the same
// instructions whichever wave runs them: a chain's bits do not depend on the deal
// (tests/test_gpu_families.py: MHX_NO_DEAL=1 against the default, bit for bit).
// BUILT, MEASURED, AND LEFT OUT OF THE PRODUCT BUILD (round 4; -DMHX_DEAL compiles it in).  What
// it costs: the estimate (a second prepare() and one pass of tile_mask) and one barrier - so a
// workgroup that finds its costs within a quarter of one another leaves everything in place and
// does not look again for 15 iterations (deal_skip).  Same-box A/B against the build without it,
// 3-5 interleaved rounds of bench.py each (kernel time of the timed launch): config 2 at the
// driver's arguments (a walk's iterations 6-25, wild proposals) 2.888 against 2.896 ms (+0.3 %;
// dealing at every iteration +1.2 %, and +3.8 % with the recurrence off, where a directly
// evaluated peak costs 14 instructions per point), but config 2 after the first adaptation tick
// 29.52 against 29.09 ms (-1.5 %) and config 3 75.17 against 73.49 ms (-2.3 %) - with
// MHX_NO_DEAL=1 as well: it is the code in the kernel (registers live across the first barrier,
// 220 more instructions per iteration in the controller), not the dealing, that costs there.
// The model that priced this (-3.8 % in the driver's window) was right about
// the gross gain; the cost estimate eats most of it.
template <class Spec>
__device__ __forceinline__ double group_logpost(const ProblemDesc& P, bool active, GroupLds& lds,
                                                int w, double* ll_out, double* lp_out) {
  bool deal = false;
  int skip = 0;
  if constexpr (Spec::kDeal) {
    deal = __builtin_amdgcn_readfirstlane(P.no_deal) == 0;
    // (written by thread 0 behind the first barrier of an earlier call: the barriers of that
    // call's sweep lie in between)
    skip = deal ? __builtin_amdgcn_readfirstlane(*(volatile int*)&lds.deal_skip) : 0;
    deal = deal && skip == 0;
    if (deal) {
      // (the wave's own proposal: written by this wave, no barrier needed to read it back)
      __builtin_amdgcn_wave_barrier();
      int cost = 0;
      if (active) {
        const double* tho = lds.prop[w];
        for (int k = 0; k < P.K; ++k) {
          const FnDesc& f = P.fn[k];
          auto pfo = [&](int j) -> double { return tho[f.idx[j]]; };
          cost += Spec::cost(f, pfo, lds.prm[w]);
        }
        cost = cost < 1 ? 1 : cost;  // (0 stands for "no proposal")
      }
      if (lane_id() == 0) lds.deal_cost[w] = cost;
    }
  }
  __syncthreads();  // proposals (and their costs) written, previous users of the tile buffers are done
  MHX_TIM(lds, 1);
  int src = w;
  bool act = active;
  if constexpr (Spec::kDeal) {
    if (deal) {
      static_assert(kWavesPerGroup % 4 == 0 && kWavesPerGroup <= 32, "snake over four SIMDs");
      const int l = lane_id();
      const int mine = l < kWavesPerGroup ? lds.deal_cost[l] : 0;
      int rank = 0;  // lane c: how many proposals of the workgroup cost more than chain c's
      int cmax = 0, cmin = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < kWavesPerGroup; ++j) {
        const int cj = __builtin_amdgcn_readlane(mine, j);
        rank += (cj > mine || (cj == mine && j < l)) ? 1 : 0;
        cmax = cj > cmax ? cj : cmax;
        cmin = (cj != 0 && cj < cmin) ? cj : cmin;
      }
      // every proposal within a quarter of the cheapest: nothing to gain, everybody stays at home
      const bool level = cmax - cmin <= (cmin >> 2);
      if (threadIdx.x == 0) lds.deal_skip = level ? 15 : 0;
      deal = !level;
      if (deal) {
        const int q = rank >> 2, p = rank & 3;
        const int slot = 4 * q + ((q & 1) ? 3 - p : p);
        const unsigned long long hit = __ballot(l < kWavesPerGroup && slot == w);
        src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
        act = __builtin_amdgcn_readlane(mine, src) != 0;
      }
    } else if (skip > 0) {
      if (threadIdx.x == 0) lds.deal_skip = skip - 1;
    }
  }
  double ll = 0.0, lp = 0.0;
  const double* th = lds.prop[src];   // the proposal this wave judges
  const double* tho = lds.prop[w];    // ... and its own chain's (the prior stays at home)
  for (int k = 0; k < P.K; ++k) {
    const FnDesc& f = P.fn[k];
    auto pf = [&](int j) -> double { return th[f.idx[j]]; };
    const double v = Spec::loglik(f, pf, act, lds, lds.prm[w]);
    ll = k == 0 ? v : ll + v;
    // Spec::logprior lets a user prior body add terms to bounds-total (M:366-369)
    const double q = Spec::logprior(f, tho, logprior_fn(f, tho));
    lp = k == 0 ? q : lp + q;
  }
  if constexpr (Spec::kDeal) {
    if (deal) {
      if (lane_id() == 0) lds.deal_ll[src] = ll;
      __syncthreads();
      ll = *(volatile double*)&lds.deal_ll[w];
    }
  }
  *ll_out = ll;
  *lp_out = lp;
  return ll + lp;
}

// PERSIST (with SPLIT): the chain's master wave of k_persist - wave 0 of workgroup (0, chain) -
// runs whole iterations in ONE launch: where the split step kernel leaves the proposal behind and
// ends, it publishes it (persist_publish) and goes on with the second half of the iteration,
// whose split_logpost waits for the partial sums of the chain's sweep workgroups.
// The handshake moves 8 ... 63 parameters one way and a few hundred partial sums the other, once
// per iteration, between XCDs whose L2s do not see each other's lines.  Acquire / release at
// agent scope would do - and invalidate the whole L2 of the reader at every poll, the dataset
// with it (measured: 18 us per step where the two launches take 16).  Instead every word of the
// handshake is written and read by memory instructions that go to the level all XCDs share (sc1:
// relaxed agent-scope atomics, and 16-byte loads / stores with the same cache policy), data and
// tag in ONE naturally aligned unit - a 128-byte line of the proposal block, a 16-byte {sum,
// generation} pair - so that who finds the tag finds the data, and nothing else leaves a cache.
// A tag word is {check << 32 | generation}: the low half the generation (or the stop bit), the
// high half a 32-bit fold of the data the tag vouches for.  A reader that finds the generation it
// waits for but another fold has caught the words of its line / pair in two states (a 128-byte
// read is not promised to be atomic against a 128-byte write: seen once per ~1e6 polls) and
// reads again.
constexpr unsigned long long kPersistStop = 1ull << 31;  // generation half: the chain's master is through
__device__ __forceinline__ unsigned persist_fold(unsigned long long v) {
  return (unsigned)v ^ (unsigned)(v >> 32) ^ 0x9E3779B9u;
}
#ifdef MHX_PERSIST_TIMING  // (measurement build: one record per workgroup, read by the launcher)
__device__ unsigned long long g_persist_trace[1024][8];
__device__ __forceinline__ void persist_trace(unsigned long long a, unsigned long long b, unsigned long long c,
                                              unsigned long long n) {
  unsigned hw = 0, xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned i = blockIdx.y * gridDim.x + blockIdx.x;
  if (i < 1024) {
    unsigned long long* r = g_persist_trace[i];
    r[0] = xcc & 15; r[1] = (hw >> 13) & 7; r[2] = (hw >> 8) & 15; r[3] = a; r[4] = b; r[5] = c; r[6] = n;
    r[7] = __builtin_amdgcn_s_memrealtime();
  }
}
#endif
#ifdef MHX_PERSIST_NOPRIO  // (A/B knob: tools/debug/persist_prio_ab.sh)
#define MHX_PERSIST_PRIO(P) do { } while (0)
#else
#define MHX_PERSIST_PRIO(P) __builtin_amdgcn_s_setprio(P)
#endif
constexpr unsigned kPersistPatience = 1u << 20;   // polls (a memory round trip apart) before giving up
// xor over each row of 16 lanes, left in all of them: four rotations within the row (DPP: a few
// cycles each, where a shuffle is a trip through the LDS crossbar - and this sits between a poll's
// load coming back and the wave knowing what it has read)
__device__ __forceinline__ unsigned persist_row_xor(unsigned f) {
  f ^= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x128, 0xf, 0xf, false);  // row_ror:8
  f ^= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x124, 0xf, 0xf, false);  // row_ror:4
  f ^= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x122, 0xf, 0xf, false);  // row_ror:2
  f ^= (unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x121, 0xf, 0xf, false);  // row_ror:1
  return f;
}
constexpr int kPersistMaxParams = 60;             // four lines of 15 parameters and a tag
typedef __attribute__((ext_vector_type(4))) unsigned int persist_u4;
__device__ __forceinline__ void persist_store_pair(void* p, double v, unsigned long long gen) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const persist_u4 q = {(unsigned)b, (unsigned)(b >> 32), (unsigned)gen, persist_fold(b)};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ persist_u4 persist_load_pair(const void* p) {
  persist_u4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=v"(r) : "v"(p) : "memory");
  return r;
}
// the proposal (lane j < d: theta'_j) and its generation into the chain's block: ONE store
// instruction of the wave, lanes 16 i .. 16 i + 15 writing line i - its 15 parameters and, from
// lane 16 i + 15, its tag - as one 128-byte write.  (With the parameters first, a release fence,
// then the tags, the fence's wait for the first stores' acknowledgement was a memory round trip
// on the critical path of every iteration: 9.5 -> 9.2 us per step of a single walker.)
__device__ __forceinline__ void persist_publish(const ChainState& S, int64_t c, int d, double thp,
                                                unsigned long long gen) {
  unsigned long long* m = S.persist_msg + c * 64;
  const int l = lane_id();
  const int e = (l >> 4) * 15 + (l & 15);  // the element this lane's word carries
  const double v = __shfl(thp, e < kWave ? e : 0, kWave);
  const unsigned long long data = ((l & 15) != 15 && e < d) ? (unsigned long long)__double_as_longlong(v) : 0ull;
  unsigned f = (l & 15) != 15 ? persist_fold(data) : 0u;  // xor over the line's 15 data words
  f = persist_row_xor(f);
  const unsigned long long word = (l & 15) == 15 ? (((unsigned long long)f << 32) | gen) : data;
  if ((l >> 4) <= (d > 0 ? (d - 1) / 15 : 3))  // the lines in use (a stop word: all four)
    __hip_atomic_store(m + l, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// split mode: the same value from the partial sums the sweep launch left in S.split_part
// PERSIST (k_persist's master wave): the partial sums arrive as {sum, generation} pairs in
// S.persist_part while this waits - every slot's pair is read until it carries generation `gen`;
// a slot that never does (patience) is reported through *lost: the caller takes the iteration
// back and ends the launch.
template <class Spec, bool PERSIST = false>
__device__ __forceinline__ double split_logpost(const ProblemDesc& P, const ChainState& S,
                                                int64_t c, bool active, GroupLds& lds, int w,
                                                double* ll_out, double* lp_out,
                                                unsigned long long gen = 0, bool* lost_out = nullptr) {
  double ll = 0.0, lp = 0.0;
  const double* th = lds.prop[w];
  const int l = lane_id();
  if constexpr (PERSIST) {
    // the priors need the proposal, not the sums: formed while the sums are on their way (their
    // reads of the bounds are a memory round trip on every iteration's critical path otherwise)
    for (int k = 0; k < P.K; ++k) {
      const FnDesc& f = P.fn[k];
      const double q = Spec::logprior(f, th, logprior_fn(f, th));
      lp = k == 0 ? q : lp + q;
    }
  }
  for (int k = 0; k < P.K; ++k) {
    const FnDesc& f = P.fn[k];
    const double* part = S.split_part + ((active ? c : 0) * P.K + k) * S.split_slots;
    double a = 0.0;
    if constexpr (PERSIST) {
      const char* pairs = (const char*)S.persist_part + (((active ? c : 0) * P.K + k) * S.split_slots) * 16;
      bool lost = false;
      MHX_PERSIST_PRIO(0);  // (waiting: see persist_poll)
      for (int s0 = 0; s0 < S.split_slots && active; s0 += kWave) {  // (uniform)
        const int sl = s0 + l;
        double v = 0.0;
        unsigned n = 0;
        for (;;) {
          bool have = true;
          if (sl < S.split_slots) {
            const persist_u4 q = persist_load_pair(pairs + (size_t)sl * 16);
            const unsigned long long b = ((unsigned long long)q.y << 32) | q.x;
            have = q.z == (unsigned)gen && q.w == persist_fold(b);
            v = __longlong_as_double((long long)b);
          }
          if (__builtin_amdgcn_readfirstlane((int)(__ballot(!have) == 0ull))) break;
          if (++n >= kPersistPatience) {
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (sl < S.split_slots) a = a + v;  // slot order per lane, as below
      }
      if (lost && lost_out) *lost_out = true;
      MHX_PERSIST_PRIO(3);  // (the controller: the serial part of the round)
    } else
    for (int s = l; s < S.split_slots; s += kWave) a = a + part[s];  // slot order, fixed
    const double v = finish_by_lik(f, wave_sum(a));
    ll = k == 0 ? v : ll + v;
    if constexpr (!PERSIST) {
      const double q = Spec::logprior(f, th, logprior_fn(f, th));
      lp = k == 0 ? q : lp + q;
    }
  }
  *ll_out = ll;
  *lp_out = lp;
  return ll + lp;
}

// ------------------------------------------------------------------------------------------
// walker-get on the device ring (newest first = entries nh-1, nh-2, ...)
// ------------------------------------------------------------------------------------------
struct Ring {
  const double* prob;   // this chain's [R]
  const double* theta;  // this chain's [R][d]
  int64_t nh;           // entries ever pushed
  int64_t length;       // (walker-length w)
  int mask;             // R - 1
  int d;
  __device__ __forceinline__ int window(int take) const {
    return (int)(length < (int64_t)take ? length : (int64_t)take);
  }
  __device__ __forceinline__ int slot(int s) const { return (int)((nh - 1 - s) & mask); }
};

__device__ __forceinline__ unsigned long long bits_of(double v) {
  return (unsigned long long)__double_as_longlong(v);
}

// M:506-508 with remove-consecutive-duplicates (M:220-223); eql on doubles = same bits
__device__ __forceinline__ void ring_acceptance(const Ring& r, int take, int* num, int* den) {
  const int t = r.window(take), l = lane_id();
  int runs = 0;
  for (int s = l; s < t; s += kWave) {
    const bool last = s == t - 1;
    const double a = r.prob[r.slot(s)];
    const double b = last ? 0.0 : r.prob[r.slot(s + 1)];
    runs += (last || bits_of(a) != bits_of(b)) ? 1 : 0;
  }
  // (every lane holds the same sum; said to the compiler, so that what is decided from it - and
  // the chain's counters updated under those decisions - stays in scalar registers)
  *num = __builtin_amdgcn_readfirstlane(wave_sum_i(runs));
  *den = t;
}
// rational acceptance num/den against the single-float literals of M:898, M:911, M:930-941
__device__ __forceinline__ bool acc_lt(int num, int den, float f) {
  return (double)num < (double)f * (double)den;
}
__device__ __forceinline__ bool acc_gt(int num, int den, float f) {
  return (double)num > (double)f * (double)den;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const double o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}

// stable-probs-p M:880-885 over (walker-get :log-liklihoods :take steps-to-settle)
__device__ __forceinline__ bool ring_stable_probs(const Ring& r, int sts) {
  const int t = r.window(sts), l = lane_id();
  if (t < 200) return false;
  const double kInf = __builtin_inf();
  double early = -kInf, late = -kInf, mn = kInf;
  for (int s = l; s < t; s += kWave) {
    const double v = r.prob[r.slot(s)];
    if (s < 200 && v > early) early = v;
    if (s >= t - 200 && v > late) late = v;
    if (v < mn) mn = v;
  }
  early = uniform_f64(wave_max(early));
  late = uniform_f64(wave_max(late));
  mn = uniform_f64(-wave_max(-mn));
  const double spread = early - mn;
  return fabs(early - late) < 0.5 && 4.0 < spread && spread < 9.0;
}

enum { L_OK = 0, L_CAUGHT = 1, L_INVALID = 2, L_EMPTY = 3 };
#ifndef MHX_LM_INLINE
#define MHX_LM_INLINE __forceinline__
#endif
// With finite inputs an inf is an overflow trap and a NaN can only follow an earlier overflow
// (inf - inf), so any non-finite result means floating-point-overflow was signalled first:
// caught by the handler-case of M:891-894.  Only the explicit 0/0 of M:597 is invalid.
__device__ __forceinline__ int trap_of(double r) { return finite_f64(r) ? L_OK : L_CAUGHT; }

// (walker-get :get :l-matrix :take take) M:543 = Cholesky of the population covariance of the
// displacements between successive forward steps.  fwd: int scratch [>= take]; cov, Lout:
// double scratch [d*d]; avg: LDS scratch [d].  Returns L_* (uniform).
// forward steps M:497-502 of the newest `take` steps, newest first, compacted in order into
// fwd[] (ring slots); returns their number (uniform)
__device__ __forceinline__ int ring_forward_list(const Ring& r, int take, int* fwd) {
  const int t = r.window(take), l = lane_id();
  int nf = 0;
  for (int base = 0; base < t - 1; base += kWave) {
    const int s = base + l;
    bool f = false;
    if (s < t - 1) {
      const double a = r.prob[r.slot(s)], b = r.prob[r.slot(s + 1)];
      f = !(a <= b);
    }
    const unsigned long long m = __ballot(f);
    const int pos = nf + __popcll(m & ((1ULL << l) - 1ULL));
    if (f) fwd[pos] = r.slot(s);
    nf += __popcll(m);
  }
  return nf;
}

// cholesky-decomp M:583-598 by ONE lane in the reference's loop order (diagonal
// sqrt(max 0 .), upper triangle 0).  Returns L_OK / L_CAUGHT (x/0, overflow) / L_INVALID (0/0).
__device__ __forceinline__ int cholesky_seq(const double* cov, double* Lout, int d) {
  int cst = L_OK;
  double zero = 0.0;  // (formed here: hoisted out of the stepping loop it would live - spilled -
  asm volatile("" : "+v"(zero));  // across every likelihood sweep)
  for (int e = 0; e < d * d; ++e) Lout[e] = zero;
  for (int i = 0; i < d && cst == L_OK; ++i)
    for (int k = 0; k <= i && cst == L_OK; ++k) {
      double tmp = 0.0;
      for (int j = 0; j < k; ++j) tmp = tmp + Lout[i * d + j] * Lout[k * d + j];
      cst = trap_of(tmp);
      if (cst != L_OK) break;
      if (i == k) {
        const double a = cov[i * d + k] - tmp;
        Lout[i * d + k] = ieee_sqrt(a > 0.0 ? a : 0.0);  // (sqrt (max 0d0 a)) M:596
      } else {
        const double num = cov[i * d + k] - tmp, den = Lout[k * d + k];
        if (den == 0.0) {
          cst = num == 0.0 ? L_INVALID : L_CAUGHT;
        } else {
          const double q = num / den;
          cst = trap_of(q);
          Lout[i * d + k] = q;
        }
      }
    }
  return cst;
}

__device__ MHX_LM_INLINE int ring_l_matrix(const Ring& r, int take, int* fwd, double* cov,
                                          double* Lout, lds_dptr_t avg, int* n_forward) {
  const int l = lane_id(), d = r.d;
  const int nf = ring_forward_list(r, take, fwd);
  *n_forward = nf;
  if (nf == 0) return L_CAUGHT;  // (elt nil 0): an index error is a type-error
  if (nf == 1) return L_EMPTY;
  __threadfence();
  const int M = nf - 1;
  const double dM = (double)M;
  // diff-lplist M:277-280: older - newer of consecutive forward steps
  auto diff = [&](int k, int p) -> double {
    return r.theta[(int64_t)fwd[k + 1] * d + p] - r.theta[(int64_t)fwd[k] * d + p];
  };
  // averages M:626: (/ (reduce #'+ x) n)
  int st = L_OK, first_bad = 0x7fffffff;
  if (l < d) {
    double s = diff(0, l);
    for (int k = 1; k < M; ++k) s = s + diff(k, l);
    const double a = s / dM;
    avg[l] = a;
    st = trap_of(a);
    if (st != L_OK) first_bad = l;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // covariance M:636-643 (the /n sits inside the accumulation)
  for (int e = l; e < d * d; e += kWave) {
    const int i = e / d, j = e - i * d;
    const double ai = avg[i], aj = avg[j];
    double mini = 0.0;
    for (int k = 0; k < M; ++k) {
      const double x = diff(k, i) - ai, y = diff(k, j) - aj;
      mini = mini + (x * y) / dM;
    }
    cov[e] = mini;
    if (st == L_OK) {
      st = trap_of(mini);
      if (st != L_OK) first_bad = d + e;
    }
  }
  // the first failing operation in the reference's order decides which condition is raised
  int fb = first_bad;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(fb, m, 64);
    fb = o < fb ? o : fb;
  }
  if (fb != 0x7fffffff) {
    const unsigned long long who = __ballot(first_bad == fb);
    const int src = __ffsll((long long)who) - 1;
    return __builtin_amdgcn_readlane(st, src);
  }
  __threadfence();
  // cholesky-decomp M:583-598, one lane, the reference's loop order
  int cst = L_OK;
  if (l == 0) cst = cholesky_seq(cov, Lout, d);
  cst = __builtin_amdgcn_readfirstlane(cst);
  __threadfence();
  return cst;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
template <class Spec>
__device__ __forceinline__ void k_init_body(const ProblemDesc* __restrict__ Pp, ChainState S);
template <class Spec>
__device__ __forceinline__ void k_step_injected_body(
    const ProblemDesc* __restrict__ Pp, ChainState S, const double* __restrict__ Lin,
    int per_chain_l, const double* __restrict__ z, const double* __restrict__ u,
    const double* __restrict__ T, unsigned char* __restrict__ accepted);
template <class Spec, bool SPLIT = false, bool PERSIST = false>
__device__ __forceinline__ void k_adaptive_body(const ProblemDesc* __restrict__ Pp, ChainState S,
                                                RunDesc R, int64_t max_iters, int plain,
                                                int mode = 1, int64_t group = -1);
template <class Spec>
__device__ __forceinline__ void k_split_sweep_body(const ProblemDesc* __restrict__ Pp,
                                                   ChainState S);

// Every Spec-dependent kernel is a __device__ body + a thin __global__ template, so that the
// run-time compiled user-expression kernels (mhx_rtc.cpp) can wrap the same bodies.
template <class Spec>
__device__ __forceinline__ void k_logpost_body(const ProblemDesc* __restrict__ Pp,
                                               const double* __restrict__ theta, int64_t n,
                                               double* __restrict__ out,
                                               double* __restrict__ parts) {
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  const bool valid = c < n;
  if (l < d) lds.prop[w][l] = valid ? theta[c * d + l] : 0.0;
  double ll, lp;
  const double v = group_logpost<Spec>(P, valid, lds, w, &ll, &lp);
  if (valid && l == 0) {
    out[c] = v;
    if (parts) {
      parts[2 * c] = ll;
      parts[2 * c + 1] = lp;
    }
  }
}
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_logpost(const ProblemDesc* __restrict__ Pp,
                                                      const double* __restrict__ theta, int64_t n,
                                                      double* __restrict__ out,
                                                      double* __restrict__ parts) {
  k_logpost_body<Spec>(Pp, theta, n, out, parts);
}

// walker-create's first step (M:1148-1163): prob, walk = (first-step), length 1, age 1
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_init(const ProblemDesc* __restrict__ Pp,
                                                   ChainState S) {
  k_init_body<Spec>(Pp, S);
}
template <class Spec>
__device__ __forceinline__ void k_init_body(const ProblemDesc* __restrict__ Pp, ChainState S) {
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  const bool valid = c < S.n_chains;
  const double th = (valid && l < d) ? S.theta[c * d + l] : 0.0;
  if (l < d) lds.prop[w][l] = th;
  double ll, lp;
  const double v = group_logpost<Spec>(P, valid, lds, w, &ll, &lp);
  if (!valid) return;
  if (l < d) {
    S.best_theta[c * d + l] = th;
    S.hist_theta[(c * S.R + 0) * d + l] = th;
  }
  if (l == 0) {
    S.prob[c] = v;
    S.best_prob[c] = v;
    S.hist_prob[c * S.R + 0] = v;
    S.n_hist[c] = 1;
    S.length[c] = 1;
    S.age[c] = 1;
    S.draw[c] = 0;
    S.status[c] = finite_f64(v) ? MHX_CHAIN_DONE : MHX_CHAIN_FP_TRAP;
    S.loop_i[c] = 0;
    S.shutting[c] = 0;
    S.temperature[c] = 1.0;
    S.reset_index[c] = 10000;
  }
}

// A wave-uniform pointer the optimiser must treat as new: used on the base pointers of the
// stepping loop once per iteration, so that the per-lane addresses formed from them (base +
// lane offset: 2 VGPRs each) are recomputed where they are used instead of being hoisted out of
// the loop and kept alive - i.e. spilled to scratch - across the likelihood sweep.
template <class T>
__device__ __forceinline__ T* fresh_ptr(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// per-wave registers of one chain: wave-uniform values only (SGPRs).  The position theta and the
// most likely position stay in HBM (S.theta, S.best_theta: rewritten when a proposal is taken /
// a new best is found) and the proposal in LDS, so that no per-lane value is alive across the
// likelihood sweep - the sweep may have all 128 VGPRs and nothing of the chain goes to scratch.
struct ChainRegs {
  double prob0, best_prob, T;
  int64_t nh, length, age, loop_i, reset_index;
  uint64_t draw;
  int shutting, status;
};
// the same, parked in the wave's LDS slot while the likelihood sweep runs (with the accept
// uniform drawn for the outstanding proposal): 96 bytes, written by lane 0, read back by
// every lane (one address: a broadcast) and returned to scalar registers
struct ChainPark {
  double prob0, best_prob, T, u, t_next;
  int64_t nh, length, age, loop_i, reset_index;
  uint64_t draw;
  int shutting, status;
};
__device__ __forceinline__ void chain_park(ChainPark& slot, const ChainRegs& r, double u,
                                           double t_next) {
  if (lane_id() == 0) {
    slot.prob0 = r.prob0; slot.best_prob = r.best_prob; slot.T = r.T; slot.u = u;
    slot.t_next = t_next;
    slot.nh = r.nh; slot.length = r.length; slot.age = r.age; slot.loop_i = r.loop_i;
    slot.reset_index = r.reset_index; slot.draw = r.draw;
    slot.shutting = r.shutting; slot.status = r.status;
  }
}
__device__ __forceinline__ void chain_unpark(const ChainPark& slot, ChainRegs& r, double& u,
                                             double& t_next) {
  t_next = uniform_f64(slot.t_next);
  r.prob0 = uniform_f64(slot.prob0);
  r.best_prob = uniform_f64(slot.best_prob);
  r.T = uniform_f64(slot.T);
  u = uniform_f64(slot.u);
  r.nh = uniform_i64(slot.nh);
  r.length = uniform_i64(slot.length);
  r.age = uniform_i64(slot.age);
  r.loop_i = uniform_i64(slot.loop_i);
  r.reset_index = uniform_i64(slot.reset_index);
  r.draw = (uint64_t)uniform_i64((int64_t)slot.draw);
  r.shutting = __builtin_amdgcn_readfirstlane(slot.shutting);
  r.status = __builtin_amdgcn_readfirstlane(slot.status);
}

__device__ __forceinline__ void chain_load(const ChainState& S, int64_t c, int d, ChainRegs& r) {
  const int l = lane_id();
  (void)l;
  r.prob0 = uniform_f64(S.prob[c]);
  r.best_prob = uniform_f64(S.best_prob[c]);
  r.T = uniform_f64(S.temperature[c]);
  r.nh = uniform_i64(S.n_hist[c]);
  r.length = uniform_i64(S.length[c]);
  r.age = uniform_i64(S.age[c]);
  r.loop_i = uniform_i64(S.loop_i[c]);
  r.reset_index = uniform_i64(S.reset_index[c]);
  r.draw = (uint64_t)uniform_i64((int64_t)S.draw[c]);
  r.shutting = __builtin_amdgcn_readfirstlane(S.shutting[c]);
  r.status = __builtin_amdgcn_readfirstlane(S.status[c]);
}
__device__ __forceinline__ void chain_store(const ChainState& S, int64_t c, int d,
                                            const ChainRegs& r) {
  const int l = lane_id();
  (void)d;
  if (l == 0) {
    S.prob[c] = r.prob0;
    S.best_prob[c] = r.best_prob;
    S.temperature[c] = r.T;
    S.n_hist[c] = r.nh;
    S.length[c] = r.length;
    S.age[c] = r.age;
    S.loop_i[c] = r.loop_i;
    S.reset_index[c] = r.reset_index;
    S.draw[c] = r.draw;
    S.shutting[c] = r.shutting;
    S.status[c] = r.status;
  }
}

// get-covariant-sample M:679-700: lane i forms sum_j L_ij z_j from 0d0 (multiply, then add),
// then adds theta_i.  z_j sits in lane j of zv.
// (propose_lz: the product alone - the persistent master forms the next proposal's while it waits
// for this one's sums, and adds theta when the accept decision has said which theta)
__device__ __forceinline__ double propose_lz(const double* L, int d, double zv);
__device__ __forceinline__ double propose(const double* L, int d, double zv, double th) {
  return propose_lz(L, d, zv) + th;
}
__device__ __forceinline__ double propose_lz(const double* L, int d, double zv) {
  const int l = lane_id();
  double mini = 0.0;
  // eight elements of the lane's row at a time, all loads before the first use: one round trip
  // to L per eight columns instead of one per column (d = 8: 5 k cycles -> 1)
  for (int j0 = 0; j0 < d; j0 += 8) {
    double lv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) lv[q] = (l < d && j0 + q < d) ? L[l * d + j0 + q] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (j0 + q < d) {  // (uniform; a term is added only where the list had one)
        const double zj = readlane_f64(zv, j0 + q);
        mini = mini + lv[q] * zj;
      }
    }
  }
  return mini;
}

// walker-modify :add-step M:549-555 on the ring.  th: the step's position (lane j: theta_j),
// which is also S.theta's content; taken: the step is a newly accepted proposal (only then can it
// be a new most likely step, and only then does S.theta change)
__device__ __forceinline__ void add_step(const ChainState& S, int64_t c, int d, ChainRegs& r,
                                         double th, bool taken) {
  const int l = lane_id();
  const int64_t slot = r.nh & (int64_t)(S.R - 1);
  if (l < d) {
    S.hist_theta[(c * S.R + slot) * d + l] = th;
    if (taken) S.theta[c * d + l] = th;
  }
  if (l == 0) S.hist_prob[c * S.R + slot] = r.prob0;
  r.nh++;
  r.length++;
  r.age++;
  if (r.prob0 > r.best_prob) {  // (a repeated step has prob0 <= best_prob already)
    r.best_prob = r.prob0;
    if (l < d) S.best_theta[c * d + l] = th;
  }
}
// theta_j of chain c in lane j (0 beyond d)
__device__ __forceinline__ double chain_theta(const ChainState& S, int64_t c, int d) {
  const int l = lane_id();
  return l < d ? S.theta[c * d + l] : 0.0;
}

// accept test of M:1091-1092; log u is evaluated only when the first clause fails in the
// reference (short-circuit `or`); the value of the test is the same either way
__device__ __forceinline__ bool mh_accept(double prob1, double prob0, double T, double u) {
  return (prob1 > prob0) || ((prob1 - prob0) / T > det_log(u));
}
// ... with det_log(u) already at hand (the fused kernel: rng_lane_value)
__device__ __forceinline__ bool mh_accept_log(double prob1, double prob0, double T, double log_u) {
  return (prob1 > prob0) || ((prob1 - prob0) / T > log_u);
}

template <class Spec>
__global__ __launch_bounds__(kThreads) void k_step_injected(
    const ProblemDesc* __restrict__ Pp, ChainState S, const double* __restrict__ Lin,
    int per_chain_l, const double* __restrict__ z, const double* __restrict__ u,
    const double* __restrict__ T, unsigned char* __restrict__ accepted) {
  k_step_injected_body<Spec>(Pp, S, Lin, per_chain_l, z, u, T, accepted);
}
template <class Spec>
__device__ __forceinline__ void k_step_injected_body(
    const ProblemDesc* __restrict__ Pp, ChainState S, const double* __restrict__ Lin,
    int per_chain_l, const double* __restrict__ z, const double* __restrict__ u,
    const double* __restrict__ T, unsigned char* __restrict__ accepted) {
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  const bool valid = c < S.n_chains;
  ChainRegs r;
  double thp = 0.0;
  bool run = false;
  if (valid) {
    chain_load(S, c, d, r);
    run = r.status != MHX_CHAIN_FP_TRAP;
    const double zv = l < d ? z[c * d + l] : 0.0;
    thp = propose(Lin + (per_chain_l ? c * d * d : 0), d, zv, chain_theta(S, c, d));
    if (l < d) lds.prop[w][l] = thp;
  }
  double ll, lp;
  const double prob1 = group_logpost<Spec>(P, valid && run, lds, w, &ll, &lp);
  if (!valid || !run) return;
  int acc = 0;
  if (!finite_f64(prob1)) {
    r.status = MHX_CHAIN_FP_TRAP;  // the reference would have signalled (traps enabled)
  } else {
    const double uu = uniform_f64(u[c]), TT = uniform_f64(T[c]);
    // here the caller's u stands for (random 1.0d0); log as the runtime's libm would
    acc = (prob1 > r.prob0) || ((prob1 - r.prob0) / TT > det_log(uu));
    if (acc) r.prob0 = prob1;
    add_step(S, c, d, r, acc ? (l < d ? lds.prop[w][l] : 0.0) : chain_theta(S, c, d), acc != 0);
  }
  if (accepted && l == 0) accepted[c] = (unsigned char)acc;
  if (l == 0 && r.status != MHX_CHAIN_FP_TRAP) atomicAdd(S.step_counter, 1ULL);
  chain_store(S, c, d, r);
}

__device__ __forceinline__ int64_t floor_mod(int64_t a, int64_t b) {
  const int64_t m = a % b;
  return m < 0 ? m + b : m;
}
// The loop index modulo 200, 1000 and 2 sts, carried along instead of divided out: three 64-bit
// divisions per iteration are nothing beside a sweep of the batch kernels, but a tenth of a
// microsecond each on the path a persistent kernel's sweep workgroups wait on.  (Tried in the
// batch kernels too: test.lisp's single walker +2.7 %, the driver's window +1 %, but four more
// live scalars across the sweep cost config 3 and config 4 16 B of scratch each and 0.5-1 %.)
struct LoopPhases {
  int m200, m1000;
  int64_t msts;
  __device__ __forceinline__ void set(int64_t i, int64_t sts) {
    m200 = (int)floor_mod(i, 200);
    m1000 = (int)floor_mod(i, 1000);
    msts = floor_mod(i, 2 * sts);
  }
  __device__ __forceinline__ void step(int64_t sts) {
    if (++m200 == 200) m200 = 0;
    if (++m1000 == 1000) m1000 = 0;
    if (++msts == 2 * sts) msts = 0;
  }
};

// The do loop of walker-adaptive-steps-full (M:902-942), up to max_iters iterations for each
// chain of the workgroup.  plain != 0: walker-many-steps (M:849-853): constant L, T = 1.
template <class Spec>
__global__ __launch_bounds__(kThreads, 4) void k_adaptive(const ProblemDesc* __restrict__ Pp,
                                                       ChainState S, RunDesc R,
                                                       int64_t max_iters, int plain) {
  k_adaptive_body<Spec>(Pp, S, R, max_iters, plain);
}
// SPLIT: the iteration is cut where the log-posterior is needed - a launch of this body does the
// second half of one iteration (judge the outstanding proposal with the partial sums of the
// sweep launch: mode 1, 2) and / or the first half of the next (end test, shut-down test, new
// proposal into S.split_prop: mode 0, 1).  mode 0 primes, 1 is the steady state, 2 ends a run
// of launches with nothing outstanding.
template <class Spec, bool SPLIT, bool PERSIST>
__device__ __forceinline__ void k_adaptive_body(const ProblemDesc* __restrict__ Pp, ChainState S,
                                                RunDesc R, int64_t max_iters, int plain,
                                                int mode, int64_t group) {
  static_assert(SPLIT || !PERSIST, "the persistent kernel is a split-mode kernel");
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  // wave slot -> chain (ChainState::slot_chain: the chains still walking, packed)
  // (PERSIST: the one wave of this workgroup that got here is the master of chain blockIdx.y)
  // (PERSIST: the chain's master wave of k_persist - chain blockIdx.y in the per-chain form,
  // wave w of the master workgroup of chain group `group` in the tile-sliced form)
  const int64_t slot = group >= 0 ? group * kWavesPerGroup + w
                                  : (PERSIST ? (int64_t)blockIdx.y
                                             : (int64_t)blockIdx.x * kWavesPerGroup + w);
  const bool in_range = slot < (S.slot_chain ? S.n_slots : S.n_chains);
  const int64_t mapped =
      in_range && S.slot_chain ? (int64_t)__builtin_amdgcn_readfirstlane(S.slot_chain[slot]) : slot;
  const bool valid = in_range && mapped >= 0;  // (-1: an empty slot of the map)
  const int64_t c = valid ? mapped : 0;
  ChainRegs r;
  r.status = MHX_CHAIN_DONE;
  if (valid) chain_load(S, c, d, r);
  LoopPhases ph = {0, 0, 0};
  if (PERSIST && valid) ph.set(r.loop_i, R.sts);
  (void)ph;
  const bool cur_in_lds = d <= kCurParams;
  if (valid && cur_in_lds && lane_id() < d) lds.cur[w][lane_id()] = S.theta[c * d + lane_id()];
  double* Lc = S.L + (valid ? c : 0) * d * d;
  const uint64_t gchain = (uint64_t)(S.chain_offset + c);
  Ring ring;
  ring.prob = S.hist_prob + (valid ? c : 0) * S.R;
  ring.theta = S.hist_theta + (valid ? c : 0) * S.R * d;
  ring.mask = S.R - 1;
  ring.d = d;
  int* fwd = S.fwd_idx + (valid ? c : 0) * R.sts;
  double* covs = S.mat_tmp + (valid ? c : 0) * 2 * d * d;
  double* lnew = covs + d * d;
  const int64_t age0 = valid ? r.age : 0;
#ifdef MHX_X_TIMING
  if (l == 0) {
    for (int k = 0; k < 8; ++k) lds.tim[w][k] = 0;
    lds.tlast[w] = __builtin_readcyclecounter();
  }
#endif

  // PERSIST: the next iteration's random numbers are drawn while this one's partial sums are on
  // their way (the draw counter does not depend on the accept decision), and the proposal factor
  // waits in LDS (the master's workgroup stages no tiles), re-read after every change
  double rv_pre = 0.0, lg_pre = 0.0, lz_pre = 0.0;
  if constexpr (PERSIST) MHX_PERSIST_PRIO(3);  // (a master's controller: see persist_poll)
#ifdef MHX_PERSIST_TIMING
  unsigned long long pt_ctrl = 0, pt_wait = 0, pt_n = 0, pt_last = __builtin_readcyclecounter();
#endif
  bool have_pre = false, have_lz = false, l_stale = true, stop_sent = false, early = false;
  (void)early;
  (void)lz_pre; (void)have_lz;
  (void)stop_sent;
  // (one copy per master wave of the workgroup where they all fit, else the factor stays in HBM)
  const bool l_in_lds = PERSIST && (group < 0 ? 1 : kWavesPerGroup) * d * d * sizeof(double) <= sizeof(lds.tiles);
  double* const l_lds = &lds.tiles[0][0][0] + (group < 0 ? 0 : w * d * d);
  static_assert(!PERSIST || sizeof(lds.tiles) >= sizeof(double) * kPersistMaxParams * kPersistMaxParams,
                "the proposal factor of the persistent kernel's master waits in the tile buffers");
  (void)rv_pre; (void)lg_pre; (void)have_pre; (void)l_stale; (void)l_lds; (void)l_in_lds;
  for (int64_t it = 0; (SPLIT && !PERSIST) || it < max_iters; ++it) {
    MHX_TIM(lds, 6);
    MHX_TIMC(lds, 7);
    const int l = lane_id();  // (formed per iteration: see lane_id())
    int d_it = P.d;  // likewise the parameter count: what the cold code derives from it (the
    asm volatile("" : "+s"(d_it));  // reciprocal behind e / d ...) is formed where it is used
    const int d = d_it;
    ring.d = d;
    Lc = fresh_ptr(Lc);
    ring.prob = fresh_ptr(ring.prob);
    ring.theta = fresh_ptr(ring.theta);
    fwd = fresh_ptr(fwd);
    covs = fresh_ptr(covs);
    lnew = fresh_ptr(lnew);
    S.hist_theta = fresh_ptr(S.hist_theta);
    S.theta = fresh_ptr(S.theta);
    S.best_theta = fresh_ptr(S.best_theta);
    S.hist_prob = fresh_ptr(S.hist_prob);
    S.L_pool = fresh_ptr(S.L_pool);
    S.split_prop = fresh_ptr(S.split_prop);
    bool running = valid && r.status == MHX_CHAIN_RUNNING;
    double thp = 0.0, u = 1.0;
    bool resumed = false;
    if constexpr (SPLIT && !PERSIST) {
      resumed = it == 0 && mode != 0;
      if (!resumed && (it > 1 || (it > 0 && mode == 2))) break;
    }
    if (resumed) {
      // second half of the iteration whose proposal an earlier launch left behind
      running = running && __builtin_amdgcn_readfirstlane(S.split_pending[valid ? c : 0]) != 0;
      if (running) {
        thp = l < d ? S.split_prop[c * d + l] : 0.0;
        u = uniform_f64(S.split_u[c]);
        if (l < d) lds.prop[w][l] = thp;
        if (l == 0) S.split_pending[c] = 0;
      }
    } else {
    if (running) {
      if (r.loop_i >= R.n) {  // end test of the do loop, M:904
        r.status = MHX_CHAIN_DONE;
        running = false;
      } else if ((it & 15) == 0 &&  // looked at every 16th iteration: one L2 round trip less
                 __builtin_amdgcn_readfirstlane(*(volatile const int*)R.stop_flag)) {
        r.status = MHX_CHAIN_STOPPED;  // mfit-walker-estop
        running = false;
      }
    }
    if constexpr (SPLIT) {
      if (valid && l == 0) S.split_pending[c] = 0;
    }
    if constexpr (PERSIST) {
      // a chain that has stopped walking says so at once: its sweep waves (tile-sliced form: the
      // group's other chains go on) must not wait for a proposal that will never come
      if (valid && !running && !stop_sent) {
        persist_publish(S, c, 0, 0.0, kPersistStop);
        stop_sent = true;
      }
    }
    // (the vote is the same in every lane; as a scalar it keeps the loop's exit - and so every
    // counter of the chain that lives across it - out of the vector registers)
    // One barrier and a flag per iteration (the library's __syncthreads_or is a workgroup
    // reduction with two): a running wave raises flag it % 3, everybody reads it after the
    // barrier, and the flag of iteration it + 2 - last read before the previous barrier, next
    // written after the next one - is lowered.
    if constexpr (PERSIST) {
      // (a master wave walks alone: nothing in this loop is shared between the waves of its
      // workgroup - no tiles, the factor and the proposal in per-wave LDS - so a chain that has
      // ended leaves, and nobody waits at a barrier for the slowest of eight chains)
      if (!running) break;
    } else {
      const int vp = (int)(it % 3);
      if (running && l == 0) lds.vote[vp] = 1;
      __syncthreads();
      const int any = __builtin_amdgcn_readfirstlane(*(volatile int*)&lds.vote[vp]);
      if (threadIdx.x == 0) lds.vote[(vp + 2) % 3] = 0;
      if (!any) break;
    }
    if (running) {
      if (!plain) {
        // M:905-917
        bool shut = !r.shutting && (R.n - r.loop_i) < R.tail;
        if (!shut && R.auto_mode && !r.shutting &&
            (PERSIST ? ph.m1000 == 0 : floor_mod(r.loop_i, 1000) == 0) && r.loop_i > 2 * R.sts) {
          __threadfence();
          ring.nh = r.nh;
          ring.length = r.length;
          int num, den;
          ring_acceptance(ring, 1000, &num, &den);
          if (acc_gt(num, den, 0.2f) && acc_lt(num, den, 0.5f) &&
              ring_stable_probs(ring, (int)R.sts))
            shut = true;
        }
        if (shut) {
          r.T = 1.0;
          r.shutting = 1;
          r.loop_i = R.n - R.tail;
          if constexpr (PERSIST) ph.set(r.loop_i, R.sts);
        }
      }
      // M:918 walker-take-step: proposal
      MHX_TIMC(lds, 0);
      double lg, rv;
      if (PERSIST && have_pre) {
        rv = rv_pre;
        lg = lg_pre;
      } else {
        rv = rng_lane_value(S.seed, gchain, r.draw, d, &lg);
      }
      r.draw++;
      // (the fused kernel carries log u to the accept test, split mode u itself)
      u = (SPLIT && !PERSIST) ? readlane_f64(rv, 63) : readlane_f64(lg, 63);
      MHX_TIMC(lds, 1);
      if (PERSIST && l_in_lds) {
        if (l_stale) {
          for (int e = l; e < d * d; e += kWave) l_lds[e] = Lc[e];
          __builtin_amdgcn_wave_barrier();
          l_stale = false;
          have_lz = false;  // (formed with the old factor)
        }
        const double lz = have_lz ? lz_pre : propose_lz(l_lds, d, rv);
        thp = lz + (cur_in_lds ? (l < d ? lds.cur[w][l] : 0.0) : chain_theta(S, c, d));
      } else {
        thp = propose(Lc, d, rv, cur_in_lds ? (l < d ? lds.cur[w][l] : 0.0) : chain_theta(S, c, d));
      }
      if (l < d) lds.prop[w][l] = thp;
      MHX_TIMC(lds, 2);
    }
    if constexpr (SPLIT && !PERSIST) {  // hand the proposal to the sweep launch and stop here
      if (running) {
        if (l < d) S.split_prop[c * d + l] = thp;
        if (l == 0) {
          S.split_u[c] = u;
          S.split_pending[c] = 1;
        }
      }
      break;
    }
    if constexpr (PERSIST) {  // ... to the chain's sweep workgroups of this same launch
      if (running) {
        if (!early) persist_publish(S, c, d, thp, (unsigned long long)(it + 1));
        early = false;
        rv_pre = rng_lane_value(S.seed, gchain, r.draw, d, &lg_pre);  // (the next iteration's)
        have_pre = true;
        if (l_in_lds && !l_stale) {  // ... and its L z (the factor changes only at adaptation ticks)
          lz_pre = propose_lz(l_lds, d, rv_pre);
          have_lz = true;
        }
      }
    }
    }  // !resumed
    double ll, lp;
    double prob1;
    // the chain's scalars wait in LDS while the sweep runs: nothing of the chain is alive across
    // it in registers, so the sweep has the whole register file and nothing goes to scratch
    static_assert(sizeof(ChainPark) <= sizeof(lds.park[0]), "ChainPark slot");
    ChainPark& slot = *reinterpret_cast<ChainPark*>(lds.park[w]);
    MHX_TIM(lds, 0);
    // the temperature the annealing step below will set (M:920-921), fetched now: the round
    // trip to the schedule overlaps the sweep instead of following it
    double t_next = r.T;
    if (running && !plain && !r.shutting && r.loop_i < R.temp_steps)
      t_next = uniform_f64(R.temps[r.loop_i - R.temps_first]);
#ifdef MHX_EARLY_REJECT
    if constexpr (!SPLIT) {
      // the threshold of sweep()'s early rejection: accept needs lik_const - S/2 + log-prior >
      // prob0 + T log u, the log-prior of bounds is <= 0 (M:353-369), S only grows
      double thr = __builtin_inf();
      const FnDesc& f0 = P.fn[0];
      if (running && P.K == 1 && f0.lik == MHX_LIK_NORMAL && f0.prior_slot < 0) {
        const double lim = 2.0 * (f0.lik_const - r.prob0 - r.T * u);
        thr = lim + 1e-9 * (__builtin_fabs(lim) + __builtin_fabs(f0.lik_const) + __builtin_fabs(r.prob0)) + 1e-6;
      }
      if (l == 0) lds.deal_ll[w] = thr;
    }
#endif
#ifndef MHX_NO_PARK  // (build knob for A/B measurements)
    if constexpr (!SPLIT) chain_park(slot, r, u, t_next);
#endif
    if constexpr (PERSIST) {
      bool lost = false;
      MHX_TIMC(lds, 7);  // (persistent master: publishing + the draws made ahead)
#ifdef MHX_PERSIST_TIMING  // (measurement build: the master's iteration = controller + wait)
      const unsigned long long tw0 = __builtin_readcyclecounter();
      pt_ctrl += tw0 - pt_last;
#endif
      prob1 = split_logpost<Spec, true>(P, S, c, running, lds, w, &ll, &lp,
                                        (unsigned long long)(it + 1), &lost);
#ifdef MHX_PERSIST_TIMING
      pt_last = __builtin_readcyclecounter();
      pt_wait += pt_last - tw0;
      ++pt_n;
#endif
      if (__builtin_amdgcn_readfirstlane((int)lost)) {
        // the sweep workgroups did not answer (not all of them were on the GPU): nothing was
        // judged - the proposal is taken back (the next launch draws it again) and this chain's
        // part of the launch ends; the host hears about it (persist_error)
        r.draw--;
        if (l == 0) __hip_atomic_store(S.persist_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    } else if constexpr (SPLIT)
      prob1 = split_logpost<Spec>(P, S, c, running, lds, w, &ll, &lp);
    else
      prob1 = group_logpost<Spec>(P, running, lds, w, &ll, &lp);
#ifndef MHX_NO_PARK
    if constexpr (!SPLIT) chain_unpark(slot, r, u, t_next);
#endif
    MHX_TIM(lds, 7);
    MHX_TIMC(lds, 3);
    // the butterfly leaves the same bits in every lane: scalar from here on, and with it the
    // accept decision and every counter of the chain changed under it
    prob1 = uniform_f64(prob1);
    if (!running) continue;
    bool er_lost = false;
#ifdef MHX_EARLY_REJECT  // (the sweep left early: -inf stands for "rejected", not for a trap)
    if constexpr (!SPLIT)
      er_lost = __builtin_amdgcn_readfirstlane((int)(*(volatile double*)&lds.deal_ll[w] == -__builtin_inf())) != 0;
#endif
    if (!finite_f64(prob1) && !er_lost) {
      r.status = MHX_CHAIN_FP_TRAP;
      continue;
    }
    {
      // the proposal is still in the wave's LDS slot (nothing of it was kept in registers
      // across the sweep)
      const bool take =
          __builtin_amdgcn_readfirstlane(
              (int)((SPLIT && !PERSIST) ? mh_accept(prob1, r.prob0, r.T, u)
                                        : mh_accept_log(prob1, r.prob0, r.T, u))) != 0;
      if (take) r.prob0 = prob1;
      MHX_TIMC(lds, 4);
      const double th_now = take ? (l < d ? lds.prop[w][l] : 0.0)
                                 : (cur_in_lds ? (l < d ? lds.cur[w][l] : 0.0) : chain_theta(S, c, d));
      if (take && cur_in_lds && l < d) lds.cur[w][l] = th_now;
      if constexpr (PERSIST) {
        // The sweep workgroups wait for the next proposal, and on most iterations everything it
        // is made of is known here: the position (th_now), the draw made ahead and its L z.  It
        // goes out NOW, before :add-step, annealing, the adaptation test and the loop's top -
        // which then run while the sweeps work.  Only where none of them can change the
        // proposal or end the walk: no adaptation tick at this index, no shut-down or
        // acceptance look at the next, not the launch's or the walk's last iteration, not an
        // iteration that looks at the stop flag.  The loop's top still forms the proposal its
        // own way (the same operands, the same bits) and only skips the publishing.
        const int64_t i1 = r.loop_i + 1;
        const bool tick = r.loop_i > 0 && (ph.m200 == 0 || (!r.shutting && ph.msts == 0));
        const bool look = !r.shutting && ((R.n - i1) < R.tail ||
                                          (R.auto_mode && ph.m1000 == 999 && i1 > 2 * R.sts));
        if (have_pre && have_lz && !l_stale && it + 1 < max_iters && i1 < R.n && ((it + 1) & 15) != 0 &&
            (plain || (!tick && !look))) {
          persist_publish(S, c, d, lz_pre + th_now, (unsigned long long)(it + 2));
          early = true;
        }
      }
      add_step(S, c, d, r, th_now, take);
      MHX_TIMC(lds, 5);
    }
    if (plain) {
      r.loop_i++;  // (nothing under `plain` looks at the loop index's phases)
      continue;
    }
    const int64_t i = r.loop_i;
    // annealing M:920-921
    if (!r.shutting && i < R.temp_steps) r.T = t_next;
    // cleaning M:923-927 (:keep-walks only shortens what :take can see)
    if (R.has_mwl && i == r.reset_index) {
      if (r.length > R.mwl) {
        r.length = R.mwl;
        r.reset_index += R.mwl + 1;
      } else {
        r.reset_index += 1 + (R.mwl - r.length);
      }
    }
    // regular l-matrix updating M:929-942
    if (i > 0) {
      const bool m200 = PERSIST ? ph.m200 == 0 : floor_mod(i, 200) == 0;
      const bool msts = !r.shutting && (PERSIST ? ph.msts == 0 : floor_mod(i, 2 * R.sts) == 0);
      if (m200 || msts) {
        l_stale = true;  // (PERSIST: the factor may change below)
        __threadfence();
        ring.nh = r.nh;
        ring.length = r.length;
        int num, den;
        ring_acceptance(ring, 200, &num, &den);
        if ((m200 && acc_lt(num, den, 0.2f)) || (m200 && acc_gt(num, den, 0.4f)) || msts) {
          if (acc_gt(num, den, 0.2f) && acc_lt(num, den, 0.4f)) {
            int nf;
            int st;
            if (R.adapt_mode == MHX_ADAPT_POOLED && *(volatile const int*)S.pool_valid != 0) {
              // extension: the factor of the covariance pooled over all chains and ranks
              // (already scaled by 2.38^2/d) replaces the walker's own estimate
              for (int e = l; e < d * d; e += kWave) Lc[e] = S.L_pool[e];
              st = L_EMPTY;  // nothing more to do below
            } else {
              st = ring_l_matrix(ring, (int)R.sts, fwd, covs, lnew, (lds_dptr_t)lds.prop[w], &nf);
            }
            // (/ (expt 2.38d0 2) num-params) M:890, formed here rather than once per launch: a
            // value kept for this cold block would sit in a vector register across every sweep
            int dq = d;
            asm volatile("" : "+s"(dq));
            const double factor = (2.38 * 2.38) / (double)dq;
            if (st == L_OK) {
              for (int e = l; e < d * d; e += kWave) Lc[e] = factor * lnew[e];
            } else if (st == L_CAUGHT) {  // handler-case returns the CURRENT l-matrix, which
              for (int e = l; e < d * d; e += kWave) Lc[e] = factor * Lc[e];  // scale-array mutates
            } else if (st == L_INVALID) {
              r.status = MHX_CHAIN_FP_TRAP;
            }  // L_EMPTY: 0x0 matrix fails the dimension test of M:936, L kept
          } else if (acc_lt(num, den, 0.2f)) {
            for (int e = l; e < d * d; e += kWave) Lc[e] = 0.1 * Lc[e];
          } else if (acc_gt(num, den, 0.4f)) {
            for (int e = l; e < d * d; e += kWave) Lc[e] = 1.9 * Lc[e];
          }
          __threadfence();
        }
      }
    }
    MHX_TIMC(lds, 6);
    if (r.status == MHX_CHAIN_RUNNING) {
      r.loop_i++;
      if constexpr (PERSIST) ph.step(R.sts);
    }
  }
#ifdef MHX_PERSIST_TIMING
  if (PERSIST && w == 0 && l == 0 && pt_n) persist_trace(pt_ctrl, pt_wait, 0, pt_n);
#endif
  if (valid) {
    if (r.status == MHX_CHAIN_RUNNING && r.loop_i >= R.n) r.status = MHX_CHAIN_DONE;
    chain_store(S, c, d, r);
    if (l == 0 && r.age != age0) atomicAdd(S.step_counter, (unsigned long long)(r.age - age0));
#ifdef MHX_X_TIMING
    if (l < 8 && l < d) {  // (d < 8: the last slot takes the rest)
      unsigned long long v = lds.tim[w][l];
      if (l == d - 1)
        for (int k = d; k < 8; ++k) v += lds.tim[w][k];
      S.best_theta[c * d + l] = (double)v;
    }
#endif
  }
}

// split mode, launch 1 of a step: block (slice g, chain c); wave w of it is slot g * waves + w
template <class Spec>
__device__ __forceinline__ void k_split_sweep_body(const ProblemDesc* __restrict__ Pp,
                                                   ChainState S) {
  SweepLds& sl = *reinterpret_cast<SweepLds*>(mhx_lds_raw);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  const int64_t c = blockIdx.y;
  if (__builtin_amdgcn_readfirstlane(S.split_pending[c]) == 0) return;  // nothing to judge
  // the tables of tlog() (Poisson terms, log() in user expressions) and mexp2_negsq() lead this
  // kernel's dynamic LDS (SweepLds) as they lead every other kernel's
  lds_tables_begin();
  __syncthreads();
  const int slot = (int)blockIdx.x * kWavesPerGroup + w;
  if (l < d) sl.prop[w][l] = S.split_prop[c * d + l];
  const double* th = sl.prop[w];
  for (int k = 0; k < P.K; ++k) {
    const FnDesc& f = P.fn[k];
    auto pf = [&](int j) -> double { return th[f.idx[j]]; };
    // contiguous chunks of whole 128-point pairs; the arrays are padded past n (mhx_types.hpp)
    const int64_t pairs = (f.n + 2 * kWave - 1) / (2 * kWave);
    const int64_t per = (pairs + S.split_slots - 1) / S.split_slots;
    const int64_t b0 = (int64_t)slot * per, b1 = b0 + per < pairs ? b0 + per : pairs;
    double v = 0.0;
    if (b0 < b1) v = Spec::loglik_part(f, pf, b0 * 2 * kWave, b1 * 2 * kWave, sl.scr[w]);
    if (l == 0) S.split_part[(c * P.K + k) * S.split_slots + slot] = v;
  }
}
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_split_sweep(const ProblemDesc* __restrict__ Pp,
                                                          ChainState S) {
  k_split_sweep_body<Spec>(Pp, S);
}
// split mode for 8 ... a few thousand chains ("tile-sliced"): block (slice g, group of chains b).
// The 8 / 16 chains of a group walk the SAME slice of every function - whole windows, described
// by a FnDesc of its own (the host's slice table: shifted pointers, its own n and n_tiles,
// lik_const 0) - through the batch kernels' sweep(): LDS tiles shared by the group, peak
// skipping, recurrence, yw tiles.  Where k_split_sweep has every wave read its points from the
// L2 for itself (one chain's points over many workgroups: right for a handful of chains, bound by
// the L2 from a few dozen on), this reads each point once per group.  The partial sums go where
// k_split_sweep's go (slot = slice) and the step kernel adds them in slot order and finishes the
// likelihood with the function's own constant.
template <class Spec>
__device__ __forceinline__ void k_split_tsweep_body(const ProblemDesc* __restrict__ Pp,
                                                    const FnDesc* __restrict__ slices,
                                                    ChainState S, int n_slices) {
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  const int slice = (int)blockIdx.x;
  // wave slot -> chain (ChainState::slot_chain: the chains still walking, packed into groups)
  const int64_t slot = (int64_t)blockIdx.y * kWavesPerGroup + w;
  const bool in_range = slot < (S.slot_chain ? S.n_slots : S.n_chains);
  const int64_t mapped =
      in_range && S.slot_chain ? (int64_t)__builtin_amdgcn_readfirstlane(S.slot_chain[slot]) : slot;
  const bool valid = in_range && mapped >= 0;
  const int64_t c = valid ? mapped : 0;
  const bool active = valid && __builtin_amdgcn_readfirstlane(S.split_pending[c]) != 0;
  // nothing to judge in this group?  (a flag of the dynamic LDS: __syncthreads_or() keeps a static
  // __shared__ word, and static LDS would sit in front of the math tables - mhx_device.hpp)
  if (active && l == 0) lds.vote[2] = 1;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(*(volatile int*)&lds.vote[2]) == 0) return;
  if (l < d) lds.prop[w][l] = active ? S.split_prop[c * d + l] : 0.0;
  __syncthreads();
  const double* th = lds.prop[w];
  for (int k = 0; k < P.K; ++k) {
    const FnDesc& f = slices[k * n_slices + slice];
    auto pf = [&](int j) -> double { return th[f.idx[j]]; };
    const double v = Spec::loglik(f, pf, active, lds, lds.prm[w]);
    // loglik() finishes the sum with the slice's constant, 0: -s/2 (exact) for the normal
    // likelihood, s for the others.  The step kernel finishes the whole sum: hand it s.
    const double raw = f.lik == MHX_LIK_NORMAL ? -2.0 * v : v;
    if (active && l == 0) S.split_part[(c * P.K + k) * S.split_slots + slice] = raw;
  }
}
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_split_tsweep(const ProblemDesc* __restrict__ Pp,
                                                           const FnDesc* __restrict__ slices,
                                                           ChainState S, int n_slices) {
  k_split_tsweep_body<Spec>(Pp, slices, S, n_slices);
}
// split mode, launch 2 of a step (and the priming / closing launches): see k_adaptive_body
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_split_step(const ProblemDesc* __restrict__ Pp,
                                                         ChainState S, RunDesc R, int mode,
                                                         int plain) {
  k_adaptive_body<Spec, true>(Pp, S, R, 1, plain, mode);
}

// PERSISTENT split mode: a handful of chains (fewer than a workgroup's worth) on long datasets, the
// reference's own way of working (test.lisp:12-24 fits ONE walker).  The two-launch split mode
// pays two kernel launches and a dozen dependent global round trips per iteration - 16 us per
// step on 1e5 points, 5 of them arithmetic.  Here ONE launch runs max_iters iterations: grid
// (1 + slices, chains) workgroups, all of them resident at once (the host checks); workgroup
// (0, c) keeps only its wave 0, the chain's master, which runs the controller of
// k_adaptive_body; workgroups (1 .., c) are the chain's sweep workgroups, wave w of workgroup g the
// slot (g - 1) * waves + w of k_split_sweep - same slots, same points, same order of the sum: the
// same bits as the two-launch mode.  Per iteration: the master publishes the proposal and a
// generation number (release, agent scope), the sweep workgroups - polling it - compute their
// partial sums and count themselves in (a fence and one atomic per workgroup), the master -
// polling that count - judges the proposal.  Every poll loop gives up after kPersistPatience
// polls (the master then freezes the chain and raises the stop word, which ends the sweep
// workgroups): no wave can spin for ever.
// the master's next word for chain c: the proposal of generation `round + 1` (lane q of *word: the
// q-th word of the chain's block), or stop.  false: stop, or patience ran out.
__device__ __forceinline__ bool persist_poll(const ChainState& S, int64_t c, int d,
                                             unsigned long long round, unsigned long long* word) {
  const unsigned long long* msg = S.persist_msg + c * 64;
  const int l = lane_id();
  const int nlines = (d + 14) / 15;  // lines of the block that carry parameters
  unsigned long long q = 0;
  bool ok = false;
  // A waiting wave must not take issue slots from a working one (VALU issue goes to the highest
  // priority, then to the oldest wave; two workgroups of the launch share most CUs): waiting is
  // done at priority 0, the masters' controllers - the serial part of every round - at 3, the
  // sweeps in between (tile_prio moves within 0 ... 3 inside a tile).  Same box, us per
  // iteration on 1e5 points with | without: 64 walkers 11.95 | 12.65, 128: 14.1 | 15.2,
  // 256: 19.2 | 20.2, 512: 30.9 | 32.4 (8 and 16 walkers, one workgroup per CU: 9.7 | 9.8).
  MHX_PERSIST_PRIO(0);
  for (unsigned n = 0; n < kPersistPatience; ++n) {
    // (only the lines that carry parameters are read: a poll is a request to the level all XCDs
    // share, and thousands of waves poll)
    q = (l >> 4) < nlines ? __hip_atomic_load(msg + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const bool is_tag = (l & 15) == 15 && (l >> 4) < nlines;
    unsigned f = (l & 15) != 15 ? persist_fold(q) : 0u;
    f = persist_row_xor(f);
    // (lane 16 i + 15 now holds the fold of line i's data words - its own contribution was 0)
    const unsigned gen = (unsigned)q;
    const bool fresh = gen > (unsigned)round && (unsigned)(q >> 32) == f;
    const unsigned long long bad = __ballot(is_tag && !fresh);
    if (__builtin_amdgcn_readfirstlane((int)(bad == 0ull))) {
      ok = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  const unsigned gen0 = (unsigned)__builtin_amdgcn_readlane((int)q, 15);
  *word = q;
  MHX_PERSIST_PRIO(1);
  return ok && !(gen0 & (unsigned)kPersistStop);
}
// TS = false: a handful of chains, grid (1 + slices, chains): workgroup (0, c) keeps only its wave
// 0, the chain's master; workgroups (1 .., c) are the chain's sweep workgroups, wave w of workgroup
// g the slot (g - 1) * waves + w of k_split_sweep.
// TS = true: the tile-sliced mode (8 ... a few hundred chains), grid (1 + slices, groups): the 8
// waves of workgroup (0, b) are the masters of chain group b's chains; workgroup (1 + g, b) walks
// slice g with the group's proposals through sweep() exactly as k_split_tsweep does - LDS tiles
// shared by the group, skipping, recurrence - round after round: every wave waits for ITS
// chain's next proposal, the sweep's own barriers bring the group together, every wave hands its
// chain's partial sum back as a {sum, round} pair.  A chain whose master has said stop leaves its
// wave helping with the tiles (active = false) until the group's last chain is through.
template <class Spec, bool TS>
__device__ __forceinline__ void k_persist_body(const ProblemDesc* __restrict__ Pp,
                                               const FnDesc* __restrict__ slices, ChainState S,
                                               RunDesc R, int n_slices, int64_t max_iters,
                                               int plain) {
  const ProblemDesc& P = *Pp;
  const int w = wave_in_group(), l = lane_id(), d = P.d;
  // this wave's chain: blockIdx.y itself, or through the group's slot -> chain map
  int64_t c = blockIdx.y;
  bool valid = true;
  if constexpr (TS) {
    const int64_t slot = (int64_t)blockIdx.y * kWavesPerGroup + w;
    const bool in_range = slot < (S.slot_chain ? S.n_slots : S.n_chains);
    const int64_t mapped =
        in_range && S.slot_chain ? (int64_t)__builtin_amdgcn_readfirstlane(S.slot_chain[slot]) : slot;
    valid = in_range && mapped >= 0;
    c = valid ? mapped : 0;
  }
  if (blockIdx.x != 0 && P.test_lose_sweepers != 0) return;  // (tests: the masters are left alone)
  if (blockIdx.x == 0) {
    if (!TS && w != 0) return;  // (s_barrier counts the waves still alive)
    k_adaptive_body<Spec, true, true>(Pp, S, R, max_iters, plain, 1, TS ? (int64_t)blockIdx.y : -1);
    if (valid) persist_publish(S, c, 0, 0.0, kPersistStop);
    return;
  }
  if constexpr (!TS) {
    SweepLds& sl = *reinterpret_cast<SweepLds*>(mhx_lds_raw);
    lds_tables_begin();
    __syncthreads();
    const int slot = ((int)blockIdx.x - 1) * kWavesPerGroup + w;
    // Every wave polls for itself (no barrier in this loop: the waves of a sweep workgroup run
    // independently).
    for (unsigned long long it = 0;; ++it) {
      unsigned long long q;
      if (!persist_poll(S, c, d, it, &q)) return;
      if ((l & 15) != 15 && (l >> 4) < (d + 14) / 15)
        sl.prop[w][(l >> 4) * 15 + (l & 15)] = __longlong_as_double((long long)q);
      __builtin_amdgcn_wave_barrier();
      const double* th = sl.prop[w];
      for (int k = 0; k < P.K; ++k) {
        const FnDesc& f = P.fn[k];
        auto pf = [&](int j) -> double { return th[f.idx[j]]; };
        const int64_t pairs = (f.n + 2 * kWave - 1) / (2 * kWave);
        const int64_t per = (pairs + S.split_slots - 1) / S.split_slots;
        const int64_t b0 = (int64_t)slot * per, b1 = b0 + per < pairs ? b0 + per : pairs;
        double v = 0.0;
        if (b0 < b1) v = Spec::loglik_part(f, pf, b0 * 2 * kWave, b1 * 2 * kWave, sl.scr[w]);
        if (l == 0)
          persist_store_pair((char*)S.persist_part + (((size_t)c * P.K + k) * S.split_slots + slot) * 16, v,
                             it + 1);
      }
    }
  } else {
    GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
    lds_begin(lds);
    const int slice = (int)blockIdx.x - 1;
    bool alive = valid;
    // One function whose slice is a resident window (build_ts_table): after the first round -
    // the tiles fetched together, behind barriers - the eight waves of the workgroup share
    // nothing but read-only LDS, and each follows its own chain: polls, sweeps, answers, leaves
    // when the chain ends.  (Otherwise the waves stage tiles together round after round.)
    const bool free_run = P.K == 1 && slices[slice].solo != 0 && slices[slice].n_tiles > 0;
#ifdef MHX_PERSIST_TIMING  // (measurement build: where a sweep workgroup's round goes)
    unsigned long long t_poll = 0, t_vote = 0, t_sweep = 0, t0 = __builtin_readcyclecounter(), rounds = 0;
#define MHX_PT(acc) do { const unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - t0; t0 = n_; } while (0)
#else
#define MHX_PT(acc) do { } while (0)
#endif
    for (unsigned long long round = 0;; ++round) {
      unsigned long long q = 0;
      if (alive) alive = persist_poll(S, c, d, round, &q);
      MHX_PT(t_poll);
      const bool active = alive;
      if (active && (l & 15) != 15 && (l >> 4) < (d + 14) / 15)
        lds.prop[w][(l >> 4) * 15 + (l & 15)] = __longlong_as_double((long long)q);
      int any;
      if (free_run && round > 0) {
        // (the window is in LDS: this wave's rounds need nothing of the other waves any more)
        any = active ? 1 : 0;
      } else {
        // anybody of the group still walking?  (a flag and ONE barrier, as in k_adaptive_body)
        const int vp = (int)(round % 3);
        if (active && l == 0) lds.vote[vp] = 1;
        __syncthreads();
        any = __builtin_amdgcn_readfirstlane(*(volatile int*)&lds.vote[vp]);
        if (threadIdx.x == 0) lds.vote[(vp + 2) % 3] = 0;
      }
      MHX_PT(t_vote);
#ifdef MHX_PERSIST_TIMING
      if (!any && threadIdx.x == 0) persist_trace(t_poll, t_vote, t_sweep, rounds);
#ifdef MHX_X_TIMING
      if (!any && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0 && rounds)
        printf("sweep wg (1,0) wave 0, MHX_TIM phases 0-7, cycles per round: %llu %llu %llu %llu %llu %llu %llu %llu\n",
               lds.tim[0][0] / rounds, lds.tim[0][1] / rounds, lds.tim[0][2] / rounds, lds.tim[0][3] / rounds,
               lds.tim[0][4] / rounds, lds.tim[0][5] / rounds, lds.tim[0][6] / rounds, lds.tim[0][7] / rounds);
#endif
      ++rounds;
#ifdef MHX_X_TIMING
      if (round == 0 && l == 0) { for (int k = 0; k < 8; ++k) lds.tim[w][k] = 0; }
      if (l == 0) lds.tlast[w] = __builtin_readcyclecounter();
#endif
#endif
      if (!any) return;
      const double* th = lds.prop[w];
      for (int k = 0; k < P.K; ++k) {
        const FnDesc& f = slices[k * n_slices + slice];
        auto pf = [&](int j) -> double { return th[f.idx[j]]; };
        const double v = Spec::loglik(f, pf, active, lds, lds.prm[w]);
        const double raw = f.lik == MHX_LIK_NORMAL ? -2.0 * v : v;  // (as k_split_tsweep)
        if (active && l == 0)
          persist_store_pair((char*)S.persist_part + (((size_t)c * P.K + k) * S.split_slots + slice) * 16,
                             raw, round + 1);
      }
      MHX_TIM(lds, 6);
      MHX_PT(t_sweep);
    }
  }
}
template <class Spec>
__global__ __launch_bounds__(kThreads) void k_persist(const ProblemDesc* __restrict__ Pp,
                                                      ChainState S, RunDesc R, int64_t max_iters,
                                                      int plain) {
  k_persist_body<Spec, false>(Pp, nullptr, S, R, 0, max_iters, plain);
}
// (registers for four waves per SIMD, as k_adaptive: two workgroups of the w8 family on a CU -
// without the bound the two-peak kernel took 132 VGPRs, ONE workgroup fitted a CU, and the 400
// workgroups of a 64-walker launch ran as two shifts: twice the time, found in the per-workgroup
// trace of tools/debug/persist_ts_timing.py)
template <class Spec>
__global__ __launch_bounds__(kThreads, 4) void k_persist_ts(const ProblemDesc* __restrict__ Pp,
                                                         const FnDesc* __restrict__ slices,
                                                         ChainState S, RunDesc R, int n_slices,
                                                         int64_t max_iters, int plain) {
  k_persist_body<Spec, true>(Pp, slices, S, R, n_slices, max_iters, plain);
}

// Initial L of M:896-901 when the caller gave none.
__global__ __launch_bounds__(kThreads) void k_initial_l(ChainState S, RunDesc R, int have_l,
                                                        double T0) {
  GroupLds& lds = *reinterpret_cast<GroupLds*>(mhx_lds_raw);
  lds_begin(lds);
  const int w = wave_in_group(), l = lane_id(), d = S.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  if (c >= S.n_chains) return;
  double* Lc = S.L + c * d * d;
  Ring ring;
  ring.prob = S.hist_prob + c * S.R;
  ring.theta = S.hist_theta + c * S.R * d;
  ring.mask = S.R - 1;
  ring.d = d;
  ring.nh = uniform_i64(S.n_hist[c]);
  ring.length = uniform_i64(S.length[c]);
  if (have_l) {  // :l-matrix given: M:896 skips the whole block
    if (l == 0) {
      if (S.status[c] != MHX_CHAIN_FP_TRAP) S.status[c] = MHX_CHAIN_RUNNING;
      S.loop_i[c] = 1;
      S.shutting[c] = 0;
      S.reset_index[c] = 10000;
      S.temperature[c] = T0;
    }
    return;
  }
  // (diagonal-covariance (get-plist-values most-likely-params)) M:899
  for (int e = l; e < d * d; e += kWave) {
    const int i = e / d, j = e - i * d;
    Lc[e] = i == j ? S.best_theta[c * d + i] : 0.0;
  }
  int num, den;
  ring_acceptance(ring, 100, &num, &den);
  int status = MHX_CHAIN_RUNNING;
  if (!(ring.length < R.sts || acc_lt(num, den, 0.1f))) {
    __threadfence();
    int* fwd = S.fwd_idx + c * R.sts;
    double* covs = S.mat_tmp + c * 2 * d * d;
    double* lnew = covs + d * d;
    const double factor = (2.38 * 2.38) / (double)d;
    int nf;
    const int st = ring_l_matrix(ring, (int)R.sts, fwd, covs, lnew, (lds_dptr_t)lds.prop[w], &nf);
    if (st == L_OK) {
      for (int e = l; e < d * d; e += kWave) Lc[e] = factor * lnew[e];
    } else if (st == L_CAUGHT) {
      for (int e = l; e < d * d; e += kWave) Lc[e] = factor * Lc[e];
    } else {
      status = MHX_CHAIN_FP_TRAP;  // invalid operation, or a 0x0 l-matrix (M:901 then M:918)
    }
  }
  if (l == 0) {
    if (S.status[c] != MHX_CHAIN_FP_TRAP) S.status[c] = status;
    S.loop_i[c] = 1;  // (do ((i 1 (+ i 1))) ...) M:902
    S.shutting[c] = 0;
    S.reset_index[c] = 10000;
    S.temperature[c] = T0;  // (rational temperature) M:876
  }
}

// (walker-get w :get :l-matrix :take take) of one chain, with the device code the controller
// itself runs.  out: [d*d] factor, info[0] = L_* status, info[1] = (length forward-steps)
__global__ __launch_bounds__(kWave) void k_l_matrix(ChainState S, int64_t c, int take, int* fwd,
                                                    double* cov, double* out, int* info) {
  __shared__ double avg[MHX_MAX_PARAMS];
  Ring ring;
  ring.prob = S.hist_prob + c * S.R;
  ring.theta = S.hist_theta + c * S.R * S.d;
  ring.mask = S.R - 1;
  ring.d = S.d;
  ring.nh = uniform_i64(S.n_hist[c]);
  ring.length = uniform_i64(S.length[c]);
  int nf = 0;
  const int st = ring_l_matrix(ring, take, fwd, cov, out, (lds_dptr_t)avg, &nf);
  if (lane_id() == 0) {
    info[0] = st;
    info[1] = nf;
  }
}

// walker-modify's list surgery on the ring (M:566-578).  action 0 :burn-walks n (drop the n
// oldest steps), 1 :keep-walks n (keep the n newest), 2 :reset (walk <- its OLDEST step:
// (last (walker-walk w)) of a newest-first list), 3 :reset-to-most-likely.  2 and 3 also move
// last-step, as the reference does; age and most-likely-step stay.
__global__ __launch_bounds__(kThreads) void k_modify(ChainState S, int action, int64_t n) {
  const int w = wave_in_group(), l = lane_id(), d = S.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  if (c >= S.n_chains) return;
  const int64_t nh = uniform_i64(S.n_hist[c]), len = uniform_i64(S.length[c]);
  if (action == 0) {
    if (l == 0) S.length[c] = len - n;
  } else if (action == 1) {
    if (l == 0) S.length[c] = n;
  } else {
    double th = 0.0, pr = 0.0;
    if (action == 2) {
      const int64_t vis = len < (int64_t)S.R ? len : (int64_t)S.R;  // what the ring still holds
      const int64_t slot = (nh - vis) & (int64_t)(S.R - 1);
      if (l < d) th = S.hist_theta[(c * S.R + slot) * d + l];
      pr = S.hist_prob[c * S.R + slot];
    } else {
      if (l < d) th = S.best_theta[c * d + l];
      pr = S.best_prob[c];
    }
    __builtin_amdgcn_wave_barrier();
    if (l < d) {
      S.hist_theta[(c * S.R + 0) * d + l] = th;
      S.theta[c * d + l] = th;
    }
    if (l == 0) {
      S.hist_prob[c * S.R + 0] = pr;
      S.prob[c] = pr;
      S.n_hist[c] = 1;
      S.length[c] = 1;
    }
  }
}

// ---- pooled adaptive covariance (extension of the north star; not in the reference) --------
// Step 1: every chain reduces the displacements between its successive forward steps (the
// vectors lplist-covariance would see, M:543) to (n, sum delta, sum delta delta^T).
__global__ __launch_bounds__(kThreads) void k_pool_stats(ChainState S, RunDesc R) {
  const int w = wave_in_group(), l = lane_id(), d = S.d;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  if (c >= S.n_chains) return;
  const int E = 1 + d + d * d;
  double* out = S.pool_stats + c * E;
  Ring ring;
  ring.prob = S.hist_prob + c * S.R;
  ring.theta = S.hist_theta + c * S.R * d;
  ring.mask = S.R - 1;
  ring.d = d;
  ring.nh = uniform_i64(S.n_hist[c]);
  ring.length = uniform_i64(S.length[c]);
  int* fwd = S.fwd_idx + c * R.sts;
  const int nf = S.status[c] == MHX_CHAIN_FP_TRAP ? 0 : ring_forward_list(ring, (int)R.sts, fwd);
  __threadfence();
  const int M = nf >= 2 ? nf - 1 : 0;
  auto diff = [&](int k, int p) -> double {
    return ring.theta[(int64_t)fwd[k + 1] * d + p] - ring.theta[(int64_t)fwd[k] * d + p];
  };
  if (l == 0) out[0] = (double)M;
  if (l < d) {
    double s = 0.0;
    for (int k = 0; k < M; ++k) s = s + diff(k, l);
    out[1 + l] = s;
  }
  for (int e = l; e < d * d; e += kWave) {
    const int i = e / d, j = e - i * d;
    double q = 0.0;
    for (int k = 0; k < M; ++k) q = __builtin_fma(diff(k, i), diff(k, j), q);
    out[1 + d + e] = q;
  }
}

// Step 2: sum over the chains of this rank, one block per entry, fixed order (reproducible)
__global__ __launch_bounds__(256) void k_pool_reduce(ChainState S) {
  __shared__ double part[256];
  const int d = S.d, E = 1 + d + d * d, e = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  for (int64_t c = t; c < S.n_chains; c += 256) s = s + S.pool_stats[c * E + e];
  part[t] = s;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if (t < h) part[t] = part[t] + part[t + h];
    __syncthreads();
  }
  if (t == 0) S.pool_vec[e] = part[0];
}

// Step 3 (after the all-reduce over ranks): covariance, clamped Cholesky, 2.38^2/d scaling
__global__ __launch_bounds__(kWave) void k_pool_factor(ChainState S) {
  const int l = lane_id(), d = S.d;
  const double* v = S.pool_vec;
  double* cov = S.mat_tmp;  // chain 0's scratch is free between step launches
  const double n = v[0];
  int st = L_CAUGHT;
  if (n >= 2.0) {
    for (int e = l; e < d * d; e += kWave) {
      const int i = e / d, j = e - i * d;
      cov[e] = v[1 + d + e] / n - (v[1 + i] / n) * (v[1 + j] / n);
    }
    __threadfence();
    st = L_OK;
    if (l == 0) st = cholesky_seq(cov, S.L_pool, d);
    st = __builtin_amdgcn_readfirstlane(st);
    __threadfence();
    const double factor = (2.38 * 2.38) / (double)d;
    if (st == L_OK)
      for (int e = l; e < d * d; e += kWave) S.L_pool[e] = factor * S.L_pool[e];
  }
  if (l == 0) *S.pool_valid = st == L_OK ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void k_acceptance(ChainState S, int take,
                                                         double* __restrict__ out) {
  const int w = wave_in_group(), l = lane_id();
  const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + w;
  if (c >= S.n_chains) return;
  Ring ring;
  ring.prob = S.hist_prob + c * S.R;
  ring.theta = nullptr;
  ring.mask = S.R - 1;
  ring.d = S.d;
  ring.nh = uniform_i64(S.n_hist[c]);
  ring.length = uniform_i64(S.length[c]);
  int num, den;
  ring_acceptance(ring, take, &num, &den);
  if (l == 0) out[c] = (double)num / (double)den;
}

}  // inline namespace MHX_FAMILY
}  // namespace mhx

