// mhx_types.hpp -- structures shared by the host engine and the gfx950 kernels.
//
// Data layout in HBM (all IEEE binary64 unless noted):
//   dataset k      x[n_pad] y[n_pad] (= y/sigma for the normal likelihoods) w[n_pad] (c[n_pad] for the cutoff likelihood), each a
//                  separate 256-B aligned array padded to a whole number of tiles; w = 1/sigma;
//                  pads are (x_last, 0, 0) so a padded point adds exactly +0 to the sum.
//   chain state    theta[C][d], prob[C], best_theta[C][d], best_prob[C], length/age/draw[C]
//   history ring   hist_prob[C][R], hist_theta[C][R][d]; n_hist[C] = entries ever pushed,
//                  entry e lives in slot e % R; the walk of the reference (M:471, newest first)
//                  is entries n_hist-1, n_hist-2, ...
//   controller     L[C][d][d] row-major, temperature[C], loop_i[C], flags
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#include "../../include/mhx.h"

namespace mhx {

constexpr int kWave = 64;
// The device code is compiled in two FAMILIES that differ in the chains (= waves) per workgroup,
// and with it in the LDS tile (2 * 64 * waves points: one 16-B LDS-DMA element per thread and
// array):
//   w8   8 waves, 1024-point tiles, two workgroups per CU: short datasets, few chains
//   w16 16 waves, 2048-point tiles, one workgroup per CU: half the per-tile fixed cost (barrier,
//        pipeline prologue), +9 % on BASELINE config 2, -17 % on test.lisp's 334 points
// The engine picks one per problem (mhx_engine.cpp: choose_family).  Datasets are padded to
// kPadPoints, a whole number of tiles of either family.
constexpr int tile_points_of(int waves_per_group) { return 2 * kWave * waves_per_group; }
constexpr int kPadPoints = tile_points_of(16);
constexpr int kMaxArrays = 4;               // x, y, w, c

struct FnDesc {
  int32_t model, lik, n_idx, n_bounds;
  int32_t shape[4];
  int32_t idx[MHX_MAX_FN_PARAMS];
  int32_t bidx[MHX_MAX_BOUNDS];
  double blo[MHX_MAX_BOUNDS];
  double bhi[MHX_MAX_BOUNDS];
  const double* x;
  const double* y;
  const double* w;  // 1/sigma (normal), unused (poisson)
  const double* c;  // -1/2 log(2 pi) - log sigma_i (cutoff only)
  int64_t n;        // points
  int64_t n_tiles;  // ceil(n / tile points of the family in use)
  double lik_const; // normal: sum_i(-1/2 log 2pi - log sigma_i); poisson: -sum_i logfact(k_i)
  double xmin, xmax; // range of x over the n points (fast-path preconditions of the models)
  double grid_H;     // 64 h when x is a uniform grid x_0 + i h (to 8 ulp of max |x|), else 0: the
                     // distance between two successive points of one lane (Gaussian recurrence)
  const double* txlo;  // [ceil(n / kPadPoints)] smallest / largest x of each 2048-point window (-inf /
  const double* txhi;  // +inf when it holds a non-finite x): what tile-level peak skipping tests against
  int32_t tile_skip;   // 0: evaluate every peak for every point (MHX_NO_TILE_SKIP=1)
  int32_t solo;        // 1: the problem's only function and a single tile - it stays in LDS
  int32_t user_slot;  // >= 0: index of the run-time compiled expression model (MHX_MODEL_EXPR)
  int32_t prior_slot; // >= 0: index of the run-time compiled prior body, else -1
  int32_t no_yw;      // 1: never take the two-array "yw" tiles of the all-recurrence steps (MHX_NO_YW=1)
  int32_t n_xcols;    // columns of x the dataset brought (mhx_set_dataset_cols): with 2, c holds x1
  // per-window grids: [ceil(n / kPadPoints)] 64 h of every 2048-point window whose x are a grid
  // x_w + i h (to 8 ulp of its max |x|), 0 where they are not; nullptr when the WHOLE dataset is
  // one grid (grid_H != 0) or no window is.  Runs of windows on one grid carry the same bits.
  const double* tgh;
};

struct ProblemDesc {
  int32_t d, K;
  int32_t no_deal;  // 1: wave w judges the proposal of its own chain (MHX_NO_DEAL=1; group_logpost)
  int32_t test_lose_sweepers;  // 1 (test library only): k_persist's sweep workgroups leave at once,
                               // as if they had never got onto the GPU
  FnDesc fn[MHX_MAX_FUNCTIONS];
};

// per-chain state, structure of arrays
struct ChainState {
  int64_t n_chains;
  int64_t chain_offset;  // global id of local chain 0
  int32_t d, R;          // R = history ring capacity
  uint64_t seed;
  double* theta;
  double* prob;
  double* best_theta;
  double* best_prob;
  int64_t* length;  // (walker-length w)
  int64_t* age;     // (walker-age w)
  uint64_t* draw;   // proposals drawn so far: Philox counter
  int64_t* n_hist;
  double* hist_prob;
  double* hist_theta;
  // controller
  double* L;
  double* temperature;
  int64_t* loop_i;
  int64_t* reset_index;
  int32_t* shutting;
  int32_t* status;
  // scratch for the adaptation tick
  int32_t* fwd_idx;  // [C][sts]
  double* mat_tmp;   // [C][2][d][d]
  // pooled-mode statistics [C][1 + d + d*d]: (n, sum delta, sum delta delta^T) of the chain's
  // forward-step displacements; pool_vec [1 + d + d*d] = their sum over chains (and ranks);
  // L_pool [d][d] = (2.38^2/d) * chol(pooled covariance); pool_valid = 1 when usable
  double* pool_stats;
  double* pool_vec;
  double* L_pool;
  int32_t* pool_valid;
  unsigned long long* step_counter;  // chain-steps taken by all chains (device atomic)
  // split mode (few chains, long datasets: one chain's likelihood sum is spread over many
  // workgroups, mhx_kernels.hpp "split mode"): the outstanding proposal of every chain and the
  // partial sums of its functions, one per (slice, wave) slot
  double* split_prop;      // [C][d]
  double* split_u;         // [C] the accept uniform drawn with the proposal
  int32_t* split_pending;  // [C] 1: split_prop / split_u hold a proposal to be judged
  double* split_part;      // [C][K][split_slots]
  int32_t split_slots;
  // the stepping kernel's wave slots -> chains (batch mode, per-walker adaptation): chains that
  // have finished give their slots up (mhx_engine.cpp, compact_slots): entry -1 = empty slot.
  // nullptr: slot s is chain s
  const int32_t* slot_chain;
  int64_t n_slots;  // entries of slot_chain
  int32_t split_pad_;
  // persistent split mode (k_persist: ONE launch runs many iterations of a handful of chains
  // spread over the GPU); both zeroed by the host before every launch
  // persist_msg [C][64] 64-bit words: the chain's proposal as four 128-byte lines of 15
  // parameters and one generation tag each (element j at word (j / 15) * 16 + j % 15, tags at
  // words 15, 31, 47, 63); persist_part [C][K][split_slots] 16-byte pairs {partial sum,
  // generation}.  A 128-byte line / a 16-byte pair is written and read by ONE memory
  // instruction that goes to the level every XCD sees (sc1): who finds the tag finds the data.
  unsigned long long* persist_msg;
  void* persist_part;
  // set by a master whose sweep workgroups did not answer within its patience (they were not all
  // on the GPU - another kernel held it): the iteration is taken back, the launch ends, the host
  // reports it (MHX_EDEVICE) and goes back to the two-launch form
  int32_t* persist_error;
};

struct RunDesc {
  int64_t n, sts, temp_steps, mwl, tail;
  int32_t auto_mode, has_mwl, adapt_mode;
  const double* temps;  // entries [temps_first, temps_first + window) of the schedule: the host
  int64_t temps_first;  // keeps the window over the loop indices of the launch (mhx_engine.cpp)
  const int32_t* stop_flag;
};

}  // namespace mhx
