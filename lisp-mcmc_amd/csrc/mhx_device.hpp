// mhx_device.hpp -- gfx950 device code of the walker-adaptive-steps path.
//
// Mapping: one wavefront (64 lanes) = one walker ("chain"); the lanes split the data points
// of the likelihood sum; a workgroup of kWavesPerGroup chains shares LDS-staged data tiles.
// Everything about one chain (proposal, prior, accept, history, controller) is wave-uniform.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit
// __builtin_fma, so the proposal / covariance / Cholesky arithmetic keeps the reference's
// multiply-then-add rounding (M:594, M:643, M:697).
//
// M: = mcmc-fitting.lisp of the reference.
#pragma once
#ifndef __HIPCC_RTC__  // hiprtc pre-includes the runtime header
#include <hip/hip_runtime.h>
#endif

#include "mhx_types.hpp"

// build knobs of one family of kernels (see mhx_types.hpp); the Makefile compiles
// mhx_kernels.hip once per family, hiprtc gets the same defines for expression kernels
#ifndef MHX_WPG
#define MHX_WPG 8
#endif
#ifndef MHX_FAMILY
#define MHX_FAMILY w8
#endif
#ifndef MHX_PPI
// data points per lane and inner-loop iteration.  2 keeps the pad evaluations of short datasets
// and the register pressure low: with 4 the 16-wave kernels run at the same speed but spill
// 224 instead of 80 bytes per lane around every likelihood sweep (measured)
#define MHX_PPI 2
#endif
#ifndef MHX_PPI_MASKED
#define MHX_PPI_MASKED 4  // the same where Gaussian peaks are skipped through run-time branches
#endif

namespace mhx {
inline namespace MHX_FAMILY {

constexpr int kWavesPerGroup = MHX_WPG;     // one wave = one chain; a workgroup shares LDS tiles
constexpr int kThreads = kWave * kWavesPerGroup;
constexpr int kTilePoints = tile_points_of(kWavesPerGroup);  // data points per LDS tile and array

// ------------------------------------------------------------------------------------------
// Every kernel's dynamic LDS starts with the two look-up tables of the device math (filled by
// lds_tables_begin()): the kernels that stage data tiles put them in front of their GroupLds, the
// split-mode sweep kernel allocates nothing else (kSweepLdsBytes).
// ------------------------------------------------------------------------------------------
struct LdsHead {
  double logtab[256];      // kLogTab: {1/c_i, log c_i} of tlog()                      2 KiB
  double exp2tab[256][2];  // kExp2Tab: {RN(2^(j/256)), its relative residual} of texp2   4 KiB
};
extern __shared__ __attribute__((aligned(16))) unsigned char mhx_lds_raw[];
typedef __attribute__((address_space(3))) const double* lds_cdptr_t;
typedef double mhx_double2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const mhx_double2* lds_cd2ptr_t;
// an LDS address whose relation to other addresses the optimiser may not use (see sweep())
__device__ __forceinline__ lds_cdptr_t opaque_lds(lds_cdptr_t p) {
#ifndef MHX_FUSED_LDS_READS  // (build knob for A/B measurements: tools/ab_build.sh)
  asm volatile("" : "+v"(p));
#endif
  return p;
}
__device__ __forceinline__ lds_cdptr_t lds_logtab() {
  return (lds_cdptr_t)reinterpret_cast<LdsHead*>(mhx_lds_raw)->logtab;
}
__device__ __forceinline__ lds_cd2ptr_t lds_exp2tab() {
  return (lds_cd2ptr_t)reinterpret_cast<LdsHead*>(mhx_lds_raw)->exp2tab;
}
// The tables are READ at absolute LDS addresses: no kernel that reads them has static LDS (the
// host checks it where it configures the kernels: mhx_launch.inc, mhx_rtc.cpp), so the dynamic
// LDS - and with it LdsHead - starts at address 0.  Through mhx_lds_raw the compiler forms
// `index * 16 + <symbol>` and, the symbol being resolved to 0 only after instruction selection,
// leaves a `v_add_u32 v, 0, v` (log) or spends the add of a v_lshl_add_u32 (exp) on it in
// every call; with the byte address formed as an integer the table's offset folds into the DS
// instruction's offset field.
constexpr unsigned kLdsLogTab = 0u;       // __builtin_offsetof(LdsHead, logtab)
constexpr unsigned kLdsExp2Tab = 2048u;   // __builtin_offsetof(LdsHead, exp2tab)
static_assert(__builtin_offsetof(LdsHead, logtab) == kLdsLogTab &&
              __builtin_offsetof(LdsHead, exp2tab) == kLdsExp2Tab, "LdsHead moved");
// {Th_j, rho_j} of kExp2Tab for j = the low byte of `lo`: the byte select and the scaling by the
// entry size in ONE instruction (an SDWA operand selects byte 0 of the register), where
// `(lo & 255) << 4` is two
__device__ __forceinline__ mhx_double2 exp2tab_entry(int lo) {
  unsigned a;
  asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD "
      "src1_sel:BYTE_0"
      : "=v"(a)
      : "v"(lo));
  return ((lds_cd2ptr_t)(__UINTPTR_TYPE__)a)[kLdsExp2Tab / 16];
}
// {1/c_i, log c_i} for the high word hx of the argument: LDS slot (hx >> 13) & 127, which
// lds_tables_begin() fills with entry i = ((hx - 0x3FE60000) >> 13) & 127 of kLogTab
__device__ __forceinline__ mhx_double2 logtab_entry(unsigned hx) {
  const unsigned a = (hx >> 9) & 0x7F0u;
  return ((lds_cd2ptr_t)(__UINTPTR_TYPE__)a)[kLdsLogTab / 16];
}

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------
// (opaque to the optimiser: every use site forms its lane number, and what it derives from it,
// afresh - one v_and_b32 - instead of keeping per-lane offsets and addresses of the cold
// controller code alive, i.e. in scratch, across the likelihood sweep)
__device__ __forceinline__ int lane_id() {
  int l = threadIdx.x & 63;
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ int wave_in_group() {
  return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}
__device__ __forceinline__ double uniform_f64(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readfirstlane((int)b);
  int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane /*uniform*/) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)b, src_lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)v);
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (unsigned int)lo;
}
// butterfly sum over the 64 lanes: every lane ends with the same bits
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ bool finite_f64(double v) {
  return (__double_as_longlong(v) & 0x7ff0000000000000LL) != 0x7ff0000000000000LL;
}

// ------------------------------------------------------------------------------------------
// the random stream (specification shared with the oracle; the reference's own stream,
// cl:random + alexandria:gaussian-random M:687/M:1092, is unseeded and unpinned)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// fdlibm-recipe log on its general path, pure IEEE mul/add/div (bit-identical to the oracle)
__device__ __forceinline__ double det_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  int hx = (int)(b >> 32);
  unsigned int lx = (unsigned int)b;
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int i = (hx + 0x95f64) & 0x100000;
  unsigned long long nb = ((unsigned long long)(unsigned int)(hx | (i ^ 0x3ff00000)) << 32) | lx;
  k += (i >> 20);
  double f = __longlong_as_double((long long)nb) - 1.0;
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = (0.5 * f) * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}
__device__ __forceinline__ double det_ksin(double x) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x + v * (S1 + z * r);
}
__device__ __forceinline__ double det_kcos(double x) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  return 1.0 - (0.5 * z - z * r);
}
__device__ __forceinline__ double det_cos2pi(double t) {
  double t4 = 4.0 * t;
  int q = (int)(t4 + 0.5);
  double r = t4 - (double)q;
  double y = r * 1.57079632679489661923;
  double c = det_kcos(y), s = det_ksin(y);
  switch (q & 3) {
    case 0: return c;
    case 1: return -s;
    case 2: return -c;
    default: return s;
  }
}
__device__ __forceinline__ unsigned long long bits53(uint32_t a, uint32_t b) {
  return ((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6);
}
// correctly rounded sqrt (IEEE), so z matches the host's sqrt bit for bit
__device__ __forceinline__ double ieee_sqrt(double x) { return __dsqrt_rn(x); }

// lane j < d returns z_j; lane 63 returns the accept uniform u in (0,1]; other lanes junk.
// *log_u1: det_log of the lane's first uniform - in lane 63 the log u of the accept test
// (M:1092), which the Box-Muller transform of the other lanes computes in passing.
__device__ __forceinline__ double rng_lane_value(uint64_t seed, uint64_t gchain, uint64_t draw,
                                                 int d, double* log_u1 = nullptr) {
  int l = lane_id();
  uint32_t slot = (l == 63) ? 0xFFFFFFFFu : (uint32_t)l;
  uint32_t r[4];
  philox4x32_10((uint32_t)gchain, slot, (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  double u1 = (double)(bits53(r[0], r[1]) + 1ULL) * 0x1p-53;
  const double lg = det_log(u1);
  if (log_u1) *log_u1 = lg;
  if (l == 63) return u1;
  double u2 = (double)bits53(r[2], r[3]) * 0x1p-53;
  double rad = ieee_sqrt(-2.0 * lg);
  (void)d;
  return rad * det_cos2pi(u2);
}

// ------------------------------------------------------------------------------------------
// device models (formula spec: include/mhx.h).  Prep holds wave-uniform values (SGPRs).
// PF = functor int -> double giving local parameter j of the function (uniform).
// ------------------------------------------------------------------------------------------
// 2^s for s <= 0 (also fine for moderate s > 0).  s = k + f, k = rint(s) by the 1.5*2^52
// trick, f in [-1/2, 1/2] EXACT (no hi/lo reduction constants: the callers fold log2(e)
// into their own scale factors, so the argument already IS a base-2 exponent);
// 2^f by a degree-11 near-minimax polynomial (tools/gen_exp2_coeffs.py: 2.0e-17 relative
// after rounding the coefficients), scaled by v_ldexp_f64.  v_cvt_i32_f64 saturates, so
// s -> -inf side underflows to 0 through ldexp; NaN propagates (a NaN log-posterior is where
// the reference would have trapped).  16 VALU instructions, < 1 ulp.
__device__ __forceinline__ double mexp2(double s) {
  const double MAGIC = 0x1.8p52;
  const double kd = s + MAGIC;
  const double kf = kd - MAGIC;
  const double f = s - kf;
  double p = 0x1.e9d3fe3952179p-32;
  p = __builtin_fma(p, f, 0x1.e6063f7217bc6p-28);
  p = __builtin_fma(p, f, 0x1.b524fae627834p-24);
  p = __builtin_fma(p, f, 0x1.62bfd47773353p-20);
  p = __builtin_fma(p, f, 0x1.ffcbfc670dcd4p-17);
  p = __builtin_fma(p, f, 0x1.430913096fd9fp-13);
  p = __builtin_fma(p, f, 0x1.5d87fe78a5276p-10);
  p = __builtin_fma(p, f, 0x1.3b2ab6fba1ddap-7);
  p = __builtin_fma(p, f, 0x1.c6b08d704a0c2p-5);
  p = __builtin_fma(p, f, 0x1.ebfbdff82c598p-3);
  p = __builtin_fma(p, f, 0x1.62e42fefa39efp-1);
  p = __builtin_fma(p, f, 1.0);
  return ldexp(p, (int)kf);
}
// 2^(-t*t), table driven: s = -t^2 = k + j/256 + r with |r| <= 2^-9, all three parts from TWO
// fmas (kd = fma(-t, t, 1.5 2^44) rounds s to a multiple of 1/256 - its low dword IS
// 256 k + j, no conversion - and r = fma(-t, t, -(kd - 1.5 2^44)) is exact to one rounding);
// 2^s = 2^k Th_j (1 + rho_j) 2^r with {Th_j, rho_j} = {RN(2^(j/256)), what that rounding lost}
// read from LDS by ONE ds_read_b128, 2^r - 1 = r q(r), q of degree 3 (tools/gen_exp2_table.py:
// 4.8e-18 relative), e = fma(r, q, rho), p = fma(Th, e, Th), v_ldexp_f64.  12 VALU instructions
// and one LDS read where the degree-11 polynomial took 15; <= 0.51 ulp (the table entry is exact
// to 2^-105 through rho, so the final fma's rounding is the error): tests/test_device_math_cpu.py.
// Valid for t^2 < 2^23 (|t| < 2896: then 256 s fits the low dword); callers establish
// |t| < kFastT once per step from the dataset's x range (Prep::fast).
// NaN propagates (the masked index keeps the table read in range).
constexpr double kFastT = 2890.0;
__device__ const double kExp2Tab[256][2] = {
#include "mhx_exp2_table.inc"
};
// The constants, pinned to register files once per step by the caller: every v_fma_f64 may read
// ONE scalar operand, so q2..q0 and the magic number live in SGPRs and q3 (which shares the
// first fma with q2) in a VGPR.  Left to itself the compiler materialises the coefficients in
// VGPRs as soon as the loop body has branches and copies one into the accumulator before every
// v_fmac.
struct Exp2K {
  double q3;     // VGPR
  double q[3];   // SGPRs: q2, q1, q0
  double magic;  // SGPR
  __device__ __forceinline__ void pin() {
    constexpr double k[3] = {0x1.c6b0902b5a0abp-5, 0x1.ebfbdff82c585p-3, 0x1.62e42fefa39d9p-1};
    double v = 0x1.3b2ab83eadfb0p-7;
    asm volatile("" : "+v"(v));
    q3 = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double c = k[i];
      asm volatile("" : "+s"(c));
      q[i] = c;
    }
    double m = 0x1.8p44;
    asm volatile("" : "+s"(m));
    magic = m;
  }
};
// In two halves, so that a caller with several independent arguments can issue all the table
// reads before the first polynomial (an LDS read takes ~64+ cycles to come back; left to the
// compiler every exp waits for its own):  head = reduction + the read, tail = the rest.
struct Exp2Head {
  double r;       // |r| <= 2^-9
  int lo;         // 256 k + j
  mhx_double2 e;  // {Th_j, rho_j}, in flight until the tail uses it
};
__device__ __forceinline__ void mexp2_negsq_head(double t, const Exp2K& K, Exp2Head& h) {
  const double kd = __builtin_fma(-t, t, K.magic);
  const double kf = kd - K.magic;
  h.r = __builtin_fma(-t, t, -kf);
  h.lo = (int)__double_as_longlong(kd);
  h.e = exp2tab_entry(h.lo);
}
// the same reduction for an argument s given as it is (|s| < 2^22): 2^s = tail(head)
__device__ __forceinline__ void mexp2_head(double s, const Exp2K& K, Exp2Head& h) {
  const double kd = s + K.magic;
  const double kf = kd - K.magic;
  h.r = s - kf;  // exact: both multiples of ulp(s), difference at most 2^-9
  h.lo = (int)__double_as_longlong(kd);
  h.e = exp2tab_entry(h.lo);
}
__device__ __forceinline__ double mexp2_negsq_tail(const Exp2Head& h, const Exp2K& K) {
  double a = __builtin_fma(h.r, K.q3, K.q[0]);
  a = __builtin_fma(h.r, a, K.q[1]);
  a = __builtin_fma(h.r, a, K.q[2]);
  const double ee = __builtin_fma(h.r, a, h.e.y);
  return ldexp(__builtin_fma(h.e.x, ee, h.e.x), h.lo >> 8);
}
__device__ __forceinline__ double mexp2_negsq(double t, const Exp2K& K) {
  Exp2Head h;
  mexp2_negsq_head(t, K, h);
  return mexp2_negsq_tail(h, K);
}
// N independent arguments at once, STAGE BY STAGE (on[i] false: argument i is skipped, v[i]
// untouched): all table reads are issued first, then every stage of the tail runs across the N
// chains before the next stage starts.  A dependent v_fma_f64 can issue only every 8 cycles, an
// independent one every 4 (tools/microbench/fma_chain): with the stages pinned by
// sched_barriers a wave issues back to back whatever the other waves of its SIMD do.  Each
// chain's operations are mexp2_negsq()'s: identical bits.
// PLAIN_ODD: the arguments with odd index are exponents s (2^s), the even ones t (2^(-t t)).
template <int N, bool PLAIN_ODD = false>
__device__ __forceinline__ void mexp2_negsq_batch(const double (&t)[N], const bool (&on)[N],
                                                  const Exp2K& K, double (&v)[N]) {
  Exp2Head h[N];
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) {
      if (PLAIN_ODD && (i & 1)) mexp2_head(t[i], K, h[i]);
      else mexp2_negsq_head(t[i], K, h[i]);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) a[i] = __builtin_fma(h[i].r, K.q3, K.q[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) a[i] = __builtin_fma(h[i].r, a[i], K.q[1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) a[i] = __builtin_fma(h[i].r, a[i], K.q[2]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) a[i] = __builtin_fma(h[i].r, a[i], h[i].e.y);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) a[i] = __builtin_fma(h[i].e.x, a[i], h[i].e.x);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (on[i]) v[i] = ldexp(a[i], h[i].lo >> 8);
}
// t = x iw + c with iw in a scalar register and c in a vector register: ONE v_fma_f64.  (Written
// as __builtin_fma the compiler sometimes picks the two-address v_fmac_f64 and copies c into
// the destination first - an extra instruction per point and peak.)
__device__ __forceinline__ double fma_svv(double x, double iw_s, double c_v) {
  double t;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "s"(iw_s), "v"(x), "v"(c_v));
  return t;
}
// any t (NaN propagates; |t| huge or inf gives 0): the guarded form for the rare step whose
// parameters put |t| beyond kFastT somewhere in the data range
__device__ __forceinline__ double mexp2_negsq_safe(double t) {
  double s = -(t * t);
  s = s < -1100.0 ? -1100.0 : s;
  return mexp2(s);
}
// log(x) for the Poisson term k*log(lambda) (M:383): the fdlibm recipe (x = 2^k (1+f),
// s = f/(2+f), log(1+f) = f - (f^2/2 - s (f^2/2 + R(s^2)))) with the quotient formed from
// v_rcp_f64 + two Newton steps instead of an IEEE division.  < 1 ulp on normal positive x.
// x <= 0, subnormal, inf or NaN -> NaN: a rate outside (0, inf) is where the reference errors
// (log of a negative number is complex, log 0 traps), and a NaN log-posterior freezes the chain.
// (out of line like dexp: it only serves the logs of user expressions within 1/16 of 1 and the
// arguments that end in NaN)
__device__ __attribute__((noinline)) double mlog(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const int hx0 = (int)(b >> 32);
  const unsigned int lx = (unsigned int)b;
  int k = (hx0 >> 20) - 1023;
  int hx = hx0 & 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;
  hx |= (i ^ 0x3ff00000);
  k += (i >> 20);
  const double m = __longlong_as_double((long long)(((unsigned long long)(unsigned int)hx << 32) | lx));
  const double f = m - 1.0;
  const double dd = 2.0 + f;
  double y = __builtin_amdgcn_rcp(dd);
  y = __builtin_fma(__builtin_fma(-dd, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-dd, y, 1.0), y, y);
  const double s = f * y;
  const double dk = (double)k;
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = (0.5 * f) * f;
  const double r = __builtin_fma(dk, ln2_hi, -((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f));
  // valid iff 2^-1022 <= x < inf: one unsigned compare on the high word (sign bit fails it)
  const bool ok = (unsigned int)(hx0 - 0x00100000) < (unsigned int)(0x7ff00000 - 0x00100000);
  return ok ? r : __builtin_nan("");
}
// log(x), table driven (Tang 1990): x = 2^k z, z in [0.6875, 1.375), the top 7 mantissa bits of
// z pick c_i; r = z/c_i - 1 (one fma, |r| < 2^-7.6);
// log x = (k ln2_hi + log c_i) + [k ln2_lo + log1p(r)].  The first sum is exact (both
// terms are multiples of 2^-43) and log c_i is tabulated to 2^-68 (tools/gen_log_table.py), so the
// result stays below 0.7 ulp (checked against 80-bit logl on 4e7 arguments) outside 1/16 of 1,
// where the table terms cancel: there the ABSOLUTE error stays below 2^-56
// (tests/test_device_math_cpu.py).  Two entry points:
//   tlog_rate(x)  the Poisson sweep's k log(lambda) (M:383): the table form everywhere (a term
//                 k log(lambda) - lambda near lambda = 1 is of size 1: 2^-56 is nothing to it);
//                 x <= 0, subnormal, inf or NaN -> NaN.  18 VALU instructions + one ds_read_b128.
//   tlog(x)       (log x) of a user expression: within 1/16 of 1 through mlog() (< 1 ulp there
//                 too), and so is everything mlog() answers with NaN.
// In instructions: the table is indexed by the mantissa bits of x itself (the LDS copy is rotated
// by OFF's 48 entries: lds_tables_begin), -k comes out of one subtraction and one shift, z out of
// one v_ldexp_f64 (the same bits as subtracting k from the exponent field, which took an and, a
// subtraction and a register copy), and A1 enters as the addend of a three-register fma (as the
// addend of the fma with the literal -1/4 it was copied into the destination first, every call).
__device__ const double kLogTab[128][2] = {
#include "mhx_log_table.inc"
};
// a3: the constant 1/5 of the polynomial handed in from a VGPR the caller keeps alive across its
// loop (tlog_a3()): fma(r, A4, A3) reads two constants and only one may come from the scalar
// file - otherwise the compiler copies one into the accumulator before a v_fmac, every call
constexpr double kTlogA3 = 0x1.999999999999ap-3;
__device__ __forceinline__ double tlog_a3() {
  double v = kTlogA3;
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ double tlog_table(double x, unsigned hx, double A3) {
  const double nLn2hi = -0x1.62e42fefa3800p-1, nLn2lo = -0x1.ef35793c76730p-45;
  const double A1 = 0x1.5555555555555p-2, A4 = -0x1.5555555555555p-3;
  const mhx_double2 ic = logtab_entry(hx);               // {1/c_i, log c_i}
  // -k, k = floor((hx - OFF) / 2^20) with OFF = 0x3FE60000: ceil((OFF - hx) / 2^20)
  const int nk = (int)(0x3FE60000u + 0x000FFFFFu - hx) >> 20;
  const double z = ldexp(x, nk);                         // exact
  const double r = __builtin_fma(z, ic.x, -1.0);
  const double nkd = (double)nk;
  const double w = __builtin_fma(nkd, nLn2hi, ic.y);     // exact
  const double r2 = r * r;
  const double p1 = __builtin_fma(r, A4, A3);
  const double p3 = __builtin_fma(r, -0.25, __builtin_fma(r2, p1, A1));
  // w + [(k ln2_lo + r) + r^2 (-1/2 + r p3)]: the bracket is below 2^-7 and carries its rounding
  // errors at that scale, the one rounding that counts is the last addition's
  const double a = __builtin_fma(nkd, nLn2lo, r);
  const double q = __builtin_fma(r, p3, -0.5);
  return w + __builtin_fma(r2, q, a);
}
__device__ __forceinline__ double tlog_rate(double x, double A3 = kTlogA3) {
  const unsigned hx = (unsigned)((unsigned long long)__double_as_longlong(x) >> 32);
  const unsigned long long rb = (unsigned long long)__double_as_longlong(tlog_table(x, hx, A3));
  // positive and normal, or the high word of a NaN (one v_cmp_class_f64, one v_cndmask_b32)
  const unsigned rh = __builtin_amdgcn_class(x, 0x100u) ? (unsigned)(rb >> 32) : 0x7FF80000u;
  return __longlong_as_double((long long)(((unsigned long long)rh << 32) | (unsigned)rb));
}
__device__ __forceinline__ double tlog(double x, double A3 = kTlogA3) {
  const unsigned hx = (unsigned)((unsigned long long)__double_as_longlong(x) >> 32);
  double res = tlog_table(x, hx, A3);
  // normal, positive, finite and not within 1/16 of 1: one unsigned compare each
  const bool ordinary = (unsigned)(hx - 0x00100000u) < (unsigned)(0x7ff00000u - 0x00100000u);
  const bool near_one = (unsigned)(hx - 0x3FEE0000u) < (unsigned)(0x3FF10000u - 0x3FEE0000u);
  if (!ordinary || near_one) res = mlog(x);
  return res;
}
// e^x in pure IEEE operations (fma, add, ldexp), so that the oracle's mirror mode reproduces it
// bit for bit: k = rint(x log2 e), r = x - k ln2 (hi/lo), degree-13 Taylor in Horner form.
// < 1 ulp for |x| < 700.  Used by the bounds prior (M:360), off the hot loop.
// NOT inlined: as part of the step kernels its thirteen coefficients were materialised once per
// launch into vector registers that then stayed occupied through every sweep (four of them
// spilled: the last scratch of the config-2 kernel); it runs only when a proposal violates a bound.
__device__ __attribute__((noinline)) double dexp(double x) {
  // beyond the range of a double the answer is decided here (inf, 0, NaN for NaN): the
  // reduction below is meaningless there - it once answered -inf for e^(1e25), a bound
  // violated by 1e30, and the oracle's restatement something else again
  if (!(x < 0x1.62e42fefa39efp+9)) return x != x ? x : __builtin_inf();
  if (x < -745.2) return 0.0;
  const double MAGIC = 0x1.8p52;
  const double kd = __builtin_fma(x, 1.4426950408889634074, MAGIC);
  const double kf = kd - MAGIC;
  double r = __builtin_fma(-kf, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// 1/d from v_rcp_f64 and ONE cubic correction, y (1 + e + e^2) with e = 1 - d y: 4 instructions
// instead of the ~25 of an IEEE fp64 division.  v_rcp_f64 is good to 2^-24.4 on gfx950
// (tools/microbench/rcp_error.hip), so e^3 = 2^-73 and the result is within 2^-53.0 of 1/d -
// what two quadratic (Newton) steps give in 5.  Used for the Lorentzian 1/(1+u^2), whose
// denominator is >= 1 and finite.
__device__ __forceinline__ double frcp(double d) {
  const double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
// e^x, < 1 ulp: the `exp` of user expressions (mhx_rtc.cpp) - 15 VALU instructions and one LDS
// read where ocml's exp takes about 37 (round 2's polynomial form: 18).  x log2 e = k + j/256 + r
// by the magic-number trick (low dword of kd = 256 k + j), r from a hi/lo split of log2 e (the
// inner fma cancels k + j/256, so r keeps an absolute error of 2^-62), 2^(j/256) from
// mexp2_negsq's table, 2^r - 1 by its cubic, the scale by v_ldexp_f64 - which overflows to inf
// (x > 709.8) and underflows to 0 (x < -745.2) by itself.  The trick needs |x log2 e| < 2^22:
// beyond |x| = 1000 the answer is known - inf or 0 for finite x; NaN -> NaN, and x = +-inf ->
// NaN as well: an infinite argument can only come from an operation SBCL would already have
// trapped on (overflow), and a NaN log-posterior marks the chain as trapped.
__device__ __forceinline__ double gexp(double x) {
  const double MAGIC = 0x1.8p44, L2E_HI = 0x1.71547652b82fep+0, L2E_LO = 0x1.777d0ffda0d24p-56;
  const double q3 = 0x1.3b2ab83eadfb0p-7, q2 = 0x1.c6b0902b5a0abp-5, q1 = 0x1.ebfbdff82c585p-3,
               q0 = 0x1.62e42fefa39d9p-1;  // (Exp2K's constants)
  const double kd = __builtin_fma(x, L2E_HI, MAGIC);
  const double kf = kd - MAGIC;
  const double r = __builtin_fma(x, L2E_LO, __builtin_fma(x, L2E_HI, -kf));
  const int lo = (int)__double_as_longlong(kd);
  const mhx_double2 e = exp2tab_entry(lo);
  double a = __builtin_fma(r, q3, q2);
  a = __builtin_fma(r, a, q1);
  a = __builtin_fma(r, a, q0);
  const double ee = __builtin_fma(r, a, e.y);
  double v = ldexp(__builtin_fma(e.x, ee, e.x), lo >> 8);
  // (a compare and two selects; as a branch it would cut the callers' batches of independent
  // exps into dependent pieces)
  if (!(fabs(x) <= 1000.0))
    v = fabs(x) < __builtin_inf() ? (x > 0.0 ? __builtin_inf() : 0.0) : __builtin_nan("");
  return v;
}
constexpr double kLog2e = 1.4426950408889634074;       // log2(e)
constexpr double kSqrtLog2e = 1.2011224087864497594;   // sqrt(log2(e))

template <int NBG, int NPK, bool LORENTZ>
struct PeaksModel {
  static constexpr bool kHasFast = !LORENTZ;
  static constexpr bool kSeedInLds = !LORENTZ && NPK > 2;  // (Prep::sc; needs prepare(pf, fn, scratch))
  static constexpr int kScratchDoubles = 4 * NPK;
  struct Prep {
    double bg[NBG > 0 ? NBG : 1];
    double A[NPK];
    double mu[kSeedInLds ? 1 : NPK], iw[kSeedInLds ? 1 : NPK];
    double cv[kSeedInLds ? 1 : NPK];  // mu[] once more, pinned in VGPRs: t = fma(x, iw, c) may read only ONE
                     // scalar operand (constant bus), so c would be re-copied for every point
    Exp2K K;         // the constants of the 2^f polynomial, pinned likewise
    bool fast;  // |t| < kFastT over the whole x range for every peak (uniform)
    bool skip;  // tile-level skipping by the relative rule allowed this step (see tile_mask)
    bool afin;  // every amplitude is finite (the exact-zero rule of tile_mask)
    // uniform-grid recurrence (below): per peak -2 D, -D^2 and 2^(-2 D^2), D = the step of t from
    // one point of a lane to its next (64 grid points on); rec: usable this step
    double rm2d[kSeedInLds ? 1 : NPK], rnd2[kSeedInLds ? 1 : NPK], rq[NPK];
    unsigned rmask;  // bit k: peak k goes by the recurrence this step
    // narrower peaks go by it too, re-seeded more often: bit k of s16 - also every kSeedSteps / 2
    // points of the lane; of s8 - every kSeedSteps / 4 (a peak in s8 is in s16 as well)
    unsigned s16, s8;
    // kSeedInLds (more than two peaks): what only the seeds, a direct-form block and tile_mask
    // read - c = -mu iw, iw, -2 D, -D^2 of every peak - waits in the wave's LDS scratch
    // (sc[4 k .. 4 k + 3]) instead of 8 scalar registers per peak.  With five peaks those were 40
    // of the ~100 scalar registers a wave has; the compiler kept them and spilled A and q - which
    // every POINT needs - to vector lanes, reloading them (4 v_readlane per peak and 4 points) in
    // the hot blocks, and materialised the constants of the Poisson term's log in vector
    // registers per point: a quarter of config 3's instructions.
    lds_cdptr_t sc;
    __device__ __forceinline__ double c_of(int k) const {
      if constexpr (kSeedInLds) return sc[4 * k]; else return mu[k];
    }
    __device__ __forceinline__ double iw_of(int k) const {
      if constexpr (kSeedInLds) return sc[4 * k + 1]; else return iw[k];
    }
    __device__ __forceinline__ double rm2d_of(int k) const {
      if constexpr (kSeedInLds) return sc[4 * k + 2]; else return rm2d[k];
    }
    __device__ __forceinline__ double rnd2_of(int k) const {
      if constexpr (kSeedInLds) return sc[4 * k + 3]; else return rnd2[k];
    }
    // ... and when EVERY peak does, so does a constant or linear background (b(x + 64 h) =
    // b(x) + 64 h b1, re-seeded with the peaks): such a step never reads x beyond the seeds, which
    // takes a third off the LDS traffic of the sweep.  Decided per step, not per tile.
    double bgH;
    bool bgrec;
  };
  // ---- Gaussians on a uniformly spaced x grid: a two-multiply recurrence ----------------------
  // Lane l evaluates the points l, l + 64, l + 128, ... of a tile, so on a grid of spacing h its t
  // advances by the constant D = 64 h iw from one point to the next, and
  //     g(t + D) = 2^-(t + D)^2 = g(t) r(t),   r(t) = 2^(-2 t D - D^2),   r(t + D) = r(t) 2^(-2 D^2):
  // per point and peak  f = fma(A, g, f); g = g r; r = r q  - 3 instructions where the direct
  // form takes 14.  g and r are re-seeded from the direct formulas (table exp, at the lane's
  // ACTUAL x) every kSeedSteps = 32 points of the lane (at the start of every 2048-point window
  // of the dataset: one tile of the 16-wave family, two of the 8-wave family), so a value is at
  // most 31 steps away from an exactly evaluated one: r_j = r_0 q^j carries (j + 1) roundings,
  // g_m = g_0 r_0 ... r_(m-1)  m + sum_j (j + 1) <= 527 of them -> |g_m / g(t_m) - 1| <=
  // 560 * 2^-53 = 6.2e-14, plus the grid's own departure from x_0 + i h (checked <= 8 ulp of
  // max |x| when the dataset is set: 2 |t| iw 8 ulp).  Both are far inside the stated tolerance
  // 1e-12 sum |term| (tests/test_gpu_recurrence.py measures them against the direct path and the
  // oracle).  A seed costs two table exps per peak: 1.6 instructions per point at 32 points per
  // seed next to the 9 of the two-peak loop itself (3.3 at 16, which round 2 started with).
  // Preconditions, per step and PEAK (else that peak keeps the direct form): the fast path,
  // 32 |D| <= 1 - a seed that underflows to 0 (|t| > 33.8) then stays beyond |t| = 32.8 for its
  // 31 steps, where the true value is below 2^-1075 as well - and a dataset on a grid
  // (FnDesc::grid_H).  (A narrow peak - large D - is left out of most tiles by tile-level
  // skipping anyway; where it is not, t runs through it in a few points and the direct form is
  // the right one.)
  // The seeding cadence is fixed in points of a lane, not in tiles, so both kernel families
  // produce the same bits.  Tile-level skipping stays exact for these values too: its bound has a
  // factor 2 in hand (an addend below f 2^-53 cannot move f; the test asks for f 2^-54), and a
  // peak's mask can only change where a window - and with it a seed - starts.
  static constexpr bool kHasRec = !LORENTZ;
#ifndef MHX_SEED_STEPS
#define MHX_SEED_STEPS 32
#endif
  static constexpr int kSeedSteps = MHX_SEED_STEPS;  // points of a lane from seed to seed
  // Peaks too narrow for that period (S |D| <= 1 fails for S = kSeedSteps) still go by the
  // recurrence when half or a quarter of it does: they are re-seeded inside the window as well
  // (sweep()).  Only for models of at most two peaks, whose tile loops are straight-line variants:
  // in the run-time-masked loops of bigger models the extra seed sites cost every step 5 %
  // (config 3, measured) for peaks those problems rarely have.
#ifdef MHX_ONE_SEED_CLASS  // (build knob for A/B measurements: round 2's single class)
  static constexpr bool kMultiSeed = false;
#else
  static constexpr bool kMultiSeed = NPK <= 2;
#endif
  struct Rec {
    double g[NPK], r[NPK];
    double b;  // the background's running value (Prep::bgrec)
  };
  static __device__ __forceinline__ unsigned rec_mask(const Prep& p) { return p.rmask; }
  static __device__ __forceinline__ bool rec_bg(const Prep& p) { return p.bgrec; }
  static __device__ __forceinline__ unsigned seed16(const Prep& p) { return p.s16; }
  static __device__ __forceinline__ unsigned seed8(const Prep& p) { return p.s8; }
  // (split mode seeds every kSeedSteps points only: the peaks of the long class, and the
  // background with them only when that is all of them)
  static __device__ __forceinline__ unsigned rec_mask_long(const Prep& p) { return p.rmask & ~p.s16; }
  static __device__ __forceinline__ bool rec_bg_long(const Prep& p) { return p.bgrec && p.s16 == 0u; }
  static __device__ __forceinline__ void rec_seed_bg(const Prep& p, double x0, Rec& rs) {
    rs.b = bg_of(p, x0);
  }
  // The same for a mask known only at run time (the seeds INSIDE a window: sweep()), peak by peak
  // behind scalar branches.  As one batch with run-time `on` flags the compiler evaluates every
  // exp of the batch and selects: 80 instructions, 56 of them v_cndmask, where one peak's seed
  // takes 28.  Per exp the operations are the batch's: identical bits.
  static __device__ __forceinline__ void rec_seed_some(const Prep& p, double x0, unsigned mask,
                                                       Rec& rs) {
    const unsigned sm = (unsigned)__builtin_amdgcn_readfirstlane((int)mask);
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      if (!((sm >> k) & 1u)) continue;
      double t[2], v[2];
      const bool on[2] = {true, true};
      if constexpr (kSeedInLds) t[0] = __builtin_fma(x0, p.iw_of(k), p.c_of(k));
      else t[0] = fma_svv(x0, p.iw[k], p.cv[k]);
      t[1] = __builtin_fma(p.rm2d_of(k), t[0], p.rnd2_of(k));
      mexp2_negsq_batch<2, true>(t, on, p.K, v);
      rs.g[k] = v[0];
      rs.r[k] = v[1];
    }
  }
  // x0: the lane's x at the first point of the window; mask: the peaks to seed (those
  // that are evaluated in this tile AND go by the recurrence)
  static __device__ __forceinline__ void rec_seed(const Prep& p, double x0, unsigned mask, Rec& rs) {
    if constexpr (NPK <= 2) {
      double t[2 * NPK], v[2 * NPK];
      bool on[2 * NPK];
#pragma unroll
      for (int k = 0; k < NPK; ++k) {
        on[2 * k] = on[2 * k + 1] = !kHasSkip || ((mask >> k) & 1u);
        const double ts = on[2 * k] ? fma_svv(x0, p.iw[k], p.cv[k]) : 0.0;
        t[2 * k] = ts;
        t[2 * k + 1] = __builtin_fma(p.rm2d[k], ts, p.rnd2[k]);
      }
      mexp2_negsq_batch<2 * NPK, true>(t, on, p.K, v);
#pragma unroll
      for (int k = 0; k < NPK; ++k)
        if (on[2 * k]) {
          rs.g[k] = v[2 * k];
          rs.r[k] = v[2 * k + 1];
        }
    } else {
#pragma unroll
      for (int k = 0; k < NPK; ++k) {
        if (kHasSkip && !((mask >> k) & 1u)) continue;  // wave-uniform
        double t[2], v[2];
        const bool on[2] = {true, true};
        t[0] = __builtin_fma(x0, p.iw_of(k), p.c_of(k));
        // (forming -2 D and -D^2 here from D = 64 h iw instead of keeping them per peak takes the
        // kernel's scratch from 224 to 188 B per lane and config 3 from 7.70e5 to 7.31e5
        // chain-steps/s: the reloads at the seeds are cheaper than the arithmetic - measured)
        t[1] = __builtin_fma(p.rm2d_of(k), t[0], p.rnd2_of(k));
        mexp2_negsq_batch<2, true>(t, on, p.K, v);
        rs.g[k] = v[0];
        rs.r[k] = v[1];
      }
    }
  }
  // The model at the lane's next P points on the fast path: peaks outside `mask` are left out
  // (tile-level skipping), peaks in `rmask` advance by the recurrence, the others are evaluated
  // directly (all of those as one batch when the masks are compile-time constants).  The peaks
  // are added in increasing k whatever their form.
  template <int P, bool BGREC = false>
  static __device__ __forceinline__ void eval_mixed(const Prep& p, const double (&x)[P],
                                                    unsigned mask, unsigned rmask, Rec& rs,
                                                    double (&f)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      if (BGREC) {
        f[i] = rs.b;
        if (NBG > 1) rs.b = rs.b + p.bgH;
      } else {
        f[i] = bg_of(p, x[i]);
      }
    }
    if constexpr (NPK * P <= 4) {
      double t[NPK * P], v[NPK * P];
      bool on[NPK * P];
      bool any = false;
#pragma unroll
      for (int k = 0; k < NPK; ++k)
#pragma unroll
        for (int i = 0; i < P; ++i) {
          on[k * P + i] = ((mask >> k) & 1u) && !((rmask >> k) & 1u);
          any = any || on[k * P + i];
          t[k * P + i] = on[k * P + i] ? fma_svv(x[i], p.iw[k], p.cv[k]) : 0.0;
        }
      if (any) mexp2_negsq_batch<NPK * P>(t, on, p.K, v);
#pragma unroll
      for (int k = 0; k < NPK; ++k) {
        if (!((mask >> k) & 1u)) continue;
#pragma unroll
        for (int i = 0; i < P; ++i) {
          if ((rmask >> k) & 1u) {
            f[i] = __builtin_fma(p.A[k], rs.g[k], f[i]);
            rs.g[k] = rs.g[k] * rs.r[k];
            rs.r[k] = rs.r[k] * p.rq[k];
          } else {
            f[i] = __builtin_fma(p.A[k], v[k * P + i], f[i]);
          }
        }
      }
    } else {
      // Wave-uniform branches on SCALAR masks, and per peak two independent `if`s instead of an
      // if / else: as one three-way diamond (left out / recurrence / direct) the compiler gave
      // each arm its own result registers and copied them back at the join - six v_mov_b64 per
      // peak and 4 points on EVERY path - and rebuilt the arms' conditions as lane masks with a
      // v_cndmask / v_cmp pair each: 22 instructions where the recurrence's own are 12 (config
      // 3's five peaks: 12 of 37 per point).  Either arm now updates f, g, r in place.
      const unsigned rm = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask & rmask));
      const unsigned dm = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask & ~rmask));
#pragma unroll
      for (int k = 0; k < NPK; ++k) {
        if ((rm >> k) & 1u) {
#pragma unroll
          for (int i = 0; i < P; ++i) {
            f[i] = __builtin_fma(p.A[k], rs.g[k], f[i]);
            rs.g[k] = rs.g[k] * rs.r[k];
            rs.r[k] = rs.r[k] * p.rq[k];
          }
        }
        if ((dm >> k) & 1u) {
          double t[P], v[P], ck = p.c_of(k);
          asm volatile("" : "+v"(ck));
          bool on[P];
          const double iwk = p.iw_of(k);
#pragma unroll
          for (int i = 0; i < P; ++i) {
            on[i] = true;
            t[i] = kSeedInLds ? __builtin_fma(x[i], iwk, ck) : fma_svv(x[i], iwk, ck);
          }
          mexp2_negsq_batch<P>(t, on, p.K, v);
#pragma unroll
          for (int i = 0; i < P; ++i) f[i] = __builtin_fma(p.A[k], v[i], f[i]);
        }
      }
    }
  }
  // Tile-level skipping of Gaussian peaks, an EXACT transformation of the fast path.
  // Over a tile whose x lie in [xlo, xhi], t_k = fma(x, iw_k, c_k) is monotone in x, so
  // |t_k| >= tmin = min(|t_k(xlo)|, |t_k(xhi)|) when both ends have the same sign, hence
  // k = floor(rint(-256 t^2) / 256) <= kmin, the same from tmin (the same fma; rounding and the
  // arithmetic shift are monotone) and the peak's value e = ldexp(p, k) with
  // p = Th (1 + e') < 2^(255/256 + 1/512) (1 + 2^-52) < 2 obeys e < 2^(kmin+1).  With 0 <= A_k < 2^ea that bounds the
  // addend of f = fma(A_k, e, f) by 2^(ea+kmin+1).  The running f is >= the background over the
  // tile, bmin = min(bg(xlo), bg(xhi)) (Horner fma, monotone for NBG <= 2; every earlier peak
  // added a non-negative amount), and bmin >= 2^(ef-1).  A double g > 0 has no neighbour closer
  // than g 2^-53, so an addend below g 2^-54 leaves fma(A, e, f) == f: the peak is a no-op for
  // every point of the tile when  ea + kmin + 1 <= ef - 55.
  // What the rule needs is a lower bound on |f| at the moment peak k is added, and the sign of
  // nothing: while every earlier peak of the window was itself left out, f IS the background,
  // whose magnitude over the window is at least lb = min(|bg(xlo)|, |bg(xhi)|) when both ends
  // have the same sign - negative backgrounds and negative amplitudes included.  An evaluated
  // peak changes f by less than 2^(ea + kmin + 1): if the background is positive and every
  // evaluated amplitude so far was >= 0, f has only grown and lb still holds; otherwise that
  // much comes off the bound (|f + a| >= |f| - |a|; the roundings of f and of the bound are
  // far inside the factor 2 the test keeps in hand), and once it reaches 0 the later peaks of
  // the window are evaluated.  Preconditions: 1 <= NBG <= 2, every amplitude finite, bg finite
  // and of one sign at both ends of the window, the window on the table-exp path.
  static constexpr bool kHasSkip = !LORENTZ && NBG >= 1 && NBG <= 2 && NPK <= 30;
  static constexpr int kPeaks = NPK;
  static __device__ __forceinline__ double bg_of(const Prep& p, double x) {
    if (NBG == 0) return 0.0;
    double f = p.bg[NBG > 0 ? NBG - 1 : 0];
#pragma unroll
    for (int j = NBG - 2; j >= 0; --j) f = __builtin_fma(f, x, p.bg[j]);
    return f;
  }
  // Bit 31 of the result (kGuardBit): the window needs the guarded exp.  The table exp wants
  // |t| < kFastT, but only where a peak is evaluated: a peak whose |t| is at least kFarT at every
  // point of the window (same side of its centre at both ends) is EXACTLY zero there in every
  // form of the Gaussian (2^-1600 is 25 binades below the smallest subnormal; A * 0 added to f
  // leaves f as it is for any finite A), so it is left out - whatever the background, whether
  // or not skipping is on - and the window is fast when every other peak has |t| < kFastT at
  // both ends.  A proposal with one very narrow peak (|t| beyond kFastT at the far end of the
  // data) therefore costs the guarded form only in the window that holds the peak, if at all,
  // not over the whole dataset.
  static constexpr unsigned kGuardBit = 0x80000000u;
  static constexpr double kFarT = 40.0;
  static __device__ __forceinline__ unsigned tile_mask(const Prep& p, double xlo, double xhi) {
    const double blo = bg_of(p, xlo), bhi = bg_of(p, xhi);
    const bool bpos = blo > 0.0 && bhi > 0.0, bneg = blo < 0.0 && bhi < 0.0;
    const double lb = fabs(blo) < fabs(bhi) ? fabs(blo) : fabs(bhi);
    // lbr: a lower bound on |f| for the running f of every point of the window (0: none);
    // mono: f has only grown from a positive background so far, so bg's bound still holds
    double lbr = (p.skip && (bpos || bneg) && finite_f64(blo) && finite_f64(bhi)) ? lb : 0.0;
    bool mono = bpos;
    unsigned m = 0;
    bool guard = false;
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      const double iwk = p.iw_of(k), ck = kSeedInLds ? p.c_of(k) : p.cv[k];
      const double tl = __builtin_fma(xlo, iwk, ck);
      const double th = __builtin_fma(xhi, iwk, ck);
      // -(binary exponent of A_k) - 56, formed here (once per 64 windows) rather than kept
      const int thr_k = -__builtin_amdgcn_frexp_exp(p.A[k]) - 56;
      const bool same_side = (tl > 0.0 && th > 0.0) || (tl < 0.0 && th < 0.0);
      const double al = fabs(tl), ah = fabs(th);
      const double tmin = al < ah ? al : ah;
      const bool far = p.afin && same_side && (tmin >= kFarT);        // NaN: false
      const bool in = (al < kFastT) && (ah < kFastT);                  // NaN: false
      guard = guard || !(far || in);
      const double kd = __builtin_fma(-tmin, tmin, p.K.magic);
      const int kmin = (int)__double_as_longlong(kd) >> 8;  // (meaningful when `in`)
      const int ef = __builtin_amdgcn_frexp_exp(lbr);       // lbr in [2^(ef-1), 2^ef)
      const bool noop =
          far || ((lbr > 0.0) && in && same_side && (kmin <= ef + thr_k));
      m |= noop ? 0u : (1u << k);
      // an evaluated peak moves f by less than 2^(ea + kmin + 1) (by less than 2^(ea + 1) when
      // its centre lies inside the window): a non-negative addend to a positive f leaves bg's
      // bound alone, anything else takes that much off it
      if (!noop && !(mono && p.A[k] >= 0.0)) {
        mono = false;
        const int km = (same_side && in) ? kmin : 0;
        const double left = lbr - ldexp(1.0, km - thr_k - 55);
        lbr = left > 0.0 ? left : 0.0;  // (NaN: 0)
      }
    }
    // per LANE: sweep() gives each lane the range of a different window (64 per pass) and later
    // broadcasts the window's word with readlane, so the tests in eval() are scalar branches
    return m | (guard ? kGuardBit : 0u);
  }
  // What a sweep over fn will cost this chain, in quarter VALU instructions per point of a lane
  // and window, summed over the windows (wave-uniform; only differences between chains matter).
  // Used to DEAL the proposals of a workgroup to its wave slots (group_logpost): the chains of a
  // workgroup meet at a barrier every window and the four waves of a SIMD share its issue slots,
  // so a SIMD that drew four expensive proposals holds everybody up.  Per window and peak, by the
  // kernel's own rules: left out 0, recurrence 3 (4.75 / 6.5 when re-seeded inside the window),
  // direct table exp 14, a guarded window 18 for every peak.
  static __device__ __forceinline__ int sweep_cost(const Prep& p, const FnDesc& fn) {
    static_assert(kHasSkip, "per-window masks");
    const int l = lane_id();
    const int64_t nw = (fn.n + kPadPoints - 1) / kPadPoints;
    int cost = 0;
    for (int64_t w0 = 0; w0 < nw; w0 += kWave) {
      const int64_t wi = w0 + l;
      const int64_t ti = wi < nw ? wi : nw - 1;
      const unsigned tw = tile_mask(p, fn.txlo[ti], fn.txhi[ti]);
      int cw = 0;
#pragma unroll
      for (int k = 0; k < NPK; ++k) {
        const unsigned b = 1u << k;
        const int ck = (p.rmask & b) ? ((p.s8 & b) ? 26 : ((p.s16 & b) ? 19 : 12)) : 56;
        cw += (tw & b) ? ck : 0;
      }
      cw = (tw & kGuardBit) ? 72 * NPK : cw;
      cost += wi < nw ? cw + (p.bgrec ? 0 : 4) : 0;
    }
    return __builtin_amdgcn_readfirstlane(wave_sum_i(cost));
  }
  // scratch: kScratchDoubles doubles of the wave's own LDS (kSeedInLds; ignored otherwise)
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc& fn, double* scratch = nullptr) {
    Prep p;
    p.sc = (lds_cdptr_t)scratch;
    bool fast = true;
    bool skip = fn.tile_skip != 0;
    bool afin = true;
#pragma unroll
    for (int j = 0; j < NBG; ++j) p.bg[j] = uniform_f64(pf(j));
    if (NBG > 1) {  // the first Horner step reads two coefficients: keep the leading one in a VGPR
      double lead = p.bg[NBG - 1];
      asm volatile("" : "+v"(lead));
      p.bg[NBG - 1] = lead;
    }
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      p.A[k] = uniform_f64(pf(NBG + 3 * k));
      const double mu = pf(NBG + 3 * k + 1);
      // Gaussian: exp(-((x-mu)/w)^2) = 2^(-t^2) with t = x*iw' - mu*iw', iw' = sqrt(log2 e)/w
      const double iw = LORENTZ ? 1.0 / pf(NBG + 3 * k + 2) : kSqrtLog2e / pf(NBG + 3 * k + 2);
      const double iwu = uniform_f64(iw);
      const double cu = uniform_f64(-mu * iw);  // additive constant of the fma below
      if constexpr (kSeedInLds) {
        if (lane_id() == 0) {
          scratch[4 * k] = cu;
          scratch[4 * k + 1] = iwu;
        }
        p.mu[0] = p.iw[0] = p.cv[0] = 0.0;
      } else {
        p.iw[k] = iwu;
        p.mu[k] = cu;
        double cvk = cu;
        if (NPK <= 2) asm volatile("" : "+v"(cvk));  // (more peaks: eval_n moves it per block)
        p.cv[k] = cvk;
      }
      // t is linear in x: its extremes sit at the ends of the data range (NaN fails the test)
      const double ta = fabs(__builtin_fma(fn.xmin, iwu, cu));
      const double tb = fabs(__builtin_fma(fn.xmax, iwu, cu));
      fast = fast && (ta < kFastT) && (tb < kFastT);
      // |A_k| < 2^ea; an infinite or NaN amplitude switches skipping off
      afin = afin && finite_f64(p.A[k]);
    }
    p.fast = fast;
    // (with kHasSkip the fast / guarded choice is made per window, tile_mask; without it per step)
    p.skip = skip && afin && kHasSkip;
    p.afin = afin;
    p.K.pin();
    unsigned rmask = 0, s16 = 0, s8 = 0;
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      // (the same iw as above: recomputed rather than kept across the loops)
      const double iwk = uniform_f64(LORENTZ ? 1.0 / pf(NBG + 3 * k + 2) : kSqrtLog2e / pf(NBG + 3 * k + 2));
      const double dl = fn.grid_H * iwk;  // D_k
      const double rm2d_k = uniform_f64(-2.0 * dl), rnd2_k = uniform_f64(-(dl * dl));
      if constexpr (kSeedInLds) {
        if (lane_id() == 0) {
          scratch[4 * k + 2] = rm2d_k;
          scratch[4 * k + 3] = rnd2_k;
        }
        p.rm2d[0] = p.rnd2[0] = 0.0;
      } else {
        p.rm2d[k] = rm2d_k;
        p.rnd2[k] = rnd2_k;
      }
      p.rq[k] = uniform_f64(mexp2(2.0 * rnd2_k));
      // three classes by the peak's width in grid points: S |D| <= 1 for the seeding period S the
      // peak gets (kSeedSteps, half of it, a quarter of it); NaN fails all three
      const bool base = kHasRec && (fast || kHasSkip) && fn.grid_H != 0.0;
      const double ad = fabs(dl);
      const bool ok32 = base && (ad * (double)kSeedSteps <= 1.0);
      const bool ok16 = kMultiSeed ? base && (ad * (double)(kSeedSteps / 2) <= 1.0) : ok32;
      const bool ok8 = kMultiSeed ? base && (ad * (double)(kSeedSteps / 4) <= 1.0) : ok32;
      rmask |= ok8 ? (1u << k) : 0u;
      s16 |= (ok8 && !ok32) ? (1u << k) : 0u;
      s8 |= (ok8 && !ok16) ? (1u << k) : 0u;
    }
    p.rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)rmask);
    p.s16 = (unsigned)__builtin_amdgcn_readfirstlane((int)s16);
    p.s8 = (unsigned)__builtin_amdgcn_readfirstlane((int)s8);
    // (<= 2 peaks: with more, the second copy of the run-time-masked tile loop costs more than
    // the reads it saves - config 3: 6.1e5 -> 4.5e5 chain-steps/s, measured)
    p.bgrec = NBG >= 1 && NBG <= 2 && NPK <= 2 && p.rmask == (1u << NPK) - 1u;
#ifdef MHX_NO_BGREC  // (build knob for A/B measurements; the oracle's mirror knows only the default)
    p.bgrec = false;
#endif
    p.bgH = NBG == 2 ? uniform_f64(p.bg[1] * fn.grid_H) : 0.0;
    if constexpr (kSeedInLds) __builtin_amdgcn_wave_barrier();  // (the wave's own LDS writes above)
    return p;
  }
  // PER-WINDOW GRIDS.  The recurrence needs the step of t from one point of a lane to its next,
  // D = 64 h iw - a property of the 2048-point window being swept, not of the dataset: real
  // spectra are often piecewise uniform (concatenated scans, a re-gridded stretch), and a seed is
  // formed at the lane's ACTUAL x anyway.  For datasets that are not one grid the host gives every
  // window its own H = 64 h (0: not a grid; FnDesc::tgh, mhx_engine.cpp), and sweep() calls this
  // when the window about to be swept has another H than the last one: what prepare() derives
  // from fn.grid_H, restated for H (same operations: a dataset whose windows all carry one H
  // gives the bits of the whole-dataset rule).  Rare by construction - runs of windows share
  // their H bit for bit - so its cost (a degree-11 exp2 per peak) does not matter.
  static __device__ __forceinline__ void regrid(Prep& p, double H) {
    unsigned rmask = 0, s16 = 0, s8 = 0;
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      const double iwk = uniform_f64(p.iw_of(k));
      const double dl = H * iwk;  // D_k
      const double rm2d_k = uniform_f64(-2.0 * dl), rnd2_k = uniform_f64(-(dl * dl));
      if constexpr (kSeedInLds) {
        if (lane_id() == 0) {
          typedef __attribute__((address_space(3))) double* lds_wptr_t;
          lds_wptr_t sc = (lds_wptr_t)p.sc;
          sc[4 * k + 2] = rm2d_k;
          sc[4 * k + 3] = rnd2_k;
        }
      } else {
        p.rm2d[k] = rm2d_k;
        p.rnd2[k] = rnd2_k;
      }
      p.rq[k] = uniform_f64(mexp2(2.0 * rnd2_k));
      const bool base = kHasRec && kHasSkip && H != 0.0;
      const double ad = fabs(dl);
      const bool ok32 = base && (ad * (double)kSeedSteps <= 1.0);
      const bool ok16 = kMultiSeed ? base && (ad * (double)(kSeedSteps / 2) <= 1.0) : ok32;
      const bool ok8 = kMultiSeed ? base && (ad * (double)(kSeedSteps / 4) <= 1.0) : ok32;
      rmask |= ok8 ? (1u << k) : 0u;
      s16 |= (ok8 && !ok32) ? (1u << k) : 0u;
      s8 |= (ok8 && !ok16) ? (1u << k) : 0u;
    }
    p.rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)rmask);
    p.s16 = (unsigned)__builtin_amdgcn_readfirstlane((int)s16);
    p.s8 = (unsigned)__builtin_amdgcn_readfirstlane((int)s8);
    p.bgrec = NBG >= 1 && NBG <= 2 && NPK <= 2 && p.rmask == (1u << NPK) - 1u;
#ifdef MHX_NO_BGREC
    p.bgrec = false;
#endif
    p.bgH = NBG == 2 ? uniform_f64(p.bg[1] * H) : 0.0;
    if constexpr (kSeedInLds) __builtin_amdgcn_wave_barrier();
  }
  static __device__ __forceinline__ bool fast_ok(const Prep& p) { return p.fast; }
  // mask (wave-uniform, from tile_mask): bit k clear = peak k is a no-op for this tile
  template <bool FAST>
  static __device__ __forceinline__ double eval(const Prep& p, double x, unsigned mask = ~0u) {
    double f = 0.0;
    if (NBG > 0) {
      f = p.bg[NBG - 1];
#pragma unroll
      for (int j = NBG - 2; j >= 0; --j) f = __builtin_fma(f, x, p.bg[j]);
    }
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      if (FAST && kHasSkip && !((mask >> k) & 1u)) continue;
      const double t = kSeedInLds ? __builtin_fma(x, p.iw_of(k), p.c_of(k))
                                  : __builtin_fma(x, p.iw[k], p.cv[k]);
      if (LORENTZ)
        f = __builtin_fma(p.A[k], frcp(__builtin_fma(t, t, 1.0)), f);
      else
        f = __builtin_fma(p.A[k], FAST ? mexp2_negsq(t, p.K) : mexp2_negsq_safe(t), f);
    }
    return f;
  }
  // The same for P points at once, peak by peak: with a run-time mask every peak is a basic
  // block of its own, and evaluating the P points inside it gives the block P independent fp64
  // chains (a dependent v_fma_f64 can issue only every other slot: tools/microbench/fma_chain).
  // Each point's operations and their order are those of eval(): identical bits.
  template <bool FAST, int P>
  static __device__ __forceinline__ void eval_n(const Prep& p, const double (&x)[P], unsigned mask,
                                                double (&f)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      f[i] = 0.0;
      if (NBG > 0) {
        f[i] = p.bg[NBG - 1];
#pragma unroll
        for (int j = NBG - 2; j >= 0; --j) f[i] = __builtin_fma(f[i], x[i], p.bg[j]);
      }
    }
    if constexpr (FAST && !LORENTZ && NPK * P <= 4) {
      // (compile-time masks, <= 2 peaks x 2 points) all chains of the iteration as one batch
      double t[NPK * P], v[NPK * P];
      bool on[NPK * P];
#pragma unroll
      for (int k = 0; k < NPK; ++k)
#pragma unroll
        for (int i = 0; i < P; ++i) {
          on[k * P + i] = !kHasSkip || ((mask >> k) & 1u);
          t[k * P + i] = on[k * P + i] ? fma_svv(x[i], p.iw[k], p.cv[k]) : 0.0;
        }
      mexp2_negsq_batch<NPK * P>(t, on, p.K, v);
#pragma unroll
      for (int k = 0; k < NPK; ++k)
#pragma unroll
        for (int i = 0; i < P; ++i)
          if (on[k * P + i]) f[i] = __builtin_fma(p.A[k], v[k * P + i], f[i]);
      return;
    }
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      if (FAST && kHasSkip && !((mask >> k) & 1u)) continue;
      if constexpr (FAST && !LORENTZ) {
        // one peak (a basic block of its own under a run-time mask): its P points as one batch.
        // The additive constant moves to a vector register HERE (one v_mov per block and P
        // points) rather than being pinned in one for the whole sweep: NPK register pairs less
        double t[P], v[P], ck = p.c_of(k);
        asm volatile("" : "+v"(ck));
        bool on[P];
        const double iwk = p.iw_of(k);
#pragma unroll
        for (int i = 0; i < P; ++i) {
          on[i] = true;
          t[i] = kSeedInLds ? __builtin_fma(x[i], iwk, ck) : fma_svv(x[i], iwk, ck);
        }
        mexp2_negsq_batch<P>(t, on, p.K, v);
#pragma unroll
        for (int i = 0; i < P; ++i) f[i] = __builtin_fma(p.A[k], v[i], f[i]);
      } else {
#pragma unroll
        for (int i = 0; i < P; ++i) {
          const double t = kSeedInLds ? __builtin_fma(x[i], p.iw_of(k), p.c_of(k))
                                      : __builtin_fma(x[i], p.iw[k], p.cv[k]);
          if (LORENTZ)
            f[i] = __builtin_fma(p.A[k], frcp(__builtin_fma(t, t, 1.0)), f[i]);
          else
            f[i] = __builtin_fma(p.A[k], mexp2_negsq_safe(t), f[i]);
        }
      }
    }
  }
};

// any (nbg, npk): parameters stay in the wave's LDS slot and are re-read per point
template <bool LORENTZ>
struct PeaksModelDyn {
  struct Prep {
    const double* q;  // LDS: bg[nbg], then (A, mu, 1/w) per peak
    int nbg, npk;
  };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc& f, double* scratch) {
    Prep p;
    p.nbg = f.shape[0];
    p.npk = f.shape[1];
    int l = lane_id();
    int np = p.nbg + 3 * p.npk;
    if (l < np) {
      // same folding as PeaksModel: slot mu holds -mu*iw, slot w holds iw (Gaussian: *sqrt(log2 e))
      const int r = l >= p.nbg ? (l - p.nbg) % 3 : -1;
      double v = pf(l);
      if (r == 2) v = (LORENTZ ? 1.0 : kSqrtLog2e) / v;
      if (r == 1) v = -v * ((LORENTZ ? 1.0 : kSqrtLog2e) / pf(l + 1));
      scratch[l] = v;
    }
    p.q = scratch;
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    double f = 0.0;
    if (p.nbg > 0) {
      f = p.q[p.nbg - 1];
      for (int j = p.nbg - 2; j >= 0; --j) f = __builtin_fma(f, x, p.q[j]);
    }
    for (int k = 0; k < p.npk; ++k) {
      const double* q = p.q + p.nbg + 3 * k;
      const double t = __builtin_fma(x, q[2], q[1]);
      if (LORENTZ)
        f = __builtin_fma(q[0], frcp(__builtin_fma(t, t, 1.0)), f);
      else
        f = __builtin_fma(q[0], mexp2_negsq_safe(t), f);
    }
    return f;
  }
};

// Models that offer a cheaper evaluation valid under a per-step precondition declare
// kHasFast, fast_ok(prep) and eval<FAST>; the others only eval.
template <class M, class = void>
struct model_has_fast { static constexpr bool value = false; };
template <class M>
struct model_has_fast<M, decltype((void)M::kHasFast, void())> {
  static constexpr bool value = M::kHasFast;
};
template <class M, class = void>
struct model_has_skip { static constexpr bool value = false; };
template <class M>
struct model_has_skip<M, decltype((void)M::kHasSkip, void())> {
  static constexpr bool value = M::kHasSkip;
};
// (a model that declares kHasFast - even as false, like the Lorentzian peaks - has the templated
// eval<FAST>)
template <class M, class = void>
struct model_eval_templated { static constexpr bool value = false; };
template <class M>
struct model_eval_templated<M, decltype((void)M::kHasFast, void())> {
  static constexpr bool value = true;
};
template <class M, bool FAST>
__device__ __forceinline__ double model_eval(const typename M::Prep& p, double x,
                                             unsigned mask = ~0u) {
  if constexpr (model_eval_templated<M>::value) {
    if constexpr (model_has_skip<M>::value)
      return M::template eval<FAST>(p, x, mask);
    else
      return M::template eval<FAST>(p, x);
  } else {
    return M::eval(p, x);
  }
}
template <class M, class = void>
struct model_peaks { static constexpr int value = 0; };
template <class M>
struct model_peaks<M, decltype((void)M::kPeaks, void())> {
  static constexpr int value = M::kPeaks;
};
template <class M, class = void>
struct model_has_rec { static constexpr bool value = false; };
template <class M>
struct model_has_rec<M, decltype((void)M::kHasRec, void())> {
  static constexpr bool value = M::kHasRec;
};
struct NoRec {};
template <class M, bool = model_has_rec<M>::value>
struct model_rec_state { typedef NoRec type; };
template <class M>
struct model_rec_state<M, true> { typedef typename M::Rec type; };
template <class M, bool = model_has_rec<M>::value>
struct model_seed_steps { static constexpr int value = 16; };
template <class M>
struct model_seed_steps<M, true> { static constexpr int value = M::kSeedSteps; };
// models that keep part of their per-step constants in the wave's LDS scratch
// (PeaksModel::kSeedInLds): prepare(pf, fn, scratch)
template <class M, class = void>
struct model_wants_scratch { static constexpr bool value = false; };
template <class M>
struct model_wants_scratch<M, decltype((void)M::kSeedInLds, void())> {
  static constexpr bool value = M::kSeedInLds;
};
template <class M, class PF>
__device__ __forceinline__ typename M::Prep model_prepare(PF pf, const FnDesc& fn, double* scratch) {
  if constexpr (model_wants_scratch<M>::value) return M::prepare(pf, fn, scratch);
  else return M::prepare(pf, fn);
}
// models that read a second column of x ("multiple or linked independent variables", M:1136-1137:
// (elt x 0), (elt x 1)): expression models whose text names xcol1 - eval2(prep, xcol0, xcol1)
template <class M, class = void>
struct model_xcols { static constexpr int value = 1; };
template <class M>
struct model_xcols<M, decltype((void)M::kXCols, void())> {
  static constexpr int value = M::kXCols;
};
template <class M, class = void>
struct model_has_eval_n { static constexpr bool value = false; };
template <class M>
struct model_has_eval_n<M, decltype((void)M::kHasEvalN, void())> {
  static constexpr bool value = M::kHasEvalN;
};
// P points of one lane at once (models without an eval_n of their own: point after point)
template <class M, bool FAST, int P>
__device__ __forceinline__ void model_eval_n(const typename M::Prep& p, const double (&x)[P],
                                             unsigned mask, double (&f)[P]) {
  if constexpr (model_has_skip<M>::value || model_has_eval_n<M>::value) {
    M::template eval_n<FAST, P>(p, x, mask, f);
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) f[i] = model_eval<M, FAST>(p, x[i], mask);
  }
}

template <int NP>
struct PolyModel {
  struct Prep { double c[NP]; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {
    Prep p;
#pragma unroll
    for (int j = 0; j < NP; ++j) p.c[j] = uniform_f64(pf(j));
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    double f = p.c[NP - 1];
#pragma unroll
    for (int j = NP - 2; j >= 0; --j) f = __builtin_fma(f, x, p.c[j]);
    return f;
  }
};
struct PolyModelDyn {
  struct Prep { const double* c; int np; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc& f, double* scratch) {
    Prep p;
    p.np = f.n_idx;
    int l = lane_id();
    if (l < p.np) scratch[l] = pf(l);
    p.c = scratch;
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    if (p.np <= 0) return 0.0;
    double f = p.c[p.np - 1];
    for (int j = p.np - 2; j >= 0; --j) f = __builtin_fma(f, x, p.c[j]);
    return f;
  }
};

struct LorderModel {
  struct Prep { double scale, cm, sm, x0, ilw, bg0, bg1; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {
    Prep p;
    double mix = pf(3);
    p.scale = uniform_f64(pf(0));
    p.cm = uniform_f64(cos(mix));
    p.sm = uniform_f64(sin(mix));
    p.x0 = uniform_f64(pf(2));
    p.ilw = uniform_f64(1.0 / pf(1));
    p.bg0 = uniform_f64(pf(4));
    p.bg1 = uniform_f64(pf(5));
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    double u = (x - p.x0) * p.ilw;
    double uu = u * u;
    double q = 1.0 + uu;
    double num = p.cm * (-2.0 * u) + p.sm * (1.0 - uu);
    return ((p.scale * num) / (q * q) + p.bg0) + p.bg1 * x;
  }
};

struct ExpDecayModel {
  struct Prep { double A, itau, c; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {
    Prep p;
    p.A = uniform_f64(pf(0));
    p.itau = uniform_f64(1.0 / pf(1));
    p.c = uniform_f64(pf(2));
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    return __builtin_fma(p.A, gexp(-(x * p.itau)), p.c);  // < 1 ulp for any argument
  }
};

struct SinusoidModel {
  struct Prep { double A, om, ph, c; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {
    Prep p;
    p.A = uniform_f64(pf(0));
    p.om = uniform_f64(pf(1));
    p.ph = uniform_f64(pf(2));
    p.c = uniform_f64(pf(3));
    return p;
  }
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    return __builtin_fma(p.A, sin(p.om * x + p.ph), p.c);
  }
};

struct PVoigt2Model {
  // per peak: u = x*iw + c (c = -mu*iw); Lorentzian eta/(1+u^2) by frcp; Gaussian
  // (1-eta) exp(-u^2) = (1-eta) 2^(-t^2), t = x*(iw g) + c g, g = sqrt(log2 e).  Fast path (the
  // low-dword exponent trick of mexp2_negsq) when |t| < kFastT over the dataset's x range.
  static constexpr bool kHasFast = true;
  struct Prep {
    double A, b0, b1, c1, iw1, eta1, om1, c2p, iw2, eta2, om2, rho, c2;
    double g1c, g1w, g2c, g2w;  // the same lines scaled by g (VGPR-pinned constants of the fma)
    double k1, k2, k3, k4;      // A eta1, A (1 - eta1), A rho eta2, A rho (1 - eta2): finish_fast
    Exp2K K;
    bool fast;
  };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc& fn) {
    Prep p;
    p.A = uniform_f64(pf(0)); p.b0 = uniform_f64(pf(1)); p.b1 = uniform_f64(pf(2));
    const double iw1 = 1.0 / pf(4), iw2 = 1.0 / pf(7);
    p.iw1 = uniform_f64(iw1); p.c1 = uniform_f64(-pf(3) * iw1); p.eta1 = uniform_f64(pf(5));
    p.iw2 = uniform_f64(iw2); p.c2p = uniform_f64(-pf(6) * iw2); p.eta2 = uniform_f64(pf(8));
    p.om1 = uniform_f64(1.0 - pf(5)); p.om2 = uniform_f64(1.0 - pf(8));
    p.rho = uniform_f64(pf(9)); p.c2 = uniform_f64(pf(10));
    p.k1 = uniform_f64(p.A * p.eta1); p.k2 = uniform_f64(p.A * p.om1);
    p.k3 = uniform_f64((p.A * p.rho) * p.eta2); p.k4 = uniform_f64((p.A * p.rho) * p.om2);
    p.g1w = uniform_f64(p.iw1 * kSqrtLog2e); p.g2w = uniform_f64(p.iw2 * kSqrtLog2e);
    double a = p.c1 * kSqrtLog2e, b = p.c2p * kSqrtLog2e;
    asm volatile("" : "+v"(a));
    asm volatile("" : "+v"(b));
    p.g1c = a; p.g2c = b;
    // likewise the additive constants of u1, u2 and of the background: an fma may read ONE scalar
    // operand, and left in scalar registers these three were copied into a vector register in
    // every iteration of the tile loop (3 v_mov_b64 per 2 points: 1.5 of 55 instructions per point)
    {
      double c1v = p.c1, c2v = p.c2p, b0v = p.b0, b1v = p.b1;
      asm volatile("" : "+v"(c1v));
      asm volatile("" : "+v"(c2v));
      asm volatile("" : "+v"(b0v));
      asm volatile("" : "+v"(b1v));  // (Horner's inner fma reads c2 from the scalar file)
      p.c1 = c1v; p.c2p = c2v; p.b0 = b0v; p.b1 = b1v;
    }
    const double e1 = fabs(__builtin_fma(fn.xmin, p.g1w, p.g1c)), e2 = fabs(__builtin_fma(fn.xmax, p.g1w, p.g1c));
    const double e3 = fabs(__builtin_fma(fn.xmin, p.g2w, p.g2c)), e4 = fabs(__builtin_fma(fn.xmax, p.g2w, p.g2c));
    p.fast = (e1 < kFastT) && (e2 < kFastT) && (e3 < kFastT) && (e4 < kFastT);  // NaN fails
    // ... and the fast path forms both Lorentzians from ONE reciprocal, 1 / ((1 + u1^2)(1 + u2^2))
    // (finish_fast): the product must not overflow anywhere in the data range.  |t| < kFastT
    // already bounds |u| by 2890 / sqrt(log2 e), so this holds whenever the line above does; it
    // is spelt out because it is what finish_fast relies on.
    {
      const double g = kSqrtLog2e;
      const double d1 = __builtin_fma(e1 > e2 ? e1 : e2, (e1 > e2 ? e1 : e2) / (g * g), 1.0);
      const double d2 = __builtin_fma(e3 > e4 ? e3 : e4, (e3 > e4 ? e3 : e4) / (g * g), 1.0);
      p.fast = p.fast && (d1 * d2 < 1e300);
    }
    p.K.pin();
    return p;
  }
  static __device__ __forceinline__ bool fast_ok(const Prep& p) { return p.fast; }
  // everything but the two Gaussians
  static __device__ __forceinline__ double finish(const Prep& p, double x, double u1, double u2,
                                                  double g1, double g2) {
    const double l1 = frcp(__builtin_fma(u1, u1, 1.0)), l2 = frcp(__builtin_fma(u2, u2, 1.0));
    const double pv1 = __builtin_fma(p.eta1, l1, p.om1 * g1);
    const double pv2 = __builtin_fma(p.eta2, l2, p.om2 * g2);
    const double bg = __builtin_fma(p.c2, x * x, __builtin_fma(p.b1, x, p.b0));
    return __builtin_fma(p.A, __builtin_fma(p.rho, pv2, pv1), bg);
  }
  // ... on the fast path with ONE reciprocal for the two Lorentzians: y = 1 / (d1 d2),
  // 1 / d1 = y d2, 1 / d2 = y d1 - 8 instructions where two reciprocals take 10 (Prep::fast
  // guarantees that d1 d2 stays finite; each quotient within 2.5 ulp instead of 1).  Within the
  // stated 1e-12 sum |term| of the reference's nested form (tests/test_gpu_fullsize.py, config 4).
  static __device__ __forceinline__ double finish_fast(const Prep& p, double x, double u1,
                                                       double u2, double g1, double g2) {
    const double d1 = __builtin_fma(u1, u1, 1.0), d2 = __builtin_fma(u2, u2, 1.0);
    const double y = frcp(d1 * d2);
    const double l1 = y * d2, l2 = y * d1;
    // A (pv1 + rho pv2) + bg with the four products of its constants formed once per step:
    // four fmas on top of the background where the nested form takes six operations
    double f = __builtin_fma(__builtin_fma(p.c2, x, p.b1), x, p.b0);  // Horner: 2, not 3
    f = __builtin_fma(p.k1, l1, f);
    f = __builtin_fma(p.k2, g1, f);
    f = __builtin_fma(p.k3, l2, f);
    return __builtin_fma(p.k4, g2, f);
  }
  template <bool FAST>
  static __device__ __forceinline__ double eval(const Prep& p, double x) {
    const double u1 = __builtin_fma(x, p.iw1, p.c1), u2 = __builtin_fma(x, p.iw2, p.c2p);
    double g1, g2;
    if (FAST) {
      g1 = mexp2_negsq(__builtin_fma(x, p.g1w, p.g1c), p.K);
      g2 = mexp2_negsq(__builtin_fma(x, p.g2w, p.g2c), p.K);
      return finish_fast(p, x, u1, u2, g1, g2);
    } else {
      g1 = mexp2_negsq_safe(u1 * kSqrtLog2e);
      g2 = mexp2_negsq_safe(u2 * kSqrtLog2e);
    }
    return finish(p, x, u1, u2, g1, g2);
  }
  // P points at once: the 2 P table reads of the Gaussians are issued before anything uses one
  // (the Lorentzians' reciprocals fill the wait); per point the operations are eval()'s
  static constexpr bool kHasEvalN = true;
  template <bool FAST, int P>
  static __device__ __forceinline__ void eval_n(const Prep& p, const double (&x)[P], unsigned,
                                                double (&f)[P]) {
    if constexpr (FAST) {
      double t[2 * P], v[2 * P];
      bool on[2 * P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        on[2 * i] = on[2 * i + 1] = true;
        t[2 * i] = fma_svv(x[i], p.g1w, p.g1c);
        t[2 * i + 1] = fma_svv(x[i], p.g2w, p.g2c);
      }
      mexp2_negsq_batch<2 * P>(t, on, p.K, v);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const double u1 = __builtin_fma(x[i], p.iw1, p.c1), u2 = __builtin_fma(x[i], p.iw2, p.c2p);
        f[i] = finish_fast(p, x[i], u1, u2, v[2 * i], v[2 * i + 1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < P; ++i) f[i] = eval<false>(p, x[i]);
    }
  }
};

}  // inline namespace MHX_FAMILY
}  // namespace mhx
