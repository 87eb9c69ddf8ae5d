#!/usr/bin/env python3
"""CPU-side check that the embedded device sources compile under hiprtc for gfx950 (no GPU
needed): mirrors what mhx_rtc.cpp generates for one expression model and one prior body."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lisp-mcmc_amd", "csrc")
SRC = r'''
#include "mhx_kernels.hpp"
namespace mhx {
__device__ __forceinline__ double mhx_ux_min(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double mhx_ux_max(double a, double b) { return a > b ? a : b; }
struct UserModel0 {
  struct Prep { double p[2]; };
  template <class PF>
  static __device__ __forceinline__ Prep prepare(PF pf, const FnDesc&) {
    Prep q; q.p[0] = uniform_f64(pf(0)); q.p[1] = uniform_f64(pf(1)); return q; }
  static __device__ __forceinline__ double eval(const Prep& q, double x) {
    const double p_b = q.p[0]; const double p_m = q.p[1];
    return (double)(p_b + p_m * x + exp(-x) + pow(x, 2.0));
  }
};
struct UserSpec {
  template <class PF>
  static __device__ __forceinline__ double loglik(const FnDesc& f, PF pf, bool active, GroupLds& lds, double* scratch) {
    switch (f.user_slot) {
      case 0: return GenericSpec::by_lik<UserModel0>(f, pf, active, lds);
      default: return GenericSpec::loglik(f, pf, active, lds, scratch);
    }
  }
  static __device__ __forceinline__ double logprior(const FnDesc& f, const double* th, double bounds_total) {
    switch (f.prior_slot) {
      case 0: { const double p_b = th[0]; return (double)(bounds_total + (p_b > 1.0 ? -1e9 : 0.0)); }
      default: return bounds_total;
    }
  }
};
}
using namespace mhx;
extern "C" __global__ __launch_bounds__(512) void mhx_user_logpost(const ProblemDesc* P, const double* theta, int64_t n, double* out, double* parts) {
  k_logpost_body<UserSpec>(P, theta, n, out, parts); }
extern "C" __global__ __launch_bounds__(512, 4) void mhx_user_adaptive(const ProblemDesc* P, ChainState S, RunDesc R, int64_t max_iters, int plain) {
  k_adaptive_body<UserSpec>(P, S, R, max_iters, plain); }
'''


def main():
    rtc = C.CDLL("/opt/rocm/lib/libhiprtc.so")
    hdr_files = [("mhx_kernels.hpp", os.path.join(CSRC, "mhx_kernels.hpp")),
                 ("mhx_device.hpp", os.path.join(CSRC, "mhx_device.hpp")),
                 ("mhx_types.hpp", os.path.join(CSRC, "mhx_types.hpp")),
                 ("../../include/mhx.h", os.path.join(ROOT, "include", "mhx.h"))]
    srcs = (C.c_char_p * 4)(*[open(p, "rb").read() for _, p in hdr_files])
    names = (C.c_char_p * 4)(*[n.encode() for n, _ in hdr_files])
    prog = C.c_void_p()
    rc = rtc.hiprtcCreateProgram(C.byref(prog), SRC.encode(), b"mhx_user.hip", 4, srcs, names)
    assert rc == 0, rc
    opts = (C.c_char_p * 4)(b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-ffp-contract=off")
    rc = rtc.hiprtcCompileProgram(prog, 4, opts)
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(n.value + 1)
    rtc.hiprtcGetProgramLog(prog, log)
    if rc != 0:
        print(log.value.decode(errors="replace")[:6000])
        sys.exit(1)
    rtc.hiprtcGetCodeSize(prog, C.byref(n))
    print("hiprtc ok, code object %d bytes" % n.value)


if __name__ == "__main__":
    main()
