// Device-side pieces of the two-sided Jacobi eigenvalue iteration shared by the in-LDS step kernel
// (kernels_narrow.hip, n <= 64) and the large-tensor path (kernels_big.hip, n <= 128).
#pragma once
#include <hip/hip_runtime.h>

namespace tnml {

constexpr int kJacobiMaxSweeps = 30;

// Workgroup barrier for hand-offs through LDS only.  __syncthreads() is a workgroup-scope fence plus s_barrier, and the fence
// drains the wave's GLOBAL stores too (s_waitcnt vmcnt(0)): behind a phase that stored its results to HBM every such barrier
// costs a memory round trip (2-5 k cycles measured per phase of the step kernel).  Here only LDS traffic is waited for;
// global hand-offs to other workgroups keep their own explicit s_waitcnt vmcnt(0) + flag protocol.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte agent-scope (sc1) stores and loads for hand-offs between workgroups of one launch (MI355X_MICROARCH.md: a scalar sc1
// store is one fabric write each -- a dword store costs ~6x the dwordx4 time per byte).  Buffer instructions with the sc1
// cache-policy bit (aux bit 4), issued through the compiler's builtins so that it tracks their completion itself.  `base`
// must be wave-uniform; `byte_off` is the lane's byte offset from it (16-byte aligned).
typedef unsigned tn_uvec4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t sc1_rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
__device__ inline void st_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off, tn_uvec4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}
__device__ inline tn_uvec4 ld_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
}

// Rotation (c, s) that annihilates g in [[a, g], [g, b]] under J = [[c, s], [-s, c]] (columns:
// a' = c a - s b, b' = s a + c b).  level: 0 negligible (identity), 1 small, 2 large -- the sweep
// loop stops once a whole sweep made only small rotations (quadratic convergence then leaves
// off-diagonals below the tolerance).
// Convergence thresholds of the Jacobi iteration, all on g^2 / scale2 with
//   scale2 = max(a b, (kKeptFrac * lambda_m)^2),   lambda_m = m-th largest diagonal entry:
// diagonal entries far below the smallest eigenvalue that is KEPT are treated as if they were at
// that level, i.e. the discarded cluster is not resolved to high relative accuracy (it only has to
// be separated from the kept subspace).  Emulated on headline-shaped matrices: 7.5 sweeps instead
// of 9.4, truncated product within 2e-8 of LAPACK's (tests/emulation/jacobi_emulation.py).
constexpr double kJacobiTol2 = 1e-14;     // below: pair left alone
// "big" rotation: g^2 / scale2 above NarrowParams::svd_stop2 (default kJacobiStop2).  A sweep without one ends
// the iteration: quadratic convergence then leaves off-diagonals of relative size ~svd_stop2.  Measured on
// the merged tensors of a C3-shaped run (tests/emulation/jacobi_correction_emulation.py, probe_svd_accuracy.py):
//   svd_stop2   sweeps (n = 40)   worst |A.C - best rank-m| / max|B|   worst relative error of a kept sigma
//   1e-4        3.9               2e-4                                  5e-6
//   1e-6        5.0               6e-6                                  7e-11
//   1e-8        5.9               1e-6                                  8e-12
constexpr double kKeptFrac = 0.2;
constexpr double kJacobiAbs = 1e-15;      // |g| / trace floor: eigenvalues under 1e-15 trace are float32 noise of B

// Rotation J = [[c, s], [-s, c]] annihilating g in [[a, g], [g, b]] (columns: a' = c a - s b,
// b' = s a + c b); t = s / c.  G is pre-scaled to trace ~ 1, so the float evaluation of
// t = 2g / (d + sign(d) sqrt(d^2 + 4 g^2)) can neither overflow nor (above the floor) underflow;
// c is refined to float64 by one Newton step on rsqrt so that c^2 + s^2 = 1 to ~1e-14.
struct Rot { double c, s, t; int level; };
__device__ inline Rot jacobi_rot(double a, double b, double g, double kept2, double abs2, double big2) {
  Rot r; r.c = 1.0; r.s = 0.0; r.t = 0.0; r.level = 0;
  const double g2 = g * g;
  const double sc = fmax(fabs(a * b), kept2);
  const bool act = g2 > fmax(kJacobiTol2 * sc, abs2);          // false for g == 0 and NaN
  const float df = (float)(b - a), gf = (float)g;
  const float hyp = __builtin_amdgcn_sqrtf(fmaf(df, df, 4.f * gf * gf));
  const float t = 2.f * gf * __builtin_amdgcn_rcpf(df + copysignf(hyp, df));
  const double td = (double)t;
  const double x = fma(td, td, 1.0);                           // in [1, 2]
  double c0 = (double)__builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
  c0 = c0 * fma(-0.5 * x, c0 * c0, 1.5);
  if (act) { r.c = c0; r.s = c0 * td; r.t = td; r.level = (g2 > big2 * sc) ? 2 : 1; }
  return r;
}

// The same rotation from float32 arithmetic only, for the look-ahead chain of the in-LDS kernel (kernels_narrow.hip phase 7).
// A dependent float64 instruction costs 40 cycles on gfx950 and a float32 one 8-14 (tools/ubench/prims.hip), and this chain
// sets the length of a Jacobi round.  Returned: t = tan(angle) and c0 ~ 1 / sqrt(1 + t^2), both float32-exact numbers; the
// threads that APPLY the rotation refine c to float64 themselves,
//     c = c0 (1.5 - 0.5 (1 + t^2) c0^2),   s = c t            (one Newton step: c^2 + s^2 = 1 to ~1e-14),
// so every rotation is orthogonal to float64 accuracy whatever the accuracy of t -- the angle only decides how well the
// off-diagonal element is annihilated (relative error ~1e-7: the next sweep sees an element 1e-7 of the old one), and the
// thresholds below are compared at float32 accuracy, which their margins (factors of 1e6 and more) absorb.
struct RotT { float t, c0; int level; };
__device__ inline RotT jacobi_rot_f32(float a, float b, float g, float kept2, float abs2, float big2) {
  RotT r; r.t = 0.f; r.c0 = 1.f; r.level = 0;
  const float g2 = g * g;
  const float sc = fmaxf(fabsf(a * b), kept2);
  const bool act = g2 > fmaxf((float)kJacobiTol2 * sc, abs2);   // false for g == 0 and NaN
  const float df = b - a;
  const float hyp = __builtin_amdgcn_sqrtf(fmaf(df, df, 4.f * g2));
  const float t = 2.f * g * __builtin_amdgcn_rcpf(df + copysignf(hyp, df));
  const float c0 = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
  if (act) { r.t = t; r.c0 = c0; r.level = (g2 > big2 * sc) ? 2 : 1; }
  return r;
}
// float64 cosine of the rotation (t, c0) published by jacobi_rot_f32: c0 * corr, corr = 1.5 - 0.5 (1 + t^2) c0^2
__device__ inline double rot_corr(double t, double c0) { return fma(-0.5 * fma(t, t, 1.0), c0 * c0, 1.5); }

// one-lane wave shift of a double: CTRL 0x138 = wave_shr:1 (lane i receives lane i-1), 0x130 = wave_shl:1
template <int CTRL> __device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

template <int CTRL> __device__ inline float dpp_f32(float v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, 0xf, 0xf, false));
}

// wave-wide maximum of a float without touching the LDS crossbar: xor-butterfly inside every 16-lane row by DPP
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row maxima by v_readlane
__device__ inline float wave_max_f32(float v) {
#define TNML_DPPMAX(ctrl)                                                                                         \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                       ctrl, 0xf, 0xf, false)))
  TNML_DPPMAX(0xB1);
  TNML_DPPMAX(0x4E);
  TNML_DPPMAX(0x141);
  TNML_DPPMAX(0x140);
#undef TNML_DPPMAX
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// value of a double held by lane `src` (wave-uniform), by v_readlane
__device__ inline double wave_read_f64(double v, int src) {
  const int s_ = __builtin_amdgcn_readfirstlane(src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), s_), hi = __builtin_amdgcn_readlane(__double2hiint(v), s_);
  return __hiloint2double(hi, lo);
}

}  // namespace tnml
