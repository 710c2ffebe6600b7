// float64 MFMA GEMM on small LDS-resident operands, and the slice-wise pre-computation of the merged tensor and
// of the L2 term built on it.  Shared by the narrow step kernel (kernels_narrow.hip) and by the extra workgroups
// the wide step kernel carries for that purpose (kernels_wide.hip).
#pragma once
#include "tnml_internal.h"

namespace tnml {

// ------------------------------------------------------------------------------------------
// C[M x N] = A[M x K] . B[K x N] on the matrix cores, float64 (v_mfma_f64_16x16x4_f64), for operands
// that live in LDS.  The 16x16 output tiles are dealt round-robin to the waves of the workgroup;
// loadA(i, k) / loadB(k, j) / store(i, j, value) are inlined index maps, called with in-range
// indices only (out-of-range rows, columns and k are fed as zeros).
// Lane maps (cdna_hip_programming.md section 3): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], C/D col = lane & 15, row = (lane >> 4) + 4 * reg.
// ------------------------------------------------------------------------------------------
typedef double dvec4 __attribute__((ext_vector_type(4)));

template <class FA, class FB, class FS>
__device__ inline void small_gemm_f64(int nbatch, int M, int N, int K, FA loadA, FB loadB, FS store) {
  // C_b[M x N] = A_b[M x K] . B_b[K x N] for b < nbatch; loadA(b, i, k), loadB(b, k, j),
  // store(b, i, j, value) must be LINEAR index maps (no run-time divisions: they are evaluated per
  // element); a composite row index such as (site index, label) is expressed through the batch.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int tn = (N + 15) >> 4, tm = (M + 15) >> 4, per = tm * tn, ntiles = nbatch * per;
  const int r = lane & 15, q = lane >> 4;
  for (int t = wave; t < ntiles; t += nw) {
    const int bt = t / per, tt = t - bt * per;
    const int ti = tt / tn;
    const int i0 = ti << 4, j0 = (tt - ti * tn) << 4;
    const bool va = i0 + r < M, vb = j0 + r < N;
    const int ia = va ? i0 + r : M - 1, jb = vb ? j0 + r : N - 1;
    // four k-steps of operands are fetched before the first MFMA of the group, two accumulators
    // break the MFMA -> MFMA dependency
    dvec4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 16) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k0 + 4 * u + q;
        const bool vk = kk < K;
        const int kc = vk ? kk : K - 1;
        a[u] = loadA(bt, ia, kc);
        b[u] = loadB(bt, kc, jb);
        a[u] = (va && vk) ? a[u] : 0.0;
        b[u] = (vb && vk) ? b[u] : 0.0;
      }
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
      if (k0 + 8 < K) {                                    // wave-uniform
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
      }
    }
    const dvec4 acc = acc0 + acc1;
    const int j = j0 + r;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = i0 + q + 4 * reg;
      if (i < M && j < N) store(bt, i, j, acc[reg]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same product for operands addressed by STRIDES (element (i, k) of A at A[b * a_bs + i * a_rs + k * a_ks], element
// (k, j) of B at B[b * b_bs + k * b_ks + j * b_cs]; float or double, converted on load).  Against the lambda form: every
// lane computes its two base addresses once per tile and then only adds the k strides, the operands of up to eight
// k-steps are in flight before the first MFMA, out-of-range rows / columns are clamped (their results are never stored)
// and only the last k-step is masked.  This is what the step kernel's critical path uses: its ten-odd products per sweep
// step are each a few hundred cycles of MFMA time, so address arithmetic and exposed LDS latency decide their cost.
// ------------------------------------------------------------------------------------------
template <class TA, class TB, class FS>
__device__ inline void mm_lds(int nbatch, int M, int N, int K, const TA *A, int a_bs, int a_rs, int a_ks, const TB *B, int b_bs,
                              int b_ks, int b_cs, FS store, bool upper_only = false) {
  // upper_only: M == N and only tiles with ti <= tj are computed (symmetric products: the caller mirrors)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int tn = (N + 15) >> 4, tm = (M + 15) >> 4;
  const int r = lane & 15, q = lane >> 4;
  const int nk = (K + 3) >> 2;                         // k-steps of 4
  const bool ktail = (K & 3) != 0;                     // the last step reaches past K
  const int sa = 4 * a_ks, sb = 4 * b_ks;
  // tiles are dealt to the waves in the order they are met; three nested counters instead of t / per, t % tn: a wave-uniform
  // integer division is ~40 scalar instructions and a per-lane one 134 cycles (tools/ubench/prims.hip), which at one or two
  // dozen tiles per product used to cost more than the MFMAs
  int slot = 0;
  for (int bt = 0; bt < nbatch; ++bt)
   for (int ti = 0; ti < tm; ++ti)
    for (int tj = upper_only ? ti : 0; tj < tn; ++tj) {
    const bool mine = slot == wave;
    slot = slot + 1 == nw ? 0 : slot + 1;
    if (!mine) continue;
    const int i0 = ti << 4, j0 = tj << 4;
    const int ia = min(i0 + r, M - 1), jb = min(j0 + r, N - 1);
    // the k index of the last step is clamped into range and its operand zeroed (ktail only)
    const int nfull = ktail ? nk - 1 : nk;
    const TA *pa = A + bt * a_bs + ia * a_rs + q * a_ks;
    const TB *pb = B + bt * b_bs + q * b_ks + jb * b_cs;
    dvec4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    int ks = 0;
    for (; ks + 4 <= nfull; ks += 4) {                  // four k-steps: all eight operands in flight before the first MFMA
      const double a0 = (double)pa[0], a1 = (double)pa[sa], a2 = (double)pa[2 * sa], a3 = (double)pa[3 * sa];
      const double b0 = (double)pb[0], b1 = (double)pb[sb], b2 = (double)pb[2 * sb], b3 = (double)pb[3 * sb];
      pa += 4 * sa; pb += 4 * sb;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc1, 0, 0, 0);
    }
    for (; ks < nfull; ++ks) {
      const double a0 = (double)pa[0], b0 = (double)pb[0];
      pa += sa; pb += sb;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
    }
    if (ktail) {
      const int kk = 4 * nfull + q, back = kk < K ? 0 : kk - (K - 1);
      double a0 = (double)pa[-back * a_ks], b0 = (double)pb[-back * b_ks];
      if (back) { a0 = 0.0; b0 = 0.0; }
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc1, 0, 0, 0);
    }
    const dvec4 acc = acc0 + acc1;
    const int j = j0 + r;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = i0 + q + 4 * reg;
      if (i < M && j < N) store(bt, i, j, acc[reg]);
    }
  }
}

// float32 form of mm_lds (v_mfma_f32_16x16x4_f32: half the cycles of the float64 instruction, operands used as stored) for
// products whose inputs are float32 data anyway and whose result is rounded to float32 precision downstream -- the
// contraction of the reduced pre-gradient with the previous step's core (the classic path accumulates the same sums in
// float32 inside its batch kernel).  C/D lane map of this instruction: col = lane & 15, row = 4 (lane >> 4) + reg.
typedef float fvec4_t __attribute__((ext_vector_type(4)));
template <class FS>
__device__ inline void mm_lds_f32(int M, int N, int K, const float *A, int a_rs, int a_ks, const float *B, int b_ks, int b_cs, FS store) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int tn = (N + 15) >> 4, tm = (M + 15) >> 4;
  const int r = lane & 15, q = lane >> 4;
  const int nk = (K + 3) >> 2;
  const bool ktail = (K & 3) != 0;
  const int nfull = ktail ? nk - 1 : nk;
  const int sa = 4 * a_ks, sb = 4 * b_ks;
  int slot = 0;
  for (int ti = 0; ti < tm; ++ti)
    for (int tj = 0; tj < tn; ++tj) {
      const bool mine = slot == wave;
      slot = slot + 1 == nw ? 0 : slot + 1;
      if (!mine) continue;
      const int i0 = ti << 4, j0 = tj << 4;
      const int ia = min(i0 + r, M - 1), jb = min(j0 + r, N - 1);
      const float *pa = A + ia * a_rs + q * a_ks;
      const float *pb = B + q * b_ks + jb * b_cs;
      fvec4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      int ks = 0;
      for (; ks + 4 <= nfull; ks += 4) {
        const float a0 = pa[0], a1 = pa[sa], a2 = pa[2 * sa], a3 = pa[3 * sa];
        const float b0 = pb[0], b1 = pb[sb], b2 = pb[2 * sb], b3 = pb[3 * sb];
        pa += 4 * sa; pb += 4 * sb;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
      }
      for (; ks < nfull; ++ks) {
        const float a0 = pa[0], b0 = pb[0];
        pa += sa; pb += sb;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      }
      if (ktail) {
        const int kk = 4 * nfull + q, back = kk < K ? 0 : kk - (K - 1);
        float a0 = pa[-back * a_ks], b0 = pb[-back * b_ks];
        if (back) { a0 = 0.f; b0 = 0.f; }
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc1, 0, 0, 0);
      }
      const fvec4_t acc = acc0 + acc1;
      const int j = j0 + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = i0 + 4 * q + reg;
        if (i < M && j < N) store(i, j, acc[reg]);
      }
    }
}

// ------------------------------------------------------------------------------------------
// Slice (dk, dk1) of the merged tensor B = A_k . A_{k+1} and of Ln.B.Rn: both factorise over the two feature
// indices, so the D*D slices are independent; one workgroup each.  Results go to HBM in the sweep-relative layout
// [h][dk][dk1][g][l].
// ------------------------------------------------------------------------------------------
// coherent: the results are read by another workgroup of the SAME launch (agent-scope stores; see narrow_helper_block)
__device__ inline void prep_slice_block(const PrepParams &p, int slice, unsigned char *smem_raw, bool coherent = false) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
  const int D = kD, h = p.h, g = p.g, s = p.s, L = p.L;
  const int dk = slice / D, dk1 = slice % D;
  const int GL = g * L;
  double *dNh = (double *)smem_raw;                      // [h][h]
  double *dNg = dNh + (((size_t)h * h + 1) & ~(size_t)1);   // [g][g]
  double *dT = dNg + (((size_t)g * g + 1) & ~(size_t)1);    // [h][g][L]
  float *sLab = (float *)(dT + (size_t)h * GL);          // [h][s][L]   (this dk)
  float *sPl = sLab + (size_t)h * s * L;                 // [s][g]      (this dk1)
  float *fBs = sPl + (size_t)s * g;                      // [h][g][L]
  // rows over waves, the contiguous index over lanes: no integer division (134 cycles each, tools/ubench/prims.hip)
  for (int h_ = wave; h_ < h; h_ += nw)
    for (int l = 0; l < L; ++l)
      for (int s_ = lane; s_ < s; s_ += 64)
        sLab[(h_ * s + s_) * L + l] = p.lab.base[h_ * p.lab.s_in + dk * p.lab.s_d + s_ * p.lab.s_out + l];
  for (int s_ = wave; s_ < s; s_ += nw)
    for (int g_ = lane; g_ < g; g_ += 64)
      sPl[s_ * g + g_] = p.pl.base[s_ * p.pl.s_in + dk1 * p.pl.s_d + g_ * p.pl.s_out];
  if (p.l2_flag) {
    for (int e = tid; e < h * h; e += NT) dNh[e] = p.Nh ? p.Nh[e] : 1.0;
    for (int e = tid; e < g * g; e += NT) dNg[e] = p.Ng ? p.Ng[e] : 1.0;
  }
  __syncthreads();
  mm_lds(L, h, g, s, sLab, 1, s * L, L, sPl, 0, g, 1,
         [&](int l, int i, int j, double v) {
           const float fv = (float)v;
           fBs[(i * g + j) * L + l] = fv;
           float *dst = p.prepB + (size_t)(((i * D + dk) * D + dk1) * g + j) * L + l;
           if (coherent) __hip_atomic_store(dst, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = fv;
         });
  if (p.l2_flag) {
    __syncthreads();
    mm_lds(1, h, GL, h, dNh, 0, 1, h, fBs, 0, GL, 1, [&](int, int i, int j, double v) { dT[i * GL + j] = v; });
    __syncthreads();
    mm_lds(L, h, g, g, dT, 1, GL, L, dNg, 0, g, 1,
           [&](int l, int i, int j, double v) {
             double *dst = p.prepG + (size_t)(((i * D + dk) * D + dk1) * g + j) * L + l;
             if (coherent) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = v;
           });
  }
}

// LDS bytes prep_slice_block needs
inline size_t prep_slice_lds_bytes(int h, int g, int s, int L) {
  return ((((size_t)h * h + 1) & ~(size_t)1) + (((size_t)g * g + 1) & ~(size_t)1) + (size_t)h * g * L) * sizeof(double) +
         ((size_t)h * s * L + (size_t)s * g + (size_t)h * g * L) * sizeof(float) + 16;
}

}  // namespace tnml
