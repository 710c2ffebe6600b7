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
// Slice (dk, dk1) of the merged tensor B = A_k . A_{k+1} and of Ln.B.Rn: both factorise over the two feature
// indices, so the D*D slices are independent; one workgroup each.  Results go to HBM in the sweep-relative layout
// [h][dk][dk1][g][l].
// ------------------------------------------------------------------------------------------
__device__ inline void prep_slice_block(const PrepParams &p, int slice, unsigned char *smem_raw) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = kD, h = p.h, g = p.g, s = p.s, L = p.L;
  const int dk = slice / D, dk1 = slice % D;
  const int GL = g * L;
  double *dNh = (double *)smem_raw;                      // [h][h]
  double *dNg = dNh + (((size_t)h * h + 1) & ~(size_t)1);   // [g][g]
  double *dT = dNg + (((size_t)g * g + 1) & ~(size_t)1);    // [h][g][L]
  float *sLab = (float *)(dT + (size_t)h * GL);          // [h][s][L]   (this dk)
  float *sPl = sLab + (size_t)h * s * L;                 // [s][g]      (this dk1)
  float *fBs = sPl + (size_t)s * g;                      // [h][g][L]
  for (int e = tid; e < h * s * L; e += NT) {
    const int l = e % L, q = e / L;
    const int s_ = q % s, h_ = q / s;
    sLab[e] = p.lab.base[h_ * p.lab.s_in + dk * p.lab.s_d + s_ * p.lab.s_out + l];
  }
  for (int e = tid; e < s * g; e += NT) {
    const int g_ = e % g, s_ = e / g;
    sPl[e] = p.pl.base[s_ * p.pl.s_in + dk1 * p.pl.s_d + g_ * p.pl.s_out];
  }
  if (p.l2_flag) {
    for (int e = tid; e < h * h; e += NT) dNh[e] = p.Nh ? p.Nh[e] : 1.0;
    for (int e = tid; e < g * g; e += NT) dNg[e] = p.Ng ? p.Ng[e] : 1.0;
  }
  __syncthreads();
  small_gemm_f64(L, h, g, s,
                 [&](int l, int i, int kk) { return (double)sLab[(i * s + kk) * L + l]; },
                 [&](int l, int kk, int j) { return (double)sPl[kk * g + j]; },
                 [&](int l, int i, int j, double v) {
                   const float fv = (float)v;
                   fBs[(i * g + j) * L + l] = fv;
                   p.prepB[(size_t)(((i * D + dk) * D + dk1) * g + j) * L + l] = fv;
                 });
  __syncthreads();
  if (p.l2_flag) {
    small_gemm_f64(1, h, GL, h,
                   [&](int, int i, int kk) { return dNh[kk * h + i]; },
                   [&](int, int kk, int j) { return (double)fBs[kk * GL + j]; },
                   [&](int, int i, int j, double v) { dT[i * GL + j] = v; });
    __syncthreads();
    small_gemm_f64(L, h, g, g,
                   [&](int l, int i, int kk) { return dT[(i * g + kk) * L + l]; },
                   [&](int l, int kk, int j) { return dNg[kk * g + j]; },
                   [&](int l, int i, int j, double v) { p.prepG[(size_t)(((i * D + dk) * D + dk1) * g + j) * L + l] = v; });
  }
}

// LDS bytes prep_slice_block needs
inline size_t prep_slice_lds_bytes(int h, int g, int s, int L) {
  return ((((size_t)h * h + 1) & ~(size_t)1) + (((size_t)g * g + 1) & ~(size_t)1) + (size_t)h * g * L) * sizeof(double) +
         ((size_t)h * s * L + (size_t)s * g + (size_t)h * g * L) * sizeof(float) + 16;
}

}  // namespace tnml
