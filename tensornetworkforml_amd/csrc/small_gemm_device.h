// float64 MFMA GEMM on small LDS-resident operands, and the slice-wise pre-computation of the merged tensor and
// of the L2 term built on it.  Shared by the narrow step kernel (kernels_narrow.hip) and by the extra workgroups
// the wide step kernel carries for that purpose (kernels_wide.hip).
#pragma once
#include "tnml_internal.h"
#include "jacobi_device.h"

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
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform for the compiler: scalar tile bookkeeping
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
__device__ inline int mm_lds(int nbatch, int M, int N, int K, const TA *A, int a_bs, int a_rs, int a_ks, const TB *B, int b_bs,
                             int b_ks, int b_cs, FS store, bool upper_only = false, int slot0 = 0) {
  // upper_only: M == N and only tiles with ti <= tj are computed (symmetric products: the caller mirrors)
  // slot0 / return value: independent products issued back to back (no barrier between them) continue the round-robin deal
  // of tiles to waves where the previous one stopped
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform for the compiler: scalar tile bookkeeping
  const int tn = (N + 15) >> 4, tm = (M + 15) >> 4;
  const int r = lane & 15, q = lane >> 4;
  const int nk = (K + 3) >> 2;                         // k-steps of 4
  const bool ktail = (K & 3) != 0;                     // the last step reaches past K
  const int sa = 4 * a_ks, sb = 4 * b_ks;
  // Tiles are dealt round-robin to the waves, starting at wave slot0.  A wave goes straight to ITS tiles (t = first, first + nw,
  // ...) and decodes (batch, row tile, column tile) with scalar compare-and-subtract loops: walking all slots cost every wave
  // -- the idle ones too -- a taken branch or two per slot (a 40-tile product: ~1.2 k cycles of bookkeeping per wave), and a
  // scalar integer division is ~40 instructions.
  const int per = upper_only ? (tm * (tm + 1)) >> 1 : tm * tn, ntiles = nbatch * per;
  int first = wave - slot0;
  if (first < 0) first += nw;
  for (int t = first; t < ntiles; t += nw) {
    int bt = 0, rem = t;
    while (rem >= per) { rem -= per; ++bt; }
    int ti = 0, tj;
    if (upper_only) { int rowlen = tn; while (rem >= rowlen) { rem -= rowlen; ++ti; --rowlen; } tj = ti + rem; }
    else { while (rem >= tn) { rem -= tn; ++ti; } tj = rem; }
    const int i0 = ti << 4, j0 = tj << 4;
    const int ia = min(i0 + r, M - 1), jb = min(j0 + r, N - 1);
    // the k index of the last step is clamped into range and its operand zeroed (ktail only)
    const int nfull = ktail ? nk - 1 : nk;
    // (24-bit multiplies: LDS element offsets are far below 2^23, and v_mul_lo_u32 issues at quarter rate)
    const TA *pa = A + (bt * a_bs + __mul24(ia, a_rs) + __mul24(q, a_ks));
    const TB *pb = B + (bt * b_bs + __mul24(q, b_ks) + __mul24(jb, b_cs));
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
  int next = slot0 + ntiles;                            // where the next independent product continues the deal
  while (next >= nw) next -= nw;
  return next;
}

// float32 form of mm_lds (v_mfma_f32_16x16x4_f32: half the cycles of the float64 instruction, operands used as stored) for
// products whose inputs are float32 data anyway and whose result is rounded to float32 precision downstream -- the
// contraction of the reduced pre-gradient with the previous step's core (the classic path accumulates the same sums in
// float32 inside its batch kernel).  C/D lane map of this instruction: col = lane & 15, row = 4 (lane >> 4) + reg.
typedef float fvec4_t __attribute__((ext_vector_type(4)));
template <class FS>
__device__ inline void mm_lds_f32(int M, int N, int K, const float *A, int a_rs, int a_ks, const float *B, int b_ks, int b_cs, FS store) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tn = (N + 15) >> 4, tm = (M + 15) >> 4;
  const int r = lane & 15, q = lane >> 4;
  const int nk = (K + 3) >> 2;
  const bool ktail = (K & 3) != 0;
  const int nfull = ktail ? nk - 1 : nk;
  const int sa = 4 * a_ks, sb = 4 * b_ks;
  for (int t = wave; t < tm * tn; t += nw) {             // straight to this wave's tiles (see mm_lds)
      int ti = 0, tj = t;
      while (tj >= tn) { tj -= tn; ++ti; }
      const int i0 = ti << 4, j0 = tj << 4;
      const int ia = min(i0 + r, M - 1), jb = min(j0 + r, N - 1);
      const float *pa = A + (__mul24(ia, a_rs) + __mul24(q, a_ks));
      const float *pb = B + (__mul24(q, b_ks) + __mul24(jb, b_cs));
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
__device__ inline void prep_slice_block(const PrepParams &p, int slice, unsigned char *smem_raw, bool coherent = false, double *hstamp = nullptr) {
  // B_dd' = lab_d . pl_d'   and   (Ln.B.Rn)_dd' = (Nh^T lab_d) . (pl_d' Ng): the three first-level products are independent
  // and run back to back without a barrier, the second level is one more product -- a dependent chain of two instead of
  // three, behind ONE round trip to memory for all four operands.
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = kD, h = p.h, g = p.g, s = p.s, L = p.L;
  // rows [i_lo, i_lo + nr) of slice (dk, dk1): a slice may be cut into row parts over several workgroups (a 20-row slice is
  // 16 + 4 rows of MFMA tiles anyway), which only repeat the small product pl . Ng
  const int nparts = p.nparts > 1 ? p.nparts : 1, part = slice / (D * D), sl = slice - part * (D * D);
  const int hh = (h + nparts - 1) / nparts, i_lo = part * hh, nr = min(h, i_lo + hh) - i_lo;
  if (nr <= 0) return;
  const int dk = sl / D, dk1 = sl % D;
  const int sL = s * L, nlab = h * sL, npl = s * g, nhh = h * h, ngg = g * g;
  double *dNh = (double *)smem_raw;                      // [h][h]
  double *dNg = dNh + (((size_t)nhh + 1) & ~(size_t)1);  // [g][g]
  double *dX = dNg + (((size_t)ngg + 1) & ~(size_t)1);   // [h][s][L]   Nh^T . lab
  double *dY = dX + (((size_t)nlab + 1) & ~(size_t)1);   // [s][g]      pl . Ng
  float *sLab = (float *)(dY + (((size_t)npl + 1) & ~(size_t)1));   // [h][s][L]   (this dk)
  float *sPl = sLab + (size_t)nlab;                      // [s][g]      (this dk1)
  // exact quotients of small integers by float reciprocal (a hardware integer division costs 134 cycles per lane)
  const float inv_sL = 1.0f / (float)sL, inv_L = 1.0f / (float)L, inv_g = 1.0f / (float)g;
  auto lab_at = [&](int e) -> float {
    const int h_ = (int)(((float)e + 0.5f) * inv_sL), x = e - h_ * sL;
    const int s_ = (int)(((float)x + 0.5f) * inv_L), l = x - s_ * L;
    return p.lab.base[h_ * p.lab.s_in + dk * p.lab.s_d + s_ * p.lab.s_out + l];
  };
  auto pl_at = [&](int e) -> float {
    const int s_ = (int)(((float)e + 0.5f) * inv_g), g_ = e - s_ * g;
    return p.pl.base[s_ * p.pl.s_in + dk1 * p.pl.s_d + g_ * p.pl.s_out];
  };
  {
    float rl[2], rp[2];
    double rh[2], rg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                        // every load of the first 2 * NT elements of each operand in flight together
      const int e = tid + u * NT;
      rl[u] = e < nlab ? lab_at(e) : 0.f;
      rp[u] = e < npl ? pl_at(e) : 0.f;
      rh[u] = (p.l2_flag && e < nhh) ? (p.Nh ? p.Nh[e] : 1.0) : 0.0;
      rg[u] = (p.l2_flag && e < ngg) ? (p.Ng ? p.Ng[e] : 1.0) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NT;
      if (e < nlab) sLab[e] = rl[u];
      if (e < npl) sPl[e] = rp[u];
      if (p.l2_flag && e < nhh) dNh[e] = rh[u];
      if (p.l2_flag && e < ngg) dNg[e] = rg[u];
    }
    for (int e = tid + 2 * NT; e < nlab; e += NT) sLab[e] = lab_at(e);
    for (int e = tid + 2 * NT; e < npl; e += NT) sPl[e] = pl_at(e);
    if (p.l2_flag) {
      for (int e = tid + 2 * NT; e < nhh; e += NT) dNh[e] = p.Nh ? p.Nh[e] : 1.0;
      for (int e = tid + 2 * NT; e < ngg; e += NT) dNg[e] = p.Ng ? p.Ng[e] : 1.0;
    }
  }
  __syncthreads();
  // Results leave through LDS: a row (i) of a slice is g * L contiguous elements of the [h][dk][dk1][g][l] layout, so the
  // hand-off to another workgroup of this launch goes out as 16-byte agent-scope stores instead of one fabric write per element.
  const int gL = g * L;
  float *oB = sLab + (((size_t)nlab + npl + 3) & ~(size_t)3);   // [h][g][L], 16-byte aligned
  const bool vecB = coherent && (gL & 3) == 0, vecG = coherent && (gL & 1) == 0;
  int slot = mm_lds(L, nr, g, s, sLab + i_lo * sL, 1, sL, L, sPl, 0, g, 1,
                    [&](int l, int i, int j, double v) {
                      if (vecB) oB[(i * g + j) * L + l] = (float)v;
                      else {
                        float *dst = p.prepB + (size_t)((((i_lo + i) * D + dk) * D + dk1) * g + j) * L + l;
                        if (coherent) __hip_atomic_store(dst, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = (float)v;
                      }
                    });
  if (p.l2_flag) {
    // X[i][x] = sum_a Nh[a][i] lab[a][x]  (x = (s_, l));   Y[s_][j] = sum_c pl[s_][c] Ng[c][j]
    slot = mm_lds(1, nr, sL, h, dNh + i_lo, 0, 1, h, sLab, 0, sL, 1, [&](int, int i, int x, double v) { dX[i * sL + x] = v; }, false, slot);
    mm_lds(1, s, g, g, sPl, 0, g, 1, dNg, 0, g, 1, [&](int, int s_, int j, double v) { dY[s_ * g + j] = v; }, false, slot);
  }
  __syncthreads();
  if (vecB) {                                            // B rows: (gL / 4) 16-byte pieces each
    const __amdgpu_buffer_rsrc_t rB = sc1_rsrc(p.prepB);
    const int pieces = gL >> 2, total = nr * pieces;
    const float inv_p = 1.0f / (float)pieces;
    for (int e = tid; e < total; e += NT) {
      const int i = (int)(((float)e + 0.5f) * inv_p), c4 = e - i * pieces;
      const tn_uvec4 v = *reinterpret_cast<const tn_uvec4 *>(oB + (size_t)i * gL + 4 * c4);
      st_sc1_b128(rB, (unsigned)(((((i_lo + i) * D + dk) * D + dk1) * gL + 4 * c4) * sizeof(float)), v);
    }
  }
  if (p.l2_flag) {
    double *oGs = (double *)(oB + (((size_t)h * gL + 3) & ~(size_t)3));     // [h][g][L] staging of the second-level result
    mm_lds(L, nr, g, s, dX, 1, sL, L, dY, 0, g, 1,
           [&](int l, int i, int j, double v) {
             if (vecG) oGs[(i * g + j) * L + l] = v;
             else {
               double *dst = p.prepG + (size_t)((((i_lo + i) * D + dk) * D + dk1) * g + j) * L + l;
               if (coherent) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = v;
             }
           });
    if (vecG) {
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rG = sc1_rsrc(p.prepG);
      const int pieces = gL >> 1, total = nr * pieces;
      const float inv_p = 1.0f / (float)pieces;
      for (int e = tid; e < total; e += NT) {
        const int i = (int)(((float)e + 0.5f) * inv_p), c2 = e - i * pieces;
        const tn_uvec4 v = *reinterpret_cast<const tn_uvec4 *>(oGs + (size_t)i * gL + 2 * c2);
        st_sc1_b128(rG, (unsigned)(((((i_lo + i) * D + dk) * D + dk1) * gL + 2 * c2) * sizeof(double)), v);
      }
    }
  }
}

// LDS bytes prep_slice_block needs
inline size_t prep_slice_lds_bytes(int h, int g, int s, int L) {
  auto ev = [](size_t x) { return (x + 1) & ~(size_t)1; };
  return (ev((size_t)h * h) + ev((size_t)g * g) + ev((size_t)h * s * L) + ev((size_t)s * g) + (size_t)h * g * L) * sizeof(double) +
         ((size_t)h * s * L + (size_t)s * g + (((size_t)h * g * L + 3) & ~(size_t)3)) * sizeof(float) + 48;
}

}  // namespace tnml
