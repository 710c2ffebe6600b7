// Large-tensor path of the sweep step's batch-independent part: the same arithmetic as
// narrow_step_kernel (kernels_narrow.hip), for merged tensors that do not fit one workgroup's LDS or
// whose short side exceeds 64 (C5 of BASELINE.json: bond 50, 10 labels -> a 100 x 1000 merged tensor,
// 400 KB in float32).  The merged tensor, the weight-decay term and the update live in HBM and are
// produced by many-workgroup kernels; only the Jacobi eigenvalue iteration on the n x n Gram matrix
// (n <= 128) runs in a single workgroup, with the symmetric matrix packed as upper-triangular 2x2
// blocks (two LDS buffers of np(np+1)/2 blocks: 82 KB at n = 100).  The eigenvectors are never formed
// in that kernel: it logs every rotation (c, s), and a replay kernel applies the log to the n unit
// vectors AND to the `len` columns of the matricised tensor, one wavefront per vector (pair k on lane k,
// the tournament move as DPP wave shifts), which yields V and W^T V -- the two SVD factors up to the
// sigma^(+-1/2) scaling -- without any further GEMM.
//
// Reference lines: update_B Network_class.py:577-763, compute_L2_reg :966-1179, tensor_svd :839-962.
#include <algorithm>
#include <cstdio>

#include <type_traits>
#include "tnml_internal.h"
#include "jacobi_device.h"

namespace tnml {

namespace {

constexpr int kBT = 256;   // threads of the element-wise kernels

// ---- merged tensor B[h,dk,dk1,g,l] = sum_s lab(h,dk,s,l) pl(s,dk1,g)  (Network_class.py:484) ----------------
__device__ inline void big_merge_body(const NarrowParams &p, float *__restrict__ Bf, int blk, int nblk) {
  const int D = kD, g = p.g, s = p.s, L = p.L;
  for (int e = blk * kBT + threadIdx.x; e < p.bsize; e += nblk * kBT) {
    const int l = e % L;
    int q = e / L;
    const int g_ = q % g; q /= g;
    const int dk1 = q % D; q /= D;
    const int dk = q % D, h_ = q / D;
    const float *la = p.lab.base + h_ * p.lab.s_in + dk * p.lab.s_d + l;
    const float *pl = p.pl.base + dk1 * p.pl.s_d + g_ * p.pl.s_out;
    // (the launch has ~1.5 waves per SIMD: a dependent load + FMA per iteration made this kernel one memory latency per k -- 16 us for
    //  s = 50.  Eight loads in flight and four accumulators: the sum is re-associated, in float64)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 8 <= s; k += 8) {
      float lv[8], pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { lv[u] = la[(k + u) * p.lab.s_out]; pv[u] = pl[(k + u) * p.pl.s_in]; }
      a0 = fma((double)lv[0], (double)pv[0], a0); a1 = fma((double)lv[1], (double)pv[1], a1);
      a2 = fma((double)lv[2], (double)pv[2], a2); a3 = fma((double)lv[3], (double)pv[3], a3);
      a0 = fma((double)lv[4], (double)pv[4], a0); a1 = fma((double)lv[5], (double)pv[5], a1);
      a2 = fma((double)lv[6], (double)pv[6], a2); a3 = fma((double)lv[7], (double)pv[7], a3);
    }
    for (; k < s; ++k) a0 = fma((double)la[k * p.lab.s_out], (double)pl[k * p.pl.s_in], a0);
    Bf[e] = (float)((a0 + a1) + (a2 + a3));
  }
}
__global__ __launch_bounds__(kBT) void big_merge_kernel(NarrowParams p, float *__restrict__ Bf) {
  big_merge_body(p, Bf, blockIdx.x, gridDim.x);
}

// ---- T[e_, rest] = sum_a Nh[a, e_] B[a, rest]   (first half of Ln.B.Rn) --------------------------------------
__global__ __launch_bounds__(kBT) void big_l2_T_kernel(NarrowParams p, const float *__restrict__ Bf, double *__restrict__ T) {
  const int h = p.h, RW = p.bsize / p.h;
  for (int e = blockIdx.x * kBT + threadIdx.x; e < p.bsize; e += gridDim.x * kBT) {
    const int rest = e % RW, e_ = e / RW;
    double acc = 0.0;
    if (p.Nh) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int a = 0;
      for (; a + 8 <= h; a += 8) {
        double nv[8]; float bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { nv[u] = p.Nh[(a + u) * h + e_]; bv[u] = Bf[(size_t)(a + u) * RW + rest]; }
        a0 = fma(nv[0], (double)bv[0], a0); a1 = fma(nv[1], (double)bv[1], a1); a2 = fma(nv[2], (double)bv[2], a2); a3 = fma(nv[3], (double)bv[3], a3);
        a0 = fma(nv[4], (double)bv[4], a0); a1 = fma(nv[5], (double)bv[5], a1); a2 = fma(nv[6], (double)bv[6], a2); a3 = fma(nv[7], (double)bv[7], a3);
      }
      for (; a < h; ++a) a0 = fma(p.Nh[a * h + e_], (double)Bf[(size_t)a * RW + rest], a0);
      acc = (a0 + a1) + (a2 + a3);
    } else acc = (double)Bf[e];
    T[e] = acc;
  }
}

// ---- the weight-decay term, dv = raw - wdterm and the three block-partial sums ----------------------------------------
//   ws layout (doubles): [0,Bs) B   [Bs,2Bs) dB_raw   [2Bs,3Bs) dv   [3Bs,4Bs) weight-decay term
//   part: [nb][3] block partials; big_update_kernel sums them with big_sum_parts: every block and every rank in the same fixed order.
//   (A last-arriver block that left the step length behind was tried: one agent-scope atomic per block on one address costs
//   ~20 ns each, serialised -- 34 us at the 1563 blocks of a C5 step.)
__device__ inline void big_wd_tail(double *__restrict__ part, double sumB, double sumD, double l2) {
  __shared__ double red[3][kBT / 64];
  for (int off = 32; off > 0; off >>= 1) { sumB += __shfl_xor(sumB, off); sumD += __shfl_xor(sumD, off); l2 += __shfl_xor(l2, off); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sumB; red[1][threadIdx.x >> 6] = sumD; red[2][threadIdx.x >> 6] = l2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < kBT / 64; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
    part[3 * blockIdx.x] = a; part[3 * blockIdx.x + 1] = b; part[3 * blockIdx.x + 2] = c;
  }
}
// sums of the partials, by all 256 threads of a workgroup (kBT): threads stride the partials, a fixed xor-shuffle tree per wave,
// the four waves in order.  out3 (LDS) = {sum|B|, sum|dv|, L2 sum}; ends with a barrier.
__device__ inline void big_sum_parts(const double *__restrict__ part, int nparts, double (*sRed)[kBT / 64], double *out3) {
  double a_ = 0.0, b_ = 0.0, c_ = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBT) { a_ += part[3 * i]; b_ += part[3 * i + 1]; c_ += part[3 * i + 2]; }
  for (int off = 32; off > 0; off >>= 1) { a_ += __shfl_xor(a_, off); b_ += __shfl_xor(b_, off); c_ += __shfl_xor(c_, off); }
  if ((threadIdx.x & 63) == 0) { sRed[0][threadIdx.x >> 6] = a_; sRed[1][threadIdx.x >> 6] = b_; sRed[2][threadIdx.x >> 6] = c_; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < kBT / 64; ++w) { a += sRed[0][w]; b += sRed[1][w]; c += sRed[2][w]; }
    out3[0] = a; out3[1] = b; out3[2] = c;
  }
  __syncthreads();
}

// (a) from a merged tensor in memory and T = Nh^T . B (a given merged tensor; steps whose preparation ran ahead on the side stream)
__global__ __launch_bounds__(kBT) void big_wd_kernel(NarrowParams p, const float *__restrict__ Bf, const double *__restrict__ T,
                                                    double *__restrict__ ws, double *__restrict__ part) {
  const int g = p.g, L = p.L, Bs = p.bsize;
  double sumB = 0.0, sumD = 0.0, l2 = 0.0;
  for (int e = blockIdx.x * kBT + threadIdx.x; e < Bs; e += gridDim.x * kBT) {
    const double bv = (double)Bf[e];
    const double raw = (double)p.red[e];
    double wdterm;
    if (p.l2_flag) {
      const int l = e % L, q = e / L;
      const int f_ = q % g, i = q / g;
      double gv = 0.0;
      if (p.Ng) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int c = 0;
        for (; c + 8 <= g; c += 8) {
          double tv[8], nv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { tv[u] = T[((size_t)i * g + c + u) * L + l]; nv[u] = p.Ng[(c + u) * g + f_]; }
          a0 = fma(tv[0], nv[0], a0); a1 = fma(tv[1], nv[1], a1); a2 = fma(tv[2], nv[2], a2); a3 = fma(tv[3], nv[3], a3);
          a0 = fma(tv[4], nv[4], a0); a1 = fma(tv[5], nv[5], a1); a2 = fma(tv[6], nv[6], a2); a3 = fma(tv[7], nv[7], a3);
        }
        for (; c < g; ++c) a0 = fma(T[((size_t)i * g + c) * L + l], p.Ng[c * g + f_], a0);
        gv = (a0 + a1) + (a2 + a3);
      } else gv = T[e];
      l2 += bv * gv;
      wdterm = 2.0 * (double)p.wd * gv;
    } else {
      wdterm = (double)p.wd * bv;
    }
    const double dv = raw - wdterm;
    ws[e] = bv; ws[(size_t)Bs + e] = raw; ws[2 * (size_t)Bs + e] = dv; ws[3 * (size_t)Bs + e] = wdterm;
    sumB += fabs(bv);
    sumD += fabs(dv);
  }
  big_wd_tail(part, sumB, sumD, l2);
}

// (b) from the two cores: Ln.B.Rn = (Nh^T . lab) . (pl . Ng) contracted over the shared bond like the merged tensor itself -- the
//     merged tensor B and the L2 gradient in ONE loop over s, no T = Nh^T . B in between (a launch of 100 k x 50-term sums less).
//       NL[h', dk, s, l] = sum_a Nh[a, h'] lab(a, dk, s, l)       PR[s, dk1, f] = sum_c pl(s, dk1, c) Ng[c, f]
//     are formed by big_front_kernel beside the contraction (float64: the norm environments range far beyond float32's exponents
//     along a 784-site chain -- stored as float32 they overflowed at full length).
//     On the float64 matrix cores (v_mfma_f64_16x16x4_f64), one 16 x 16 output tile per wave:
//       Bt[n][(m, l)] = sum_s pl(s, n) lab(m, s, l)        GVt[n][(m, l)] = sum_s PR[s, n] NL[m, s, l]       n = (dk1, g), m = (h, dk)
//     -- ONE plain product each, K = shared bond, with the label index riding in the column index; transposed so that a store
//     instruction's 16 columns x 4 rows land in contiguous runs of the merged tensor ((m N + n) L + l).  Lane-per-element forms
//     came first (one lane per element with both sums: 23 us at C5; the two sums on lanes of one wave: 38 us, divergence; two waves
//     per sum, two lanes per element, 72 VGPRs: 18 us): each re-reads every operand row through L1 for every element -- 120 MB of
//     L2 -> L1 traffic for a 10 MFLOP product.  Here each operand element is loaded once per tile (16-fold reuse in the matrix core).  Lane maps as in small_gemm_device.h:
//     A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], C col = lane & 15, row = (lane >> 4) + 4 reg.
typedef double big_dvec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBT) void big_merge_wd_mfma_kernel(NarrowParams p, float *__restrict__ Bf, const double *__restrict__ NL,
                                                               const double *__restrict__ PR, double *__restrict__ ws,
                                                               double *__restrict__ part) {
  const int D = kD, g = p.g, sb = p.s, L = p.L, Bs = p.bsize;
  const int N = D * g, MC = p.h * D * L;                 // rows n, columns c = m L + l
  const int tn = (N + 15) >> 4, tc = (MC + 15) >> 4, ntiles = tn * tc;
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = (sb + 3) >> 2;
  double sumB = 0.0, sumD = 0.0, l2 = 0.0;
  for (int t = blockIdx.x * (kBT / 64) + wave; t < ntiles; t += gridDim.x * (kBT / 64)) {
    const int ti = t / tc, tj = t - ti * tc;             // (wave-uniform)
    const int n_a = min(16 * ti + r, N - 1);             // this lane's operand row of the A side
    const int c_b = min(16 * tj + r, MC - 1);            // ... column of the B side
    const int dk1 = n_a / g, g_ = n_a - dk1 * g;
    const int m_b = c_b / L, l_b = c_b - m_b * L;
    const float *pa = p.pl.base + dk1 * p.pl.s_d + g_ * p.pl.s_out;                       // + s * pl.s_in
    const float *pb = p.lab.base + (m_b / D) * p.lab.s_in + (m_b % D) * p.lab.s_d + l_b;  // + s * lab.s_out
    const double *qa = PR + n_a;                                                          // + s * N
    const double *qb = NL + (size_t)m_b * sb * L + l_b;                                   // + s * L
    big_dvec4 b0 = {0.0, 0.0, 0.0, 0.0}, b1 = b0, g0 = b0, g1 = b0;
    for (int k0 = 0; k0 < nk; k0 += 4) {                 // four k-steps: their operands in flight before the first MFMA
      double av[4], bv[4], cv[4], dv_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = 4 * (k0 + u) + q, kc = min(kk, sb - 1);
        const bool ok = kk < sb;
        av[u] = ok ? (double)pa[kc * p.pl.s_in] : 0.0;
        bv[u] = (double)pb[kc * p.lab.s_out];
        if (p.l2_flag) { cv[u] = ok ? qa[(size_t)kc * N] : 0.0; dv_[u] = qb[(size_t)kc * L]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u < nk) {                               // wave-uniform
          if (u & 1) b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], b1, 0, 0, 0);
          else b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], b0, 0, 0, 0);
          if (p.l2_flag) {
            if (u & 1) g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[u], dv_[u], g1, 0, 0, 0);
            else g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[u], dv_[u], g0, 0, 0, 0);
          }
        }
      }
    }
    const big_dvec4 bacc = b0 + b1, gacc = g0 + g1;
    const int c = 16 * tj + r;
    if (c < MC) {
      const int m = c / L, l = c - m * L;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = 16 * ti + q + 4 * reg;
        if (n < N) {
          const int e = (m * N + n) * L + l;
          const float bf = (float)bacc[reg];              // the merged tensor is a float32 tensor (as on the classic path)
          Bf[e] = bf;
          const double bval = (double)bf;
          const double raw = (double)p.red[e];
          double wdterm;
          if (p.l2_flag) { l2 += bval * gacc[reg]; wdterm = 2.0 * (double)p.wd * gacc[reg]; }
          else wdterm = (double)p.wd * bval;
          const double dv = raw - wdterm;
          ws[e] = bval; ws[(size_t)Bs + e] = raw; ws[2 * (size_t)Bs + e] = dv; ws[3 * (size_t)Bs + e] = wdterm;
          sumB += fabs(bval);
          sumD += fabs(dv);
        }
      }
    }
  }
  big_wd_tail(part, sumB, sumD, l2);
}

// ---- clip + update (Network_class.py:755-761); every block sums the partials in the same fixed order ----------
__global__ __launch_bounds__(kBT) void big_update_kernel(NarrowParams p, double *__restrict__ ws, const double *__restrict__ part,
                                                        int nparts) {
  const int Bs = p.bsize;
  __shared__ double sSum[3];
  __shared__ double sRed[3][kBT / 64];
  big_sum_parts(part, nparts, sRed, sSum);
  const double sumB = sSum[0], sumD = sSum[1], l2 = sSum[2];
  double factor = (double)p.lr;
  if (sumD > sumB) factor = (double)p.lr * (sumB / sumD);
  for (int e = blockIdx.x * kBT + threadIdx.x; e < Bs; e += gridDim.x * kBT) {
    const float v = (float)(ws[e] + factor * ws[2 * (size_t)Bs + e]);
    p.Bnew[e] = v;
    ws[2 * (size_t)Bs + e] = (double)v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (!isfinite(sumD) || !isfinite(sumB)) atomicOr(p.status, 1);
    double *sc = ws + 4 * (size_t)Bs + kDbgSigma;
    sc[0] = (double)p.wd * l2; sc[1] = sumB; sc[2] = sumD;
    if (p.stop_after_update && p.metrics) {
      const double cnt = (double)p.red[Bs + 3];
      const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
      p.metrics[0] = (float)((double)p.red[Bs] * inv);
      p.metrics[1] = (float)((double)p.red[Bs + 1] * inv / (double)p.L);
    }
  }
}

// ---- Gram matrix of the short side, float64 accumulation of the float32 B_new -----------------------------------
//   W(i, x) = Bn[i * si + x * sx];  G[i][j] = sum_x W(i,x) W(j,x), 16x16 output tile per block, upper tiles only
//   (Applying the clipped step inside this launch was tried in round 3: it saves the update launch on this stream but the side stream,
//   which waits for B_new, then starts 8 us later -- and its chain is as long as the Jacobi kernel: removed.)
constexpr int kGramKS = 8;      // slices along the long index; the Jacobi kernel sums the partial matrices in slice order
__global__ __launch_bounds__(256) void big_gram_kernel(const float *__restrict__ Bn, int n, int len, int si, int sx,
                                                      double *__restrict__ Gpart, unsigned *sig_flag, unsigned sig_val) {
  const int ti = blockIdx.y, tj = blockIdx.x, ks = blockIdx.z;
  // "B_new is ready" for the side stream (big_gate_kernel): this kernel starts when the update before it in the stream is complete and
  // visible, so its first workgroup may say so at once -- a signal kernel of its own between the two costs this stream 5-6 us
  if (sig_flag && ti == 0 && tj == 0 && ks == 0 && threadIdx.x == 0)
    __hip_atomic_store(sig_flag, sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ti > tj) return;
  __shared__ float sA[16][65], sB[16][65];
  const int r = threadIdx.x >> 4, cidx = threadIdx.x & 15;
  const int chunk = ((len + kGramKS - 1) / kGramKS + 63) & ~63;
  const int x_lo = ks * chunk, x_hi = min(len, x_lo + chunk);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  for (int x0 = x_lo; x0 < x_hi; x0 += 64) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int row = e >> 6, x = e & 63;
      const int ia = ti * 16 + row, ib = tj * 16 + row;
      sA[row][x] = (ia < n && x0 + x < x_hi) ? Bn[(size_t)ia * si + (size_t)(x0 + x) * sx] : 0.f;
      sB[row][x] = (ib < n && x0 + x < x_hi) ? Bn[(size_t)ib * si + (size_t)(x0 + x) * sx] : 0.f;
    }
    __syncthreads();
    // (four accumulators: a dependent float64 FMA costs 40 cycles; the sum is re-associated, in float64)
#pragma unroll 4
    for (int x = 0; x < 64; x += 4) {
      acc0 = fma((double)sA[r][x], (double)sB[cidx][x], acc0); acc1 = fma((double)sA[r][x + 1], (double)sB[cidx][x + 1], acc1);
      acc2 = fma((double)sA[r][x + 2], (double)sB[cidx][x + 2], acc2); acc3 = fma((double)sA[r][x + 3], (double)sB[cidx][x + 3], acc3);
    }
    __syncthreads();
  }
  const double acc = (acc0 + acc1) + (acc2 + acc3);
  const int i = ti * 16 + r, j = tj * 16 + cidx;
  double *G = Gpart + (size_t)ks * n * n;
  if (i < n && j < n && i <= j) { G[(size_t)i * n + j] = acc; G[(size_t)j * n + i] = acc; }
}

// ---- two-sided Jacobi on the packed symmetric matrix, rotations logged ------------------------------------------
// Block (P, Q), P <= Q, of the current pairing has the components [e(2P,2Q), e(2P,2Q+1), e(2P+1,2Q), e(2P+1,2Q+1)]
// (component 2 of a diagonal block is unused).  Same schedule, look-ahead parameters and stopping rule as the
// in-LDS kernel (kernels_narrow.hip phase 7); see there for the derivation.
__device__ inline int blk_index(int P, int Q, int np) { return P * np - P * (P - 1) / 2 + (Q - P); }
// Storage: FOUR PLANES of nblk doubles -- component c = 2 (row & 1) + (col & 1) of block b at [c * nblk + b].  (Until late in round 3
// the four components of a block sat together, 32 bytes per block: the 16-byte reads of consecutive blocks were 2-way and the
// scattered 8-byte writes ~4-way bank conflicts -- SQ_LDS_BANK_CONFLICT was 56 % of SQ_LDS_IDX_ACTIVE, a quarter of the kernel's
// cycles.  Consecutive blocks are now 8 bytes apart in every plane.)
__device__ inline int elem_slot(int a, int b, int np, int nblk) {       // element (a, b) of the symmetric matrix, any order
  const int lo = min(a, b), hi = max(a, b);
  return (2 * (lo & 1) + (hi & 1)) * nblk + blk_index(lo >> 1, hi >> 1, np);
}

struct BigJacobiArgs {
  const double *G;       // kGramKS partial n x n matrices, row-major (input)
  int n, m;
  double stop2;
  double2 *rotlog;       // [round][np] (c, s)
  double *lam;           // [n] eigenvalues (diag of the rotated matrix, original scale) by final position
  int *info;             // [0] rounds applied, [1] sweeps, [2] converged
  unsigned long long *counters;
  int *status;
  // The replay of the rotation log rides in the same launch (round 3): workgroups 1.. apply the rotations to the unit vectors and to
  // the columns of W WHILE workgroup 0 produces them, sixteen vectors per workgroup (one per wave).  Hand-off: the log is written
  // with 16-byte agent-scope stores; once the entries of round r are known to be in memory (they were stored a whole round
  // earlier: `s_waitcnt vmcnt(1)` costs nothing) lane 0 publishes `token << 12 | rounds ready` in prog[0]; prog[1] gets the final
  // round count the same way when the iteration has ended.  The token makes words of earlier launches read as "nothing yet".
  const float *Bn;
  int len, si, sx;
  double *VW;
  unsigned *prog;
  unsigned token;
  // eigenvalue order, kept rank, sigma^(+-1/4) and the step's metrics: the tail of workgroup 0 (a launch of its own until round 3)
  int keep_max;          // kept rank under the fixed / reference policy (= m)
  double trunc_thr;      // > 0: adaptive truncation
  int *m_out;
  float *metrics;
  const float *red;      // the four metric slots sit behind its Bs gradient elements
  int Bs, L;
  double *ws;            // capture block / workspace (sigma at 4 Bs, scalars at 4 Bs + kDbgSigma)
};

// A replay workgroup: sixteen vectors, one per wave (pair k on lane k, the tournament move by one-lane wave shifts, as the eigenvector
// waves of the in-LDS kernel).  ONE thread of the workgroup polls the two progress words and hands what it saw to the others
// through LDS: with every wave polling (1100 of them, the same cache line) the update workgroup's own stores to that line
// queued behind the polls and the iteration crawled -- an intermittent time-out in the first build.
__device__ inline void big_replay_block(const BigJacobiArgs &a, int *sPoll) {
  const int n = a.n, np = n / 2, k = threadIdx.x & 63;
  const int v = 16 * ((int)blockIdx.x - 1) + (int)(threadIdx.x >> 6);
  const bool have = v < n + a.len;                       // (waves without a vector still meet the barriers)
  const int kk = k < np ? k : np - 1;
  double top = 0.0, bot = 0.0;
  if (have) {
    if (v < n) { top = (2 * kk == v) ? 1.0 : 0.0; bot = (2 * kk + 1 == v) ? 1.0 : 0.0; }
    else {
      const size_t x = (size_t)(v - n) * a.sx;
      top = (double)a.Bn[(size_t)(2 * kk) * a.si + x];
      bot = (double)a.Bn[(size_t)(2 * kk + 1) * a.si + x];
    }
  }
  const __amdgpu_buffer_rsrc_t rlog = sc1_rsrc(a.rotlog);
  constexpr int CH = 8;
  int done_r = 0;                      // rounds applied
  while (true) {
    if (threadIdx.x == 0) {
      // how far may this workgroup go?  (bounded: a producer that never comes sets a status bit and the workgroup gives up)
      int avail = 0, ended = 0;
      for (int spin = 0; spin < (1 << 21); ++spin) {
        const unsigned v1 = __hip_atomic_load(a.prog + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v1 >> 12) == a.token) { ended = 1; avail = (int)(v1 & 4095u); break; }
        const unsigned v0 = __hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        avail = (v0 >> 12) == a.token ? (int)(v0 & 4095u) : 0;
        if (avail >= done_r + CH) break;
        __builtin_amdgcn_s_sleep(64);
      }
      if (!ended && avail < done_r + CH) {               // timed out: leave what was seen for the host's message
        if (atomicOr(a.status, 16) == 0) {
          a.prog[2] = a.token; a.prog[3] = __hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          a.prog[4] = __hip_atomic_load(a.prog + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a.prog[5] = (unsigned)done_r; a.prog[6] = (unsigned)v;
        }
        ended = 2;                                       // give up
      }
      sPoll[0] = avail; sPoll[1] = ended;
    }
    __syncthreads();
    const int avail = sPoll[0], ended = sPoll[1];
    __syncthreads();                                     // (the words are rewritten by the next poll)
    if (ended == 2) return;
    const int upto = ended ? avail : done_r + ((avail - done_r) / CH) * CH;
    while (done_r < upto) {
      double2 cs[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int rr = min(done_r + u, upto - 1);
        const tn_uvec4 raw = ld_sc1_b128(rlog, (unsigned)(((size_t)rr * np + kk) * sizeof(double2)));
        cs[u] = __builtin_bit_cast(double2, raw);
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (done_r + u >= upto) break;                  // block-uniform
        const double nt = cs[u].x * top - cs[u].y * bot, nb = cs[u].y * top + cs[u].x * bot;
        if (np > 1) {
          const double t1 = dpp_f64<0x138>(nt), b1 = dpp_f64<0x138>(nb), c1 = dpp_f64<0x130>(nb);
          top = k == 0 ? nt : (k == 1 ? b1 : t1);
          bot = k == np - 1 ? nt : c1;
        } else {
          top = nt; bot = nb;
        }
      }
      done_r = min(done_r + CH, upto);
    }
    if (ended) break;
  }
  if (have && k < np) {
    a.VW[(size_t)v * n + 2 * k] = top;
    a.VW[(size_t)v * n + 2 * k + 1] = bot;
  }
}

// ---- eigenvalue order, kept rank, sigma^(+-1/2), metrics: the tail of the Jacobi workgroup --------------------------
//   lam3: [0,128) eigenvalues by position, [128,256) sigma^(1/2) and [256,384) sigma^(-1/2) of the kept columns (out)
//   info: [3] kept rank, [4..) position of the sp-th largest eigenvalue (out)
//   sLam[n] holds the eigenvalues by position (LDS, written by the caller, barrier passed); sOrd[n], sM: LDS scratch
__device__ inline void big_order_tail(const BigJacobiArgs &a, int sweeps, const double *sLam, int *sOrd, int *sM) {
  const int tid = threadIdx.x, NT = blockDim.x, n = a.n, Bs = a.Bs;
  double *lam3 = a.lam;
  int *info = a.info;
  for (int j = tid; j < n; j += NT) {
    const double lj = sLam[j];
    int rank = 0;
    for (int i = 0; i < n; ++i) { const double li = sLam[i]; rank += (li > lj) || (li == lj && i < j); }
    sOrd[rank] = j;
    info[4 + rank] = j;
    a.ws[4 * (size_t)Bs + rank] = sqrt(lj);
  }
  __syncthreads();
  if (tid == 0) {
    int me = a.keep_max;
    if (a.trunc_thr > 0.0) {                       // adaptive truncation (see narrow_step_kernel)
      double tot = 0.0;
      for (int j = 0; j < n; ++j) tot += sqrt(sLam[sOrd[j]]);
      double cum = 0.0;
      int idx = 0;
      bool found = false;
      for (int j = 0; j < n && !found; ++j) {
        cum += sqrt(sLam[sOrd[j]]);
        if (cum / tot > a.trunc_thr) { idx = j; found = true; }
      }
      me = min(a.keep_max, idx + 1);
      if (a.m_out) *a.m_out = me;
    }
    *sM = me;
    info[3] = me;
    double *sc = a.ws + 4 * (size_t)Bs + kDbgSigma;
    sc[3] = (double)sweeps;
    sc[4] = (double)n;
    if (a.metrics) {
      const double cnt = (double)a.red[Bs + 3];
      const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
      a.metrics[0] = (float)((double)a.red[Bs] * inv);
      a.metrics[1] = (float)((double)a.red[Bs + 1] * inv / (double)a.L);
      if (a.red[Bs + 2] != 0.f) atomicOr(a.status, 1);
    }
  }
  __syncthreads();
  const double lam_max = sLam[sOrd[0]];
  const int me = *sM;
  for (int sp = tid; sp < me; sp += NT) {
    const double l_ = sLam[sOrd[sp]];
    const bool ok = l_ > 1e-300 && l_ > 1e-30 * lam_max;
    const double sq = ok ? sqrt(sqrt(l_)) : 0.0;
    lam3[kBigMaxN + sp] = sq;
    lam3[2 * kBigMaxN + sp] = ok ? 1.0 / sq : 0.0;
  }
}

__global__ __launch_bounds__(1024) void big_jacobi_kernel(BigJacobiArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, NT = 1024;
  if (blockIdx.x > 0) {
    big_replay_block(a, (int *)smem_raw);
    return;
  }
  const int n = a.n, m = a.m, np = n / 2, ne = n;
  const int nblk = np * (np + 1) / 2;
  double *G0 = (double *)smem_raw, *G1 = G0 + 4 * (size_t)nblk;
  double *dCS = G1 + 4 * (size_t)nblk;            // [2][4 np]: per buffer t[np], c0[np] as doubles, then (t, c0)[np] as float2
  double *dRed = dCS + 8 * (size_t)np;            // 64
  int *sPi = (int *)(dRed + 64), *sPiInv = sPi + ne, *sFlag = sPiInv + ne;     // sFlag[2..3]: round index + 1 of the last big rotation

  // load + trace
  double trp = 0.0;
  for (int b = tid; b < nblk; b += NT) {
    int P = 0, rem = b;
    while (rem >= np - P) { rem -= np - P; ++P; }
    const int Q = P + rem;
    double e11 = 0.0, e12 = 0.0, e21 = 0.0, e22 = 0.0;
    for (int ks = 0; ks < kGramKS; ++ks) {                 // partial Gram matrices, fixed order
      const double *Gp = a.G + (size_t)ks * n * n;
      e11 += Gp[(size_t)(2 * P) * n + 2 * Q]; e12 += Gp[(size_t)(2 * P) * n + 2 * Q + 1];
      e21 += Gp[(size_t)(2 * P + 1) * n + 2 * Q]; e22 += Gp[(size_t)(2 * P + 1) * n + 2 * Q + 1];
    }
    G0[b] = e11; G0[nblk + b] = e12; G0[2 * nblk + b] = e21; G0[3 * nblk + b] = e22;
    if (P == Q) trp += e11 + e22;
  }
  for (int pos = tid; pos < ne; pos += NT) {
    const int kq = pos >> 1;
    int nxt;
    if (pos & 1) nxt = (kq == 0) ? (np > 1 ? 2 : 1) : 2 * (kq - 1) + 1;
    else nxt = (kq == 0) ? 0 : (kq == np - 1 ? 2 * kq + 1 : 2 * (kq + 1));
    sPi[pos] = nxt;
    sPiInv[nxt] = pos;
  }
  for (int off = 32; off > 0; off >>= 1) trp += __shfl_xor(trp, off);
  if ((tid & 63) == 0) dRed[tid >> 6] = trp;
  __syncthreads();
  double tr = 0.0;
  for (int w = 0; w < NT / 64; ++w) tr += dRed[w];
  const int sc_exp = (tr > 0.0 && isfinite(tr)) ? __builtin_amdgcn_frexp_exp(tr) : 0;
  __syncthreads();
  for (int e = tid; e < 4 * nblk; e += NT) G0[e] = __builtin_amdgcn_ldexp(G0[e], -sc_exp);
  __syncthreads();

  auto diag = [&](const double *G, int j) { return G[3 * (j & 1) * nblk + blk_index(j >> 1, j >> 1, np)]; };
  auto kept_scale = [&](const double *G) -> double {
    for (int j = tid; j < n; j += NT) {
      const double lj = diag(G, j);
      int rank = 0;
      for (int i = 0; i < n; ++i) { const double li = diag(G, i); rank += (li > lj) || (li == lj && i < j); }
      if (rank == m - 1) dRed[60] = lj;
    }
    __syncthreads();
    const double lm = kKeptFrac * fmax(dRed[60], 0.0);
    return lm * lm;
  };

  // items: blocks P <= Q, up to 3 per worker thread (np <= 64: 2080 blocks on 960 threads)
  constexpr int T0 = 64, MAXI = 3;
  const int NW = NT - T0;
  bool itValid[MAXI], itDiag[MAXI];
  int itSrc[MAXI], itQ[MAXI], itP[MAXI], itD11[MAXI], itD12[MAXI], itD21[MAXI], itD22[MAXI];
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int it = (tid - T0) + u * NW;
    itValid[u] = tid >= T0 && it < nblk;
    int P = 0, Q = 0;
    if (itValid[u]) {
      int rem = it;
      while (rem >= np - P) { rem -= np - P; ++P; }
      Q = P + rem;
    }
    const int c1 = sPi[2 * Q], c2 = sPi[2 * Q + 1], o1 = sPi[2 * P], o2 = sPi[2 * P + 1];
    itSrc[u] = blk_index(P, Q, np);
    itQ[u] = Q; itP[u] = P;
    itDiag[u] = P == Q;
    itD11[u] = elem_slot(o1, c1, np, nblk); itD12[u] = elem_slot(o1, c2, np, nblk);
    itD21[u] = elem_slot(o2, c1, np, nblk); itD22[u] = elem_slot(o2, c2, np, nblk);
  }
  // item slots this WAVE runs (wave-uniform: slot u is valid for a prefix of the worker threads)
  int wave_items = 0;
#pragma unroll
  for (int u = 0; u < MAXI; ++u) wave_items += __builtin_amdgcn_readfirstlane(__ballot(itValid[u]) != 0ull ? 1 : 0);
  const double abs2 = kJacobiAbs * kJacobiAbs;
  const bool isParam = tid < np;
  int pa = 0, pb = 1;
  if (isParam) { pa = sPiInv[2 * tid]; pb = sPiInv[2 * tid + 1]; }
  const int pA = pa >> 1, ra = pa & 1, pB = pb >> 1, rb = pb & 1;
  const int slotAA = blk_index(pA, pA, np), slotBB = blk_index(pB, pB, np);
  const int slotAB = blk_index(min(pA, pB), max(pA, pB), np);

  int sweeps = 0, converged = 0, cur = 0, rounds = 0;
  double *Gc = G0, *Gn = G1;
  double kept2 = 0.0;
  if (n > 1) {
    kept2 = kept_scale(Gc);
    if (tid == 0) { sFlag[2] = 0; sFlag[3] = 0; }
    // Rotation of pair k in a dCS buffer: t at [k], c0 at [np + k] as doubles (float32-exact values), and the same two at [2 np ..) as a
    // float2 for the look-ahead chain -- the scheme of the in-LDS kernel (kernels_narrow.hip phase 7, jacobi_rot_f32): the
    // dependent chain that sets the length of a round runs in float32, the threads that APPLY a rotation refine its cosine
    // to float64 themselves (rot_corr), and the log for the replay kernel gets the refined (c, s) off the critical path.
    const float kept_lo = 1e-36f;
    const __amdgpu_buffer_rsrc_t rlog = sc1_rsrc(a.rotlog);
    auto publish = [&](double *buf, int pair, const RotT &r, int applied, size_t log_at) {
      buf[pair] = (double)r.t; buf[np + pair] = (double)r.c0;
      reinterpret_cast<float2 *>(buf + 2 * np)[pair] = make_float2(r.t, r.c0);
      if (r.level >= 2) sFlag[2 + (applied & 1)] = applied + 1;
      const double c = (double)r.c0 * rot_corr((double)r.t, (double)r.c0);
      st_sc1_b128(rlog, (unsigned)(log_at * sizeof(double2)), __builtin_bit_cast(tn_uvec4, make_double2(c, c * (double)r.t)));
    };
    if (isParam) {
      const int sl = blk_index(tid, tid, np);
      const RotT r = jacobi_rot_f32((float)Gc[sl], (float)Gc[3 * nblk + sl], (float)Gc[nblk + sl], fmaxf((float)kept2, kept_lo), (float)abs2,
                                    (float)a.stop2);
      publish(dCS + cur * np * 4, tid, r, 0, (size_t)tid);
    }
    __syncthreads();
    // sliding window, as in the in-LDS kernel (kernels_narrow.hip phase 7): stop as soon as ne - 1 consecutive rounds applied no
    // big rotation
    int last_big1 = 0;
    for (; sweeps < kJacobiMaxSweeps && !converged; ++sweeps) {
      for (int rnd = 0; rnd < ne - 1; ++rnd) {
        const double *csc = dCS + cur * np * 4;
        const int big_slot = sFlag[2 + (rounds & 1)];
        if (isParam) {
          // look-ahead: pair `tid` of the NEXT round is (a, b) in today's positions; its three elements after today's
          // rotations, in float32
          const float2 fA = reinterpret_cast<const float2 *>(csc + 2 * np)[pA];      // (t, c0) of pair A
          const float2 fB = reinterpret_cast<const float2 *>(csc + 2 * np)[pB];
          const double2 dA = make_double2(Gc[slotAA], Gc[nblk + slotAA]);
          const double bA = Gc[3 * nblk + slotAA];
          const double2 dB = make_double2(Gc[slotBB], Gc[nblk + slotBB]);
          const double bB = Gc[3 * nblk + slotBB];
          double2 r0, r1;
          if (pA < pB) {
            r0 = make_double2(Gc[slotAB], Gc[nblk + slotAB]);
            r1 = make_double2(Gc[2 * nblk + slotAB], Gc[3 * nblk + slotAB]);
          } else if (pA > pB) {
            r0 = make_double2(Gc[slotAB], Gc[2 * nblk + slotAB]);               // (the stored block is the transpose)
            r1 = make_double2(Gc[nblk + slotAB], Gc[3 * nblk + slotAB]);
          } else {
            r0 = dA;
            r1 = make_double2(dA.y, bA);
          }
          const float tA = fA.x, cA = fA.y, sA = fA.x * fA.y, tB = fB.x, cB = fB.y, sB = fB.x * fB.y;
          const float aAx = (float)dA.x, aAy = (float)dA.y, aAb = (float)bA, aBx = (float)dB.x, aBy = (float)dB.y, aBb = (float)bB;
          const float q0x = (float)r0.x, q0y = (float)r0.y, q1x = (float)r1.x, q1y = (float)r1.y;
          const float na = ra ? fmaf(tA, aAy, aAb) : fmaf(-tA, aAy, aAx);
          const float nb = rb ? fmaf(tB, aBy, aBb) : fmaf(-tB, aBy, aBx);
          const float h0 = ra ? fmaf(sA, q0x, cA * q1x) : fmaf(cA, q0x, -sA * q1x);
          const float h1 = ra ? fmaf(sA, q0y, cA * q1y) : fmaf(cA, q0y, -sA * q1y);
          const float ng = rb ? fmaf(sB, h0, cB * h1) : fmaf(cB, h0, -sB * h1);
          const RotT r = jacobi_rot_f32(na, nb, ng, fmaxf((float)kept2, kept_lo), (float)abs2, (float)a.stop2);
          publish(dCS + (cur ^ 1) * np * 4, tid, r, rounds + 1, (size_t)(rounds + 1) * np + tid);
        }
        if (tid < 64) {
          // wave 0 holds every parameter thread: all of its log stores but the one just issued are complete -> the entries of
          // rounds 0 .. `rounds` are in memory (they were stored one round ago and earlier)
          asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          if (tid == 0) __hip_atomic_store(a.prog, (a.token << 12) | (unsigned)(rounds + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // The items of a lane as INDEPENDENT chains: all operands are requested first, the (dependent, 40-cycle) float64 steps of the
        // items interleave, stores last.  A per-item `if (!valid) continue` made them run one after the other: a round of the
        // n = 100 matrix (1275 blocks on 960 worker threads: two items on a third of them) cost 2760 cycles.  The number of
        // item slots a WAVE runs is wave-uniform (items are dealt slot by slot), so waves without a second item skip it with a
        // scalar branch.
        auto run_items = [&](auto KC) {
          constexpr int K = decltype(KC)::value;
          double2 tq[K], tp[K], r0[K], r1[K];
#pragma unroll
          for (int u = 0; u < K; ++u) {
            tq[u] = make_double2(csc[itQ[u]], csc[np + itQ[u]]);            // (t, c0) of the column pair
            tp[u] = make_double2(csc[itP[u]], csc[np + itP[u]]);            // ... of the row pair
            r0[u] = make_double2(Gc[itSrc[u]], Gc[nblk + itSrc[u]]);
            r1[u] = make_double2(Gc[2 * nblk + itSrc[u]], Gc[3 * nblk + itSrc[u]]);
            if (itDiag[u]) r1[u].x = r0[u].y;
          }
          double n11[K], n12[K], n21[K], n22[K];
#pragma unroll
          for (int u = 0; u < K; ++u) {
            // R_P^T . blk . R_Q = cP cQ [[1, -tP], [tP, 1]] . blk . [[1, tQ], [-tQ, 1]]: the tangents act first, the product of the
            // two refined cosines is formed meanwhile and multiplied in last
            const double a11 = fma(-tp[u].x, r1[u].x, r0[u].x), a12 = fma(-tp[u].x, r1[u].y, r0[u].y);
            const double a21 = fma(tp[u].x, r0[u].x, r1[u].x), a22 = fma(tp[u].x, r0[u].y, r1[u].y);
            const double b11 = fma(-tq[u].x, a12, a11), b12 = fma(tq[u].x, a11, a12);
            const double b21 = fma(-tq[u].x, a22, a21), b22 = fma(tq[u].x, a21, a22);
            const double c00 = tp[u].y * tq[u].y, corr = rot_corr(tp[u].x, tp[u].y) * rot_corr(tq[u].x, tq[u].y);
            n11[u] = (b11 * c00) * corr; n12[u] = (b12 * c00) * corr; n21[u] = (b21 * c00) * corr; n22[u] = (b22 * c00) * corr;
            if (itDiag[u] && tq[u].x != 0.0) { n12[u] = 0.0; n21[u] = 0.0; }
          }
#pragma unroll
          for (int u = 0; u < K; ++u) {
            if (itValid[u]) {
              Gn[itD11[u]] = n11[u]; Gn[itD12[u]] = n12[u];
              if (!itDiag[u]) Gn[itD21[u]] = n21[u];
              Gn[itD22[u]] = n22[u];
            }
          }
        };
        if (wave_items == 1) run_items(std::integral_constant<int, 1>());
        else if (wave_items == 2) run_items(std::integral_constant<int, 2>());
        else if (wave_items == 3) run_items(std::integral_constant<int, 3>());
        // LDS-only barrier: __syncthreads() would also drain the parameter threads' stores to the rotation log (global memory,
        // read by the next kernel) -- a memory round trip per round
        lds_barrier();
        double *tsw = Gc; Gc = Gn; Gn = tsw;
        cur ^= 1;
        ++rounds;
        last_big1 = max(last_big1, big_slot);
        // block-uniform, and told so (the flag word comes out of LDS): a scalar loop exit
        if (rounds - __builtin_amdgcn_readfirstlane(last_big1) >= ne - 1) { converged = 1; break; }
      }
      if (!converged) kept2 = kept_scale(Gc);
    }
  } else {
    converged = 1;
  }
  if (tid < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every log entry is in memory
    if (tid == 0) {
      __hip_atomic_store(a.prog + 1, (a.token << 12) | (unsigned)rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.prog, (a.token << 12) | (unsigned)rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // eigenvalues by position -> memory (the cores kernel) and LDS (their order, below)
  double *sLam = (double *)(sFlag + 8);            // (2 n + 8 ints before it: 8-byte aligned)
  int *sOrd = (int *)(sLam + kBigMaxN), *sM = sOrd + kBigMaxN;
  for (int j = tid; j < n; j += NT) {
    const double lj = __builtin_amdgcn_ldexp(fmax(diag(Gc, j), 0.0), sc_exp);
    a.lam[j] = lj;
    sLam[j] = lj;
  }
  __syncthreads();
  big_order_tail(a, sweeps, sLam, sOrd, sM);
  if (tid == 0) {
    a.info[0] = rounds; a.info[1] = sweeps; a.info[2] = converged;
    if (a.counters) {
      atomicAdd(a.counters, (unsigned long long)sweeps);
      atomicAdd(a.counters + 1, 1ull);
      atomicAdd(a.counters + 2, (unsigned long long)rounds);
    }
    if (!converged) atomicOr(a.status, 2);
  }
}

// ---- the two new cores: short side V[kk][pos_j] sigma_j^(1/2), long side (W^T V)[x][pos_j] sigma_j^(-1/2) ------------
__device__ inline void big_cores_body(const NarrowParams &p, const double *__restrict__ lam3, const int *__restrict__ info,
                                      const double *__restrict__ VW, float *__restrict__ Cb, int blk, int nblk) {
  const int D = kD, h = p.h, g = p.g, L = p.L;
  const int r = D * h, c = D * g * L;
  const bool short_rows = r <= c;
  const int n = short_rows ? r : c, len = short_rows ? c : r;
  const int m = info[3];
  const int *ord = info + 4;
  int ob_s_h = p.ob_s_h, ob_s_d = p.ob_s_d, oa_s_d = p.oa_s_d, oa_s_g = p.oa_s_g;
  if (p.trunc_thr > 0.0) { if (!p.left_dir) { ob_s_h = D * m; ob_s_d = m; } else { oa_s_d = m * L; oa_s_g = D * m * L; } }
  for (int e = blk * kBT + threadIdx.x; e < (n + len) * m; e += nblk * kBT) {
    const int row = e / m, sp = e - row * m;
    const bool is_short = row < n;
    const float v = (float)(VW[(size_t)row * n + ord[sp]] * lam3[(is_short ? 1 : 2) * kBigMaxN + sp]);
    const int x = is_short ? row : row - n;              // index on its own side
    if (is_short == short_rows) {                        // behind core: index (h_, dk)
      Cb[x * m + sp] = v;
      p.out_behind[(x / D) * ob_s_h + (x % D) * ob_s_d + sp * p.ob_s_m] = v;
    } else {                                             // ahead core: index (dk1, g_, l)
      const int l = x % L, q = x / L;
      p.out_ahead[sp * p.oa_s_m + (q / g) * oa_s_d + (q % g) * oa_s_g + l] = v;
    }
  }
}
__global__ __launch_bounds__(kBT) void big_cores_kernel(NarrowParams p, const double *__restrict__ lam3, const int *__restrict__ info,
                                                       const double *__restrict__ VW, float *__restrict__ Cb) {
  big_cores_body(p, lam3, info, VW, Cb, blockIdx.x, gridDim.x);
}

// ---- behind norm environment of the next step: Nh_new = Cb^T (Nh (x) 1_d) Cb -----------------------------------
// 16 lanes share one output element and split its inner sum (the outputs alone are too few to fill the chip);
// the partial sums meet by xor shuffles inside the 16-lane group, in a fixed order
// The cores launch and T2 = (Nh (x) 1_d) . Cb in ONE launch: an element of the behind core is one product away from what the replay
// left -- Cb[x][j] = float(VW[row(x)][ord_j] sigma_j^(+-1/2)) -- so T2 does not have to wait for the cores (a launch of one thread
// takes 4.4 us here; the same sums as the separate launch it replaces, operand for operand).
__global__ __launch_bounds__(kBT) void big_cores_normT_kernel(NarrowParams p, const double *__restrict__ lam3, const int *__restrict__ info,
                                                             const double *__restrict__ VW, float *__restrict__ Cb,
                                                             double *__restrict__ T2, int nt_blocks) {
  if ((int)blockIdx.x >= nt_blocks) {
    big_cores_body(p, lam3, info, VW, Cb, (int)blockIdx.x - nt_blocks, (int)gridDim.x - nt_blocks);
    return;
  }
  const int D = kD, h = p.h, m = info[3], DM = D * m;
  const int rr = D * h, cc = D * p.g * p.L;
  const bool short_rows = rr <= cc;
  const int n = short_rows ? rr : cc;
  const double *vb = VW + (short_rows ? (size_t)0 : (size_t)n * n);     // the behind side's block of VW
  const double *lam = lam3 + (short_rows ? 1 : 2) * kBigMaxN;
  const int *ord = info + 4;
  const int sub = threadIdx.x & 15;
  for (int e = (blockIdx.x * kBT + threadIdx.x) >> 4; e < h * DM; e += (nt_blocks * kBT) >> 4) {
    const int j = e % DM, i = e / DM;
    const int d = j / m, sp = j - d * m;                   // column (d, sp) of Cb viewed as [h][D m]
    const double *col = vb + ord[sp];
    const double ls = lam[sp];
    double acc = 0.0;
    if (p.Nh) { for (int kk = sub; kk < h; kk += 16) acc += p.Nh[i * h + kk] * (double)(float)(col[(size_t)(kk * D + d) * n] * ls); }
    else if (sub == 0) acc = (double)(float)(col[(size_t)(i * D + d) * n] * ls);
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (sub == 0) T2[e] = acc;
  }
}
__global__ __launch_bounds__(kBT) void big_norm_out_kernel(NarrowParams p, const float *__restrict__ Cb, const double *__restrict__ T2,
                                                          const int *__restrict__ m_dev) {
  const int D = kD, h = p.h, m = m_dev[0];
  const int sub = threadIdx.x & 15;
  for (int e = (blockIdx.x * kBT + threadIdx.x) >> 4; e < m * m; e += (gridDim.x * kBT) >> 4) {
    const int j = e % m, i = e / m;
    double acc = 0.0;
    for (int kk = sub; kk < h * D; kk += 16) acc += (double)Cb[(size_t)kk * m + i] * T2[(size_t)kk * m + j];
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (sub == 0) p.Nh_new[e] = acc;
  }
}
// (one launch with a workgroup per output column was tried in round 3 -- 14 us against 5.9 + 5.5 for these two: removed)

}  // namespace

size_t big_jacobi_lds_bytes(int n) {
  const int np = n / 2;
  const size_t nblk = (size_t)np * (np + 1) / 2;
  return (8 * nblk + 8 * (size_t)np + 64) * sizeof(double) + (2 * (size_t)n + 8) * sizeof(int) + 16 +
         kBigMaxN * (sizeof(double) + sizeof(int)) + 16;       // + the order tail's eigenvalues, positions, kept rank
}

// Every launch of this path goes through big_launch: grid, block and dynamic-LDS sizes are validated against the
// device limits BEFORE the launch (an illegal AQL packet aborts the queue, it does not return an error), and in
// checking mode (tnml_debug_enable bit 2) the launch status is read back after each one, so that a failure names its kernel.
static thread_local char g_big_err[256];
const char *big_launch_error() { return g_big_err; }

// (two halves around a launch that names its kernel directly: a launch through a function-pointer parameter is dropped by clang
// when the host side is built with -fsanitize=undefined, see san/hip_stub.cpp)
static bool big_launch_ok(const char *name, dim3 grid, dim3 block, size_t lds) {
  const unsigned long long nthreads = (unsigned long long)block.x * block.y * block.z;
  if (grid.x < 1 || grid.y < 1 || grid.z < 1 || grid.x > 0x7fffffffu || grid.y > 65535u || grid.z > 65535u ||
      nthreads < 1 || nthreads > 1024 || lds > 160 * 1024) {
    snprintf(g_big_err, sizeof g_big_err, "%s: illegal launch grid (%u,%u,%u) block (%u,%u,%u) dynamic LDS %zu", name, grid.x,
             grid.y, grid.z, block.x, block.y, block.z, lds);
    return false;
  }
  return true;
}
static bool big_launch_done(const char *name, bool check, dim3 grid, dim3 block, size_t lds) {
  if (check) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      snprintf(g_big_err, sizeof g_big_err, "%s: launch failed: %s (grid %u,%u,%u block %u lds %zu)", name, hipGetErrorString(e),
               grid.x, grid.y, grid.z, block.x, lds);
      return false;
    }
  }
  return true;
}

// ---- pipelined large-tensor step (round 3): the two small kernels around the batch kernel that runs beside the SVD -----------
// (1) Behind environment of step k and the (h, d) operand of the pre-gradient Z_{k+1}:
//       E_k[h'][s]        = sum_{(h, d)} E_{k-1}[h][s] x_{k-1}[s][d] A_{k-1}(h, d, h')                 (Network_class.py:637-652)
//       P'_k[(h', d')][s] = E_k[h'][s] x_k[s][d']
//     64 samples per workgroup, the core in LDS (every lane of a wave reads the same element: a broadcast), four groups of
//     bond indices; plain FMAs: 25 MFLOP, a few microseconds beside a 170 us Jacobi kernel.
struct BigExtArgs { const float *Eprev, *x_km1, *x_k; CoreView A; int b_pad; float *Ecur, *Pk; };
//     (`bx`: which 64 samples, `by`: which sixteen bond indices; the loads of eight environment rows are in flight together -- with one
//      per loop trip the kernel was one L2 latency per row: 15 us at C5)
__device__ inline void big_ext_body(const BigExtArgs &a, float *sAext, int bx, int by) {
  const CoreView &A = a.A;
  const int hp = A.n_in, h = A.n_out, nI = hp * kD, HS = (h + 3) & ~3, b_pad = a.b_pad;
  for (int e = threadIdx.x; e < nI * HS; e += 256) {
    const int i = e / HS, o = e - i * HS;
    sAext[e] = o < h ? A.base[(size_t)(i >> 1) * A.s_in + (i & 1) * A.s_d + (size_t)o * A.s_out] : 0.f;
  }
  __syncthreads();
  // 64 samples x 16 bond indices per workgroup: a lane owns one sample and four consecutive h' (one 16-byte LDS read per core row)
  const int s = bx * 64 + (threadIdx.x & 63), o = by * 16 + 4 * (threadIdx.x >> 6);
  if (o >= h) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float xm0 = a.x_km1[(size_t)s * kD], xm1 = a.x_km1[(size_t)s * kD + 1];
  const float x0 = a.x_k[(size_t)s * kD], x1 = a.x_k[(size_t)s * kD + 1];
  for (int h0 = 0; h0 < hp; h0 += 8) {
    float ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ev[u] = a.Eprev[(size_t)min(h0 + u, hp - 1) * b_pad + s];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int hh = h0 + u;
      if (hh < hp) {
        const float e0 = ev[u] * xm0, e1 = ev[u] * xm1;
        const float4 a0 = *reinterpret_cast<const float4 *>(sAext + (size_t)(2 * hh) * HS + o);
        const float4 a1 = *reinterpret_cast<const float4 *>(sAext + (size_t)(2 * hh + 1) * HS + o);
        acc.x = fmaf(e1, a1.x, fmaf(e0, a0.x, acc.x)); acc.y = fmaf(e1, a1.y, fmaf(e0, a0.y, acc.y));
        acc.z = fmaf(e1, a1.z, fmaf(e0, a0.z, acc.z)); acc.w = fmaf(e1, a1.w, fmaf(e0, a0.w, acc.w));
      }
    }
  }
  const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (o + u < h) {
      a.Ecur[(size_t)(o + u) * b_pad + s] = v[u];
      a.Pk[(size_t)(2 * (o + u)) * b_pad + s] = v[u] * x0;
      a.Pk[(size_t)(2 * (o + u) + 1) * b_pad + s] = v[u] * x1;
    }
}
__global__ __launch_bounds__(256) void big_ext_kernel(BigExtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sAext[];          // [(h, d)][h' padded to a multiple of 4]
  big_ext_body(a, sAext, blockIdx.x, blockIdx.y);
}

// (2) The raw gradient of step k+1 from the reduced pre-gradient and the behind core the SVD of step k has just left:
//       dB_{k+1}[h', c] = sum_{i = (h, d)} A_k(h, d, h') Z_{k+1}[i][c],      c = (d', d'', g, l)                 
//     one column and eight h' per thread, the core in LDS (16-byte broadcast reads); the four metric slots behind Z are carried
//     over.  Fixed summation order: every rank gets the same bits.
//     `chunk`: which 64 columns this WAVE takes; `by`: which eight h'; all waves of the workgroup share `by` and stage the core together
__device__ inline void big_contract_body(const float *__restrict__ Z, const CoreView &A, int ncols, float *__restrict__ red, float *sAc,
                                         int chunk, int by, bool carry_metrics) {
  const int hp = A.n_in, h = A.n_out, nI = hp * kD;
  const int o0 = by * 8;
  for (int e = threadIdx.x; e < nI * 8; e += blockDim.x) {                // the eight columns of the core this workgroup uses
    const int i = e >> 3, o = o0 + (e & 7);
    sAc[e] = o < h ? A.base[(size_t)(i >> 1) * A.s_in + (i & 1) * A.s_d + (size_t)o * A.s_out] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int c = chunk * 64 + lane;
  if (c < ncols) {
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    int i = 0;
    for (; i + 8 <= nI; i += 8) {                   // eight rows of Z in flight (few waves per CU: nothing else hides the latency)
      float z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = Z[(size_t)(i + u) * ncols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 a = *reinterpret_cast<const float4 *>(sAc + 8 * (i + u)), b2 = *reinterpret_cast<const float4 *>(sAc + 8 * (i + u) + 4);
        lo.x = fmaf(a.x, z[u], lo.x); lo.y = fmaf(a.y, z[u], lo.y); lo.z = fmaf(a.z, z[u], lo.z); lo.w = fmaf(a.w, z[u], lo.w);
        hi.x = fmaf(b2.x, z[u], hi.x); hi.y = fmaf(b2.y, z[u], hi.y); hi.z = fmaf(b2.z, z[u], hi.z); hi.w = fmaf(b2.w, z[u], hi.w);
      }
    }
    for (; i < nI; ++i) {
      const float z = Z[(size_t)i * ncols + c];
      const float4 a = *reinterpret_cast<const float4 *>(sAc + 8 * i), b2 = *reinterpret_cast<const float4 *>(sAc + 8 * i + 4);
      lo.x = fmaf(a.x, z, lo.x); lo.y = fmaf(a.y, z, lo.y); lo.z = fmaf(a.z, z, lo.z); lo.w = fmaf(a.w, z, lo.w);
      hi.x = fmaf(b2.x, z, hi.x); hi.y = fmaf(b2.y, z, hi.y); hi.z = fmaf(b2.z, z, hi.z); hi.w = fmaf(b2.w, z, hi.w);
    }
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (o0 + u < h) red[(size_t)(o0 + u) * ncols + c] = v[u];
  }
  if (carry_metrics && threadIdx.x < kMetricSlots) red[(size_t)h * ncols + threadIdx.x] = Z[(size_t)nI * ncols + threadIdx.x];
}
__global__ __launch_bounds__(64) void big_contract_kernel(const float *__restrict__ Z, CoreView A, int ncols, float *__restrict__ red) {
  extern __shared__ __attribute__((aligned(16))) float sAc[];             // [(h, d)][8]
  big_contract_body(Z, A, ncols, red, sAc, blockIdx.x, blockIdx.y, blockIdx.x == 0 && blockIdx.y == 0);
}
// ---- hand-offs between the context's stream and the side stream without events (round 3) -----------------------------------------
// A cross-queue event dependency costs 6-7 us on the stream that records or waits even when it is already satisfied, and 12-14 us
// from the producer's end to the consumer's start (tools/c5_gaps.py).  Instead: the producer's stream runs a one-thread kernel that
// stores a sequence number (big_signal_kernel: everything before it in its stream is complete and visible at agent scope), and the
// consumer either polls that word from the workgroups that need the data (big_front_kernel) or runs a one-wave kernel in front of them
// that does (big_gate_kernel).  A waiting kernel is always enqueued AFTER the kernels it waits for, the waits are bounded, and a
// time-out sets a status bit the host reports (TNML_ERR_STATE).
__global__ void big_signal_kernel(unsigned *flag, unsigned value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool big_flag_reached(const unsigned *flag, unsigned want) {
  return (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0;      // (sequence numbers wrap)
}
__device__ inline void big_spin(const unsigned *flag, unsigned want, int *status, int bit) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (big_flag_reached(flag, want)) return;
    __builtin_amdgcn_s_sleep(32);
  }
  atomicOr(status, bit);
}
__global__ void big_gate_kernel(const unsigned *flag, unsigned want, int *status) {
  if (threadIdx.x == 0) big_spin(flag, want, status, 32);
}
// inside a workgroup: thread 0 waits, everybody leaves with what was written before the signal visible (agent-scope acquire: the
// L2 of this XCD may hold lines of the same buffer from an earlier step)
// (acquire = false: the caller only must not WRITE before the signal -- nothing it reads was written by the other stream)
__device__ inline void big_poll(const unsigned *flag, unsigned want, int *status, bool acquire) {
  if (threadIdx.x == 0) big_spin(flag, want, status, 64);
  __syncthreads();
  if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// The contraction and the two small products of the L2 term need nothing from each other: one launch, the contraction's workgroups
// first (theirs is the longer latency chain), four column chunks each; then NL = Nh^T . lab and PR = pl . Ng (big_merge_wd_mfma_kernel).
__device__ inline void big_nlpr_body(const NarrowParams &p, double *__restrict__ NL, double *__restrict__ PR, int blk, int nblk) {
  const int D = kD, h = p.h, g = p.g, sb = p.s, L = p.L;
  const int nNL = h * D * sb * L, nPR = sb * D * g;
  for (int o = blk * kBT + threadIdx.x; o < nNL + nPR; o += nblk * kBT) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (o < nNL) {                                     // NL[h', dk, s, l] = sum_a Nh[a, h'] lab(a, dk, s, l)
      const int l = o % L;
      int q = o / L;
      const int s_ = q % sb; q /= sb;
      const int dk = q % D, hq = q / D;
      const float *la = p.lab.base + dk * p.lab.s_d + s_ * p.lab.s_out + l;
      if (!p.Nh) { NL[o] = (double)la[hq * p.lab.s_in]; continue; }
      const double *nh = p.Nh + hq;
      int a = 0;
      for (; a + 8 <= h; a += 8) {
        double nv[8]; float lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { nv[u] = nh[(size_t)(a + u) * h]; lv[u] = la[(a + u) * p.lab.s_in]; }
        a0 = fma(nv[0], (double)lv[0], a0); a1 = fma(nv[1], (double)lv[1], a1); a2 = fma(nv[2], (double)lv[2], a2); a3 = fma(nv[3], (double)lv[3], a3);
        a0 = fma(nv[4], (double)lv[4], a0); a1 = fma(nv[5], (double)lv[5], a1); a2 = fma(nv[6], (double)lv[6], a2); a3 = fma(nv[7], (double)lv[7], a3);
      }
      for (; a < h; ++a) a0 = fma(nh[(size_t)a * h], (double)la[a * p.lab.s_in], a0);
      NL[o] = (a0 + a1) + (a2 + a3);
    } else {                                           // PR[s, dk1, f] = sum_c pl(s, dk1, c) Ng[c, f]
      const int oo = o - nNL;
      const int f_ = oo % g;
      int q = oo / g;
      const int dk1 = q % D, s_ = q / D;
      const float *pl = p.pl.base + s_ * p.pl.s_in + dk1 * p.pl.s_d;
      if (!p.Ng) { PR[oo] = (double)pl[f_ * p.pl.s_out]; continue; }
      const double *ng = p.Ng + f_;
      int c = 0;
      for (; c + 8 <= g; c += 8) {
        double nv[8]; float pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { nv[u] = ng[(size_t)(c + u) * g]; pv[u] = pl[(c + u) * p.pl.s_out]; }
        a0 = fma((double)pv[0], nv[0], a0); a1 = fma((double)pv[1], nv[1], a1); a2 = fma((double)pv[2], nv[2], a2); a3 = fma((double)pv[3], nv[3], a3);
        a0 = fma((double)pv[4], nv[4], a0); a1 = fma((double)pv[5], nv[5], a1); a2 = fma((double)pv[6], nv[6], a2); a3 = fma((double)pv[7], nv[7], a3);
      }
      for (; c < g; ++c) a0 = fma((double)pl[c * p.pl.s_out], ng[(size_t)c * g], a0);
      PR[oo] = (a0 + a1) + (a2 + a3);
    }
  }
}
// All three jobs are small matrix products: on the matrix cores, one 16 x 16 output tile per wave, operands straight from memory
// (each element loaded once per tile, four k-steps in flight).  Lane maps: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15];
// C col = lane & 15 and row = 4 (lane >> 4) + reg for v_mfma_f32_16x16x4_f32, row = (lane >> 4) + 4 reg for v_mfma_f64_16x16x4_f64.
// As lane-per-output FMA loops (with the core in LDS) the launch took 19 us at C5, the environment job the longest.
typedef float big_fvec4 __attribute__((ext_vector_type(4)));
template <class FA, class FB>
__device__ inline big_fvec4 big_tile_f32(int nk, FA loadA, FB loadB) {      // loadX(kk): this lane's operand of k index kk (masked by the caller's map)
  const int q = (threadIdx.x & 63) >> 4;
  big_fvec4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int k0 = 0; k0 < nk; k0 += 4) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = loadA(4 * (k0 + u) + q); b[u] = loadB(4 * (k0 + u) + q); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < nk) {                                                     // wave-uniform
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc0, 0, 0, 0);
      }
  }
  return acc0 + acc1;
}
template <class FA, class FB>
__device__ inline big_dvec4 big_tile_f64(int nk, FA loadA, FB loadB) {
  const int q = (threadIdx.x & 63) >> 4;
  big_dvec4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  for (int k0 = 0; k0 < nk; k0 += 4) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = loadA(4 * (k0 + u) + q); b[u] = loadB(4 * (k0 + u) + q); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < nk) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc0, 0, 0, 0);
      }
  }
  return acc0 + acc1;
}

struct BigFrontTiles { int contract_blocks, ext_blocks; };       // workgroups of the first two jobs (four tiles each); the rest: NL, PR
__global__ __launch_bounds__(kBT) void big_front_kernel(NarrowParams p, double *__restrict__ NL, double *__restrict__ PR,
                                                       const float *__restrict__ Z, CoreView A, int ncols, float *__restrict__ red,
                                                       BigFrontTiles ft, BigExtArgs ext,
                                                       const unsigned *poll_flag, unsigned poll_want, int ext_acquire) {
  const int bid = blockIdx.x, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nc0 = ft.contract_blocks, nc1 = nc0 + ft.ext_blocks;
  // Z, and the buffers the extension rewrites, belong to the side stream until its chain of the previous step has signalled.  The
  // contraction READS what that stream wrote (acquire); the extension reads it only behind the first pipelined step of a sweep,
  // whose own extension ran over there (an acquire in each of its workgroups empties this XCD's L2 as often: +9 us).
  if (poll_flag && bid < nc1) big_poll(poll_flag, poll_want, p.status, bid < nc0 || ext_acquire != 0);
  if (bid < nc0) {
    // ---- raw gradient dB[h'][c] = sum_{i = (h, d)} A(i, h') Z[i][c]  (float32, as the batch kernel's own sums) ------------------
    const int hp = A.n_in, h = A.n_out, nI = hp * kD, tcn = (ncols + 15) >> 4, ntiles = ((h + 15) >> 4) * tcn;
    const int t = bid * (kBT / 64) + wave;
    if (t < ntiles) {
      const int ti = t / tcn, tj = t - ti * tcn;
      const int o_a = min(16 * ti + r, h - 1), c_b = min(16 * tj + r, ncols - 1);
      const float *pa = A.base + (size_t)o_a * A.s_out;
      const float *pz = Z + c_b;
      const big_fvec4 acc = big_tile_f32((nI + 3) >> 2,
          [&](int i) { const int ic = min(i, nI - 1); const float v = pa[(size_t)(ic >> 1) * A.s_in + (ic & 1) * A.s_d]; return i < nI ? v : 0.f; },
          [&](int i) { return pz[(size_t)min(i, nI - 1) * ncols]; });
      const int c = 16 * tj + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int o = 16 * ti + 4 * q + reg;
        if (o < h && c < ncols) red[(size_t)o * ncols + c] = acc[reg];
      }
    }
    if (bid == 0 && threadIdx.x < kMetricSlots) red[(size_t)h * ncols + threadIdx.x] = Z[(size_t)nI * ncols + threadIdx.x];
  } else if (bid < nc1) {
    // ---- behind environment E_k[h'][s] = sum_{i = (h, d)} A(i, h') (E_{k-1}[h][s] x_{k-1}[s][d]) and P'_k = E_k (x) x_k for the batch
    //      kernel "of step k+1" on the side stream: they need only what step k-1 left --------------------------------------------------
    const CoreView &EA = ext.A;
    const int hp = EA.n_in, h = EA.n_out, nI = hp * kD, b_pad = ext.b_pad, tcn = b_pad >> 4, ntiles = ((h + 15) >> 4) * tcn;
    const int t = (bid - nc0) * (kBT / 64) + wave;
    if (t < ntiles) {
      const int ti = t / tcn, tj = t - ti * tcn;
      const int o_a = min(16 * ti + r, h - 1), s_b = 16 * tj + r;               // (b_pad is a multiple of 64)
      const float *pa = EA.base + (size_t)o_a * EA.s_out;
      const float *pe = ext.Eprev + s_b;
      const float xm0 = ext.x_km1[(size_t)s_b * kD], xm1 = ext.x_km1[(size_t)s_b * kD + 1];
      const big_fvec4 acc = big_tile_f32((nI + 3) >> 2,
          [&](int i) { const int ic = min(i, nI - 1); const float v = pa[(size_t)(ic >> 1) * EA.s_in + (ic & 1) * EA.s_d]; return i < nI ? v : 0.f; },
          [&](int i) { const int ic = min(i, nI - 1); return pe[(size_t)(ic >> 1) * b_pad] * ((ic & 1) ? xm1 : xm0); });
      const float x0 = ext.x_k[(size_t)s_b * kD], x1 = ext.x_k[(size_t)s_b * kD + 1];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int o = 16 * ti + 4 * q + reg;
        if (o < h) {
          ext.Ecur[(size_t)o * b_pad + s_b] = acc[reg];
          ext.Pk[(size_t)(2 * o) * b_pad + s_b] = acc[reg] * x0;
          ext.Pk[(size_t)(2 * o + 1) * b_pad + s_b] = acc[reg] * x1;
        }
      }
    }
  } else {
    // ---- NL[h'][(dk, s, l)] = sum_a Nh[a, h'] lab(a, dk, s, l)   and   PR[(s, dk1)][f] = sum_c pl(s, dk1, c) Ng[c, f]   (float64) -----
    //      (a missing norm environment is the identity: fed as a unit operand, the product is then an exact copy)
    const int D = kD, h = p.h, g = p.g, sb = p.s, L = p.L;
    const int J = D * sb * L, tnl_c = (J + 15) >> 4, tnl = ((h + 15) >> 4) * tnl_c;
    const int I2 = sb * D, tpr_c = (g + 15) >> 4, tpr = ((I2 + 15) >> 4) * tpr_c;
    for (int t = (bid - nc1) * (kBT / 64) + wave; t < tnl + tpr; t += ((int)gridDim.x - nc1) * (kBT / 64)) {
      if (t < tnl) {
        const int ti = t / tnl_c, tj = t - ti * tnl_c;
        const int o_a = min(16 * ti + r, h - 1), j_b = min(16 * tj + r, J - 1);
        const int l = j_b % L, rest = j_b / L, s_ = rest % sb, dk = rest / sb;
        const float *pb = p.lab.base + dk * p.lab.s_d + s_ * p.lab.s_out + l;
        const big_dvec4 acc = big_tile_f64((h + 3) >> 2,
            [&](int a) { const int ac = min(a, h - 1); const double v = p.Nh ? p.Nh[(size_t)ac * h + o_a] : (ac == o_a ? 1.0 : 0.0); return a < h ? v : 0.0; },
            [&](int a) { return (double)pb[(size_t)min(a, h - 1) * p.lab.s_in]; });
        const int j = 16 * tj + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int o = 16 * ti + q + 4 * reg;
          if (o < h && j < J) NL[(size_t)o * J + j] = acc[reg];
        }
      } else {
        const int tt = t - tnl, ti = tt / tpr_c, tj = tt - ti * tpr_c;
        const int i_a = min(16 * ti + r, I2 - 1), f_b = min(16 * tj + r, g - 1);
        const float *pa = p.pl.base + (i_a / D) * p.pl.s_in + (i_a % D) * p.pl.s_d;
        const big_dvec4 acc = big_tile_f64((g + 3) >> 2,
            [&](int c) { const double v = (double)pa[(size_t)min(c, g - 1) * p.pl.s_out]; return c < g ? v : 0.0; },
            [&](int c) { const int cc = min(c, g - 1); return p.Ng ? p.Ng[(size_t)cc * g + f_b] : (cc == f_b ? 1.0 : 0.0); });
        const int f = 16 * tj + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = 16 * ti + q + 4 * reg;
          if (i < I2 && f < g) PR[(size_t)i * g + f] = acc[reg];
        }
      }
    }
  }
}

__global__ __launch_bounds__(kBT) void big_nlpr_kernel(NarrowParams p, double *__restrict__ NL, double *__restrict__ PR) {
  big_nlpr_body(p, NL, PR, blockIdx.x, gridDim.x);
}

bool launch_big_ext(const float *Eprev, const float *x_km1, const float *x_k, const CoreView &A, int b_pad, float *Ecur, float *Pk,
                    hipStream_t st) {
  const size_t lds = (size_t)A.n_in * kD * ((A.n_out + 3) & ~3) * sizeof(float);
  if (A.n_out < 1 || lds > 160 * 1024 || b_pad % 64) {
    snprintf(g_big_err, sizeof g_big_err, "big_ext_kernel: core %d x %d x %d, b_pad %d", A.n_in, kD, A.n_out, b_pad);
    return false;
  }
  BigExtArgs a{Eprev, x_km1, x_k, A, b_pad, Ecur, Pk};
  hipLaunchKernelGGL(big_ext_kernel, dim3(b_pad / 64, (A.n_out + 15) / 16), dim3(256), lds, st, a);
  return true;
}
bool launch_big_signal(unsigned *flag, unsigned value, hipStream_t st) {
  hipLaunchKernelGGL(big_signal_kernel, dim3(1), dim3(64), 0, st, flag, value);
  return true;
}
bool launch_big_gate(const unsigned *flag, unsigned want, int *status, hipStream_t st) {
  hipLaunchKernelGGL(big_gate_kernel, dim3(1), dim3(64), 0, st, flag, want, status);
  return true;
}
bool launch_big_contract(const float *Zred, const CoreView &A, int ncols, float *red, hipStream_t st) {
  const size_t lds = (size_t)A.n_in * kD * 8 * sizeof(float);
  if (A.n_out < 1 || lds > 160 * 1024 || ncols < 1) {
    snprintf(g_big_err, sizeof g_big_err, "big_contract_kernel: core %d x %d x %d, %d columns", A.n_in, kD, A.n_out, ncols);
    return false;
  }
  hipLaunchKernelGGL(big_contract_kernel, dim3((ncols + 63) / 64, (A.n_out + 7) / 8), dim3(64), lds, st, Zred, A, ncols, red);
  return true;
}

// The part of the chain that needs neither the batch-summed gradient nor anything else the batch kernel of this step
// produces: the merged tensor and T = Nh^T . B.  The host may enqueue it on a second stream beside the batch kernel
// (`prep_only`), and then asks the chain proper to skip it (`skip_prep`).
bool launch_narrow_big(const NarrowParams &p, const BigScratch &s, hipStream_t st, bool check, bool prep_only, bool skip_prep,
                       hipEvent_t after_update, const BigFront *front, unsigned *sig_flag, unsigned sig_val) {
  const int D = kD, Bs = p.bsize;
  const int r = D * p.h, c = D * p.g * p.L;
  const bool short_rows = r <= c;
  const int n = short_rows ? r : c, len = short_rows ? c : r;
  const int si = short_rows ? c : 1, sx = short_rows ? 1 : c;
  if (n < 2 || n > kBigMaxN || (n & 1) || p.m < 1 || p.m > n || Bs < 1) {
    snprintf(g_big_err, sizeof g_big_err, "large-tensor path: unsupported matrix %d x %d, kept rank %d", r, c, p.m);
    return false;
  }
  const int nbe = std::min((Bs + kBT - 1) / kBT, 2048);          // one element per thread
  const int nb = std::min((Bs + kBT - 1) / kBT, kBigParts);      // kernels that leave block partials
  // ... one 16 x 16 tile of [D g] x [h D L] per wave, four waves per workgroup
  const int nbm = std::min((((D * p.g + 15) / 16) * ((p.h * D * p.L + 15) / 16) + kBT / 64 - 1) / (kBT / 64), kBigParts);
  const float *Bf = p.Bdirect;
#define BIG(kern, grid, block, lds, ...)                                          \
  do {                                                                            \
    const dim3 g_ = (grid), b_ = (block);                                         \
    const size_t l_ = (lds);                                                      \
    if (!big_launch_ok(#kern, g_, b_, l_)) return false;                          \
    hipLaunchKernelGGL(kern, g_, b_, l_, st, __VA_ARGS__);                        \
    if (!big_launch_done(#kern, check, g_, b_, l_)) return false;                 \
  } while (0)
  // Steps that form their merged tensor here, on this stream, take it and the L2 term straight from the two cores ("factored");
  // a given merged tensor and a preparation that ran ahead on the side stream keep T = Nh^T . B.
  const bool factored = !Bf && !skip_prep && !prep_only;
  double *ws = p.dbg;                          // the capture block doubles as the workspace of this path
  double *NL = s.T, *PR = s.T + (size_t)p.h * D * p.s * p.L;
  if (front && !factored) {                    // no products to form beside it: the contraction alone
    if (front->poll_flag && (!front->wait_ev || hipStreamWaitEvent(st, front->wait_ev, 0) != hipSuccess)) {
      snprintf(g_big_err, sizeof g_big_err, "large-tensor path: cannot join the side stream in front of the contraction");
      return false;
    }
    if (!launch_big_contract(front->Z, front->A, front->ncols, front->red, st)) return false;
    front = nullptr;
  }
  if (factored) {
    const int nlpr = p.l2_flag ? std::min((p.h * D * p.s * p.L + p.s * D * p.g + kBT - 1) / kBT, 2048) : 0;
    if (front) {
      const int wpb = kBT / 64;                                      // one 16 x 16 output tile per wave
      if (front->A.n_out < 1 || front->ncols < 1) {
        snprintf(g_big_err, sizeof g_big_err, "big_front_kernel: core %d x %d x %d, %d columns", front->A.n_in, kD, front->A.n_out, front->ncols);
        return false;
      }
      BigFrontTiles ft{};
      ft.contract_blocks = (((front->A.n_out + 15) / 16) * ((front->ncols + 15) / 16) + wpb - 1) / wpb;
      BigExtArgs ext{};
      if (front->ext_Ecur) {
        const CoreView &EA = front->ext_A;
        if (EA.n_out < 1 || front->b_pad % 64) {
          snprintf(g_big_err, sizeof g_big_err, "big_front_kernel: extension core %d x %d x %d, b_pad %d", EA.n_in, kD, EA.n_out, front->b_pad);
          return false;
        }
        ext = BigExtArgs{front->ext_Eprev, front->ext_x_km1, front->ext_x_k, EA, front->b_pad, front->ext_Ecur, front->ext_Pk};
        ft.ext_blocks = (((EA.n_out + 15) / 16) * (front->b_pad / 16) + wpb - 1) / wpb;
      }
      const int tl2 = p.l2_flag ? ((p.h + 15) / 16) * ((D * p.s * p.L + 15) / 16) + ((p.s * D + 15) / 16) * ((p.g + 15) / 16) : 0;
      const int l2_blocks = std::min((tl2 + wpb - 1) / wpb, 1024);
      BIG(big_front_kernel, dim3(ft.contract_blocks + ft.ext_blocks + l2_blocks), dim3(kBT), 0, p, NL, PR, front->Z, front->A, front->ncols,
          front->red, ft, ext, (const unsigned *)front->poll_flag, front->poll_want, front->ext_acquire ? 1 : 0);
    } else if (nlpr) BIG(big_nlpr_kernel, dim3(nlpr), dim3(kBT), 0, p, NL, PR);
    BIG(big_merge_wd_mfma_kernel, dim3(nbm), dim3(kBT), 0, p, s.Bf, (const double *)NL, (const double *)PR, ws, s.part);
  } else {
    if (!Bf) {
      if (!skip_prep) BIG(big_merge_kernel, dim3(nbe), dim3(kBT), 0, p, s.Bf);
      Bf = s.Bf;
    }
    if (p.l2_flag && !skip_prep) BIG(big_l2_T_kernel, dim3(nbe), dim3(kBT), 0, p, Bf, s.T);
    if (prep_only) return true;
    BIG(big_wd_kernel, dim3(nb), dim3(kBT), 0, p, Bf, (const double *)s.T, ws, s.part);
  }
  BIG(big_update_kernel, dim3(nb), dim3(kBT), 0, p, ws, (const double *)s.part, factored ? nbm : nb);
  // B_new is complete: the batch kernel of the NEXT step may start beside the SVD of this one (pipelined large-tensor step)
  if (!sig_flag && after_update && hipEventRecord(after_update, st) != hipSuccess) {
    snprintf(g_big_err, sizeof g_big_err, "hipEventRecord behind big_update_kernel failed");
    return false;
  }
  if (p.stop_after_update) return true;
  const int nt = (n + 15) / 16;
  BIG(big_gram_kernel, dim3(nt, nt, kGramKS), dim3(256), 0, (const float *)p.Bnew, n, len, si, sx, s.gram, sig_flag, sig_val);
  BigJacobiArgs a{};
  a.G = s.gram; a.n = n; a.m = p.m; a.stop2 = p.svd_stop2; a.rotlog = s.rotlog; a.lam = s.lam; a.info = s.info;
  a.counters = p.counters; a.status = p.status;
  a.Bn = p.Bnew; a.len = len; a.si = si; a.sx = sx; a.VW = s.VW; a.prog = s.prog; a.token = p.token & 0xfffffu;
  a.keep_max = p.m; a.trunc_thr = p.trunc_thr; a.m_out = p.m_out; a.metrics = p.metrics; a.red = p.red; a.Bs = Bs; a.L = p.L; a.ws = ws;
  // workgroup 0 iterates; workgroups 1.. replay its rotation log on the n unit vectors and the len columns of W while it does
  // (they need no LDS but share the launch's request: all of them resident beside the batch kernel of the next step, whose 157
  // workgroups leave 99 CUs)
  BIG(big_jacobi_kernel, dim3(1 + (n + len + 15) / 16), dim3(1024), big_jacobi_lds_bytes(n), a);
  const int ncores = std::min(((n + len) * p.m + kBT - 1) / kBT, 1024);
  if (p.Nh_new) {
    const int nb2 = std::min((16 * p.h * D * p.m + kBT - 1) / kBT, 1024);
    BIG(big_cores_normT_kernel, dim3(nb2 + ncores), dim3(kBT), 0, p, (const double *)s.lam, (const int *)s.info, (const double *)s.VW, s.Cb,
        s.T2, nb2);
    BIG(big_norm_out_kernel, dim3(std::min((16 * p.m * p.m + kBT - 1) / kBT, 1024)), dim3(kBT), 0, p, (const float *)s.Cb,
        (const double *)s.T2, (const int *)(s.info + 3));
  } else {
    BIG(big_cores_kernel, dim3(ncores), dim3(kBT), 0, p, (const double *)s.lam, (const int *)s.info, (const double *)s.VW, s.Cb);
  }
#undef BIG
  return true;
}

}  // namespace tnml
