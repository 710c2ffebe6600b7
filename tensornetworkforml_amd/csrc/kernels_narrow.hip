// Single-workgroup kernels of the MPS sweep: everything that touches only the (small) merged
// two-site tensor.  One launch per sweep step does
//   B = A_k . A_{k+1}                                   (Network_class.py:484)
//   dB = dB_raw - 2 wd Ln.B.Rn   (or - wd B)            (:728-734, compute_L2_reg :966-1179)
//   clip by the sum|.| ratio, B_new = B + lr dB         (:755-761)
//   truncated SVD of the matricised B_new, sqrt(S) on both factors   (:528-563, :839-962)
//   the two new cores, the behind norm environment of the next step, (accuracy, MAE)
// all in LDS.  The update and the SVD run in float64: the norm environments of a 784-site chain
// reach 1e196, and the SVD goes through the Gram matrix, whose float64 accumulation
// keeps the squared condition number harmless for float32 data.
//
// SVD method: G = W^T W (n x n, n = min(rows, cols) <= 64) in float64, then one-sided (Hestenes)
// Jacobi on the columns of [G; I] with a round-robin pair schedule: n/2 pairs rotate concurrently,
// 32 lanes per pair, one barrier per round.  Columns of the bottom half converge to the
// eigenvectors q_j of G, top-half column norms to the eigenvalues sigma_j^2.  The short-side factor
// is q_j sqrt(sigma_j), the long-side one W q_j / sqrt(sigma_j), so that their product is the
// projection W Q Q^T whatever the accuracy of the small sigma_j.
#include "tnml_internal.h"

namespace tnml {

__device__ inline double group32_sum(double v) {
  // all-reduce inside an aligned group of 32 lanes
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// block-wide sum of up to 3 doubles; result valid in every thread.  scratch: >= 3*16 doubles.
__device__ inline void block_sum3(double &a, double &b, double &c, double *scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
    c += __shfl_xor(c, off);
  }
  __syncthreads();
  if (lane == 0) { scratch[wave] = a; scratch[16 + wave] = b; scratch[32 + wave] = c; }
  __syncthreads();
  a = 0; b = 0; c = 0;
  for (int w = 0; w < nw; ++w) { a += scratch[w]; b += scratch[16 + w]; c += scratch[32 + w]; }
  __syncthreads();
}

struct NarrowCarve {
  double *dT, *dG, *Z, *dNh, *dNg, *dLam, *dRed, *dT2;
  float *fB, *sLab, *sPl, *sCb;
  int *sOrd, *sFlag;
  size_t bytes;
};

__host__ __device__ inline NarrowCarve narrow_carve(unsigned char *base, int h, int g, int s, int L, int m) {
  const int D = kD;
  const size_t Bs = (size_t)h * D * D * g * L;
  const int r = D * h, c = D * g * L;
  const int n = r <= c ? r : c, ne = n + (n & 1);
  size_t zreg = 2 * Bs;
  if ((size_t)2 * n * ne > zreg) zreg = (size_t)2 * n * ne;
  NarrowCarve k;
  double *d = (double *)base;
  k.dT = d; k.dG = d + Bs; k.Z = d; d += zreg;
  k.dNh = d; d += (size_t)h * h;
  k.dNg = d; d += (size_t)g * g;
  k.dLam = d; d += ne;
  k.dRed = d; d += 64;
  k.dT2 = d; d += (size_t)h * D * m;
  float *f = (float *)d;
  k.fB = f; f += Bs;
  k.sLab = f; f += (size_t)h * D * s * L;
  k.sPl = f; f += (size_t)s * D * g;
  k.sCb = f; f += (size_t)r * m;
  int *ip = (int *)f;
  k.sOrd = ip; ip += ne;
  k.sFlag = ip; ip += 4;
  k.bytes = (size_t)((unsigned char *)ip - base);
  return k;
}

size_t narrow_lds_bytes(int h, int g, int s, int L, int m) {
  return narrow_carve(nullptr, h, g, s, L, m).bytes + 16;
}

constexpr double kJacobiTol = 1e-11;
constexpr int kJacobiMaxSweeps = 30;

__global__ __launch_bounds__(kNarrowThreads) void narrow_step_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const NarrowCarve k = narrow_carve(smem_raw, p.h, p.g, p.s, p.L, p.m);
  const int tid = threadIdx.x, NT = kNarrowThreads;
  const int D = kD, h = p.h, g = p.g, s = p.s, L = p.L, m = p.m, Bs = p.bsize;
  const int r = D * h, c = D * g * L;
  const bool short_rows = (r <= c);
  const int n = short_rows ? r : c, ne = n + (n & 1), len = short_rows ? c : r;
  const int twoN = 2 * n;

  // ---- phase 0: stage the two cores and the norm environments ---------------------------------
  for (int e = tid; e < h * D * s * L; e += NT) {
    const int l = e % L, q = e / L;
    const int s_ = q % s, q2 = q / s;
    const int d = q2 % D, h_ = q2 / D;
    k.sLab[e] = p.lab.base[h_ * p.lab.s_in + d * p.lab.s_d + s_ * p.lab.s_out + l];
  }
  for (int e = tid; e < s * D * g; e += NT) {
    const int g_ = e % g, q = e / g;
    const int d = q % D, s_ = q / D;
    k.sPl[e] = p.pl.base[s_ * p.pl.s_in + d * p.pl.s_d + g_ * p.pl.s_out];
  }
  for (int e = tid; e < h * h; e += NT) k.dNh[e] = p.Nh ? p.Nh[e] : 1.0;
  for (int e = tid; e < g * g; e += NT) k.dNg[e] = p.Ng ? p.Ng[e] : 1.0;
  __syncthreads();

  // ---- phase 1: B[h,dk,dk1,g,l] = sum_s lab(h,dk,s,l) * pl(s,dk1,g) ---------------------------
  const int RW = D * D * g * L;  // elements per behind-bond index
  for (int e = tid; e < Bs; e += NT) {
    const int l = e % L, q = e / L;
    const int g_ = q % g, q2 = q / g;
    const int dk1 = q2 % D, q3 = q2 / D;   // q3 = h_*D + dk
    double acc = 0.0;
    for (int s_ = 0; s_ < s; ++s_)
      acc += (double)k.sLab[(q3 * s + s_) * L + l] * (double)k.sPl[(s_ * D + dk1) * g + g_];
    k.fB[e] = (float)acc;
  }
  __syncthreads();

  // ---- phases 2-3: weight decay term ------------------------------------------------------------
  if (p.l2_flag) {
    for (int e = tid; e < Bs; e += NT) {          // T = Nh^T . B over the behind bond
      const int a_out = e / RW, rest = e % RW;
      double acc = 0.0;
      for (int a = 0; a < h; ++a) acc += k.dNh[a * h + a_out] * (double)k.fB[a * RW + rest];
      k.dT[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < Bs; e += NT) {          // G = T . Ng over the ahead bond
      const int l = e % L, q = e / L;
      const int f_ = q % g, pre = q / g;
      double acc = 0.0;
      for (int cc = 0; cc < g; ++cc) acc += k.dT[(pre * g + cc) * L + l] * k.dNg[cc * g + f_];
      k.dG[e] = acc;
    }
    __syncthreads();
  }
  double sumB = 0.0, sumD = 0.0, l2 = 0.0;
  for (int e = tid; e < Bs; e += NT) {
    const double bv = (double)k.fB[e];
    const double raw = (double)p.red[e];
    double wdterm;
    if (p.l2_flag) {
      const double gv = k.dG[e];
      l2 += bv * gv;
      wdterm = 2.0 * (double)p.wd * gv;
    } else {
      wdterm = (double)p.wd * bv;
    }
    const double dv = raw - wdterm;
    if (p.dbg) {
      p.dbg[e] = bv;
      p.dbg[Bs + e] = raw;
      p.dbg[3 * (size_t)Bs + e] = wdterm;
    }
    k.dG[e] = dv;
    sumB += fabs(bv);
    sumD += fabs(dv);
  }
  block_sum3(sumB, sumD, l2, k.dRed);

  // ---- phase 5: clip + update (Network_class.py:755-761) ----------------------------------------
  double factor = (double)p.lr;
  if (sumD > sumB) factor = (double)p.lr * (sumB / sumD);
  const bool bad = !isfinite(sumD) || !isfinite(sumB);
  for (int e = tid; e < Bs; e += NT) {
    const float v = (float)((double)k.fB[e] + factor * k.dG[e]);
    k.fB[e] = v;
    p.Bnew[e] = v;
    if (p.dbg) p.dbg[2 * (size_t)Bs + e] = (double)v;
  }
  if (tid == 0) {
    if (bad) atomicOr(p.status, 1);
    if (p.dbg) {
      double *sc = p.dbg + 4 * (size_t)Bs + 64;
      sc[0] = (double)p.wd * l2;
      sc[1] = sumB;
      sc[2] = sumD;
    }
  }
  __syncthreads();   // dT/dG are dead from here on; Z aliases them

  // ---- phase 6: Gram matrix in float64 -> top half of Z, identity -> bottom half ----------------
  //   W(x, kk) = short_rows ? Bm[kk][x] : Bm[x][kk]   with Bm = B_new as (r x c) row-major
  for (int e = tid; e < n * n; e += NT) {
    const int col = e / n, kk = e % n;
    double acc = 0.0;
    if (short_rows) {
      const float *ra = k.fB + (size_t)kk * c, *rb = k.fB + (size_t)col * c;
      for (int x = 0; x < len; ++x) acc += (double)ra[x] * (double)rb[x];
    } else {
      for (int x = 0; x < len; ++x) acc += (double)k.fB[(size_t)x * c + kk] * (double)k.fB[(size_t)x * c + col];
    }
    k.Z[(size_t)col * twoN + kk] = acc;
    k.Z[(size_t)col * twoN + n + kk] = (kk == col) ? 1.0 : 0.0;
  }
  __syncthreads();

  // ---- phase 7: one-sided Jacobi, round-robin schedule, 32 lanes per pair -----------------------
  const int grp = tid >> 5, lane32 = tid & 31;
  int sweeps = 0, converged = 0;
  if (n > 1) {
    for (; sweeps < kJacobiMaxSweeps; ++sweeps) {
      if (tid == 0) k.sFlag[0] = 0;
      __syncthreads();
      for (int rnd = 0; rnd < ne - 1; ++rnd) {
        int pc, qc;
        if (grp == 0) { pc = ne - 1; qc = rnd; }
        else { pc = (rnd + grp) % (ne - 1); qc = (rnd - grp + (ne - 1)) % (ne - 1); }
        if (grp < ne / 2 && pc < n && qc < n) {
          double a[4], b[4];
          double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int kk = lane32 + 32 * i;
            a[i] = 0.0; b[i] = 0.0;
            if (kk < twoN) {
              a[i] = k.Z[(size_t)pc * twoN + kk];
              b[i] = k.Z[(size_t)qc * twoN + kk];
              if (kk < n) { al += a[i] * a[i]; be += b[i] * b[i]; ga += a[i] * b[i]; }
            }
          }
          al = group32_sum(al); be = group32_sum(be); ga = group32_sum(ga);
          if (fabs(ga) > kJacobiTol * sqrt(al * be)) {
            const double zeta = (be - al) / (2.0 * ga);
            const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int kk = lane32 + 32 * i;
              if (kk < twoN) {
                k.Z[(size_t)pc * twoN + kk] = cs * a[i] - sn * b[i];
                k.Z[(size_t)qc * twoN + kk] = sn * a[i] + cs * b[i];
              }
            }
            if (lane32 == 0) k.sFlag[0] = 1;
          }
        }
        __syncthreads();
      }
      const int rotated = k.sFlag[0];
      __syncthreads();
      if (!rotated) { converged = 1; ++sweeps; break; }
    }
  } else {
    converged = 1;
  }

  // ---- phase 8: eigenvalues (top-half column norms), descending order ---------------------------
  for (int j = tid; j < n; j += NT) {
    double acc = 0.0;
    for (int kk = 0; kk < n; ++kk) { const double v = k.Z[(size_t)j * twoN + kk]; acc += v * v; }
    k.dLam[j] = sqrt(acc);
  }
  __syncthreads();
  for (int j = tid; j < n; j += NT) {
    const double lj = k.dLam[j];
    int rank = 0;
    for (int i = 0; i < n; ++i) {
      const double li = k.dLam[i];
      rank += (li > lj) || (li == lj && i < j);
    }
    k.sOrd[rank] = j;
    if (p.dbg) p.dbg[4 * (size_t)Bs + rank] = sqrt(lj);
  }
  if (tid == 0) {
    if (!converged) atomicOr(p.status, 2);
    if (p.dbg) {
      double *sc = p.dbg + 4 * (size_t)Bs + 64;
      sc[3] = (double)sweeps;
      sc[4] = (double)n;
    }
  }
  __syncthreads();

  // ---- phase 9: the two new cores -----------------------------------------------------------------
  const double lam_max = k.dLam[k.sOrd[0]];
  // short-side factor: q_j * sigma_j^(1/2)
  for (int e = tid; e < n * m; e += NT) {
    const int sp = e % m, kk = e / m;
    const int j = k.sOrd[sp];
    const double lam = k.dLam[j];
    const double sq = (lam > 1e-300 && lam > 1e-30 * lam_max) ? sqrt(sqrt(lam)) : 0.0;
    const float v = (float)(k.Z[(size_t)j * twoN + n + kk] * sq);
    if (short_rows) {                       // kk = row index i = h_*D + dk  -> behind core
      k.sCb[kk * m + sp] = v;
      p.out_behind[(kk / D) * p.ob_s_h + (kk % D) * p.ob_s_d + sp * p.ob_s_m] = v;
    } else {                                // kk = column index (dk1*g + g_)*L + l -> ahead core
      const int l = kk % L, q = kk / L;
      p.out_ahead[sp * p.oa_s_m + (q / g) * p.oa_s_d + (q % g) * p.oa_s_g + l] = v;
    }
  }
  // long-side factor: (W q_j) / sigma_j^(1/2)
  for (int e = tid; e < len * m; e += NT) {
    const int sp = e % m, x = e / m;
    const int j = k.sOrd[sp];
    const double lam = k.dLam[j];
    const double *q = k.Z + (size_t)j * twoN + n;
    double acc = 0.0;
    if (short_rows) {                       // x = column index, sum over rows
      for (int kk = 0; kk < n; ++kk) acc += (double)k.fB[(size_t)kk * c + x] * q[kk];
    } else {                                // x = row index, sum over columns
      const float *row = k.fB + (size_t)x * c;
      for (int kk = 0; kk < n; ++kk) acc += (double)row[kk] * q[kk];
    }
    const double isq = (lam > 1e-300 && lam > 1e-30 * lam_max) ? 1.0 / sqrt(sqrt(lam)) : 0.0;
    const float v = (float)(acc * isq);
    if (short_rows) {
      const int l = x % L, qq = x / L;
      p.out_ahead[sp * p.oa_s_m + (qq / g) * p.oa_s_d + (qq % g) * p.oa_s_g + l] = v;
    } else {
      k.sCb[x * m + sp] = v;
      p.out_behind[(x / D) * p.ob_s_h + (x % D) * p.ob_s_d + sp * p.ob_s_m] = v;
    }
  }
  __syncthreads();

  // ---- phase 10: behind norm environment of the next step ------------------------------------------
  if (p.Nh_new) {
    for (int e = tid; e < h * D * m; e += NT) {     // T2[(h_,d), s''] = sum_h' Nh[h_,h'] Cb[(h',d), s'']
      const int sp = e % m, q = e / m;
      const int d = q % D, h_ = q / D;
      double acc = 0.0;
      for (int hq = 0; hq < h; ++hq) acc += k.dNh[h_ * h + hq] * (double)k.sCb[(hq * D + d) * m + sp];
      k.dT2[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += NT) {
      const int s2 = e % m, s1 = e / m;
      double acc = 0.0;
      for (int q = 0; q < h * D; ++q) acc += (double)k.sCb[q * m + s1] * k.dT2[q * m + s2];
      p.Nh_new[e] = acc;
    }
  }

  // ---- phase 11: metrics of this step (var_hist, Network_class.py:739-750) --------------------------
  if (tid == 0 && p.metrics) {
    const double cnt = (double)p.red[Bs + 3];
    const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
    p.metrics[0] = (float)((double)p.red[Bs] * inv);
    p.metrics[1] = (float)((double)p.red[Bs + 1] * inv / (double)L);
    if (p.red[Bs + 2] != 0.f) atomicOr(p.status, 1);
  }
}

void launch_narrow(const NarrowParams &p, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(narrow_step_kernel, dim3(1), dim3(kNarrowThreads), lds_bytes, st, p);
}

// ------------------------------------------------------------------------------------------
// Norm-environment chain (the batch-independent part of compute_L2_reg, Network_class.py:1004-1061):
//   env_out[o][o'] = sum_{in,in',d} A(in,d,o) env_in[in][in'] A(in',d,o')
// One workgroup walks the sites sequentially; float64 throughout.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_chain_kernel(const NormChainSite *__restrict__ sites, int n_sites,
                                                         const float *__restrict__ cores,
                                                         double *__restrict__ env_base, int Mmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *env = (double *)smem_raw;                   // [Mmax][Mmax]
  double *T = env + (size_t)Mmax * Mmax;              // [Mmax][D][Mmax]
  float *sA = (float *)(T + (size_t)Mmax * kD * Mmax);  // [Mmax][D][Mmax]
  const int tid = threadIdx.x;
  if (tid == 0) env[0] = 1.0;
  __syncthreads();
  for (int i = 0; i < n_sites; ++i) {
    const NormChainSite cs = sites[i];
    const int ni = cs.n_in, no = cs.n_out;
    for (int e = tid; e < ni * kD * no; e += 256) {
      const int o = e % no, q = e / no;
      sA[e] = cores[cs.core_off + (q / kD) * cs.s_in + (q % kD) * cs.s_d + o * cs.s_out];
    }
    __syncthreads();
    for (int e = tid; e < ni * kD * no; e += 256) {   // T[in][d][o'] = sum_in' env[in][in'] A[in'][d][o']
      const int o = e % no, q = e / no;
      const int d = q % kD, in = q / kD;
      double acc = 0.0;
      for (int j = 0; j < ni; ++j) acc += env[in * ni + j] * (double)sA[(j * kD + d) * no + o];
      T[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < no * no; e += 256) {        // env_out[o][o'] = sum_{in,d} A[in][d][o] T[in][d][o']
      const int o2 = e % no, o1 = e / no;
      double acc = 0.0;
      for (int q = 0; q < ni * kD; ++q) acc += (double)sA[q * no + o1] * T[q * no + o2];
      env_base[cs.env_out_off + e] = acc;
      env[e] = acc;                                   // nobody reads env in this phase
    }
    __syncthreads();
  }
}

void launch_norm_chain(const NormChainSite *sites_dev, int n_sites, const float *cores, double *env_base,
                       int Mmax, hipStream_t st) {
  size_t lds = ((size_t)Mmax * Mmax + (size_t)Mmax * kD * Mmax) * sizeof(double) +
               (size_t)Mmax * kD * Mmax * sizeof(float);
  hipLaunchKernelGGL(norm_chain_kernel, dim3(1), dim3(256), lds, st, sites_dev, n_sites, cores, env_base, Mmax);
}

}  // namespace tnml
