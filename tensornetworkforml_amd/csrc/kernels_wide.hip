// Batch-parallel kernels of the MPS sweep: input re-tiling, the forward environment chain,
// the per-step wide kernel (f from the previous B, activation / loss derivative / metrics,
// environment extension, partial bond gradient) and the slab reduction.
//
// v1 = plain FMA + LDS formulation, written for correctness and coalesced HBM access; the MFMA
// formulation of the two GEMM-shaped parts replaces it where profiling says so.
//
// Layouts: features x[site][b_pad][D]; environments env[slot][m][b_pad] (bond-major, the batch is
// the contiguous axis, so a wave reads 64 consecutive samples of one bond index); f [L][b_pad].
#include "tnml_internal.h"

namespace tnml {

// ------------------------------------------------------------------------------------------
// X [b][N][D] -> x [N][b_pad][D]   (D == 2: one float2 per (sample, site))
// ------------------------------------------------------------------------------------------
__global__ void transpose_input_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int b,
                                       int b_pad, int N) {
  __shared__ float2 tile[32][33];
  const int s0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r, n = n0 + tx;
    float2 v = make_float2(0.f, 0.f);
    if (s < b && n < N) v = in[(size_t)s * N + n];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, s = s0 + tx;
    if (n < N && s < b_pad) out[(size_t)n * b_pad + s] = tile[tx][r];
  }
}

void launch_transpose_input(const float *X_bnd, float *X_nbd, int b, int b_pad, int N, hipStream_t st) {
  dim3 grid((b_pad + 31) / 32, (N + 31) / 32);
  hipLaunchKernelGGL(transpose_input_kernel, grid, dim3(256), 0, st, (const float2 *)X_bnd, (float2 *)X_nbd,
                     b, b_pad, N);
}

// ------------------------------------------------------------------------------------------
// Forward environment chain (Network.forward, Network_class.py:227-255):
//   env_out[s][o] = sum_{in,d} env_in[s][in] * x[s][d] * A(in, d, o)        for every site of the chain
// One workgroup owns 16 samples and walks all sites; the core of the current site is staged in
// LDS, the running environment ping-pongs between two LDS buffers and is streamed to HBM once.
// ------------------------------------------------------------------------------------------
constexpr int kChainTS = 16;
constexpr int kChainThreads = 512;

// LOGMODE (calibration, Network_class.py:168-170): the running environment of every sample is
// renormalised by its max |.| after each site and the logs are accumulated, so that chains whose
// output under- or overflows float32 (1e-66 for the un-calibrated 784-site chain) still yield
// log max|f| exactly.  Nothing is written to the environment stack in this mode; the per-workgroup
// maxima of log|f| go to logmax_out[blockIdx.x].
template <bool LOGMODE>
__global__ __launch_bounds__(kChainThreads) void env_chain_kernel(
    const ChainSite *__restrict__ sites, int n_sites, const float *__restrict__ cores,
    const float *__restrict__ labcore, const float *__restrict__ X, float *__restrict__ env_base,
    float *__restrict__ f, int b, int b_pad, int L, int Mmax, float *__restrict__ logmax_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int mo = Mmax > L ? Mmax : L;
  float *sA = smem;                               // [n_in][D][n_out]
  float *sE0 = sA + (size_t)Mmax * kD * mo;       // [Mmax][TS]
  float *sE1 = sE0 + (size_t)mo * kChainTS;
  float *sX = sE1 + (size_t)mo * kChainTS;        // [TS][D]
  float *sScale = sX + kChainTS * kD;             // [TS] accumulated log scale   (LOGMODE)
  unsigned *sMax = (unsigned *)(sScale + kChainTS);  // [TS] max |v| of the site, as float bits
  const int tid = threadIdx.x;
  const int sl = tid % kChainTS, og = tid / kChainTS;
  constexpr int OG = kChainThreads / kChainTS;
  const int s = blockIdx.x * kChainTS + sl;

  float *cur = sE0, *nxt = sE1;
  if (tid < kChainTS) { cur[tid] = 1.0f; sScale[tid] = 0.f; sMax[tid] = 0u; }
  for (int i = 0; i < n_sites; ++i) {
    const ChainSite cs = sites[i];
    const float *src = (cs.is_label ? labcore : cores) + cs.core_off;
    const int na = cs.n_in * kD * cs.n_out;
    for (int e = tid; e < na; e += kChainThreads) {
      const int o = e % cs.n_out, r = e / cs.n_out;
      const int d = r % kD, in = r / kD;
      sA[e] = src[in * cs.s_in + d * cs.s_d + o * cs.s_out];
    }
    if (tid < kChainTS * kD) {
      const int ss = blockIdx.x * kChainTS + tid / kD;
      sX[tid] = X[((size_t)cs.x_site * b_pad + ss) * kD + (tid % kD)];
    }
    __syncthreads();
    const float x0 = sX[sl * kD], x1 = sX[sl * kD + 1];
    for (int o = og; o < cs.n_out; o += OG) {
      float a0 = 0.f, a1 = 0.f;
      for (int in = 0; in < cs.n_in; ++in) {
        const float e = cur[in * kChainTS + sl];
        a0 = fmaf(e, sA[(in * kD) * cs.n_out + o], a0);
        a1 = fmaf(e, sA[(in * kD + 1) * cs.n_out + o], a1);
      }
      const float v = x0 * a0 + x1 * a1;
      nxt[o * kChainTS + sl] = v;
      if (LOGMODE) {
        atomicMax(&sMax[sl], __float_as_uint(fabsf(v)));   // non-negative floats order like their bits
      } else if (cs.env_out_off >= 0) {
        env_base[cs.env_out_off + (size_t)o * b_pad + s] = v;
      } else {
        f[(size_t)o * b_pad + s] = v;
      }
    }
    __syncthreads();
    if (LOGMODE) {
      const float mx = __uint_as_float(sMax[sl]);
      const float inv = (mx > 0.f && isfinite(mx)) ? 1.0f / mx : 1.0f;
      for (int o = og; o < cs.n_out; o += OG) nxt[o * kChainTS + sl] *= inv;
      __syncthreads();
      if (tid < kChainTS) {
        const float m2 = __uint_as_float(sMax[tid]);
        sScale[tid] += (m2 > 0.f && isfinite(m2)) ? logf(m2) : (m2 > 0.f ? INFINITY : -INFINITY);
        sMax[tid] = 0u;
      }
      __syncthreads();
    }
    float *t = cur; cur = nxt; nxt = t;
  }
  if (LOGMODE) {
    // after the label site the renormalised max |f| of every sample is 1: log max|f| = sScale
    if (tid < kChainTS) {
      float v = (blockIdx.x * kChainTS + tid < b) ? sScale[tid] : -INFINITY;
      for (int off = kChainTS / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
      if (tid == 0) logmax_out[blockIdx.x] = v;
    }
  }
}

void launch_env_chain(const ChainSite *sites_dev, int n_sites, const float *cores, const float *labcore,
                      const float *X, float *env_base, float *f, int b, int b_pad, int L, int Mmax,
                      float *logmax_out, hipStream_t st) {
  const int mo = Mmax > L ? Mmax : L;
  size_t lds = ((size_t)Mmax * kD * mo + 2 * (size_t)mo * kChainTS + kChainTS * kD + 2 * kChainTS) * sizeof(float);
  if (logmax_out)
    hipLaunchKernelGGL(env_chain_kernel<true>, dim3(b_pad / kChainTS), dim3(kChainThreads), lds, st, sites_dev,
                       n_sites, cores, labcore, X, env_base, f, b, b_pad, L, Mmax, logmax_out);
  else
    hipLaunchKernelGGL(env_chain_kernel<false>, dim3(b_pad / kChainTS), dim3(kChainThreads), lds, st, sites_dev,
                       n_sites, cores, labcore, X, env_base, f, b, b_pad, L, Mmax, logmax_out);
}

// ------------------------------------------------------------------------------------------
// activation + loss derivative of one sample (Network_class.py:767-835).  fa and g are written in
// place over L values held in LDS at stride `st`.
// ------------------------------------------------------------------------------------------
__device__ inline void act_and_lossder(const float *fin, int st_in, float *fa, float *g, int st, int L,
                                       int y, int act_fn, int loss_fn, float T, float &sumabs,
                                       int &correct, int &nonfinite) {
  // bit 8 of act_fn: the input already went through the activation (compute_loss_derivate's
  // argument); the low bits still select the cross-entropy formula (Network_class.py:826-830)
  const bool pre_activated = (act_fn & 0x100) != 0;
  act_fn &= 0xff;
  // activation
  if (pre_activated) {
    for (int l = 0; l < L; ++l) fa[l * st] = fin[l * st_in];
  } else if (act_fn == TNML_ACT_SOFTMAX) {
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l) mx = fmaxf(mx, fin[l * st_in]);
    float sum = 0.f;
    for (int l = 0; l < L; ++l) {
      const float e = __expf((fin[l * st_in] - mx) / T);
      fa[l * st] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int l = 0; l < L; ++l) fa[l * st] *= inv;
  } else if (act_fn == TNML_ACT_SIGMOID) {
    for (int l = 0; l < L; ++l) fa[l * st] = 1.0f / (1.0f + __expf(-fin[l * st_in] / T));
  } else {
    for (int l = 0; l < L; ++l) fa[l * st] = fin[l * st_in];
  }
  // metrics: argmax (first maximum, as np.argmax) and sum |y - fa|.  All three activations are
  // monotonic, so the argmax is taken on f itself: identical in exact arithmetic, and immune to the
  // float32 saturation of sigmoid/softmax that would create ties the float64 reference does not see.
  int am = 0;
  float best = fin[0];
  float sa = 0.f;
  for (int l = 0; l < L; ++l) {
    const float v = fa[l * st];
    const float fv = fin[l * st_in];
    if (fv > best) { best = fv; am = l; }
    sa += fabsf((l == y ? 1.0f : 0.0f) - v);
    if (!isfinite(v)) nonfinite = 1;
  }
  sumabs = sa;
  correct = (am == y) ? 1 : 0;
  // loss derivative
  for (int l = 0; l < L; ++l) {
    const float v = fa[l * st];
    const float yy = (l == y) ? 1.0f : 0.0f;
    float d;
    if (loss_fn == TNML_LOSS_MSE) {
      d = yy - v;
    } else if (loss_fn == TNML_LOSS_CROSS_ENTROPY) {
      d = (act_fn == TNML_ACT_SOFTMAX) ? (yy - yy * v) / T : yy / v;
    } else {
      d = 1.0f / ((l == y ? v : v - 1.0f) + 1e-4f);
    }
    g[l * st] = d;
  }
}

// ------------------------------------------------------------------------------------------
// f[l][s] = sum H'[s,h'] x[s,d] x'[s,d'] G'[s,g'] Bprev[h',d,d',g',l]   (Network_class.py:494-523)
// thread (jj, s): rows j = (h', d) strided by the 8 row groups; partial sums meet in LDS.
// sHp [hp][TS], sGp [gp][TS], sXa/sXb [TS][D] are staged by the caller.
// ------------------------------------------------------------------------------------------
__device__ inline void f_from_B(const WideParams &p, const float *sHp, const float *sGp, const float *sXa,
                                const float *sXb, float *sRed /*[8][TS]*/, float *sF /*[L][TS]*/) {
  const int tid = threadIdx.x;
  const int sl = tid % kTS, jj = tid / kTS;
  constexpr int JG = kWideThreads / kTS;  // 8
  const int rows = p.hp * kD;
  const int inner = kD * p.gp;            // (d', g')
  const float xb0 = sXb[sl * kD], xb1 = sXb[sl * kD + 1];
  for (int l = 0; l < p.L; ++l) {
    float acc = 0.f;
    for (int j = jj; j < rows; j += JG) {
      const int hq = j / kD, d = j % kD;
      const float w = sHp[hq * kTS + sl] * sXa[sl * kD + d];
      const float *brow = p.Bprev + (size_t)j * inner * p.L + l;
      float u = 0.f;
      for (int q = 0; q < p.gp; ++q) {
        const float gq = sGp[q * kTS + sl];
        u = fmaf(xb0 * gq, brow[(size_t)q * p.L], u);
        u = fmaf(xb1 * gq, brow[(size_t)(p.gp + q) * p.L], u);
      }
      acc = fmaf(w, u, acc);
    }
    sRed[jj * kTS + sl] = acc;
    __syncthreads();
    if (tid < kTS) {
      float t = 0.f;
      for (int k = 0; k < JG; ++k) t += sRed[k * kTS + tid];
      sF[l * kTS + tid] = t;
    }
    __syncthreads();
  }
}

// LDS carve shared by wide_step_kernel and f_only_kernel
struct WideSmem {
  float *sHp, *sGp, *sG, *sH, *sX, *sF, *sGl, *sRed, *sA, *sP, *sQ;
};
__device__ inline WideSmem carve(float *smem, int hmax, int gmax, int L, int core_elems) {
  WideSmem w;
  float *q = smem;
  w.sHp = q; q += hmax * kTS;
  w.sGp = q; q += gmax * kTS;
  w.sG = q; q += gmax * kTS;
  w.sH = q; q += hmax * kTS;
  w.sX = q; q += 3 * kTS * kD;
  w.sF = q; q += L * kTS;
  w.sGl = q; q += L * kTS;
  w.sRed = q; q += (kWideThreads / kTS) * kTS;
  w.sA = q; q += core_elems;
  w.sP = q; q += kTS * hmax * kD;
  w.sQ = q;  // kTS * gmax * kD
  return w;
}
static size_t wide_lds_bytes(int hmax, int gmax, int L, int core_elems) {
  size_t n = (size_t)2 * hmax * kTS + (size_t)2 * gmax * kTS + 3 * kTS * kD + (size_t)2 * L * kTS +
             kWideThreads + core_elems + (size_t)kTS * hmax * kD + (size_t)kTS * gmax * kD;
  return n * sizeof(float);
}

// ------------------------------------------------------------------------------------------
// The wide step kernel: one workgroup = kTS samples.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWideThreads) void wide_step_kernel(WideParams p, int hmax, int gmax,
                                                                int core_elems) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideSmem w = carve(smem, hmax, gmax, p.L, core_elems);
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * kTS;

  // ---- stage per-sample operands (coalesced: 32 consecutive samples per bond index) ----------
  if (p.do_f || (p.do_ext && !p.first_ext))
    for (int e = tid; e < p.hp * kTS; e += kWideThreads)
      w.sHp[e] = p.Hprev ? p.Hprev[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  if (p.do_f)
    for (int e = tid; e < p.gp * kTS; e += kWideThreads)
      w.sGp[e] = p.Gprev ? p.Gprev[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  for (int e = tid; e < p.g * kTS; e += kWideThreads)
    w.sG[e] = p.Gcur ? p.Gcur[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  for (int e = tid; e < 3 * kTS * kD; e += kWideThreads) {
    const int which = e / (kTS * kD), r = e % (kTS * kD);
    const float *src = which == 0 ? p.x_km1 : (which == 1 ? p.x_k : p.x_kp1);
    w.sX[e] = src ? src[(size_t)s0 * kD + r] : 0.f;
  }
  if (p.do_ext) {
    const int na = p.ext_core.n_in * kD * p.ext_core.n_out;
    for (int e = tid; e < na; e += kWideThreads) {
      const int o = e % p.ext_core.n_out, r = e / p.ext_core.n_out;
      w.sA[e] = p.ext_core.base[(r / kD) * p.ext_core.s_in + (r % kD) * p.ext_core.s_d + o * p.ext_core.s_out];
    }
  }
  __syncthreads();
  const float *sXm = w.sX, *sXk = w.sX + kTS * kD, *sXp = w.sX + 2 * kTS * kD;

  // ---- f of the previous step from its updated, un-truncated B --------------------------------
  if (p.do_f) {
    f_from_B(p, w.sHp, w.sGp, sXm, sXk, w.sRed, w.sF);
    for (int e = tid; e < p.L * kTS; e += kWideThreads)
      p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] = w.sF[e];
  } else {
    for (int e = tid; e < p.L * kTS; e += kWideThreads)
      w.sF[e] = p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)];
  }
  __syncthreads();

  // ---- activation, metrics, loss derivative (one thread per sample) ---------------------------
  float m_abs = 0.f;
  int m_cor = 0, m_nf = 0;
  if (tid < kTS) {
    const int s = s0 + tid;
    if (s < p.b) {
      act_and_lossder(w.sF + tid, kTS, w.sGl + tid /*fa scratch*/, w.sGl + tid, kTS, p.L, p.y[s], p.act_fn,
                      p.loss_fn, p.T, m_abs, m_cor, m_nf);
    } else {
      for (int l = 0; l < p.L; ++l) w.sGl[l * kTS + tid] = 0.f;  // padded samples carry no gradient
    }
    // wave-level sum over the 32 sample lanes (lanes 32..63 of this wave hold zeros)
    for (int off = 16; off > 0; off >>= 1) {
      m_abs += __shfl_xor(m_abs, off);
      m_cor += __shfl_xor(m_cor, off);
      m_nf += __shfl_xor(m_nf, off);
    }
    if (tid == 0) {
      float *tail = p.slabs + (size_t)blockIdx.x * p.slab_stride + p.bsize;
      tail[0] = (float)m_cor;
      tail[1] = m_abs;
      tail[2] = (float)m_nf;
      const int valid = p.b - s0;
      tail[3] = (float)(valid < 0 ? 0 : (valid > kTS ? kTS : valid));   // samples counted
    }
  }

  // ---- extend the behind environment with the core the previous step produced -----------------
  if (p.do_ext) {
    for (int e = tid; e < p.h * kTS; e += kWideThreads) {
      const int o = e / kTS, sl = e % kTS;
      float a0 = 0.f, a1 = 0.f;
      for (int in = 0; in < p.ext_core.n_in; ++in) {
        const float ev = p.first_ext ? 1.0f : w.sHp[in * kTS + sl];
        a0 = fmaf(ev, w.sA[(in * kD) * p.h + o], a0);
        a1 = fmaf(ev, w.sA[(in * kD + 1) * p.h + o], a1);
      }
      const float v = sXm[sl * kD] * a0 + sXm[sl * kD + 1] * a1;
      w.sH[e] = v;
      p.Hcur[(size_t)o * p.b_pad + s0 + sl] = v;
    }
  } else {
    for (int e = tid; e < p.h * kTS; e += kWideThreads)
      w.sH[e] = p.Hcur ? p.Hcur[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  }
  __syncthreads();

  // ---- partial bond gradient over this workgroup's samples (Network_class.py:710) --------------
  //   dB[h,dk,dk1,g,l] = sum_s gl[l,s] * (H[s,h] x_k[s,dk]) * (x_{k+1}[s,dk1] G[s,g])
  const int PW = p.h * kD, QW = kD * p.g;
  for (int e = tid; e < kTS * PW; e += kWideThreads) {
    const int sl = e / PW, c = e % PW;
    w.sP[e] = w.sH[(c / kD) * kTS + sl] * sXk[sl * kD + (c % kD)];
  }
  for (int e = tid; e < kTS * QW; e += kWideThreads) {
    const int sl = e / QW, c = e % QW;
    w.sQ[e] = sXp[sl * kD + (c / p.g)] * w.sG[(c % p.g) * kTS + sl];
  }
  __syncthreads();
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  const int QL = QW * p.L;
  for (int e = tid; e < p.bsize; e += kWideThreads) {
    const int row = e / QL, r = e % QL;
    const int col = r / p.L, l = r % p.L;
    float acc = 0.f;
#pragma unroll 8
    for (int sl = 0; sl < kTS; ++sl)
      acc = fmaf(w.sGl[l * kTS + sl] * w.sP[sl * PW + row], w.sQ[sl * QW + col], acc);
    slab[e] = acc;
  }
}

// f-only variant: recompute f from the last updated B (end of a tnml_sweep call).
__global__ __launch_bounds__(kWideThreads) void f_only_kernel(WideParams p, int hmax, int gmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideSmem w = carve(smem, hmax, gmax, p.L, 0);
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * kTS;
  for (int e = tid; e < p.hp * kTS; e += kWideThreads)
    w.sHp[e] = p.Hprev ? p.Hprev[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  for (int e = tid; e < p.gp * kTS; e += kWideThreads)
    w.sGp[e] = p.Gprev ? p.Gprev[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] : 1.0f;
  for (int e = tid; e < 2 * kTS * kD; e += kWideThreads) {
    const int which = e / (kTS * kD), r = e % (kTS * kD);
    const float *src = which == 0 ? p.x_km1 : p.x_k;
    w.sX[e] = src[(size_t)s0 * kD + r];
  }
  __syncthreads();
  f_from_B(p, w.sHp, w.sGp, w.sX, w.sX + kTS * kD, w.sRed, w.sF);
  for (int e = tid; e < p.L * kTS; e += kWideThreads)
    p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] = w.sF[e];
}

void launch_wide(const WideParams &p, int nblk, hipStream_t st) {
  const int hmax = p.h > p.hp ? p.h : p.hp;
  const int gmax = p.g > p.gp ? p.g : p.gp;
  const int core_elems = p.do_ext ? p.ext_core.n_in * kD * p.ext_core.n_out : 0;
  hipLaunchKernelGGL(wide_step_kernel, dim3(nblk), dim3(kWideThreads), wide_lds_bytes(hmax, gmax, p.L, core_elems),
                     st, p, hmax, gmax, core_elems);
}

void launch_f_only(const WideParams &p, int nblk, hipStream_t st) {
  hipLaunchKernelGGL(f_only_kernel, dim3(nblk), dim3(kWideThreads), wide_lds_bytes(p.hp, p.gp, p.L, 0), st, p,
                     p.hp, p.gp);
}

// ------------------------------------------------------------------------------------------
// red[e] = sum over slabs, in slab order (deterministic).  One thread per element.
// ------------------------------------------------------------------------------------------
// 64 consecutive elements x 16 slab chunks per workgroup; chunk partials meet in LDS in chunk order.
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float *__restrict__ slabs, int nblk,
                                                            int slab_stride, int n, float *__restrict__ red) {
  __shared__ float part[16][64];
  const int el = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int per = (nblk + 15) / 16;
  const int k0 = chunk * per, k1 = min(nblk, k0 + per);
  float a0 = 0.f, a1 = 0.f;
  if (e < n) {
    int k = k0;
    for (; k + 2 <= k1; k += 2) {
      a0 += slabs[(size_t)k * slab_stride + e];
      a1 += slabs[(size_t)(k + 1) * slab_stride + e];
    }
    if (k < k1) a0 += slabs[(size_t)k * slab_stride + e];
  }
  part[chunk][el] = a0 + a1;
  __syncthreads();
  if (chunk == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) t += part[c][el];
    red[e] = t;
  }
}

void launch_reduce(const float *slabs, int nblk, int slab_stride, int n, float *red, hipStream_t st) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, slabs, nblk, slab_stride, n,
                     red);
}

// ------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------
__global__ void scale_kernel(float *p, size_t n, float factor) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] *= factor;
}
void launch_scale(float *p, size_t n, float factor, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, factor);
}

__global__ void absmax_kernel(const float *__restrict__ f, int L, int b, int b_pad, float *out) {
  __shared__ float red[256];
  float m = 0.f;
  for (int e = threadIdx.x; e < L * b; e += 256) m = fmaxf(m, fabsf(f[(size_t)(e / b) * b_pad + (e % b)]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}
void launch_absmax(const float *f, int L, int b, int b_pad, float *out, hipStream_t st) {
  hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, st, f, L, b, b_pad, out);
}

__global__ void activation_kernel(const float *__restrict__ f, const int *__restrict__ y, int L, int b,
                                  int b_pad, int act_fn, int loss_fn, float T, float *act_out,
                                  float *der_out) {
  // one thread per sample; L values live in global scratch rows of act_out/der_out themselves
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= b) return;
  float sa;
  int c, nf = 0;
  act_and_lossder(f + s, b_pad, act_out + s, der_out + s, b_pad, L, y ? y[s] : 0, act_fn, loss_fn, T, sa, c, nf);
}
void launch_activation(const float *f, const int *y, int L, int b, int b_pad, int act_fn, int loss_fn,
                       float T, float *act_out, float *der_out, hipStream_t st) {
  hipLaunchKernelGGL(activation_kernel, dim3((b + 127) / 128), dim3(128), 0, st, f, y, L, b, b_pad, act_fn,
                     loss_fn, T, act_out, der_out);
}

}  // namespace tnml
