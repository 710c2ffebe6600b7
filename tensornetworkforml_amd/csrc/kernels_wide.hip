// Batch-parallel kernels of the MPS sweep: input re-tiling, the forward environment chain,
// the per-step wide kernel (f from the previous B, activation / loss derivative / metrics,
// environment extension, partial bond gradient) and the slab reduction.
//
// The per-step kernels here are the CLASSIC launch sequence (batch kernel -> [reduction] -> update/SVD kernel); the default
// path since round 2 is the pipelined single-launch step (wide_pipe_device.h + kernels_narrow.hip), which keeps these for
// steps whose merged tensor does not fit one workgroup's LDS (large-tensor path) and for the standalone entry points.
//
// Layouts: features x[site][b_pad][D]; environments env[slot][m][b_pad] (bond-major, the batch is
// the contiguous axis, so a wave reads 64 consecutive samples of one bond index); f [L][b_pad].
#include "tnml_internal.h"
#include "small_gemm_device.h"
#include "act_device.h"

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
typedef float fvec4 __attribute__((ext_vector_type(4)));
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
  // Software pipeline over the sites: the descriptor of site i + 2 and the core elements + features of site i + 1 are
  // requested while site i is computed (a site used to cost two dependent memory round trips -- descriptor, then core --
  // and three integer divisions per staged element: 2.5 us of a 784-site chain's 1.9 ms per site).
  constexpr int kPre = 4;                         // core elements per thread held in registers (bond <= 32); the rest is staged directly
  ChainSite cs = sites[0], cs1 = sites[n_sites > 1 ? 1 : 0];
  float pre[kPre], prex = 0.f;
  auto core_at = [&](const ChainSite &c, int e, float inv_o) -> float {
    const float *src = (c.is_label ? labcore : cores) + c.core_off;
    const int r = (int)(((float)e + 0.5f) * inv_o), o = e - r * c.n_out;      // exact quotient of small integers
    return src[(r / kD) * c.s_in + (r % kD) * c.s_d + o * c.s_out];
  };
  auto issue = [&](const ChainSite &c) {
    const int na = c.n_in * kD * c.n_out;
    const float inv_o = 1.0f / (float)c.n_out;
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int e = tid + u * kChainThreads;
      pre[u] = e < na ? core_at(c, e, inv_o) : 0.f;
    }
    if (tid < kChainTS * kD) prex = X[((size_t)c.x_site * b_pad + blockIdx.x * kChainTS + tid / kD) * kD + (tid % kD)];
  };
  issue(cs);
  for (int i = 0; i < n_sites; ++i) {
    const int na = cs.n_in * kD * cs.n_out;
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int e = tid + u * kChainThreads;
      if (e < na) sA[e] = pre[u];
    }
    if (na > kPre * kChainThreads) {
      const float inv_o = 1.0f / (float)cs.n_out;
      for (int e = tid + kPre * kChainThreads; e < na; e += kChainThreads) sA[e] = core_at(cs, e, inv_o);
    }
    if (tid < kChainTS * kD) sX[tid] = prex;
    const ChainSite cs2 = sites[i + 2 < n_sites ? i + 2 : n_sites - 1];
    lds_barrier();                                // LDS only: nobody waits for the environment stores of the previous site
    if (i + 1 < n_sites) issue(cs1);              // in flight while this site is computed
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
        if (env_base) env_base[cs.env_out_off + (size_t)o * b_pad + s] = v;   // nullptr: prediction only, nothing kept
      } else {
        f[(size_t)o * b_pad + s] = v;
      }
    }
    lds_barrier();
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
    cs = cs1; cs1 = cs2;
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

// ------------------------------------------------------------------------------------------
// The same chain on the matrix cores (round 3), for bonds <= 32: sixteen samples per workgroup, ONE wave walks all sites and never
// meets a workgroup barrier.  Per site
//     env_out[s][o] = sum_{(in, d)} (env_in[s][in] x[s][d]) A[(in, d)][o]                    v_mfma_f32_16x16x4_f32
// with the samples as rows: A-operand lane (s = lane & 15, q = lane >> 4) holds, for k-step ks, row index (in, d) = 4 ks + q, i.e.
// in = 2 ks + (q >> 1), d = q & 1; the result tile (rows = samples 4 q + reg, column o = lane & 15) leaves as ONE 16-byte store per
// lane and column tile (the environment stack is [bond][sample]) and is turned into the next site's A-operand layout through a
// 2 KB LDS tile (written and read by the same wave: in order, no barrier).
// A single in-order wave pays 4-8 cycles for every instruction it issues (nothing else runs on its SIMD).  The first form of this
// kernel (one wave doing everything: 0.66 ms per 784-site C3 chain against 1.44 ms for the plain-FMA kernel) spent more than half
// of a site's ~2000 cycles on operand staging: core loads and their address arithmetic, the copy into LDS, twenty strided LDS reads.
// The vector-memory counter of gfx950 also retires IN ORDER, so a far-ahead request issued by the computing wave would put the
// latency of HBM on its very next wait.  Hence a workgroup of five waves with three jobs:
//   wave 0 (computes)  per site: waits for the loader's flag, reads its B operands (NT x NKS values at consecutive 256-byte
//                      offsets from one base register) and its feature value, multiplies the environment tile of the previous
//                      site into A operands, 2 NT chains of MFMAs, stores the environment (16 bytes per lane and tile) and writes
//                      the tile to LDS for the next site.  No loads from memory at all: its stores are never waited for.
//   waves 1..3 (load)  each takes every third site, together up to kChainRing - 1 sites ahead: core as it lies in memory (16-byte
//                      loads, requested one own site ahead) -> staging area -> gathered into the operand layout
//                      [tile][k-step][lane] of a ring slot, zeroed where a row / column lies outside the core (so the computing
//                      wave masks nothing); the lane's feature value.  (One loader kept the computing wave waiting: 0.76 ms; two
//                      0.61; three 0.55, where the loaders wait for ring room 1100 of their 5000 cycles per own site.)
//   wave 4 (warms L2)  touches one dword of every 128-byte line the loaders will want (cores, this workgroup's features) four
//                      blocks of eight sites ahead, paced by the progress word.
// Measured (tools/ubench/chain_bench.hip, -DTNML_CHAIN_STAMPS): the computing wave waits 240 cycles per site (the LDS round trip of
// its flag poll) and spends ~1450 on the site: 640 MFMA issue, the rest the serial tail accumulators -> tile -> rows -> times x.
// Two traps: a RELEASE store of a hand-off word also waits for the wave's GLOBAL stores / loads in flight (a trip to memory per
// site): the words are stored relaxed behind an explicit wait for the LDS counter; a scalar load shares that counter, so the site
// descriptor is requested one site ahead.
// Hand-offs are LDS words that only grow (sites ready / sites consumed); a wave that waits 2^22 polls sets the abort word and
// every wave leaves.
// ------------------------------------------------------------------------------------------
constexpr int kChainLD = 36;            // row stride of the wave's result tile (16-byte reads along a row of bond indices)
constexpr int kChainBlock = 8;          // sites per block of cache-warming requests
#ifdef TNML_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[16];
#endif
constexpr int kChainRing = 4;
#ifndef TNML_CHAIN_LOADERS
#define TNML_CHAIN_LOADERS 3
#endif
constexpr int kChainLoaders = TNML_CHAIN_LOADERS;          // loader waves (each takes every kChainLoaders-th site); kChainRing >= kChainLoaders + 1
__device__ inline bool chain_wait(const int *flag, int want, int *abort_word) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) { asm volatile("" ::: "memory"); return true; }
    if ((spin & 255) == 255 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return false;
}

#ifdef TNML_CHAIN_STAMPS
#define TNML_RSTAMP_BEGIN() rs0_ = __builtin_amdgcn_s_memtime()
#define TNML_RSTAMP_END(k) { if (blockIdx.x == 7 && lane == 0) { const unsigned long long d_ = __builtin_amdgcn_s_memtime() - rs0_; if ((k) == 0 || role == 1) atomicAdd(&g_chain_stamps[8 + (k)], d_); } }
#else
#define TNML_RSTAMP_BEGIN()
#define TNML_RSTAMP_END(k)
#endif
template <int NV4, int NT, int NKS>
__global__ __launch_bounds__(64 * (kChainLoaders + 2)) void env_chain_roles_kernel(const ChainSite *__restrict__ sites, int n_sites, const float *__restrict__ cores,
                                                              const float *__restrict__ labcore, const float *__restrict__ X,
                                                              float *__restrict__ env_base, float *__restrict__ f, int b_pad) {
  __shared__ __attribute__((aligned(16))) float sRawAll[kChainLoaders][NV4 * 256];
  __shared__ __attribute__((aligned(16))) float sB[kChainRing][NT * NKS * 64];
  __shared__ float sXr[kChainRing][64];
  __shared__ __attribute__((aligned(16))) float sE[16 * kChainLD];
  __shared__ int sReadyS[kChainRing], sDone, sAbort;
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int s0 = blockIdx.x * 16;
  const int half = q >> 1, dsel = q & 1;
#ifdef TNML_CHAIN_STAMPS
  unsigned long long rs0_ = 0;
#endif
  if (threadIdx.x < kChainRing) sReadyS[threadIdx.x] = 0;
  if (threadIdx.x == 0) { sDone = 0; sAbort = 0; }
  for (int e = threadIdx.x; e < 16 * kChainLD; e += 64 * (kChainLoaders + 2)) sE[e] = 0.f;
  __syncthreads();

  if (role == kChainLoaders + 1) {
    // ---- last wave: cache warming ----
    float warm = 0.f;
    for (int i0 = 0; i0 < n_sites; i0 += kChainBlock) {
      for (int spin = 0; spin < (1 << 16); ++spin) {
        if (i0 <= __hip_atomic_load(&sDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 4 * kChainBlock) break;
        __builtin_amdgcn_s_sleep(16);
      }
      float w[kChainBlock + 1];
      w[kChainBlock] = X[((size_t)sites[min(i0 + (lane & (kChainBlock - 1)), n_sites - 1)].x_site * b_pad + s0) * kD];
#pragma unroll
      for (int j = 0; j < kChainBlock; ++j) {
        const ChainSite &c = sites[min(i0 + j, n_sites - 1)];
        w[j] = (c.is_label ? labcore : cores)[c.core_off + min(lane * 32, c.n_in * kD * c.n_out - 1)];
      }
#pragma unroll
      for (int j = 0; j <= kChainBlock; ++j) warm += w[j];
    }
    if (n_sites < 0) f[lane] = warm;      // never taken: keeps the loads alive
    return;
  }

  if (role >= 1) {
    // ---- loader wave l = role - 1: operands of its sites j = l, l + kChainLoaders, ... into ring slot j % kChainRing ----
    float *sRaw = sRawAll[role - 1];
    fvec4 pre[NV4];
    float xpre;
    int goff[NT][NKS];                    // byte offsets into the staging area; -1: outside the core (operand 0)
#define TNML_LOAD_SITE(c)                                                                                          \
    {                                                                                                              \
      const fvec4 *src_ = reinterpret_cast<const fvec4 *>(((c).is_label ? labcore : cores) + (c).core_off);        \
      const int n4_ = ((c).n_in * kD * (c).n_out + 3) >> 2;                                                        \
      _Pragma("unroll") for (int u = 0; u < NV4; ++u) pre[u] = src_[min(lane + 64 * u, n4_ - 1)];                  \
      xpre = X[((size_t)(c).x_site * b_pad + s0 + r) * kD + dsel];                                                 \
    }
#define TNML_GATHER_OFFSETS(c)                                                                                     \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                               \
      const int o_ = 16 * t + r;                                                                                   \
      _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                         \
        const int in_ = 2 * ks + half;                                                                             \
        goff[t][ks] = (in_ < (c).n_in && o_ < (c).n_out) ? 4 * (dsel * (c).s_d + in_ * (c).s_in + o_ * (c).s_out) : -1; \
      }                                                                                                            \
    }
    const int j0 = role - 1;
    if (j0 >= n_sites) return;
    ChainSite cs = sites[j0], cn = sites[min(j0 + kChainLoaders, n_sites - 1)];
    TNML_LOAD_SITE(cs);
    TNML_GATHER_OFFSETS(cs);
    for (int j = j0; j < n_sites; j += kChainLoaders) {
      // the core of site j is in `pre` (requested one site ago); room in the ring?
      TNML_RSTAMP_BEGIN();
      if (j >= kChainRing && !chain_wait(&sDone, j - kChainRing + 1, &sAbort)) return;
      TNML_RSTAMP_END(1);
      TNML_RSTAMP_BEGIN();
#pragma unroll
      for (int u = 0; u < NV4; ++u) reinterpret_cast<fvec4 *>(sRaw)[lane + 64 * u] = pre[u];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TNML_RSTAMP_END(2);
      const float xj = xpre;
      const ChainSite c2 = sites[min(j + 2 * kChainLoaders, n_sites - 1)];
      if (j + kChainLoaders < n_sites) { TNML_LOAD_SITE(cn); }     // travels while this site is gathered
      float *slot = sB[j & (kChainRing - 1)];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int off = goff[t][ks];
          const float v = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sRaw) + max(off, 0));
          slot[(t * NKS + ks) * 64 + lane] = off >= 0 ? v : 0.f;
        }
      sXr[j & (kChainRing - 1)][lane] = xj;
      // (a RELEASE store would also wait for this wave's loads of the next core -- `vmcnt(0)` -- only the LDS writes above matter)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __hip_atomic_store(&sReadyS[j & (kChainRing - 1)], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (j + kChainLoaders < n_sites) {
        if (cn.s_in != cs.s_in || cn.s_d != cs.s_d || cn.s_out != cs.s_out || cn.n_out != cs.n_out || cn.n_in != cs.n_in) { TNML_GATHER_OFFSETS(cn); }
        cs = cn; cn = c2;
      }
    }
#undef TNML_LOAD_SITE
#undef TNML_GATHER_OFFSETS
    return;
  }

  // ---- wave 0: the chain ----
  // Per site, ONE block of straight-line code holds the MFMAs of this site AND the LDS reads of the next site's operands (the loader
  // is normally a site or more ahead: its flag is polled before the block), so that the reads travel in the shadow of the matrix
  // pipeline; what remains serial is: accumulators -> environment store + tile to LDS -> tile back as rows -> times the feature.
  int soff[NT];                           // this lane's place in an environment slot: [bond o = 16 t + r][samples s0 + 4 q ..]
#pragma unroll
  for (int t = 0; t < NT; ++t) soff[t] = (16 * t + r) * b_pad + s0 + 4 * q;
  float av[NKS], bv[NT][NKS], bvn[NT][NKS];
  if (!chain_wait(&sReadyS[0], 1, &sAbort)) return;
  {
    const float x0 = sXr[0][lane];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) av[ks] = 0.f;
    if (half == 0) av[0] = x0;            // the chain starts from the scalar 1 (n_in == 1)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bv[t][ks] = sB[0][(t * NKS + ks) * 64 + lane];
  }
  // what the computing wave needs of a site's descriptor, requested one site ahead (a scalar load shares its counter with the LDS
  // reads: requested where it is used it put a trip to L2 on every site)
  bool seen_next = false;
  int cur_nout = sites[0].n_out, nxt_nout = sites[min(1, n_sites - 1)].n_out;
  long long cur_off = sites[0].env_out_off, nxt_off = sites[min(1, n_sites - 1)].env_out_off;
  // one site; BV = this site's operand registers, BN = the next site's (the loop is unrolled by two so that they swap by name)
#define TNML_ROLE_SITE(BV, BN)                                                                                     \
  {                                                                                                                \
    const bool more = i + 1 < n_sites;                                                                             \
    const int n_out_i = cur_nout;                                                                                  \
    const long long off_i = cur_off;                                                                               \
    cur_nout = nxt_nout; cur_off = nxt_off;                                                                        \
    nxt_nout = sites[min(i + 2, n_sites - 1)].n_out; nxt_off = sites[min(i + 2, n_sites - 1)].env_out_off;         \
    TNML_RSTAMP_BEGIN();                                                                                           \
    /* the flag of site i + 1 was read during site i - 1 (its LDS round trip hidden behind that site's MFMAs); poll only if it */ \
    /* had not arrived then */                                                                                     \
    if (more && !seen_next && !chain_wait(&sReadyS[(i + 1) & (kChainRing - 1)], i + 2, &sAbort)) return;           \
    TNML_RSTAMP_END(0);                                                                                            \
    const int peek = __hip_atomic_load(&sReadyS[(i + 2) & (kChainRing - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    const float *nslot = sB[(i + 1) & (kChainRing - 1)];                                                           \
    /* NT == 2: one accumulator per tile, the tiles alternating (a chain's consecutive MFMAs are two issue slots apart: no */ \
    /* stall); NT == 1: two accumulators over the k-steps */                                                       \
    fvec4 acc[2];                                                                                                  \
    acc[0] = fvec4{0.f, 0.f, 0.f, 0.f}; acc[1] = fvec4{0.f, 0.f, 0.f, 0.f};                                       \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                           \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                             \
        BN[t][ks] = nslot[(t * NKS + ks) * 64 + lane];          /* (a finished chain reads a stale slot: never used) */ \
        const int ai = NT == 2 ? t : (ks & 1);                                                                     \
        acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], BV[t][ks], acc[ai], 0, 0, 0);                       \
      }                                                                                                            \
    }                                                                                                              \
    const float xnext = sXr[(i + 1) & (kChainRing - 1)][lane];                                                     \
    seen_next = peek >= i + 3;                                                                                     \
    if (NT == 1) acc[0] += acc[1];                                                                                 \
    float *out = off_i >= 0 ? (env_base ? env_base + off_i : nullptr) : f;                                         \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                               \
      const int o = 16 * t + r;                                                                                    \
      if (out && o < n_out_i) *reinterpret_cast<fvec4 *>(out + soff[t]) = acc[t];                                  \
      /* the next site reads env[s][in = o] as av[ks] of the lanes with half == (o & 1), ks = o >> 1 (columns beyond n_out */ \
      /* are exact zeros: the loader zeroed those B operands) */                                                   \
      const int colp = (o & 1) * 16 + (o >> 1);                                                                    \
      _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) sE[(4 * q + reg) * kChainLD + colp] = acc[t][reg];       \
    }                                                                                                              \
    if (more) {                                                                                                    \
      const fvec4 *erow = reinterpret_cast<const fvec4 *>(sE + r * kChainLD + half * 16);                          \
      _Pragma("unroll") for (int k4 = 0; k4 < (NKS + 3) / 4; ++k4) {                                               \
        const fvec4 v = erow[k4];                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) if (4 * k4 + e < NKS) av[4 * k4 + e] = v[e] * xnext;         \
      }                                                                                                            \
    }                                                                                                              \
    /* (this site's operands and the next site's are in registers: the loader may have slot i back) */            \
    /* (not a RELEASE store: that waits for the environment stores too -- a trip to memory per site) */           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __hip_atomic_store(&sDone, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                             \
  }
  for (int i = 0; i < n_sites; ++i) {
    TNML_ROLE_SITE(bv, bvn);
    if (++i >= n_sites) break;
    TNML_ROLE_SITE(bvn, bv);
  }
#undef TNML_ROLE_SITE
}

void launch_env_chain(const ChainSite *sites_dev, int n_sites, const float *cores, const float *labcore,
                      const float *X, float *env_base, float *f, int b, int b_pad, int L, int Mmax,
                      float *logmax_out, hipStream_t st, bool force_plain) {
  const int mo = Mmax > L ? Mmax : L;
  size_t lds = ((size_t)Mmax * kD * mo + 2 * (size_t)mo * kChainTS + kChainTS * kD + 2 * kChainTS) * sizeof(float);
  if (logmax_out)
    hipLaunchKernelGGL(env_chain_kernel<true>, dim3(b_pad / kChainTS), dim3(kChainThreads), lds, st, sites_dev,
                       n_sites, cores, labcore, X, env_base, f, b, b_pad, L, Mmax, logmax_out);
  else if (!force_plain && mo <= 32 && (Mmax & 1) == 0) {    // (core slots of an even bond capacity are 16-byte aligned: the cores travel as 16-byte loads)
    const int n4 = (Mmax * kD * mo + 3) / 4;
    const dim3 grid(b_pad / 16), block(64 * (kChainLoaders + 2));
#define TNML_CHAIN_GO(NV4, NT, NKS) hipLaunchKernelGGL((env_chain_roles_kernel<NV4, NT, NKS>), grid, block, 0, st, sites_dev, n_sites, cores, labcore, X, env_base, f, b_pad)
    (void)n4;
    if (mo <= 10) TNML_CHAIN_GO(1, 1, 5);
    else if (mo <= 16) TNML_CHAIN_GO(2, 1, 8);
    else if (mo <= 20) TNML_CHAIN_GO(4, 2, 10);
    else if (mo <= 24) TNML_CHAIN_GO(5, 2, 12);
    else TNML_CHAIN_GO(8, 2, 16);
#undef TNML_CHAIN_GO
  }
  else
    hipLaunchKernelGGL(env_chain_kernel<false>, dim3(b_pad / kChainTS), dim3(kChainThreads), lds, st, sites_dev,
                       n_sites, cores, labcore, X, env_base, f, b, b_pad, L, Mmax, logmax_out);
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
// Last-resort batch kernel (plain FMA, operands streamed from global memory): one workgroup = kTS samples.  Taken only when a
// tile's operands fit neither MFMA kernel's LDS budget (bond 64 with three labels: tests/test_true_shapes_gpu.py::
// test_largest_matrix_side_of_the_large_path runs it).
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

// ------------------------------------------------------------------------------------------
// Wide step kernel, MFMA formulation (v_mfma_f32_16x16x4_f32: exact float32 FMA chains).
// One workgroup = kTS samples, 8 waves.  All three GEMM-shaped parts put the SAMPLE on the MFMA
// column (= lane & 15), so per-sample operands are read at consecutive LDS addresses:
//   (f)   T_l[i, s]  = sum_j B'_l[i, j] Q'[j, s]          i = (h', d), j = (d', g')     waves 0-3
//         f[l, s]    = sum_i P'[i, s] T_l[i, s]                                         (epilogue)
//   (env) H[hn, s]   = sum_i A[i, hn] P'[i, s]            P'[i, s] = H'[h', s] x_{k-1}[s, d]   waves 4-7
//   (dB)  dB_l[i, j] = sum_s (g[l, s] P[i, s]) Q[j, s]    P = H x_k, Q = x_{k+1} G      all waves
// Operand products are formed once into zero-padded LDS arrays (row counts padded to 16, inner
// dimensions to 4), so the MFMA loops are two ds_read_b32 + one MFMA per k-step with no guards.
// Lane maps: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// C/D col = lane & 15, row = 4 (lane >> 4) + reg.
// ------------------------------------------------------------------------------------------
constexpr int kMfmaThreads = 512;
constexpr int kTSP = kTS + 1;           // sample stride of the LDS operand arrays (bank spread)
constexpr int kTSA = kTS + 2;           // ... of the arrays the tiled kernel's gradient GEMM reads as [row][sample] (2 mod 32)

__host__ __device__ inline int up(int x, int m) { return (x + m - 1) / m * m; }

struct WideMfmaDims {
  int IP, RS, JP, KA, HS, I3, J3;      // padded sizes (see carve)
};
__host__ __device__ inline WideMfmaDims wide_mfma_dims(int hp, int gp, int h, int g) {
  WideMfmaDims d;
  d.IP = up(hp * kD, 16);               // rows i = (h', d) of B'_l and of P'
  d.JP = up(kD * gp, 4);                // inner index j = (d', g') of the f product
  d.RS = 0;                             // row stride of B' in LDS, set by the caller (needs L)
  d.KA = up(hp * kD, 4);                // inner index of the env product
  d.HS = up(h, 16) + 1;                 // row stride of the extension core A[i][hn]
  d.I3 = up(h * kD, 16);                // rows of dB
  d.J3 = up(kD * g, 16);                // columns of dB
  return d;
}
struct WideMfmaSmem { float *sX, *sF, *sGl, *sPp, *sQp, *sBp, *sA, *sH, *sPg, *sQ, *rH, *rGp, *rG, *rB, *rA; size_t floats; };
__host__ __device__ inline WideMfmaSmem wide_mfma_carve(float *base, WideMfmaDims &d, int L, int h, int hp = 0, int gp = 0,
                                                        int g = 0) {
  WideMfmaSmem w;
  d.RS = (d.JP * L) | 1;                // rows of B'[i][(j, l)] at an odd stride: 16 rows -> 16 banks
  float *q = base;
  w.sX = q; q += 3 * kTS * kD;
  w.sF = q; q += L * kTS;
  w.sGl = q; q += L * kTS;
  w.sPp = q; q += d.IP * kTSP;                       // P'[i][s]   (also the B operand of the env product)
  w.sQp = q; q += d.JP * kTSP;                       // Q'[j][s]
  w.sBp = q; q += (size_t)d.IP * d.RS;               // B'[i][(j, l)] as stored, zero padded
  w.sA = q; q += d.KA * d.HS;                        // extension core A[i][hn]
  w.sH = q; q += up(h, 16) * kTSP;                   // H[hn][s]
  w.sPg = q; q += (size_t)L * d.I3 * kTSP;           // g[l][s] P[i][s]
  w.sQ = q; q += d.J3 * kTSP;                        // Q[j][s]
  // raw landing zones of the asynchronous global -> LDS loads (64 floats per wave-instruction)
  constexpr int RPI = 64 / kTS;                      // sample rows per wave-instruction
  w.rH = q; q += up(hp, RPI) * kTS;                  // H' rows   [h'][s]
  w.rGp = q; q += up(gp, RPI) * kTS;                 // G' rows   [g'][s]
  w.rG = q; q += up(g, RPI) * kTS;                   // G rows    [g][s]
  w.rB = q; q += up(hp * kD * kD * gp * L, 64);      // B' as stored
  w.rA = q; q += up(hp * kD * h, 64);                // extension core A[i][hn], unpadded
  w.floats = (size_t)(q - base);
  return w;
}

__global__ __launch_bounds__(kMfmaThreads) void wide_step_mfma_kernel(WideParams p, PrepParams prep, int nblk_samples) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= nblk_samples) {
    // batch-independent side job on CUs the sample tiles leave idle: slice (dk, dk1) of the merged tensor of THIS
    // step and of its L2 term, for the narrow kernel that follows (small_gemm_device.h)
    prep_slice_block(prep, blockIdx.x - nblk_samples, (unsigned char *)smem);
    return;
  }
  WideMfmaDims dm = wide_mfma_dims(p.hp, p.gp, p.h, p.g);
  const WideMfmaSmem w = wide_mfma_carve(smem, dm, p.L, p.h, p.hp, p.gp, p.g);
  const int tid = threadIdx.x, NT = kMfmaThreads;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int s0 = blockIdx.x * kTS;
  const int L = p.L, h = p.h, g = p.g, hp = p.hp, gp = p.gp;
  const int nI = hp * kD, nJ = kD * gp;               // live rows / inner size of the f product
  constexpr int ST = kTS / 16;                        // sample tiles
  const bool stamp = p.stamps && blockIdx.x == 0 && tid == 0;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WSTAMP(i) if (stamp) ts[i] = __builtin_amdgcn_s_memtime()
  WSTAMP(0);

  // ---- stage 1: every global read of the workgroup as an asynchronous global -> LDS load
  // (global_load_lds_dword: 64 consecutive floats of LDS per wave-instruction, per-lane global
  // address), all in flight together; ONE wait; then the padded operand arrays are built LDS -> LDS.
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
#define GLDS(gp_, lp_) __builtin_amdgcn_global_load_lds((gptr_t)(gp_), (lptr_t)(lp_), 4, 0, 0)
  const int NWV = NT / 64;
  static_assert(kTS == 32 || kTS == 64, "a wave-instruction carries 64 / kTS whole sample rows");
  constexpr int RPI = 64 / kTS;
  const int half = lane / kTS, col = lane % kTS;
  const bool haveH = p.Hprev && !p.first_ext && (p.do_f || p.do_ext);
  {
    // x of the three sites: 3 * kTS * kD = 192 floats = 3 wave-instructions
    for (int c = wave; c < 3 * kTS * kD / 64; c += NWV) {     // x of the three sites, kTS * kD floats each
      const int which = c / (kTS * kD / 64), off = (c % (kTS * kD / 64)) * 64;
      const float *src = which == 0 ? p.x_km1 : (which == 1 ? p.x_k : p.x_kp1);
      if (src) GLDS(src + (size_t)s0 * kD + off + lane, w.sX + c * 64);
    }
    if (haveH)
      for (int pr = wave; pr < up(hp, RPI) / RPI; pr += NWV)
        GLDS(p.Hprev + (size_t)min(RPI * pr + half, hp - 1) * p.b_pad + s0 + col, w.rH + pr * 64);
    if (p.do_f && p.Gprev)
      for (int pr = wave; pr < up(gp, RPI) / RPI; pr += NWV)
        GLDS(p.Gprev + (size_t)min(RPI * pr + half, gp - 1) * p.b_pad + s0 + col, w.rGp + pr * 64);
    if (p.Gcur)
      for (int pr = wave; pr < up(g, RPI) / RPI; pr += NWV)
        GLDS(p.Gcur + (size_t)min(RPI * pr + half, g - 1) * p.b_pad + s0 + col, w.rG + pr * 64);
    if (p.do_f) {
      const int nB = nI * nJ * L;
      for (int c = wave; c < up(nB, 64) / 64; c += NWV) GLDS(p.Bprev + min(c * 64 + lane, nB - 1), w.rB + c * 64);
    }
    if (p.do_ext) {
      const int nA = nI * h;
      for (int c = wave; c < up(nA, 64) / 64; c += NWV) {
        const int idx = min(c * 64 + lane, nA - 1);
        const int i = idx / h, o = idx - i * h;
        GLDS(p.ext_core.base + (i >> 1) * p.ext_core.s_in + (i & 1) * p.ext_core.s_d + o * p.ext_core.s_out, w.rA + c * 64);
      }
    }
    if (!p.do_f)
      for (int e = tid; e < L * kTS; e += NT) w.sF[e] = p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)];
    if (!p.do_ext)
      for (int e = tid; e < up(h, 16) * kTS; e += NT) {
        const int sl = e % kTS, hn = e / kTS;
        w.sH[hn * kTSP + sl] = hn < h ? (p.Hcur ? p.Hcur[(size_t)hn * p.b_pad + s0 + sl] : 1.0f) : 0.f;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  WSTAMP(1);
  __syncthreads();

  // ---- stage 2: padded operand arrays, LDS -> LDS ---------------------------------------------------
  const float *sXm = w.sX, *sXk = w.sX + kTS * kD, *sXp = w.sX + 2 * kTS * kD;
  if (p.do_f || p.do_ext)
    for (int e = tid; e < dm.IP * kTS; e += NT) {       // P'[i][s] = H'[h'][s] x_{k-1}[s][d]
      const int sl = e % kTS, i = e / kTS;
      float v = 0.f;
      if (i < nI) v = (haveH ? w.rH[(i >> 1) * kTS + sl] : 1.0f) * sXm[sl * kD + (i & 1)];
      w.sPp[i * kTSP + sl] = v;
    }
  if (p.do_f) {
    for (int e = tid; e < dm.JP * kTS; e += NT) {       // Q'[j][s] = x_k[s][d'] G'[g'][s],  j = d' * gp + g'
      const int sl = e % kTS, j = e / kTS;
      float v = 0.f;
      if (j < nJ) {
        const int dd = j >= gp ? 1 : 0;                 // D == 2
        v = sXk[sl * kD + dd] * (p.Gprev ? w.rGp[(j - dd * gp) * kTS + sl] : 1.0f);
      }
      w.sQp[j * kTSP + sl] = v;
    }
    const int rowlen = nJ * L;
    for (int i = wave; i < dm.IP; i += NWV)             // B'[i][(j, l)] at the odd row stride RS
      for (int x = lane; x < dm.RS; x += 64)
        w.sBp[i * dm.RS + x] = (i < nI && x < rowlen) ? w.rB[i * rowlen + x] : 0.f;
  }
  if (p.do_ext)
    for (int i = wave; i < dm.KA; i += NWV)
      for (int o = lane; o < dm.HS; o += 64)
        w.sA[i * dm.HS + o] = (i < nI && o < h) ? w.rA[i * h + o] : 0.f;
  for (int e = tid; e < dm.J3 * kTS; e += NT) {         // Q[j][s] = x_{k+1}[s][dk1] G[g][s],  j = dk1 * g + g_
    const int sl = e % kTS, j = e / kTS;
    float v = 0.f;
    if (j < kD * g) {
      const int dd = j >= g ? 1 : 0;                    // D == 2
      v = sXp[sl * kD + dd] * (p.Gcur ? w.rG[(j - dd * g) * kTS + sl] : 1.0f);
    }
    w.sQ[j * kTSP + sl] = v;
  }
  __syncthreads();

  WSTAMP(2);
  // ---- f of the previous step (waves 0-3)  ||  extension of the behind environment (waves 4-7) ----
  if (wave < 4) {
    if (p.do_f) {
      for (int cidx = wave; cidx < L * ST; cidx += 4) {
        const int l = cidx / ST, st = cidx % ST;
        const float *bq = w.sQp + q * kTSP + st * 16 + r;
        float facc = 0.f;
        for (int it = 0; it < dm.IP / 16; ++it) {
          const float *ap = w.sBp + (it * 16 + r) * dm.RS + q * L + l;      // B'[i][(j, l)], j = 4 kk + q
          fvec4 acc = {0.f, 0.f, 0.f, 0.f};
// (run-time trip count: `#pragma unroll 5` here was refused by the optimizer -- "loop not unrolled" -- and is gone)
          for (int kk = 0; kk < dm.JP / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * L], bq[kk * 4 * kTSP], acc, 0, 0, 0);
          const float *pp = w.sPp + (it * 16 + 4 * q) * kTSP + st * 16 + r;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) facc = fmaf(acc[reg], pp[reg * kTSP], facc);
        }
        facc += __shfl_xor(facc, 16);
        facc += __shfl_xor(facc, 32);
        if (q == 0) {
          w.sF[l * kTS + st * 16 + r] = facc;
          p.f[(size_t)l * p.b_pad + s0 + st * 16 + r] = facc;
        }
      }
    }
  } else if (p.do_ext) {
    const int HT = up(h, 16) / 16;
    for (int cidx = wave - 4; cidx < HT * ST; cidx += 4) {
      const int ht = cidx / ST, st = cidx % ST;
      const float *ap = w.sA + q * dm.HS + ht * 16 + r;
      const float *bq = w.sPp + q * kTSP + st * 16 + r;
      fvec4 acc = {0.f, 0.f, 0.f, 0.f};
// (run-time trip count: `#pragma unroll 5` here was refused by the optimizer -- "loop not unrolled" -- and is gone)
      for (int kk = 0; kk < dm.KA / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * dm.HS], bq[kk * 4 * kTSP], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int hn = ht * 16 + 4 * q + reg;
        w.sH[hn * kTSP + st * 16 + r] = acc[reg];                 // rows >= h of the padded tile are zero
        if (hn < h) p.Hcur[(size_t)hn * p.b_pad + s0 + st * 16 + r] = acc[reg];
      }
    }
  }
  __syncthreads();

  WSTAMP(3);
  // ---- activation, metrics, loss derivative (one thread per sample) ------------------------------
  {
    // per-sample work on the first kTS threads; the per-sample metric terms go to LDS and one thread
    // adds them in sample order (deterministic)
    __shared__ float sMet[3][kTS];
    if (tid < kTS) {
      const int s = s0 + tid;
      float m_abs = 0.f;
      int m_cor = 0, m_nf = 0;
      if (s < p.b) {
        act_and_lossder(w.sF + tid, kTS, w.sGl + tid, w.sGl + tid, kTS, L, p.y[s], p.act_fn, p.loss_fn, p.T, m_abs,
                        m_cor, m_nf);
      } else {
        for (int l = 0; l < L; ++l) w.sGl[l * kTS + tid] = 0.f;    // padded samples carry no gradient
      }
      sMet[0][tid] = (float)m_cor; sMet[1][tid] = m_abs; sMet[2][tid] = (float)m_nf;
    }
    if (stamp) ts[7] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid < 3) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < kTS; ++i) t += sMet[tid][i];
      p.slabs[(size_t)blockIdx.x * p.slab_stride + p.bsize + tid] = t;
    } else if (tid == 3) {
      const int valid = p.b - s0;
      p.slabs[(size_t)blockIdx.x * p.slab_stride + p.bsize + 3] = (float)(valid < 0 ? 0 : (valid > kTS ? kTS : valid));
    }
  }
  WSTAMP(4);
  // ---- (g P)[l][i][s] ---------------------------------------------------------------------------------
  for (int l = 0; l < L; ++l)
    for (int e = tid; e < dm.I3 * kTS; e += NT) {
      const int sl = e % kTS, i = e / kTS;
      float v = 0.f;
      if (i < h * kD) v = w.sGl[l * kTS + sl] * w.sH[(i >> 1) * kTSP + sl] * sXk[sl * kD + (i & 1)];
      w.sPg[((size_t)l * dm.I3 + i) * kTSP + sl] = v;
    }
  __syncthreads();

  WSTAMP(5);
  // ---- partial bond gradient: one 16x16 tile of dB_l per wave-iteration, K = the kTS samples ---------
  {
    const int IT = dm.I3 / 16, JT = dm.J3 / 16, QW = kD * g;
    float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
    for (int cidx = wave; cidx < L * IT * JT; cidx += NT / 64) {
      const int jt = cidx % JT, t = cidx / JT;
      const int it = t % IT, l = t / IT;
      const float *ap = w.sPg + ((size_t)l * dm.I3 + it * 16 + r) * kTSP + q;
      const float *bq = w.sQ + (jt * 16 + r) * kTSP + q;
      fvec4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < kTS / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4], bq[kk * 4], acc, 0, 0, 0);
      const int j = jt * 16 + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = it * 16 + 4 * q + reg;
        if (i < h * kD && j < QW) slab[((size_t)i * QW + j) * L + l] = acc[reg];
      }
    }
  }
  if (stamp) {
    ts[6] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 6; ++i) p.stamps[i] = (double)(ts[i + 1] - ts[i]);
    p.stamps[6] = (double)(ts[7] - ts[3]);
  }
}

// ------------------------------------------------------------------------------------------
// Wide step kernel, MFMA formulation for tiles whose operands do not fit LDS at once (bond 50 with ten labels:
// the previous step's merged tensor alone is 400 KB).  Same three products as wide_step_mfma_kernel, with
//   * the merged tensor B' streamed through LDS in chunks of 16 rows i = (h', d) (asynchronous global -> LDS
//     loads straight into the padded layout); every wave keeps the partial f of its (label, sample-tile) pairs in
//     registers across the chunks;
//   * the bond gradient as ONE GEMM over the flattened column index n = (j, l):
//        dB[i, n] = sum_s P[i, s] (Q[j, s] g[l, s]),
//     the label factor applied to the B operand on the fly (no per-label operand array), and 16 consecutive n per
//     tile so that the slab is written in 64-byte runs;
//   * the staging buffers of the first phase (environments, extension core) aliased with the chunk buffer.
// ------------------------------------------------------------------------------------------
struct WideTiledSmem { float *sX, *sF, *sGl, *sPp, *sQp, *sH, *sP, *sQ, *U, *rH, *rGp, *rG, *rA, *sA, *sBc; size_t floats; };
// (ext = false: a launch that extends no environment -- the pipelined large-tensor step, whose row operand has D h rows -- needs no room
//  for the extension core)
__host__ __device__ inline WideTiledSmem wide_tiled_carve(float *base, WideMfmaDims &d, int L, int h, int hp, int gp, int g, bool ext = true) {
  WideTiledSmem w;
  d.RS = (d.JP * L) | 1;
  float *q = base;
  w.sX = q; q += 3 * kTS * kD;
  w.sF = q; q += L * kTS;
  w.sGl = q; q += L * kTSA;            // (operands of the gradient GEMM at the stride kTSA: see there)
  w.sPp = q; q += d.IP * kTSP;
  w.sQp = q; q += d.JP * kTSP;
  w.sH = q; q += up(h, 16) * kTSP;
  w.sP = q; q += d.I3 * kTSA;
  w.sQ = q; q += d.J3 * kTSA;
  w.U = q;
  constexpr int RPI = 64 / kTS;
  float *a = q;
  w.rH = a; a += up(hp, RPI) * kTS;
  w.rGp = a; a += up(gp, RPI) * kTS;
  w.rG = a; a += up(g, RPI) * kTS;
  w.rA = a; a += ext ? up(hp * kD * h, 64) : 0;
  w.sA = a; a += ext ? d.KA * d.HS : 0;
  const size_t ua = (size_t)(a - q), ub = (size_t)16 * d.RS + 64;
  w.sBc = q;
  q += ua > ub ? ua : ub;
  w.floats = (size_t)(q - base);
  return w;
}

__global__ __launch_bounds__(kMfmaThreads) void wide_step_mfma_tiled_kernel(WideParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WideMfmaDims dm = wide_mfma_dims(p.hp, p.gp, p.h, p.g);
  const WideTiledSmem w = wide_tiled_carve(smem, dm, p.L, p.h, p.hp, p.gp, p.g, p.do_ext != 0 || p.Hcur == nullptr);
  const int tid = threadIdx.x, NT = kMfmaThreads;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int s0 = blockIdx.x * kTS;
  const int L = p.L, h = p.h, g = p.g, hp = p.hp, gp = p.gp;
  const int nI = hp * kD, nJ = kD * gp;
  constexpr int ST = kTS / 16;
  constexpr int NWV = kMfmaThreads / 64;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
#define GLDS(gp_, lp_) __builtin_amdgcn_global_load_lds((gptr_t)(gp_), (lptr_t)(lp_), 4, 0, 0)
  constexpr int RPI = 64 / kTS;
  const int half = lane / kTS, col = lane % kTS;
  const bool haveH = p.Hprev && !p.first_ext && (p.do_f || p.do_ext);

  // ---- phase A: stage the per-sample operands and the extension core; padded operand arrays; env extension ----
  for (int c = wave; c < 3 * kTS * kD / 64; c += NWV) {
    const int which = c / (kTS * kD / 64), off = (c % (kTS * kD / 64)) * 64;
    const float *src = which == 0 ? p.x_km1 : (which == 1 ? p.x_k : p.x_kp1);
    if (src) GLDS(src + (size_t)s0 * kD + off + lane, w.sX + c * 64);
  }
  if (haveH)
    for (int pr = wave; pr < up(hp, RPI) / RPI; pr += NWV)
      GLDS(p.Hprev + (size_t)min(RPI * pr + half, hp - 1) * p.b_pad + s0 + col, w.rH + pr * 64);
  if (p.do_f && p.Gprev)
    for (int pr = wave; pr < up(gp, RPI) / RPI; pr += NWV)
      GLDS(p.Gprev + (size_t)min(RPI * pr + half, gp - 1) * p.b_pad + s0 + col, w.rGp + pr * 64);
  if (p.Gcur)
    for (int pr = wave; pr < up(g, RPI) / RPI; pr += NWV)
      GLDS(p.Gcur + (size_t)min(RPI * pr + half, g - 1) * p.b_pad + s0 + col, w.rG + pr * 64);
  if (p.do_ext) {
    const int nA = nI * h;
    for (int c = wave; c < up(nA, 64) / 64; c += NWV) {
      const int idx = min(c * 64 + lane, nA - 1);
      const int i = idx / h, o = idx - i * h;
      GLDS(p.ext_core.base + (i >> 1) * p.ext_core.s_in + (i & 1) * p.ext_core.s_d + o * p.ext_core.s_out, w.rA + c * 64);
    }
  }
  if (!p.do_f)
    for (int e = tid; e < L * kTS; e += NT) w.sF[e] = p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)];
  if (!p.do_ext)
    for (int e = tid; e < up(h, 16) * kTS; e += NT) {
      const int sl = e % kTS, hn = e / kTS;
      w.sH[hn * kTSP + sl] = hn < h ? (p.Hcur ? p.Hcur[(size_t)hn * p.b_pad + s0 + sl] : 1.0f) : 0.f;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float *sXm = w.sX, *sXk = w.sX + kTS * kD, *sXp = w.sX + 2 * kTS * kD;
  if (p.do_f || p.do_ext)
    for (int e = tid; e < dm.IP * kTS; e += NT) {       // P'[i][s] = H'[h'][s] x_{k-1}[s][d]
      const int sl = e % kTS, i = e / kTS;
      float v = 0.f;
      if (i < nI) v = (haveH ? w.rH[(i >> 1) * kTS + sl] : 1.0f) * sXm[sl * kD + (i & 1)];
      w.sPp[i * kTSP + sl] = v;
    }
  if (p.do_f)
    for (int e = tid; e < dm.JP * kTS; e += NT) {       // Q'[j][s] = x_k[s][d'] G'[g'][s],  j = d' * gp + g'
      const int sl = e % kTS, j = e / kTS;
      float v = 0.f;
      if (j < nJ) {
        const int dd = j >= gp ? 1 : 0;
        v = sXk[sl * kD + dd] * (p.Gprev ? w.rGp[(j - dd * gp) * kTS + sl] : 1.0f);
      }
      w.sQp[j * kTSP + sl] = v;
    }
  if (p.do_ext)
    for (int i = wave; i < dm.KA; i += NWV)
      for (int o = lane; o < dm.HS; o += 64)
        w.sA[i * dm.HS + o] = (i < nI && o < h) ? w.rA[i * h + o] : 0.f;
  for (int e = tid; e < dm.J3 * kTS; e += NT) {         // Q[j][s] = x_{k+1}[s][dk1] G[g][s],  j = dk1 * g + g_
    const int sl = e % kTS, j = e / kTS;
    float v = 0.f;
    if (j < kD * g) {
      const int dd = j >= g ? 1 : 0;
      v = sXp[sl * kD + dd] * (p.Gcur ? w.rG[(j - dd * g) * kTS + sl] : 1.0f);
    }
    w.sQ[j * kTSA + sl] = v;
  }
  __syncthreads();
  if (p.do_ext) {                                       // H[hn][s] = sum_i A[i][hn] P'[i][s], all waves
    const int HT = up(h, 16) / 16;
    for (int cidx = wave; cidx < HT * ST; cidx += NWV) {
      const int ht = cidx / ST, st = cidx % ST;
      const float *ap = w.sA + q * dm.HS + ht * 16 + r;
      const float *bq = w.sPp + q * kTSP + st * 16 + r;
      fvec4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < dm.KA / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * dm.HS], bq[kk * 4 * kTSP], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int hn = ht * 16 + 4 * q + reg;
        w.sH[hn * kTSP + st * 16 + r] = acc[reg];
        if (hn < h) p.Hcur[(size_t)hn * p.b_pad + s0 + st * 16 + r] = acc[reg];
      }
    }
  }
  __syncthreads();                                      // staging buffers and sA are dead: the chunk buffer takes over

  // ---- phase B: f of the previous step, B' streamed in chunks of 16 rows ----------------------------------------
  if (p.do_f) {
    constexpr int kMaxCombo = 4;                        // (label, sample tile) pairs per wave: L * ST <= 32
    float facc[kMaxCombo] = {0.f, 0.f, 0.f, 0.f};
    const int rowlen = nJ * L;
    for (int it = 0; it < dm.IP / 16; ++it) {
      // rows it*16 .. +15 of B'[i][(j, l)] into the padded layout (row stride RS); tails and rows >= nI are zero
      for (int rr = wave; rr < 16; rr += NWV) {
        const int i = it * 16 + rr;
        float *dst = w.sBc + rr * dm.RS;
        if (i < nI) {
          const float *src = p.Bprev + (size_t)i * rowlen;
          for (int x0 = 0; x0 < rowlen; x0 += 64)
            if (x0 + lane < rowlen) GLDS(src + x0 + lane, dst + x0);
          for (int x = rowlen + lane; x < dm.RS; x += 64) dst[x] = 0.f;
        } else {
          for (int x = lane; x < dm.RS; x += 64) dst[x] = 0.f;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < kMaxCombo; ++cc) {
        const int cidx = wave + cc * NWV;
        if (cidx >= L * ST) break;
        const int l = cidx / ST, st = cidx % ST;
        const float *bq = w.sQp + q * kTSP + st * 16 + r;
        const float *ap = w.sBc + r * dm.RS + q * L + l;                    // B'[i][(j, l)], j = 4 kk + q
        fvec4 acc = {0.f, 0.f, 0.f, 0.f};
// (run-time trip count: `#pragma unroll 5` here was refused by the optimizer -- "loop not unrolled" -- and is gone)
        for (int kk = 0; kk < dm.JP / 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * L], bq[kk * 4 * kTSP], acc, 0, 0, 0);
        const float *pp = w.sPp + (it * 16 + 4 * q) * kTSP + st * 16 + r;
        float fa = facc[cc];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) fa = fmaf(acc[reg], pp[reg * kTSP], fa);
        facc[cc] = fa;
      }
      __syncthreads();
    }
#pragma unroll
    for (int cc = 0; cc < kMaxCombo; ++cc) {
      const int cidx = wave + cc * NWV;
      if (cidx >= L * ST) break;
      const int l = cidx / ST, st = cidx % ST;
      float fa = facc[cc];
      fa += __shfl_xor(fa, 16);
      fa += __shfl_xor(fa, 32);
      if (q == 0) {
        w.sF[l * kTS + st * 16 + r] = fa;
        p.f[(size_t)l * p.b_pad + s0 + st * 16 + r] = fa;
      }
    }
    __syncthreads();
  }

  // ---- activation, metrics, loss derivative (one thread per sample) ------------------------------
  {
    __shared__ float sMet[3][kTS];
    if (tid < kTS) {
      const int s = s0 + tid;
      float m_abs = 0.f;
      int m_cor = 0, m_nf = 0;
      if (s < p.b) {
        act_and_lossder(w.sF + tid, kTS, w.sGl + tid, w.sGl + tid, kTSA, L, p.y[s], p.act_fn, p.loss_fn, p.T, m_abs,
                        m_cor, m_nf);
      } else {
        for (int l = 0; l < L; ++l) w.sGl[l * kTSA + tid] = 0.f;
      }
      sMet[0][tid] = (float)m_cor; sMet[1][tid] = m_abs; sMet[2][tid] = (float)m_nf;
    }
    // P[i][s] = H[h][s] x_k[s][d] (the A operand of the gradient GEMM)
    for (int e = tid; e < dm.I3 * kTS; e += NT) {
      const int sl = e % kTS, i = e / kTS;
      w.sP[i * kTSA + sl] = i < h * kD ? w.sH[(i >> 1) * kTSP + sl] * sXk[sl * kD + (i & 1)] : 0.f;
    }
    __syncthreads();
    if (tid < 3) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < kTS; ++i) t += sMet[tid][i];
      p.slabs[(size_t)blockIdx.x * p.slab_stride + p.bsize + tid] = t;
    } else if (tid == 3) {
      const int valid = p.b - s0;
      p.slabs[(size_t)blockIdx.x * p.slab_stride + p.bsize + 3] = (float)(valid < 0 ? 0 : (valid > kTS ? kTS : valid));
    }
  }

  // ---- partial bond gradient: dB[i, n] = sum_s P[i, s] (Q[j, s] g[l, s]),  n = j * L + l --------------------------
  {
    const int IT = dm.I3 / 16, QW = kD * g, NN = QW * L, NTL = (NN + 15) / 16;
    float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
    for (int cidx = wave; cidx < IT * NTL; cidx += NWV) {
      const int it = cidx % IT, nt = cidx / IT;
      const int n = nt * 16 + r;
      const int nc = min(n, NN - 1);
      const int j = nc / L, l = nc - j * L;
      // The three operands are read as [row(lane & 15)][4 kk + (lane >> 4)]: a 32-lane bank group holds 16 rows x 2 consecutive
      // samples, which sit on distinct banks when the row stride is 2 mod 32 (kTSA = 34).  At the odd stride kTSP = 33 of the other
      // arrays rows r, q = 1 and r + 1, q = 0 collide (2-way); the label rows of the loss derivative at stride 32 all shared their
      // banks (10-way at ten labels): SQ_LDS_BANK_CONFLICT was 74 % of SQ_LDS_IDX_ACTIVE, half of the kernel's cycles.
      const float *ap = w.sP + (it * 16 + r) * kTSA + q;
      const float *bq = w.sQ + j * kTSA + q;
      const float *gq = w.sGl + l * kTSA + q;
      fvec4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < kTS / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4], bq[kk * 4] * gq[kk * 4], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = it * 16 + 4 * q + reg;
        if (i < h * kD && n < NN) slab[(size_t)i * NN + n] = acc[reg];
      }
    }
  }
#undef GLDS
}

size_t wide_tiled_lds_bytes(int L, int h, int hp, int gp, int g, bool ext) {
  alignas(16) static float origin[4];
  WideMfmaDims dt = wide_mfma_dims(hp, gp, h, g);
  if (L * (kTS / 16) > 4 * (kMfmaThreads / 64)) return 0;
  return wide_tiled_carve(origin, dt, L, h, hp, gp, g, ext).floats * sizeof(float);
}
void launch_wide_tiled(const WideParams &p, int nblk, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(wide_step_mfma_tiled_kernel, dim3(nblk), dim3(kMfmaThreads), lds_bytes, st, p);
}

// false: the operands of a sample tile fit neither MFMA kernel's LDS budget (bond dimensions beyond what this build handles)
bool launch_wide(const WideParams &p, int nblk, const PrepParams *prep, hipStream_t st, bool *prep_done) {
  *prep_done = false;
  WideMfmaDims dm = wide_mfma_dims(p.hp, p.gp, p.h, p.g);
  alignas(16) static float origin[4];                 // only distances from it are used (arithmetic on a null pointer is undefined)
  size_t lds = wide_mfma_carve(origin, dm, p.L, p.h, p.hp, p.gp, p.g).floats * sizeof(float);
  if (lds <= 160 * 1024) {
    PrepParams q{};
    int extra = 0;
    if (prep) {
      const size_t need = prep_slice_lds_bytes(prep->h, prep->g, prep->s, prep->L);
      if (need <= 160 * 1024) { q = *prep; extra = kD * kD; if (need > lds) lds = need; }
    }
    hipLaunchKernelGGL(wide_step_mfma_kernel, dim3(nblk + extra), dim3(kMfmaThreads), lds, st, p, q, nblk);
    *prep_done = extra > 0;
    return true;
  }
  // operands of a tile exceed LDS: stream the merged tensor through it in chunks
  WideMfmaDims dt = wide_mfma_dims(p.hp, p.gp, p.h, p.g);
  const size_t ldst = wide_tiled_carve(origin, dt, p.L, p.h, p.hp, p.gp, p.g).floats * sizeof(float);
  if (ldst <= 160 * 1024 && p.L * (kTS / 16) <= 4 * (kMfmaThreads / 64)) {
    hipLaunchKernelGGL(wide_step_mfma_tiled_kernel, dim3(nblk), dim3(kMfmaThreads), ldst, st, p);
    return true;
  }
  const int hmax = p.h > p.hp ? p.h : p.hp;
  const int gmax = p.g > p.gp ? p.g : p.gp;
  const int core_elems = p.do_ext ? p.ext_core.n_in * kD * p.ext_core.n_out : 0;
  const size_t ldsv = wide_lds_bytes(hmax, gmax, p.L, core_elems);
  if (ldsv > 160 * 1024) return false;
  hipLaunchKernelGGL(wide_step_kernel, dim3(nblk), dim3(kWideThreads), ldsv, st, p, hmax, gmax, core_elems);
  return true;
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
