// Batch-parallel half of the PIPELINED sweep step (kernels_narrow.hip: step_pipe_kernel).
//
// The dependency chain of the classic step is  wide(k) -> reduction -> update(k) -> SVD(k) -> wide(k+1): the batch-sized
// work waits for the SVD and the SVD waits for the batch-sized work.  Here the bond gradient is taken BEFORE the
// extension of the behind environment with the core the SVD produces:
//
//   dB_{k+1}[h', d, d', g, l] = sum_s gl[l,s] E_{k+1}[s,h'] x_{k+1}[s,d] x_{k+2}[s,d'] G_{k+1}[s,g]          (Network_class.py:710)
//   E_{k+1}[s, h']            = sum_{h,dk} E_k[s,h] x_k[s,dk] A_k[(h,dk), h']                                (:637-652)
//   =>  dB_{k+1} = A_k^T . Z_{k+1},   Z_{k+1}[(h,dk), d, d', g, l] = sum_s gl[l,s] (E_k[s,h] x_k[s,dk]) x_{k+1}[s,d] x_{k+2}[s,d'] G_{k+1}[s,g]
//
// Z_{k+1} needs f of step k (from the updated, un-truncated merged tensor, available BEFORE the SVD of step k) and
// environments that exist already -- not the new core A_k.  So the workgroups of this file run inside the launch of
// step k, next to the workgroup that performs the SVD: they extend E_k (with A_{k-1}, a product of the previous
// launch), wait for B_new(k) on a flag, form f, activation, loss derivative and metrics, accumulate Z_{k+1} over their
// samples and reduce the partial tensors in two fixed-order levels (last arriver of a group, then last group); the next
// launch starts from the reduced Z and contracts it with A_k (2 h x h doubles of work).  The batch-sized work is off the
// critical path as long as it is shorter than the SVD.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", first table row): every handed-off byte is stored with an
// agent-scope (sc1) store and loaded with an agent-scope (sc1) load; every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane signals (flag store or returning atomic add); the
// consumer polls / learns from the returned count, then meets at a barrier before any of its loads.
#pragma once
#include "tnml_internal.h"
#include "act_device.h"
#include "jacobi_device.h"

namespace tnml {

typedef float fvec4 __attribute__((ext_vector_type(4)));
constexpr int kPipeThreads = 1024;      // the launch is shared with the narrow workgroup
constexpr int kPTSP = kTS + 1;          // sample stride of the LDS operand arrays (bank spread)
constexpr int kPipeGroupMax = 16;       // members of a reduction group (and number of groups): up to 256 batch-side workgroups

__host__ __device__ inline int upm(int x, int m) { return (x + m - 1) / m * m; }

__device__ inline void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Everything one batch-side workgroup needs, in the sweep-relative frame.  "j" is the step whose updated tensor B_new(j)
// yields f; Z is the pre-gradient of step j+1.
struct WidePipeParams {
  int b, b_pad, L;
  int hj, gj;            // behind / ahead bond of B_new(j)             (first: 1, unused)
  int gn;                // ahead bond of step j+1
  int hprev;             // bond of E_{j-1} (do_ext)
  int first;             // j == -1: start of a sweep, no environment and no site j: Z rows are (d_0) only
  int do_ext;            // E_j = ext(E_{j-1}, x_{j-1}, A_{j-1}) is computed and stored (else E_j is read; nullptr == 1)
  int first_ext;         // E_{j-1} is the scalar 1
  int do_f;              // f from B_new(j) (else f is read from p.f)
  int wait_flag;         // B_new(j) is produced by the narrow workgroup of THIS launch: poll p.flag for p.token first
  int do_z;              // Z_{j+1} and the metric partials are produced
  int act_fn, loss_fn;
  float T;
  const float *x_jm1, *x_j, *x_jp1, *x_jp2;   // [b_pad][D]
  const float *Eprev;    // [hprev][b_pad]
  float *Ecur;           // [hj][b_pad]
  CoreView ext_core;     // A_{j-1}(hprev, d, hj)
  const float *Gj;       // ahead environment of step j    [gj][b_pad]   (nullptr == 1)
  const float *Gn;       // ahead environment of step j+1  [gn][b_pad]   (nullptr == 1)
  const float *Bnew;     // [hj][D][D][gj][L]
  const int *y;
  float *f;              // [L][b_pad]
  // partial tensors and their two-level reduction
  int zsize;             // nI * D * D * gn * L,  nI = first ? 1 : hj * D   (+ kMetricSlots behind it)
  int slab_stride;
  float *slabs;          // [nwide][slab_stride]
  float *gslabs;         // [ngroups][slab_stride]
  float *zred;           // [slab_stride]   reduced Z + metric tail, read by the next launch
  unsigned *gcnt;        // [ngroups] arrival counters, zero between launches
  unsigned *tcnt;        // top-level arrival counter
  int nwide, gsz, ngroups;
  int one_level;         // small pre-gradient: the last arriver of the single group sums all partials through its LDS (host checked the room)
  int wg0;               // blockIdx of the first batch-side workgroup
  int tiles_per_wg;      // sample tiles (kTS samples) a workgroup accumulates before it writes its partial tensor
  int ntiles;            // b_pad / kTS
  const unsigned *flag;  // set to `token` by the narrow workgroup once B_new(j) is stored
  unsigned token;
  int *status;
  // persistent sweep (sweep_persist_kernel): the workgroup loops over the steps; producers and consumers of the same launch
  int persist;
  const unsigned *coreflag;   // >= corewant: the extension core A_{j-1} is stored (agent scope) -- read with agent-scope loads
  unsigned corewant;
  unsigned *zready;           // set to zpublish once the reduced pre-gradient is stored (agent scope)
  unsigned zpublish;
  unsigned *abort_flag;
  double *stamps;             // diagnostics (batch-side workgroup 0 of the stamped step): 100 MHz real-time stamps
};

// Bounded wait of ONE lane until *flag >= want (flags of a persistent sweep only grow).  0: seen, 1: timed out, 2: another workgroup
// gave up first (abort word set).  Relaxed agent-scope polls; the caller meets its workgroup at a barrier before any load of the
// handed-off data.
__device__ inline int spin_wait_ge(const unsigned *flag, unsigned want, const unsigned *abort_flag, int tight = 0) {
  for (int spins = 0; spins < (1 << 19); ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return 0;
    if ((spins & 63) == 63 && abort_flag && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 2;
    if (tight) __builtin_amdgcn_s_sleep(1);      // the only poller of this word
    else __builtin_amdgcn_s_sleep(24);           // ~1500 cycles between polls: ~90 workgroups poll the same few words
  }
  return 1;
}

// the same for two flags at once (both polls in flight together)
__device__ inline int spin_wait_ge2(const unsigned *f1, unsigned w1, const unsigned *f2, unsigned w2, const unsigned *abort_flag) {
  for (int spins = 0; spins < (1 << 19); ++spins) {
    const unsigned a = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a >= w1 && b >= w2) return 0;
    if ((spins & 63) == 63 && abort_flag && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 2;
    __builtin_amdgcn_s_sleep(24);                // ~1500 cycles between polls: ~90 workgroups poll the same few words
  }
  return 1;
}

struct WidePipeDims { int nI, IP, JP, RS, KA, HS, I3, J3, EH; };
template <class WP>
__host__ __device__ inline WidePipeDims wide_pipe_dims(const WP &p) {
  WidePipeDims d;
  d.nI = p.first ? 1 : p.hj * kD;            // rows i = (h_j, d_j) of B_new(j) and of P'
  d.IP = upm(d.nI, 16);
  d.JP = upm(kD * p.gj, 4);                  // inner index jj = (d_{j+1}, g_j) of the f product
  d.RS = (d.JP * p.L) | 1;                   // row stride of B'[i][(jj, l)]: odd -> 16 rows on 16 banks
  d.KA = upm(p.hprev * kD, 4);               // inner index of the environment product
  d.EH = upm(p.hj, 16);
  d.HS = d.EH + 1;                           // row stride of the extension core A[i''][hn]
  d.I3 = upm(d.nI * kD, 16);                 // rows (i, d_{j+1}) of Z
  d.J3 = upm(kD * p.gn, 16);                 // columns (d_{j+2}, g) of Z
  return d;
}

struct WidePipeSmem { float *sX, *sF, *sGl, *sPpp, *sA, *sE, *sPp, *sQp, *sBp, *sPg, *sQn, *sFp, *sMet; size_t floats; };
template <class WP>
__host__ __device__ inline WidePipeSmem wide_pipe_carve(float *base, const WP &p, const WidePipeDims &d) {
  WidePipeSmem w;
  float *q = base;
  w.sX = q; q += 4 * kTS * kD;                                  // features of sites j-1, j, j+1, j+2
  w.sF = q; q += p.L * kTS;
  w.sGl = q; q += p.L * kTS;
  w.sPpp = q; q += (p.do_ext ? d.KA : 0) * kPTSP;               // P''[i''][s] = E_{j-1}[h''][s] x_{j-1}[s][d'']
  w.sA = q; q += (p.do_ext ? d.KA * d.HS : 0);                  // extension core A[i''][hn], zero padded
  w.sE = q; q += d.EH * kPTSP;                                  // E_j[hn][s]
  w.sPp = q; q += d.IP * kPTSP;                                 // P'[i][s] = E_j[h][s] x_j[s][d]
  w.sQp = q; q += (p.do_f ? d.JP : 0) * kPTSP;                  // Q'[jj][s] = x_{j+1}[s][d'] G_j[g'][s]
  w.sBp = q; q += (p.do_f ? (size_t)d.IP * d.RS : 0);           // B'[i][(jj, l)]
  w.sPg = q; q += (p.do_z ? (size_t)p.L * d.I3 * kPTSP : 0);    // gl[l][s] P'[i][s] x_{j+1}[s][d]
  w.sQn = q; q += (p.do_z ? d.J3 * kPTSP : 0);                  // Qn[jn][s] = x_{j+2}[s][d'] G_{j+1}[g][s]
  w.sFp = q; q += (p.do_f ? (size_t)(d.IP / 16) * p.L * kTS : 0);   // partial f per 16-row block of B'
  w.sMet = q; q += 4 * kTS;
  w.floats = (size_t)(q - base);
  return w;
}
inline size_t wide_pipe_lds_bytes(const WidePipeParams &p) {
  const WidePipeDims d = wide_pipe_dims(p);
  alignas(16) static float origin[4];                 // only distances from it are used (arithmetic on a null pointer is undefined)
  return wide_pipe_carve(origin, p, d).floats * sizeof(float) + 16;
}
constexpr int kPipeMaxZT = 4;           // 16 x 16 tiles of Z a wave accumulates in registers
// number of 16 x 16 tiles of Z_l over all labels: must not exceed 16 waves x kPipeMaxZT
inline int wide_pipe_ztiles(const WidePipeParams &p) {
  const WidePipeDims d = wide_pipe_dims(p);
  return p.L * (d.I3 / 16) * (d.J3 / 16);
}
// grid = w.wg0 + w.nwide workgroups of 1024 threads (kernels_narrow.hip)
void launch_step_pipe(const NarrowParams &p, const WidePipeParams &w, size_t lds_bytes, hipStream_t st);
// the update side alone: workgroups 0 .. w.wg0 - 1 (the batch-side workgroups of the same step are launched on another stream with w.wg0 = 0)
void launch_step_pipe_update(const NarrowParams &p, const WidePipeParams &w, size_t lds_bytes, hipStream_t st);

// persistent sweep (kernels_narrow.hip: sweep_persist_kernel): what a helper workgroup needs for step k, and one record per step
struct PersistHelperParams {
  int zr, s, g, L, h;        // rows of T_k (D * behind bond of step k-1; 1 at k == 0), shared / ahead / behind bond of step k
  int l2_flag;
  const float *W;            // B_new(k-1) (contiguous [zr][D][s][L]), or nullptr at k == 0: the label core `lab` is W
  CoreView lab, pl;
  const double *Ng;          // ahead norm environment of step k (nullptr == 1)
  double *T, *TN;            // T_k, T_k . Ng   [zr][D][D][g][L] float64
  const float *Z;            // reduced pre-gradient Z_k [zr][RW]
  float *prepRaw, *prepB;    // out [h][RW]
  double *prepG;
  const double *Apub;        // A' [zr][h], 1 / sigma [h], Nh [h][h] of step k-1 (update workgroup)
  const unsigned *flag;      // >= want: B_new(k-1) is stored (want == 0: nothing to wait for)
  unsigned want;
  const unsigned *aflag;     // >= awant: Apub is stored (awant == 0 at k == 0: the identity)
  unsigned awant;
  const unsigned *zready;    // >= zwant: Z_k is reduced
  unsigned zwant;
  unsigned *tcnt, *pcnt;     // arrival counters of this step: part 1 (T slices), part 2 (projections)
  unsigned *abort_flag;
  int *status;
  double *stamps;            // diagnostics (helper 0 of the stamped step): 100 MHz real-time stamps of part 2
};
struct PersistStep { NarrowParams n; WidePipeParams w; PersistHelperParams t; };
constexpr int kPersistHelpers = 8;
inline size_t persist_helper_lds_bytes(int zr, int s, int g, int L, int h, int nH) {
  const size_t DG = (size_t)kD * g, RW = kD * DG * L;
  const size_t per = ((size_t)zr * kD + nH - 1) / nH, cw = (RW + nH - 1) / nH;
  const size_t p1 = (((per * s * L + 3) & ~(size_t)3) + (((size_t)s * DG + 3) & ~(size_t)3)) * sizeof(float) +
                    ((((size_t)g * g + 1) & ~(size_t)1) + 2 * per * DG * L) * sizeof(double);
  const size_t p2 = ((size_t)zr * h + ((h + 1) & ~1) + (((size_t)h * h + 1) & ~(size_t)1) + 3 * (size_t)zr * cw + (size_t)h * cw) * sizeof(double) +
                    ((size_t)zr * cw + (size_t)zr * h) * sizeof(float);
  return (p1 > p2 ? p1 : p2) + 64;
}
// steps_dev: n_steps + 1 records, the last one carrying the prologue of the batch side in its `w`
void launch_sweep_persist(const PersistStep *steps_dev, int n_steps, int n_helpers, int grid, size_t lds_bytes, hipStream_t st);
// one launch per role on three streams (kernels_narrow.hip)
void launch_sweep_persist_split(const PersistStep *steps_dev, int n_steps, int n_helpers, int n_wide, size_t lds_update, size_t lds_helper,
                                size_t lds_wide, int rec_off, hipStream_t st_update, hipStream_t st_helper, hipStream_t st_wide);


// ------------------------------------------------------------------------------------------------------------------
// One batch-side workgroup (1024 threads, 16 waves).  MFMA lane maps (v_mfma_f32_16x16x4_f32): A[row = lane & 15]
// [k = lane >> 4], B[k = lane >> 4][col = lane & 15], C/D col = lane & 15, row = 4 (lane >> 4) + reg.
// ------------------------------------------------------------------------------------------------------------------
// returns true when the workgroup gave up (persistent sweep: a wait timed out here or elsewhere)
template <class WP>
__device__ __forceinline__ bool wide_pipe_block(const WP &p, float *smem) {
  const WidePipeDims dm = wide_pipe_dims(p);
  const WidePipeSmem w = wide_pipe_carve(smem, p, dm);
  const int tid = threadIdx.x, NT = kPipeThreads;
  const int lane = tid & 63, NWV = NT / 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: tile loops on it become SALU control flow
  const int r = lane & 15, q = lane >> 4;
  const int L = p.L, hj = p.hj, gj = p.gj, gn = p.gn, hprev = p.hprev;
  const int nI = dm.nI, nJ = kD * gj;
  constexpr int ST = kTS / 16;
  const int wg = (int)blockIdx.x - p.wg0;
  const int IT3 = dm.I3 / 16, JT3 = dm.J3 / 16, QW = kD * gn;
  // accumulators of Z: tile t = wave + u * NWV of the L * IT3 * JT3 tiles, kept across the sample tiles of this workgroup
  constexpr int kMaxZT = kPipeMaxZT;        // tiles per wave the launcher guarantees (L * IT3 * JT3 <= 16 * kMaxZT)
  fvec4 zacc[kMaxZT];
#pragma unroll
  for (int u = 0; u < kMaxZT; ++u) zacc[u] = fvec4{0.f, 0.f, 0.f, 0.f};
  float met[4] = {0.f, 0.f, 0.f, 0.f};      // thread 0..3: running metric sums of this workgroup

  bool flag_seen = !p.wait_flag;
  __shared__ int sGiveUp;
  double *bst = (p.persist && p.stamps && blockIdx.x == (unsigned)p.wg0 && tid == 0) ? p.stamps : nullptr;
  auto rt = []() { return (double)(__builtin_amdgcn_s_memrealtime() & ((1ull << 40) - 1)); };
  if (bst) bst[32] = rt();
  if (p.persist) {
    // the extension core A_{j-1} is written by the update workgroup of THIS launch (end of its step j-1)
    if (tid == 0) {
      const int bad = p.do_ext ? spin_wait_ge(p.coreflag, p.corewant, p.abort_flag) : 0;
      if (bad == 1) { atomicOr(p.status, 8); __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      sGiveUp = bad;
    }
    lds_barrier();
    if (sGiveUp) return true;
    if (bst) bst[33] = rt();
  }
  for (int tt = 0; tt < p.tiles_per_wg; ++tt) {
    const int tile = wg + tt * p.nwide;
    if (tile >= p.ntiles) break;            // block-uniform
    const int s0 = tile * kTS;
    if (tt > 0) lds_barrier();            // the previous tile's operand arrays are dead
    // ---- stage 1: everything that does not need B_new(j) ------------------------------------------------------------
    for (int e = tid; e < 4 * kTS * kD; e += NT) {
      const int which = e / (kTS * kD), rr = e % (kTS * kD);
      const float *src = which == 0 ? p.x_jm1 : (which == 1 ? p.x_j : (which == 2 ? p.x_jp1 : p.x_jp2));
      w.sX[e] = src ? src[(size_t)s0 * kD + rr] : 0.f;
    }
    if (!p.do_f)
      for (int e = tid; e < L * kTS; e += NT) w.sF[e] = p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)];
    if (p.do_ext) {
      for (int e = tid; e < dm.KA * dm.HS; e += NT) {             // A[i''][hn], zero padded
        const int o = e % dm.HS, i = e / dm.HS;
        float v = 0.f;
        if (i < hprev * kD && o < hj) {
          const float *src = p.ext_core.base + ((i >> 1) * p.ext_core.s_in + (i & 1) * p.ext_core.s_d + o * p.ext_core.s_out);
          v = p.persist ? ld_sc1(src) : *src;
        }
        w.sA[e] = v;
      }
    } else {
      for (int e = tid; e < dm.EH * kTS; e += NT) {               // E_j read back (or 1)
        const int sl = e % kTS, hn = e / kTS;
        w.sE[hn * kPTSP + sl] = hn < hj ? ((p.Ecur && !p.first) ? p.Ecur[(size_t)hn * p.b_pad + s0 + sl] : 1.0f) : 0.f;
      }
    }
    lds_barrier();
    const float *sXm = w.sX, *sXj = w.sX + kTS * kD, *sXp = w.sX + 2 * kTS * kD, *sXq = w.sX + 3 * kTS * kD;
    if (p.do_ext) {
      for (int e = tid; e < dm.KA * kTS; e += NT) {               // P''[i''][s]
        const int sl = e % kTS, i = e / kTS;
        float v = 0.f;
        if (i < hprev * kD) v = (p.first_ext ? 1.0f : p.Eprev[(size_t)(i >> 1) * p.b_pad + s0 + sl]) * sXm[sl * kD + (i & 1)];
        w.sPpp[i * kPTSP + sl] = v;
      }
      lds_barrier();
      const int HT = dm.EH / 16;
      for (int cidx = wave; cidx < HT * ST; cidx += NWV) {        // E_j[hn][s] = sum_i'' A[i''][hn] P''[i''][s]
        const int ht = cidx / ST, st = cidx % ST;
        const float *ap = w.sA + q * dm.HS + ht * 16 + r;
        const float *bq = w.sPpp + q * kPTSP + st * 16 + r;
        fvec4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < dm.KA / 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * dm.HS], bq[kk * 4 * kPTSP], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int hn = ht * 16 + 4 * q + reg;
          w.sE[hn * kPTSP + st * 16 + r] = acc[reg];             // rows >= hj of the padded tile are zero
          if (hn < hj) p.Ecur[(size_t)hn * p.b_pad + s0 + st * 16 + r] = acc[reg];
        }
      }
      lds_barrier();
    }
    for (int e = tid; e < dm.IP * kTS; e += NT) {                 // P'[i][s] = E_j[h][s] x_j[s][d]   (first: the scalar 1)
      const int sl = e % kTS, i = e / kTS;
      float v = 0.f;
      if (i < nI) v = p.first ? 1.0f : w.sE[(i >> 1) * kPTSP + sl] * sXj[sl * kD + (i & 1)];
      w.sPp[i * kPTSP + sl] = v;
    }
    if (p.do_f)
      for (int e = tid; e < dm.JP * kTS; e += NT) {               // Q'[jj][s] = x_{j+1}[s][d'] G_j[g'][s]
        const int sl = e % kTS, jj = e / kTS;
        float v = 0.f;
        if (jj < nJ) {
          const int dd = jj >= gj ? 1 : 0;
          v = sXp[sl * kD + dd] * (p.Gj ? p.Gj[(size_t)(jj - dd * gj) * p.b_pad + s0 + sl] : 1.0f);
        }
        w.sQp[jj * kPTSP + sl] = v;
      }
    if (p.do_z)
      for (int e = tid; e < dm.J3 * kTS; e += NT) {               // Qn[jn][s] = x_{j+2}[s][d'] G_{j+1}[g][s]
        const int sl = e % kTS, jn = e / kTS;
        float v = 0.f;
        if (jn < QW) {
          const int dd = jn >= gn ? 1 : 0;
          v = sXq[sl * kD + dd] * (p.Gn ? p.Gn[(size_t)(jn - dd * gn) * p.b_pad + s0 + sl] : 1.0f);
        }
        w.sQn[jn * kPTSP + sl] = v;
      }
    // ---- stage 2: f of step j from its updated, un-truncated merged tensor (Network_class.py:494-523) --------------
    if (p.do_f) {
      if (!flag_seen) {
        if (tid == 0) {
          if (p.persist) {
            const int bad = spin_wait_ge(p.flag, p.token, p.abort_flag);
            if (bad == 1) { atomicOr(p.status, 8); __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            sGiveUp = bad;
          } else {
            int spins = 0;
            while (__hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.token && spins < (1 << 21)) {
              __builtin_amdgcn_s_sleep(16);
              ++spins;
            }
            if (spins >= (1 << 21)) atomicOr(p.status, 8);
          }
        }
        flag_seen = true;
      }
      lds_barrier();                                            // the poll is over; P', Q', Qn are complete
      if (p.persist && sGiveUp) return true;
      if (bst && tt == 0) bst[34] = rt();
      const int rowlen = nJ * L;
      if (tt == 0) {                                              // B'[i][(jj, l)] at the odd row stride, zero padded
        if (p.wait_flag && (rowlen & 3) == 0) {
          // produced by the update workgroup of THIS launch: 16-byte agent-scope loads (rows are rowlen contiguous floats),
          // up to four in flight per lane; the padding of the odd row stride and of the rows beyond nI is zeroed separately
          const __amdgpu_buffer_rsrc_t rB = sc1_rsrc(p.Bnew);
          const int pieces = rowlen >> 2;
          for (int i0 = wave; i0 < nI; i0 += 4 * NWV) {
            tn_uvec4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = i0 + u * NWV;
              if (i < nI && lane < pieces) v[u] = ld_sc1_b128(rB, (unsigned)((i * rowlen + 4 * lane) * sizeof(float)));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = i0 + u * NWV;
              if (i < nI && lane < pieces) {
                float *d = w.sBp + i * dm.RS + 4 * lane;
                d[0] = __uint_as_float(v[u].x); d[1] = __uint_as_float(v[u].y); d[2] = __uint_as_float(v[u].z); d[3] = __uint_as_float(v[u].w);
              }
            }
          }
          for (int i = wave; i < dm.IP; i += NWV)
            for (int x = lane; x < dm.RS; x += 64)
              if (i >= nI || x >= rowlen || x >= 256) w.sBp[i * dm.RS + x] = (i < nI && x < rowlen) ? ld_sc1(p.Bnew + i * rowlen + x) : 0.f;
        } else {
          for (int i = wave; i < dm.IP; i += NWV)
            for (int x = lane; x < dm.RS; x += 64)
              w.sBp[i * dm.RS + x] = (i < nI && x < rowlen) ? (p.wait_flag ? ld_sc1(p.Bnew + i * rowlen + x) : p.Bnew[i * rowlen + x]) : 0.f;
        }
        lds_barrier();
      }
      const int ITF = dm.IP / 16;
      for (int cidx = wave; cidx < L * ST * ITF; cidx += NWV) {   // T_l[i][s] = sum_jj B'_l[i][jj] Q'[jj][s];  f += P' . T
        const int it = cidx % ITF, rest = cidx / ITF;
        const int l = rest / ST, st = rest % ST;
        const float *bq = w.sQp + q * kPTSP + st * 16 + r;
        const float *ap = w.sBp + (it * 16 + r) * dm.RS + q * L + l;
        fvec4 acc = {0.f, 0.f, 0.f, 0.f};
// (run-time trip count: `#pragma unroll 5` here was refused by the optimizer -- "loop not unrolled" -- and is gone)
        for (int kk = 0; kk < dm.JP / 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * L], bq[kk * 4 * kPTSP], acc, 0, 0, 0);
        const float *pp = w.sPp + (it * 16 + 4 * q) * kPTSP + st * 16 + r;
        float facc = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) facc = fmaf(acc[reg], pp[reg * kPTSP], facc);
        facc += __shfl_xor(facc, 16);
        facc += __shfl_xor(facc, 32);
        if (q == 0) w.sFp[(it * L + l) * kTS + st * 16 + r] = facc;
      }
      lds_barrier();
      for (int e = tid; e < L * kTS; e += NT) {                   // partial sums in block order (deterministic)
        float t = 0.f;
        for (int it = 0; it < ITF; ++it) t += w.sFp[it * L * kTS + e];
        w.sF[e] = t;
        p.f[(size_t)(e / kTS) * p.b_pad + s0 + (e % kTS)] = t;
      }
    }
    lds_barrier();
    if (!p.do_z) continue;
    // ---- stage 3: activation, metrics, loss derivative (one thread per sample) ------------------------------------------
    if (tid < kTS) {
      const int s = s0 + tid;
      float m_abs = 0.f;
      int m_cor = 0, m_nf = 0;
      if (s < p.b) {
        act_and_lossder(w.sF + tid, kTS, w.sGl + tid, w.sGl + tid, kTS, L, p.y[s], p.act_fn, p.loss_fn, p.T, m_abs, m_cor, m_nf);
      } else {
        for (int l = 0; l < L; ++l) w.sGl[l * kTS + tid] = 0.f;  // padded samples carry no gradient
      }
      w.sMet[tid] = (float)m_cor; w.sMet[kTS + tid] = m_abs; w.sMet[2 * kTS + tid] = (float)m_nf;
    }
    lds_barrier();
    if (tid < 3) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < kTS; ++i) t += w.sMet[tid * kTS + i];  // sample order: deterministic
      met[tid] += t;
    } else if (tid == 3) {
      const int valid = p.b - s0;
      met[3] += (float)(valid < 0 ? 0 : (valid > kTS ? kTS : valid));
    }
    // ---- stage 4: (gl P' x_{j+1})[l][ii][s],  ii = i * D + d -------------------------------------------------------------
    for (int l = 0; l < L; ++l)
      for (int e = tid; e < dm.I3 * kTS; e += NT) {
        const int sl = e % kTS, ii = e / kTS;
        float v = 0.f;
        if (ii < nI * kD) v = w.sGl[l * kTS + sl] * w.sPp[(ii >> 1) * kPTSP + sl] * sXp[sl * kD + (ii & 1)];
        w.sPg[((size_t)l * dm.I3 + ii) * kPTSP + sl] = v;
      }
    lds_barrier();
    // ---- stage 5: Z_l[ii][jn] += sum_s Pg_l[ii][s] Qn[jn][s] -------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < kMaxZT; ++u) {
      const int cidx = wave + u * NWV;
      if (cidx < L * IT3 * JT3) {
        const int jt = cidx % JT3, t = cidx / JT3;
        const int it = t % IT3, l = t / IT3;
        const float *ap = w.sPg + ((size_t)l * dm.I3 + it * 16 + r) * kPTSP + q;
        const float *bq = w.sQn + (jt * 16 + r) * kPTSP + q;
        fvec4 acc = zacc[u];
#pragma unroll
        for (int kk = 0; kk < kTS / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4], bq[kk * 4], acc, 0, 0, 0);
        zacc[u] = acc;
      }
    }
  }
  if (!p.do_z) return false;
  if (bst) bst[35] = rt();

  // ---- partial tensor of this workgroup -> slab (agent-scope stores), then the two-level fixed-order reduction ----------
  float *slab = p.slabs + (size_t)wg * p.slab_stride;
  // through LDS when it fits (it does for every shape the launcher admits at two labels): the partial leaves as 16-byte
  // agent-scope stores -- a scalar sc1 store is one fabric write per element (MI355X_MICROARCH.md)
  const bool staged = (size_t)(p.zsize + kMetricSlots) <= w.floats && (p.zsize & 3) == 0;
  if (staged) lds_barrier();                                     // the operand arrays of the last tile are dead
#pragma unroll
  for (int u = 0; u < kMaxZT; ++u) {
    const int cidx = wave + u * NWV;
    if (cidx < L * IT3 * JT3) {
      const int jt = cidx % JT3, t = cidx / JT3;
      const int it = t % IT3, l = t / IT3;
      const int jn = jt * 16 + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ii = it * 16 + 4 * q + reg;
        if (ii < nI * kD && jn < QW) {
          if (staged) smem[((size_t)ii * QW + jn) * L + l] = zacc[u][reg];
          else st_sc1(slab + ((size_t)ii * QW + jn) * L + l, zacc[u][reg]);
        }
      }
    }
  }
  if (staged) {
    if (tid < 4) smem[p.zsize + tid] = met[tid];
    lds_barrier();
    const __amdgpu_buffer_rsrc_t rS = sc1_rsrc(slab);
    for (int e = tid; e < (p.zsize + kMetricSlots) / 4; e += NT)
      st_sc1_b128(rS, (unsigned)(16 * e), *reinterpret_cast<const tn_uvec4 *>(smem + 4 * e));
  } else if (tid < 4) st_sc1(slab + p.zsize + tid, met[tid]);
  const int n = p.zsize + kMetricSlots;
  __shared__ unsigned sTicket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  const int grp = wg / p.gsz;
  const int g_lo = grp * p.gsz, g_n = min(p.gsz, p.nwide - g_lo);
  if (tid == 0) sTicket = __hip_atomic_fetch_add(p.gcnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lds_barrier();
  if (bst) bst[36] = rt();
  if (sTicket != (unsigned)(g_n - 1)) return false;               // not the last arriver of its group
  // The last arriver reads what the other workgroups stored: one agent-scope acquire (invalidates this CU's L1 and the
  // non-coherent lines of its L2), drained, then a barrier, then ordinary 16-byte loads, all of an element's summands
  // in flight together (a group has at most kPipeGroupMax members).
  auto acquire_all = [&]() {
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
  };
  // persistent sweep: the reduced tensor is read by the update workgroup of the same launch -- agent-scope stores, drained, then
  // the ready flag
  auto publish_zred = [&]() {
    if (!p.persist) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tid == 0) __hip_atomic_store(p.zready, p.zpublish, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.stamps && tid == 0) p.stamps[37] = rt();
  };
  auto sum_slabs = [&](const float *src, int count, float *dst, bool publish) {
    const int n4 = (n + 3) / 4;
    for (int e = tid; e < n4; e += NT) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k0 = 0; k0 < kPipeGroupMax; k0 += 8) {              // eight 16-byte loads in flight, summed in slab order
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          v[k] = k0 + k < count ? *reinterpret_cast<const float4 *>(src + (size_t)(k0 + k) * p.slab_stride + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
      }
      if (publish) { tn_uvec4 o; o.x = __float_as_uint(a.x); o.y = __float_as_uint(a.y); o.z = __float_as_uint(a.z); o.w = __float_as_uint(a.w); st_sc1_b128(sc1_rsrc(dst), (unsigned)(16 * e), o); }
      else *reinterpret_cast<float4 *>(dst + 4 * e) = a;
    }
  };
  if (p.one_level) {
    // Small pre-gradients (bonds of a few: the chain ends, and every step under the reference's truncation rule, where the
    // batch side IS the step): ONE level.  The last arriver sums all partials -- thread (chunk c, element e) adds the c-th
    // sixteenth of them in slab order, the sixteen chunk sums meet in LDS and are added in chunk order (fixed order:
    // deterministic) -- and writes the reduced tensor itself: no group sums, no second ticket, no second acquire.
    acquire_all();
    const int n4 = (n + 3) / 4, per = (g_n + 15) / 16;
    const float inv_n4 = 1.0f / (float)n4;
    float4 *part = reinterpret_cast<float4 *>(smem);
    for (int idx = tid; idx < 16 * n4; idx += NT) {
      const int c = (int)(((float)idx + 0.5f) * inv_n4), e = idx - c * n4;       // exact quotient of small integers
      const int k_lo = c * per, cnt = min(g_n, k_lo + per) - k_lo;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        v[k] = k < cnt ? *reinterpret_cast<const float4 *>(p.slabs + (size_t)(k_lo + k) * p.slab_stride + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 16; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
      part[idx] = a;
    }
    lds_barrier();
    for (int e = tid; e < n4; e += NT) {
      float4 a = part[e];
#pragma unroll
      for (int c = 1; c < 16; ++c) { const float4 b = part[c * n4 + e]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      if (p.persist) {
        tn_uvec4 o; o.x = __float_as_uint(a.x); o.y = __float_as_uint(a.y); o.z = __float_as_uint(a.z); o.w = __float_as_uint(a.w);
        st_sc1_b128(sc1_rsrc(p.zred), (unsigned)(16 * e), o);
      } else {
        *reinterpret_cast<float4 *>(p.zred + 4 * e) = a;
      }
    }
    if (tid == 0) __hip_atomic_store(p.gcnt + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    publish_zred();
    return false;
  }
  // level 1: the group's slabs in slab order
  acquire_all();
  sum_slabs(p.slabs + (size_t)g_lo * p.slab_stride, g_n, p.gslabs + (size_t)grp * p.slab_stride, true);
  if (tid == 0) __hip_atomic_store(p.gcnt + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0) sTicket = __hip_atomic_fetch_add(p.tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lds_barrier();
  if (sTicket != (unsigned)(p.ngroups - 1)) return false;         // not the last group
  // level 2: the group sums in group order -> the reduced tensor the next launch reads (ordinary stores: the kernel
  // boundary publishes them)
  acquire_all();
  sum_slabs(p.gslabs, p.ngroups, p.zred, p.persist != 0);
  if (tid == 0) __hip_atomic_store(p.tcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  publish_zred();
  return false;
}

}  // namespace tnml
