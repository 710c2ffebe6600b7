// The batch-independent half of a sweep step, for merged two-site tensors that fit one workgroup's LDS
// (kernels_big.hip is the HBM-resident variant).  One launch per sweep step does
//   B = A_k . A_{k+1}                                   (Network_class.py:484)
//   dB = dB_raw - 2 wd Ln.B.Rn   (or - wd B)            (:728-734, compute_L2_reg :966-1179)
//   clip by the sum|.| ratio, B_new = B + lr dB         (:755-761)
//   truncated SVD of the matricised B_new, sqrt(S) on both factors   (:528-563, :839-962)
//   the two new cores, the behind norm environment of the next step, (accuracy, MAE)
// in workgroup 0's LDS.  On a single GPU the launch carries helper workgroups that sum the gradient slabs of
// the wide kernel (and, when the wide launch did not do it, compute B and Ln.B.Rn slice by slice); workgroup 0
// waits for their arrival counter.  The update and the SVD run in float64: the norm environments of a 784-site
// chain reach 1e196, and the SVD goes through the Gram matrix, whose float64 accumulation keeps the
// squared condition number harmless for float32 data.
//
// SVD method: G = W^T W (n x n, n = min(rows, cols) <= 64) accumulated in float64, then the classical
// two-sided Jacobi eigenvalue iteration G <- J^T G J, V <- V J with a round-robin pair schedule:
// n/2 disjoint rotations per round, pairs always at positions (2k, 2k+1) ("position space").  A worker thread
// owns one 2x2 block of the symmetric G (P <= Q) and applies both rotations to it, reading the old matrix and
// writing the new one, already permuted for the next round, into a second LDS buffer; V lives in registers
// (2x2 blocks, tournament move by DPP wave shifts); parameter threads prepare the next round's rotations
// concurrently, so a round costs one barrier and no reduction.  The rotation angle comes from a float
// evaluation of t = 2g / (d + sign(d) sqrt(d^2 + 4 g^2)); c is then refined to float64 (one Newton step on
// rsqrt) so that c^2 + s^2 = 1 to 1e-14 and V stays orthogonal.  Eigenvalues sigma_j^2 = diag(G), eigenvectors
// q_j = columns of V.  The short-side factor is q_j sqrt(sigma_j), the long-side one W q_j / sqrt(sigma_j), so
// that their product is the projection W Q Q^T whatever the accuracy of the small sigma_j.
#include "tnml_internal.h"
#include "jacobi_device.h"
#include "small_gemm_device.h"
#include "wide_pipe_device.h"

namespace tnml {

// sum of a double over the 64 lanes of a wave, identical in every lane, without the LDS crossbar: xor-butterfly inside
// every 16-lane row by DPP (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: fp addition is
// commutative, so all lanes of a row end with the same bits), then the four row totals by v_readlane
__device__ inline double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  const double r0 = wave_read_f64(v, 0), r1 = wave_read_f64(v, 16), r2 = wave_read_f64(v, 32), r3 = wave_read_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}

// block-wide sum of up to 3 doubles; result valid (and bitwise identical) in every thread.  One barrier: wave totals by DPP,
// 16 partials per value through LDS, then the same 4-step xor-butterfly over the 16 partials in every 16-lane row of every
// wave (fp addition is commutative: every lane ends with the same bits).  scratch: 128 doubles; BUF selects one of two
// 48-double areas -- a call needs no barrier of its own before or after as long as two calls that share a BUF are
// separated by another barrier of the workgroup.
template <int BUF>
__device__ inline void block_sum3(double &a, double &b, double &c, double *scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double *s = scratch + BUF * 64;
  a = wave_sum_f64(a); b = wave_sum_f64(b); c = wave_sum_f64(c);
  if (lane == 0) { s[wave] = a; s[16 + wave] = b; s[32 + wave] = c; }
  lds_barrier();
  const int w = lane & 15;
  double ta = w < nw ? s[w] : 0.0, tb = w < nw ? s[16 + w] : 0.0, tc = w < nw ? s[32 + w] : 0.0;
  ta += dpp_f64<0xB1>(ta); tb += dpp_f64<0xB1>(tb); tc += dpp_f64<0xB1>(tc);
  ta += dpp_f64<0x4E>(ta); tb += dpp_f64<0x4E>(tb); tc += dpp_f64<0x4E>(tc);
  ta += dpp_f64<0x141>(ta); tb += dpp_f64<0x141>(tb); tc += dpp_f64<0x141>(tc);
  ta += dpp_f64<0x140>(ta); tb += dpp_f64<0x140>(tb); tc += dpp_f64<0x140>(tc);
  a = ta; b = tb; c = tc;
}

struct NarrowCarve {
  double *dT, *dG, *Z, *dNh, *dNg, *dLam, *dRed, *dT2, *dCS, *dSq, *dVs;
  float *fB, *fBp, *sLab, *sPl, *sCb;
  int *sOrd, *sFlag, *sPi, *sPiInv, *sWave;
  float *sTail;
  size_t bytes;
};

__host__ __device__ inline NarrowCarve narrow_carve(unsigned char *base, int h, int g, int s, int L, int m) {
  const int D = kD;
  const size_t Bs = (size_t)h * D * D * g * L;
  const int r = D * h, c = D * g * L;
  const int n = r <= c ? r : c, ne = n + (n & 1);
  size_t zreg = 2 * Bs;
  if ((size_t)4 * ne * ne > zreg) zreg = (size_t)4 * ne * ne;   // G and V, two buffers each, ne x ne
  NarrowCarve k;
  double *d = (double *)base;
  k.dT = d; k.dG = d + Bs; k.Z = d; d += zreg;
  k.dNh = d; d += ((size_t)h * h + 1) & ~(size_t)1;
  k.dNg = d; d += ((size_t)g * g + 1) & ~(size_t)1;
  k.dLam = d; d += ne;
  k.dRed = d; d += 128;        // block_sum3: two areas of 48 at 0 and 64; slots 60..62: single values
  k.dT2 = d; d += (size_t)h * D * m;
  k.dCS = d; d += 4 * ne;                   // (c, s, t, -) per pair, two rounds in flight
  k.dSq = d; d += 2 * ne;                   // sigma^(1/2) and sigma^(-1/2) of the kept columns
  k.dVs = d; d += (size_t)ne * m;           // kept eigenvectors, columns in descending order of the eigenvalue
  float *f = (float *)d;
  k.fB = f; f += Bs;
  k.fBp = f; f += Bs + (size_t)(D * h < D * g * L ? D * h : D * g * L) + 4;   // rows at stride (cols + 1): bank-conflict-free
  k.sLab = f; f += (size_t)h * D * s * L;
  k.sPl = f; f += (size_t)s * D * g;
  k.sCb = f; f += (size_t)r * m;
  f += (size_t)m * c;                       // room: the pipelined step stages the reduced pre-gradient in [fBp, sOrd) before its contraction
  int *ip = (int *)f;
  k.sOrd = ip; ip += ne;
  k.sFlag = ip; ip += 8;
  k.sPi = ip; ip += ne;
  k.sPiInv = ip; ip += ne;
  k.sTail = (float *)ip; ip += kMetricSlots;
  k.sWave = ip; ip += 16;
  k.bytes = (size_t)((unsigned char *)ip - base);
  return k;
}

size_t narrow_lds_bytes(int h, int g, int s, int L, int m) {
  alignas(16) static unsigned char origin[16];          // only distances from it are used (arithmetic on a null pointer is undefined)
  return narrow_carve(origin, h, g, s, L, m).bytes + 16;
}

// ------------------------------------------------------------------------------------------
// Helper workgroups of a fused launch (blockIdx.x >= 1).  They never wait for anybody, so the launch
// cannot deadlock whatever the residency; workgroup 0 waits for their arrival counter.
//   blocks 1 .. nred          : deterministic sum of the gradient slabs, 64 elements each (the arithmetic of
//                               reduce_slabs_kernel, kernels_wide.hip)
//   blocks nred+1 .. nred+D*D : slice (dk, dk1) of the merged tensor and of Ln.B.Rn -- both factorise over the
//                               two feature indices, so the four slices are independent
// ------------------------------------------------------------------------------------------
__device__ inline void narrow_helper_block(const NarrowParams &p, unsigned char *smem_raw) {
  const int tid = threadIdx.x;
  const int h = p.h, g = p.g, s = p.s, L = p.L, Bs = p.bsize;
  const int blk = blockIdx.x - 1;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  if (blk < p.nred) {
    float *part = (float *)smem_raw;                       // [16][64]
    const int el = tid & 63, chunk = tid >> 6;
    const int e = blk * 64 + el, n = Bs + kMetricSlots;
    const int per = (p.nslabs + 15) / 16;
    const int k0 = chunk * per, k1 = min(p.nslabs, k0 + per);
    float a0 = 0.f, a1 = 0.f;
    if (e < n) {
      int kk = k0;
      for (; kk + 2 <= k1; kk += 2) {
        a0 += p.slabs[(size_t)kk * p.slab_stride + e];
        a1 += p.slabs[(size_t)(kk + 1) * p.slab_stride + e];
      }
      if (kk < k1) a0 += p.slabs[(size_t)kk * p.slab_stride + e];
    }
    part[chunk * 64 + el] = a0 + a1;
    __syncthreads();
    if (chunk == 0 && e < n) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) t += part[c * 64 + el];
      __hip_atomic_store(p.red_out + e, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    PrepParams q{};
    q.lab = p.lab; q.pl = p.pl; q.Nh = p.Nh; q.Ng = p.Ng; q.h = h; q.g = g; q.s = s; q.L = L; q.l2_flag = p.l2_flag;
    q.prepB = p.prepB; q.prepG = p.prepG;
    q.nparts = p.pipe ? (p.wait_count - p.nred) / (kD * kD) : 1;
    prep_slice_block(q, blk - p.nred, smem_raw, true);
  }
  // hand-off to workgroup 0 (other CU, possibly other XCD: L1 and L2 are not coherent across them), first row of the
  // hand-off table of MI355X_MICROARCH.md: every handed-off byte was stored with an agent-scope (sc1, write-through)
  // store, every storing wave drains its stores, the workgroup meets, ONE lane bumps the counter (relaxed: no L2
  // write-back to wait for); workgroup 0 polls it relaxed and reads the data with agent-scope (sc1) loads
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (p.stamps && p.counters) {      // diagnostic: first start and last finish of either helper role, 100 MHz ticks
      atomicMin(p.counters + (blk < p.nred ? 4 : 5), t_start);
      atomicMax(p.counters + (blk < p.nred ? 6 : 7), __builtin_amdgcn_s_memrealtime());
    }
    __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// returns true when a wait of a persistent sweep timed out (the caller leaves its step loop)
// NP: NarrowParams (kernel argument) or its constant-address-space twin (persistent sweep: the per-step records are read with scalar
// loads only if the compiler knows nothing in the kernel writes them)
template <class NP>
__device__ __forceinline__ bool narrow_body(const NP &p, unsigned char *smem_raw) {
  // value ranges the launcher guarantees (narrow_lds_bytes / narrow_path): with them the compiler turns the index products into
  // full-rate 24-bit multiplies (a 32-bit v_mul_lo_u32 issues at quarter rate, and this workgroup is issue-bound)
  __builtin_assume(p.h >= 1 && p.h <= 64 && p.g >= 1 && p.g <= 64 && p.s >= 1 && p.s <= 128 && p.m >= 1 && p.m <= 64);
  __builtin_assume(p.L >= 1 && p.L <= 2048 && p.bsize >= 1 && p.bsize <= 8192);
  const NarrowCarve k = narrow_carve(smem_raw, p.h, p.g, p.s, p.L, p.m);
  const int tid = threadIdx.x, NT = kNarrowThreads;
  // the wave index as a scalar: loops and role tests built on it become SALU control flow instead of EXEC-mask bookkeeping
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = kD, h = p.h, g = p.g, s = p.s, L = p.L, m = p.m, Bs = p.bsize;
  // reduced gradient + metric tail: written by other workgroups of THIS launch when fused -> coherent loads
  auto ldred = [&](int e) -> float {
    return (p.fused && !p.pipe) ? __hip_atomic_load(p.red + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.red[e];
  };
  if (p.zpoll_flag) {
    // zred belongs to the side stream (batch-side launch + all-reduce) until its sequence number says otherwise; bounded, and a
    // time-out is reported (status bit 64).  The acquire drops what this XCD's L2 holds of the buffer from the previous step.
    if (threadIdx.x == 0) {
      bool ok = false;
      for (int spin = 0; spin < (1 << 22) && !ok; ++spin) {
        ok = (int)(__hip_atomic_load(p.zpoll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.zpoll_want) >= 0;
        if (!ok) __builtin_amdgcn_s_sleep(16);
      }
      if (!ok) atomicOr(p.status, 64);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // metric sums of the batch-side workgroups: behind the gradient (classic) or behind the reduced pre-gradient (pipelined)
  // (pipelined: read NOW -- the batch-side workgroups of this launch overwrite zred once B_new is published)
  if (p.pipe && !p.persist && threadIdx.x == 0)
    for (int i = 0; i < kMetricSlots; ++i) k.sTail[i] = p.zred[p.zsize + i];      // LDS: the loads complete here
  auto ldtail = [&](int i) -> float { return p.pipe ? k.sTail[i] : ldred(p.bsize + i); };
  const int r = D * h, c = D * g * L;
  const bool short_rows = (r <= c);
  const int n = short_rows ? r : c, ne = n + (n & 1), len = short_rows ? c : r;

  unsigned long long t_c0 = 0, t_r0 = 0, t_c1 = 0, t_c2 = 0, t_c2b = 0, t_c2c = 0;
  unsigned long long t_p[5] = {0, 0, 0, 0, 0};
#define TNML_STAMP(i) if (p.stamps && tid == 0) t_p[i] = __builtin_amdgcn_s_memtime()
  if (p.stamps && tid == 0) { t_c0 = __builtin_amdgcn_s_memtime(); t_r0 = __builtin_amdgcn_s_memrealtime(); }
  // ---- pipelined step: raw gradient dB[h_, rest] = sum_i' A_{k-1}[i', h_] Z_k[i', rest] (wide_pipe_device.h); both operands
  // were completed by the previous launch, so this runs before anything of this launch is waited for
  const int RWz = kD * kD * p.g * p.L;
  // Pipelined launch whose merged tensor / L2 term were prepared by the previous launch's tail: EVERY global operand of
  // this workgroup is requested at once (16-byte loads where the layout allows) and lands in LDS after one round trip --
  // a loop of dependent load -> LDS-store iterations costs one round trip per iteration (7 for the reduced pre-gradient).
  const bool fast0 = p.pipe && !p.persist && (size_t)p.z_rows * RWz + (size_t)p.z_rows * h <= (size_t)((float *)k.sOrd - k.fBp) &&
                     Bs <= 8 * NT && p.z_rows * RWz <= 16 * NT && p.z_rows * h <= 4 * NT && h * h <= 2 * NT && g * g <= 2 * NT;
  // ---- persistent sweep: merged tensor, L2 term and raw gradient of this step are projections with the previous step's behind core,
  //   B_k = diag(1 / sigma) A'^T . T_k,   (Ln.B.Rn)_k = Nh^T . diag(1 / sigma) A'^T . (T_k . Ng),   dB_raw_k = A'^T . Z_k,
  // formed by the helper workgroups of the launch (persist_helper_block), each for a slice of the columns, from operands they
  // prepared beside the previous SVD; this workgroup waits for their arrival counter and loads the three results.
  const PersistLds PL = persist_lds(smem_raw + p.persist_off, p.Mcap);
  float *sRaw = k.fBp;                          // raw gradient [h][RW] (float): dead before phase 5 rewrites fBp
  if (p.persist) {
    if (tid == 0) {
      const int bad = spin_wait_ge(p.pready, p.pwant, p.abort_flag, 1);
      if (bad == 1) { atomicOr(p.status, 4); __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      k.sFlag[3] = bad;
    }
    lds_barrier();
    if (k.sFlag[3]) return true;
    if (p.stamps && tid == 0) { p.stamps[20] = (double)(__builtin_amdgcn_s_memtime() - t_c0); p.stamps[28] = (double)(t_r0 & ((1ull << 40) - 1)); p.stamps[29] = (double)(__builtin_amdgcn_s_memrealtime() & ((1ull << 40) - 1)); }
    if (tid < kMetricSlots) k.sTail[tid] = ld_sc1(p.zred + p.zsize + tid);
    for (int e = tid; e < h * h; e += NT) k.dNh[e] = p.Nh ? PL.Nh[e] : 1.0;
    const __amdgpu_buffer_rsrc_t rR = sc1_rsrc(p.prepRaw), rB = sc1_rsrc(p.prepB), rG = sc1_rsrc(p.prepG);
    tn_uvec4 qr[2], qb[2], qg[4];                // Bs <= 8192 (launcher), a multiple of 4
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); if (e < Bs) { qr[u] = ld_sc1_b128(rR, (unsigned)e * 4u); qb[u] = ld_sc1_b128(rB, (unsigned)e * 4u); } }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); if (p.l2_flag && e < Bs) qg[u] = ld_sc1_b128(rG, (unsigned)e * 8u); }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); if (e < Bs) { *reinterpret_cast<tn_uvec4 *>(sRaw + e) = qr[u]; *reinterpret_cast<tn_uvec4 *>(k.fB + e) = qb[u]; } }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); if (p.l2_flag && e < Bs) *reinterpret_cast<tn_uvec4 *>(k.dG + e) = qg[u]; }
  }
  float *sZ = k.fBp, *sZc = sZ + (size_t)p.z_rows * RWz;
  if (fast0) {
    const int zr = p.z_rows, nz = p.z_first ? 0 : zr * RWz, nzc = p.z_first ? 0 : zr * h;
    const float *zc = p.zcore.base;
    float4 rz[4], rb[2];
    double2 rg[4];
    float rc[4];
    double rnh[2], rng[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 4 * (tid + u * NT); rz[u] = e < nz ? *reinterpret_cast<const float4 *>(p.zred + e) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); rb[u] = (p.prep_ready && e < Bs) ? *reinterpret_cast<const float4 *>(p.prepB + e) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); rg[u] = (p.prep_ready && p.l2_flag && e < Bs) ? *reinterpret_cast<const double2 *>(p.prepG + e) : make_double2(0.0, 0.0); }
    const float inv_h = 1.0f / (float)h;   // exact quotients for these small integers
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * NT, i2 = (int)(((float)e + 0.5f) * inv_h), hh = e - i2 * h;
      rc[u] = e < nzc ? zc[(i2 >> 1) * p.zcore.s_in + (i2 & 1) * p.zcore.s_d + hh * p.zcore.s_out] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NT;
      rnh[u] = (p.Nh && e < h * h) ? p.Nh[e] : 1.0;
      rng[u] = (p.Ng && e < g * g) ? p.Ng[e] : 1.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 4 * (tid + u * NT); if (e < nz) *reinterpret_cast<float4 *>(sZ + e) = rz[u]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); if (p.prep_ready && e < Bs) *reinterpret_cast<float4 *>(k.fB + e) = rb[u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); if (p.prep_ready && p.l2_flag && e < Bs) *reinterpret_cast<double2 *>(k.dG + e) = rg[u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + u * NT; if (e < nzc) sZc[e] = rc[u]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NT;
      if (e < h * h) k.dNh[e] = rnh[u];
      if (e < g * g) k.dNg[e] = rng[u];
    }
    lds_barrier();
    if (!p.z_first)
      mm_lds_f32(h, RWz, zr, sZc, 1, h, sZ, RWz, 1, [&](int i, int j, float v) { k.dT[i * RWz + j] = (double)v; });
    if (!p.prep_ready) {
      // merged tensor and L2 term from the slice workgroups of THIS launch: wait for their arrivals, then agent-scope loads,
      // all in flight together
      if (tid == 0) {
        const unsigned want = (unsigned)p.wait_count;
        int spins = 0;
        // relaxed agent-scope poll: every load of the handed-off bytes below is an agent-scope (sc1) load, so no acquire --
        // an acquire invalidates this CU's L1 and costs ~1.7 us per poll (MI355X_MICROARCH.md, hand-off table, first row)
        while (__hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 22)) {
          __builtin_amdgcn_s_sleep(2);
          ++spins;
        }
        if (spins >= (1 << 22)) atomicOr(p.status, 4);
        __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      lds_barrier();
      // 16-byte agent-scope loads, all in flight together (Bs is a multiple of D * D = 4)
      const __amdgpu_buffer_rsrc_t rB = sc1_rsrc(p.prepB), rG = sc1_rsrc(p.prepG);
      tn_uvec4 qb[2], qg[4];
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); if (e < Bs) qb[u] = ld_sc1_b128(rB, (unsigned)e * 4u); }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); if (p.l2_flag && e < Bs) qg[u] = ld_sc1_b128(rG, (unsigned)e * 8u); }
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int e = 4 * (tid + u * NT); if (e < Bs) *reinterpret_cast<tn_uvec4 *>(k.fB + e) = qb[u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = 2 * (tid + u * NT); if (p.l2_flag && e < Bs) *reinterpret_cast<tn_uvec4 *>(k.dG + e) = qg[u]; }
    }
  } else if (p.pipe && !p.persist && !p.z_first) {
    const float *zc = p.zcore.base;
    const int zs_in = p.zcore.s_in, zs_d = p.zcore.s_d, zs_out = p.zcore.s_out;
    const int zr = p.z_rows;
    small_gemm_f64(1, h, RWz, zr,
                   [&](int, int i, int kk) { return (double)zc[(kk >> 1) * zs_in + (kk & 1) * zs_d + i * zs_out]; },
                   [&](int, int kk, int j) { return (double)p.zred[(size_t)kk * RWz + j]; },
                   [&](int, int i, int j, double v) { k.dT[i * RWz + j] = v; });
  }
  // ---- phase 0: stage the two cores and the norm environments ---------------------------------
  if (fast0 || p.persist) {
    // everything is in LDS already
  } else if (p.fused) {
    // B and Ln.B.Rn come from the slice workgroups of the preceding wide launch (prep_ready: plain loads, issued
    // before the wait) or of this launch; red always from the reduce workgroups of this launch
    if (p.prep_ready)
      for (int e = tid; e < Bs; e += NT) {
        k.fB[e] = p.prepB[e];
        if (p.l2_flag) k.dG[e] = p.prepG[e];
      }
    if (tid == 0) {
      const unsigned want = (unsigned)p.wait_count;
      int spins = 0;
      while (__hip_atomic_load(p.sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 22)) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
      }
      if (spins >= (1 << 22)) atomicOr(p.status, 4);
      __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.stamps && p.counters) {      // ticks of the 100 MHz counter relative to this workgroup's start
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        p.stamps[28] = (double)(now - t_r0);                                   // wait over
        p.stamps[29] = (double)(long long)(p.counters[4] - t_r0);              // first reduce helper started
        p.stamps[30] = (double)(long long)(p.counters[6] - t_r0);              // last reduce helper done
        p.stamps[31] = (double)(long long)(p.counters[5] - t_r0);              // first slice helper started
        p.stamps[32] = (double)(long long)(p.counters[7] - t_r0);              // last slice helper done
        p.stamps[33] = (double)spins;
        p.counters[4] = ~0ull; p.counters[5] = ~0ull; p.counters[6] = 0ull; p.counters[7] = 0ull;
      }
    }
    lds_barrier();
    if (!p.prep_ready) for (int e = tid; e < Bs; e += NT) {
      k.fB[e] = __hip_atomic_load(p.prepB + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.l2_flag) k.dG[e] = __hip_atomic_load(p.prepG + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (p.prep_ready) {
    // multi-GPU sequence (separate reduction + all-reduce): B and Ln.B.Rn still come from the wide launch's slices
    for (int e = tid; e < Bs; e += NT) {
      k.fB[e] = p.prepB[e];
      if (p.l2_flag) k.dG[e] = p.prepG[e];
    }
  } else if (!p.Bdirect) {
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
  } else {
    for (int e = tid; e < Bs; e += NT) k.fB[e] = p.Bdirect[e];
  }
  if (!fast0 && !p.persist) {
    for (int e = tid; e < h * h; e += NT) k.dNh[e] = p.Nh ? p.Nh[e] : 1.0;
    for (int e = tid; e < g * g; e += NT) k.dNg[e] = p.Ng ? p.Ng[e] : 1.0;
  }
  lds_barrier();

  TNML_STAMP(0);
  // ---- phase 1: B[h,dk,dk1,g,l] = sum_s lab(h,dk,s,l) * pl(s,dk1,g) ---------------------------
  const int RW = D * D * g * L;  // elements per behind-bond index
  if (!p.Bdirect && !p.fused && !p.prep_ready && !p.persist) {   // rows i = (h_, dk, l), columns j = (dk1, g_), inner index the shared bond
    const int QW = D * g;
    small_gemm_f64(L, h * D, QW, s,
                   [&](int l, int i, int kk) { return (double)k.sLab[(i * s + kk) * L + l]; },
                   [&](int l, int kk, int j) { return (double)k.sPl[kk * QW + j]; },
                   [&](int l, int i, int j, double v) { k.fB[(i * QW + j) * L + l] = (float)v; });
  }
  lds_barrier();

  TNML_STAMP(1);
  // ---- phases 2-3: weight decay term ------------------------------------------------------------
  if (p.l2_flag && !p.fused && !p.prep_ready && !p.persist) {
    // T = Nh^T . B over the behind bond:  T[e_, rest] = sum_a Nh[a, e_] B[a, rest]
    mm_lds(1, h, RW, h, k.dNh, 0, 1, h, k.fB, 0, RW, 1, [&](int, int i, int j, double v) { k.dT[i * RW + j] = v; });
    lds_barrier();
    // G = T . Ng over the ahead bond: rows i = (e_, dk, dk1), columns f_, the label is the batch
    mm_lds(L, Bs / (g * L), g, g, k.dT, 1, g * L, L, k.dNg, 0, g, 1,
           [&](int l, int i, int j, double v) { k.dG[(i * g + j) * L + l] = v; });
    lds_barrier();
  }
  TNML_STAMP(2);
  double sumB = 0.0, sumD = 0.0, l2 = 0.0;
  constexpr int kMaxPer = 8;                              // Bs <= 8192 in this kernel (LDS budget)
  float redv[kMaxPer];
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {                     // all loads of the reduced gradient in flight together
    const int e = tid + u * NT;
    redv[u] = (e < Bs && !p.persist && !(p.pipe && !p.z_first)) ? ldred(e) : 0.f;
  }
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    const int e = tid + u * NT;
    if (e >= Bs) break;
    const double bv = (double)k.fB[e];
    const double raw = p.persist ? (double)sRaw[e] : ((p.pipe && !p.z_first) ? k.dT[e] : (double)redv[u]);
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
  block_sum3<0>(sumB, sumD, l2, k.dRed);

  // ---- phase 5: clip + update (Network_class.py:755-761) ----------------------------------------
  double factor = (double)p.lr;
  if (sumD > sumB) factor = (double)p.lr * (sumB / sumD);
  const bool bad = !isfinite(sumD) || !isfinite(sumB);
  if ((c & 3) == 0) {
    // four consecutive elements of a row per thread: 16-byte LDS accesses, and the hand-off to the batch-side workgroups of
    // this launch leaves as ONE 16-byte agent-scope store per thread straight from registers (a scalar sc1 store is one
    // fabric write per element); row index by an exact float quotient
    const float inv_c = 1.0f / (float)c;
    const __amdgpu_buffer_rsrc_t rN = sc1_rsrc(p.Bnew);
    for (int e = 4 * tid; e < Bs; e += 4 * NT) {
      const int row = (int)(((float)e + 0.5f) * inv_c), x = e - row * c;
      const float4 b4 = *reinterpret_cast<const float4 *>(k.fB + e);
      const double2 g01 = *reinterpret_cast<const double2 *>(k.dG + e), g23 = *reinterpret_cast<const double2 *>(k.dG + e + 2);
      float4 v;
      v.x = (float)((double)b4.x + factor * g01.x); v.y = (float)((double)b4.y + factor * g01.y);
      v.z = (float)((double)b4.z + factor * g23.x); v.w = (float)((double)b4.w + factor * g23.y);
      *reinterpret_cast<float4 *>(k.fB + e) = v;
      float *bp = k.fBp + row * (c + 1) + x;
      bp[0] = v.x; bp[1] = v.y; bp[2] = v.z; bp[3] = v.w;
      if (p.flag) {
        tn_uvec4 o; o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
        st_sc1_b128(rN, (unsigned)(4 * e), o);
      } else {
        *reinterpret_cast<float4 *>(p.Bnew + e) = v;
      }
      if (p.dbg) { double *dd = p.dbg + 2 * (size_t)Bs + e; dd[0] = v.x; dd[1] = v.y; dd[2] = v.z; dd[3] = v.w; }
    }
  } else {
    for (int row = wave_u; row < r; row += NT >> 6)       // rows over waves, columns over lanes
      for (int x = tid & 63; x < c; x += 64) {
        const int e = row * c + x;
        const float v = (float)((double)k.fB[e] + factor * k.dG[e]);
        k.fB[e] = v;
        k.fBp[row * (c + 1) + x] = v;
        if (p.flag) st_sc1(p.Bnew + e, v); else p.Bnew[e] = v;
        if (p.dbg) p.dbg[2 * (size_t)Bs + e] = (double)v;
      }
  }
  if (p.flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its stores before the barrier
  if (tid == 0) {
    if (bad) atomicOr(p.status, 1);
    if (p.dbg) {
      double *sc = p.dbg + 4 * (size_t)Bs + kDbgSigma;
      sc[0] = (double)p.wd * l2;
      sc[1] = sumB;
      sc[2] = sumD;
    }
  }
  lds_barrier();   // dT/dG are dead from here on; Z aliases them
  // B_new is complete in memory: the batch-side workgroups of this launch may form f and the next pre-gradient from it
  // while this workgroup goes on to the SVD
  if (p.flag && tid == 0) __hip_atomic_store(p.flag, p.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.stop_after_update) {                       // standalone update_B / compute_L2_reg
    if (tid == 0 && p.metrics) {
      const double cnt = (double)ldtail(3);
      const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
      p.metrics[0] = (float)((double)ldtail(0) * inv);
      p.metrics[1] = (float)((double)ldtail(1) * inv / (double)L);
    }
    return false;
  }

  TNML_STAMP(3);
  // ---- phase 6: Gram matrix in float64 (padded to ne x ne with a zero row/column), V = I ----------
  //   W(x, kk) = short_rows ? Bm[kk][x] : Bm[x][kk]   with Bm = B_new as (r x c) row-major
  double *G0 = k.Z, *G1 = k.Z + ne * ne, *V0 = k.Z + 2 * ne * ne;
  // float64 matrix and vector pipes of gfx950 both retire 16 FMA per cycle per SIMD, so the MFMA count per SIMD is
  // what matters: the upper 16x16 tiles (G is symmetric) are cut into slices along the long index -- as many (4, 2 or 1) as
  // still give every wave at most ONE (tile, slice) item (n = 40: 6 tiles x 2 slices; n = 20: 3 x 4), else 4 -- and the
  // items dealt round-robin to the 16 waves; slice partials land in the four ne x ne buffers of the Jacobi region and are
  // summed in slice order (deterministic).
  double g_tr = 0.0, g_dg2 = 0.0, g_off2 = 0.0;
  auto gram_phase = [&]() {
    double a_tr = 0.0, a_dg2 = 0.0, a_off2 = 0.0;            // locals: the captured sums would live in memory
    double *P0 = G0, *P1 = G1, *P2 = V0, *P3 = k.Z + 3 * ne * ne;
    const int lane = tid & 63, wave = wave_u, rr = lane & 15, qq = lane >> 4;
    const int tm = (n + 15) >> 4, ntile_g = (tm * (tm + 1)) >> 1;
    const int gshift = ntile_g * 4 <= (NT >> 6) ? 2 : (ntile_g * 2 <= (NT >> 6) ? 1 : (ntile_g <= (NT >> 6) ? 0 : 2));
    const int gsplit = 1 << gshift;
    const int kchunk = ((len + gsplit - 1) / gsplit + 3) & ~3;       // multiple of the MFMA k = 4
    // (tile, slice) items dealt round-robin to the waves: nested counters, no integer division (a per-lane division costs
    // 134 cycles, a wave-uniform one ~40 scalar instructions: tools/ubench/prims.hip)
    // this wave's (tile, slice) items, decoded from a wave-uniform index with scalar instructions only (the wave index is
    // made uniform for the compiler by v_readfirstlane); W(x, kk) = Wb[x * rs + kk * cs]
    const int wave_s = wave;
    const float *Wb = short_rows ? k.fBp : k.fB;
    const int rs = short_rows ? c + 1 : 1, cs = short_rows ? 1 : c;
    const int nitems = ntile_g * gsplit;
    for (int item = wave_s; item < nitems; item += NT >> 6) {
      const int ks = item & (gsplit - 1);
      int t = item >> gshift, ti = 0, rowlen = tm;
      while (t >= rowlen) { t -= rowlen; ++ti; --rowlen; }
      const int tj = ti + t;
      const int i0 = ti << 4, j0 = tj << 4;
      const bool va = i0 + rr < n, vb = j0 + rr < n;
      const float *pa = Wb + min(i0 + rr, n - 1) * rs, *pb = Wb + min(j0 + rr, n - 1) * rs;
      const int k_lo = ks * kchunk, k_hi = min(len, k_lo + kchunk);
      dvec4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      for (int k0 = k_lo; k0 < k_hi; k0 += 16) {           // up to four k-steps per trip, all LDS reads issued first
        float fa[4], fb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kc = min(k0 + 4 * u + qq, k_hi - 1) * cs;
          fa[u] = pa[kc];
          fb[u] = pb[kc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool vk = k0 + 4 * u + qq < k_hi;
          const double av = (va && vk) ? (double)fa[u] : 0.0, bv = (vb && vk) ? (double)fb[u] : 0.0;
          if (k0 + 4 * u < k_hi) {
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
          }
        }
      }
      const dvec4 acc = acc0 + acc1;
      const int j = j0 + rr;
      double *Pk = P0 + (size_t)ks * ne * ne;              // the four ne x ne buffers are contiguous
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = i0 + qq + 4 * reg;
        if (i < ne && j < ne) Pk[i * ne + j] = (i < n && j < n) ? acc[reg] : 0.0;     // padding row / column = 0
      }
    }
    lds_barrier();
    // sum of the slices, in slice order, written to both triangles (exact symmetry; tiles below the diagonal were never
    // written, and inside a diagonal tile only i <= j is read).  Rows over waves, columns over lanes: no division.  The
    // same pass collects trace, sum of squared diagonal and of squared off-diagonal entries.
    for (int i = wave_u; i < ne; i += NT >> 6)
      for (int j = tid & 63; j < ne; j += 64)
        if (i <= j) {
          const int e = i * ne + j;
          double v = P0[e];
          if (gsplit > 1) v += P1[e];
          if (gsplit > 2) v = (v + P2[e]) + P3[e];
          P0[e] = v;                                              // P0 == G0: in place
          if (i < j) { G0[j * ne + i] = v; a_off2 += 2.0 * v * v; }
          else { a_tr += v; a_dg2 += v * v; }
        }
    g_tr = a_tr; g_dg2 = a_dg2; g_off2 = a_off2;
  };
  gram_phase();
  // Tournament in POSITION space: the pairs of a round are always the positions (2k, 2k+1); after
  // the rotations every row/column moves to position pi(pos) of the next round (circle method:
  // position 0 fixed, top row shifts right, bottom row shifts left).  The move is free: the updated
  // blocks are written into the other buffer anyway.
  const int np = ne / 2;                      // pairs per round
  for (int pos = tid; pos < ne; pos += NT) {
    const int kq = pos >> 1;
    int nxt;
    if (pos & 1) nxt = (kq == 0) ? (np > 1 ? 2 : 1) : 2 * (kq - 1) + 1;          // bottom row
    else nxt = (kq == 0) ? 0 : (kq == np - 1 ? 2 * kq + 1 : 2 * (kq + 1));       // top row
    k.sPi[pos] = nxt;
    k.sPiInv[nxt] = pos;
  }
  TNML_STAMP(4);
  // scale G to trace ~ 1 by an exact power of two (undone on the eigenvalues in phase 8); the statistics of the summation
  // pass decide about the Cholesky step (a ratio: the scale drops out)
  double tr = 0.0, off2 = 0.0;
  int sc_exp = 0;
  auto scale_phase = [&]() {
    block_sum3<1>(g_tr, g_dg2, g_off2, k.dRed);            // its barrier also publishes the summed G and the tables
    tr = g_tr; off2 = g_off2;
    sc_exp = (tr > 0.0 && isfinite(tr)) ? __builtin_amdgcn_frexp_exp(tr) : 0;
    for (int i = wave_u; i < ne; i += NT >> 6)
      for (int j = tid & 63; j < ne; j += 64) G0[i * ne + j] = __builtin_amdgcn_ldexp(G0[i * ne + j], -sc_exp);
    lds_barrier();
  };
  scale_phase();
  const bool chol_possible = p.chol_thr > 0.0 && n > 4;   // tiny matrices (chain ends, reference policy): never worth it

  // ---- thread roles of the Jacobi iterations (phase 7; the float32 stage of phase 7a uses the same ones) ------------
  constexpr int kVR = 2;                                 // register blocks per V lane (upper bound); vr of them are used:
  const int gpw = 64 / np;                               // groups of np lanes per V wave (np <= 32)
  // one block per lane when the V and G waves then still fit the workgroup (n <= 40: 7 + 4 of 15), two otherwise
  const int vr = ((np + gpw - 1) / gpw + (np * (np + 1) / 2 + 127) / 128 <= (NT >> 6) - 1) ? 1 : 2;
  const int NVW = (np + vr * gpw - 1) / (vr * gpw);      // V waves
  // worker waves = every wave but wave 0 (the parameter wave).  (Measured, round 2: once the parameter chain runs in float32
  // the round is bound by the float64 issue rate of the working waves; keeping wave 0's SIMD free of them does not pay.)
  const int nwork = (NT >> 6) - 1, wrank = wave_u - 1;
  // threads that own G items: as few waves as hold them at one item per lane (np(np+1)/2 = 210 items at n = 40: 4 waves, not
  // the 8 that are left) -- a round is bound by the float64 issue slots the worker waves share per SIMD, and a wave costs
  // its slots whatever the number of active lanes
  const int nGw = min(nwork - NVW, (np * (np + 1) / 2 + 63) / 64);
  const int NW = nGw * 64;
  const int gtid = (wrank - NVW) * 64 + (tid & 63);      // index among them (negative: not a G wave)
  const int vlane = tid & 63, vgrp = vlane / np, vQ = vlane - vgrp * np;
  const bool isVwave = wrank >= 0 && wrank < NVW;
  const int vP0 = (wrank * gpw + vgrp) * vr;
  const bool vLaneOk = isVwave && vgrp < gpw;
  // G items: the np(np+1)/2 blocks P <= Q of the symmetric G (only entries with row <= column are kept
  // up to date); at most 2 per worker thread (np <= 32: 528 items on >= 448 threads)
  constexpr int MAXI = 2;
  const int nG = np * (np + 1) / 2;
  bool itValid[MAXI], itDiag[MAXI];
  int itSrc[MAXI], itCsQ[MAXI], itCsP[MAXI], itD11[MAXI], itD12[MAXI], itD21[MAXI], itD22[MAXI];
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int it = gtid + u * NW;
    itValid[u] = wrank >= NVW && gtid < NW && it < nG;
    int P = 0, Q = 0;
    if (itValid[u]) {                         // it-th pair (P <= Q) in row-major order of the upper triangle
      int rem = it;
      while (rem >= np - P) { rem -= np - P; ++P; }
      Q = P + rem;
    }
    const int c1 = k.sPi[2 * Q], c2 = k.sPi[2 * Q + 1];
    const int o1 = k.sPi[2 * P], o2 = k.sPi[2 * P + 1];
    itSrc[u] = (2 * P) * ne + 2 * Q;
    itCsQ[u] = 4 * Q; itCsP[u] = 4 * P;
    itDiag[u] = P == Q;
    // every element lands at (min, max) of its new position
    itD11[u] = min(o1, c1) * ne + max(o1, c1); itD12[u] = min(o1, c2) * ne + max(o1, c2);
    itD21[u] = min(o2, c1) * ne + max(o2, c1); itD22[u] = min(o2, c2) * ne + max(o2, c2);
  }
  const double abs2 = kJacobiAbs * kJacobiAbs;      // trace is ~1 after scaling
  // parameter-thread constants
  const bool isParam = tid < np;
  int pa = 0, pb = 1;
  if (isParam) { pa = k.sPiInv[2 * tid]; pb = k.sPiInv[2 * tid + 1]; }
  const int pA = pa >> 1, ra = pa & 1, pB = pb >> 1, rb = pb & 1;
  // element offsets of the parameter thread's reads, formed once: left inline they are eight 32-bit multiplies per round on
  // the critical chain of the parameter wave (v_mul_lo_u32 issues at quarter rate)
  const int oAd = (2 * pA) * ne + 2 * pA, oAb = (2 * pA + 1) * ne + 2 * pA + 1;
  const int oBd = (2 * pB) * ne + 2 * pB, oBb = (2 * pB + 1) * ne + 2 * pB + 1;
  const int oR0 = pA <= pB ? (2 * pA) * ne + 2 * pB : (2 * pB) * ne + 2 * pA, oR1 = oR0 + ne;   // stored block (min, max)

  if (p.stamps && tid == 0) t_c1 = __builtin_amdgcn_s_memtime();
  // ---- phase 6b: one pivoted-Cholesky step when G is far from diagonal ------------------------------------------
  // G = L L^T (diagonal pivoting), G' = L^T L has the same eigenvalues and is graded: the Jacobi iteration then needs
  // 2-3 sweeps fewer on the merged tensors of the first training passes (off / trace >= 0.25) and the same number
  // once the chain has settled (off / trace ~ 0.16), where the step is skipped (tests/emulation/jacobi_cholesky_emulation.py).
  // Eigenvectors: G' u = lambda u  =>  G (L u) = lambda (L u), |L u|^2 = lambda.  No physical pivoting: column k
  // of L is stored against the ORIGINAL row index (Lm[k][i] = L[i][k]), so G = L L^T and G' = L^T L hold as plain
  // products.
  double *Lm = k.Z + 3 * ne * ne;                          // free once the Gram partials are summed
  // threshold: measured break-even of the step against the sweeps it saves -- 0.22 for n >= 32 (C3: 8.1 k vs 7.8 k
  // steps/s, cold start +8 %), 0.35 for smaller matrices whose sweeps are cheaper (C2, n = 20: 0.22 costs 17 %, 0.35 is
  // neutral in the steady state and +2 % cold)
  const double cthr = n >= 32 ? p.chol_thr : fmax(p.chol_thr, kCholThrSmall);
  // (block-uniform: every thread holds the same sums; told to the compiler so that the step below is scalar control flow)
  const bool use_chol = __builtin_amdgcn_readfirstlane((int)(chol_possible && off2 > cthr * cthr * tr * tr)) != 0;
  if (use_chol) {
    for (int e = tid; e < ne * ne; e += NT) Lm[e] = 0.0;
    // The factorisation is bound by LDS traffic, so a worker thread (waves 1..15) owns up to kCholPer fixed PAIRS of
    // adjacent elements (i, 2jj), (i, 2jj+1) of the trailing matrix: one scalar and two 16-byte accesses per pair and
    // step.  Lane i of wave 0 tracks diagonal entry i in a register instead, so the pivot search needs no LDS and
    // runs while the other waves update.
    constexpr int kCholPer = 3;                            // n <= 64: 2048 pairs on 960 threads
    const int hn = n / 2, npairs = n * hn;
    int ei[kCholPer], ej[kCholPer];
#pragma unroll
    for (int u = 0; u < kCholPer; ++u) {
      const int pe = (tid - 64) + u * (NT - 64);
      const bool ok = tid >= 64 && pe < npairs;
      ei[u] = ok ? pe / hn : -1;
      ej[u] = ok ? 2 * (pe - (pe / hn) * hn) : 0;
    }
    double dgi = (tid < n) ? G0[tid * ne + tid] : -1.0;   // wave 0: remaining diagonal entry of row `tid`, -1 once chosen
    // pivot = largest remaining diagonal entry (a float key is enough to choose; lowest index on ties) and
    // 1 / sqrt(pivot) by hardware float rsq + one Newton step in float64.  Wave 0 searches the NEXT pivot while the
    // other waves update the trailing matrix: results alternate between two flag slots.
    auto pivot_search = [&](int slot) {
      const float key = (float)dgi;
      const float mx = wave_max_f32(key);
      const unsigned long long hit = __ballot(key == mx);
      const int jp = (mx > 1e-26f && hit) ? __ffsll((long long)hit) - 1 : -1;
      const double piv = wave_read_f64(dgi, jp < 0 ? 0 : jp);
      double inv0 = (double)__builtin_amdgcn_rsqf((float)piv);
      inv0 = inv0 * fma(-0.5 * piv, inv0 * inv0, 1.5);          // one Newton step: 1e-7 -> 1.5e-14 (this chain paces the factorisation)
      if (tid == 0) { k.sFlag[4 + slot] = jp; k.dRed[61 + slot] = inv0; }
    };
    if (wave_u == 0) pivot_search(0);
    for (int kc = 0; kc < n; ++kc) {
      lds_barrier();                                     // pivot known; the previous update is complete
      // The trailing matrix ping-pongs between the two G buffers (G1 is free until the Jacobi iteration): a step reads the
      // pivot column and its own elements from one and writes the updated elements to the other, so nobody has to wait
      // until everybody holds its column values -- ONE barrier per pivot instead of two.
      const double *src = (kc & 1) ? G1 : G0;
      double *dst = (kc & 1) ? G0 : G1;
      const int jp = __builtin_amdgcn_readfirstlane(k.sFlag[4 + (kc & 1)]);      // block-uniform: scalar loop exit, scalar row offset
      if (jp < 0) break;                                   // numerically rank deficient: the remaining columns stay zero
      const double inv = k.dRed[61 + (kc & 1)];
      if (wave_u == 0) {                                   // wave 0: column kc of L (original row index), the diagonal, next pivot
        const double l = (tid < n && dgi >= 0.0) ? src[jp * ne + tid] * inv : 0.0;
        if (tid < n) Lm[kc * ne + tid] = l;                // L stored transposed: row kc = column kc of L
        dgi = (tid == jp) ? -1.0 : (dgi >= 0.0 ? dgi - l * l : dgi);
        if (kc + 1 < n) pivot_search((kc + 1) & 1);
      } else {
#pragma unroll
        for (int u = 0; u < kCholPer; ++u)
          if (ei[u] >= 0) {                                // (row jp == column jp: symmetric, conflict-free)
            const double li = src[jp * ne + ei[u]] * inv;
            const double2 c2 = *reinterpret_cast<const double2 *>(src + jp * ne + ej[u]);
            double2 v = *reinterpret_cast<const double2 *>(src + ei[u] * ne + ej[u]);
            v.x -= li * (c2.x * inv);
            v.y -= li * (c2.y * inv);
            *reinterpret_cast<double2 *>(dst + ei[u] * ne + ej[u]) = v;
          }
      }
    }
    lds_barrier();
    // G' = L^T L into G0 (the trailing matrix is dead)
    mm_lds(1, n, n, n, Lm, 0, ne, 1, Lm, 0, 1, ne, [&](int, int a, int b, double v) { G0[a * ne + b] = v; });
    lds_barrier();
    for (int i = wave_u; i < n; i += NT >> 6)              // exact symmetry (rows over waves, columns over lanes)
      for (int j = tid & 63; j < i; j += 64) G0[i * ne + j] = G0[j * ne + i];
    lds_barrier();
  }

  // ---- phase 7: two-sided Jacobi, ONE barrier per round ----------------------------------------------
  // Workers (tid >= 64) own one 2x2 block of G and one of V per item: rows by R_P^T, columns by R_Q,
  // written into the other buffer at the next round's positions.  Meanwhile parameter thread k
  // (tid < np) prepares the rotation of pair k of the NEXT round: that pair is (a, b) =
  // (piInv(2k), piInv(2k+1)) in today's positions, its new diagonal follows from today's diagonal
  // blocks (alpha' = alpha - t gamma, beta' = beta + t gamma) and its new off-diagonal element from
  // one element of the updated block (A, B), which the thread recomputes itself.
  // Thread roles.  wave 0: parameter threads (tid < np).  waves 1..NVW: V.  The rest: G items.
  //   V never touches LDS during the iteration: a thread keeps kVR 2x2 blocks (row pairs P0..P0+kVR-1,
  //   column pair Q = its lane within a group of np lanes) in registers; the column rotation is local and
  //   the tournament move (top element of pair Q -> pair Q+1, bottom element -> pair Q-1) is a one-lane
  //   wave shift (DPP wave_shr / wave_shl, tools/ubench/dpp_wave_shift.hip).
  double vb[kVR][4];                                     // {v11, v12, v21, v22} per block
#pragma unroll
  for (int r = 0; r < kVR; ++r) {
    const double one = (vP0 + r == vQ) ? 1.0 : 0.0;      // V = I
    vb[r][0] = one; vb[r][1] = 0.0; vb[r][2] = 0.0; vb[r][3] = one;
  }
  int sweeps = 0, converged = 0;
  double *Gc = G0, *Gn = G1;
  auto kept_scale = [&](const double *G) -> double {
    // (kKeptFrac * m-th largest diagonal entry)^2, block-wide; ends with a barrier
    const int i = tid & 63;
    const double li = i < n ? G[i * ne + i] : 0.0;
    for (int j = wave_u; j < n; j += NT >> 6) {         // one wave per entry, ballot = rank (n <= 64)
      const double lj = G[j * ne + j];
      const int rank = __popcll(__ballot(i < n && ((li > lj) || (li == lj && i < j))));
      if (i == 0 && rank == m - 1) k.dRed[60] = lj;
    }
    lds_barrier();
    const double lm = kKeptFrac * fmax(k.dRed[60], 0.0);
    return lm * lm;
  };

  int cur = 0;
  double kept2 = 0.0;
  // Rotation slot of pair k in dCS (4 doubles): [0] t, [1] c0 as doubles (float32-exact values), [2] the same two as a
  // float2 for the look-ahead chain.  See jacobi_rot_f32 (jacobi_device.h) for the arithmetic.
  const float kept_lo = 1e-36f;
  // `applied` = index of the round in which the rotation will be applied.  A "big" rotation (jacobi_device.h) leaves its
  // round index + 1 in slot applied & 1 of sFlag[6..7]: every thread reads that slot at the top of round `applied` (the
  // barrier of the previous round completed the writes; the slot is next written two rounds later), so all threads know
  // the last round that applied a big rotation without any reduction.
  auto publish = [&](double *o, const RotT &r, int applied) {
    *reinterpret_cast<double2 *>(o) = make_double2((double)r.t, (double)r.c0);
    *reinterpret_cast<float2 *>(o + 2) = make_float2(r.t, r.c0);
    if (r.level >= 2) k.sFlag[6 + (applied & 1)] = applied + 1;
  };
  int round_idx = 0, last_big1 = 0;          // rounds applied so far; 1 + index of the last round with a big rotation
  auto jacobi_round = [&]() {
        const double *csc = k.dCS + cur * np * 4;
        const int big_slot = k.sFlag[6 + (round_idx & 1)];      // consumed after this round's barrier
        if (isParam) {
          // look-ahead: pair `tid` of the NEXT round is (a, b) in today's positions; its three elements after today's
          // rotations, in float32 (the inputs are read as float64 and converted)
          const float2 fA = *reinterpret_cast<const float2 *>(csc + 4 * pA + 2);     // (t, c0) of pair A
          const float2 fB = *reinterpret_cast<const float2 *>(csc + 4 * pB + 2);
          const double2 dA = *reinterpret_cast<const double2 *>(Gc + oAd);
          const double bA = Gc[oAb];
          const double2 dB = *reinterpret_cast<const double2 *>(Gc + oBd);
          const double bB = Gc[oBb];
          double2 r0, r1;                                       // rows of block (A, B)
          if (pA < pB) {
            r0 = *reinterpret_cast<const double2 *>(Gc + oR0);
            r1 = *reinterpret_cast<const double2 *>(Gc + oR1);
          } else if (pA > pB) {                                 // stored as (B, A): transpose
            const double2 s0 = *reinterpret_cast<const double2 *>(Gc + oR0);
            const double2 s1 = *reinterpret_cast<const double2 *>(Gc + oR1);
            r0 = make_double2(s0.x, s1.x);
            r1 = make_double2(s0.y, s1.y);
          } else {                                              // n == 2: the pair meets itself again
            r0 = dA;
            r1 = make_double2(dA.y, bA);
          }
          const float tA = fA.x, cA = fA.y, sA = fA.x * fA.y, tB = fB.x, cB = fB.y, sB = fB.x * fB.y;
          const float aAx = (float)dA.x, aAy = (float)dA.y, aAb = (float)bA, aBx = (float)dB.x, aBy = (float)dB.y, aBb = (float)bB;
          const float q0x = (float)r0.x, q0y = (float)r0.y, q1x = (float)r1.x, q1y = (float)r1.y;
          const float na = ra ? fmaf(tA, aAy, aAb) : fmaf(-tA, aAy, aAx);
          const float nb = rb ? fmaf(tB, aBy, aBb) : fmaf(-tB, aBy, aBx);
          // element (ra, rb) of R_A^T . blk . R_B
          const float h0 = ra ? fmaf(sA, q0x, cA * q1x) : fmaf(cA, q0x, -sA * q1x);
          const float h1 = ra ? fmaf(sA, q0y, cA * q1y) : fmaf(cA, q0y, -sA * q1y);
          const float ng = rb ? fmaf(sB, h0, cB * h1) : fmaf(cB, h0, -sB * h1);
          const RotT r = jacobi_rot_f32(na, nb, ng, fmaxf((float)kept2, kept_lo), (float)abs2, (float)p.svd_stop2);
          publish(k.dCS + ((cur ^ 1) * np + tid) * 4, r, round_idx + 1);
        }
        if (isVwave) {                                    // whole waves: every lane runs the shifts
          const double2 tc = *reinterpret_cast<const double2 *>(csc + 4 * (vLaneOk ? vQ : 0));     // (t, c0) of pair Q
          const double cq = tc.y * rot_corr(tc.x, tc.y);
#pragma unroll
          for (int r = 0; r < kVR; ++r) {
            if (r >= vr) break;                                 // wave-uniform
            // columns by R_Q: (v1, v2) -> c (v1 - t v2), c (t v1 + v2)
            const double n11 = cq * fma(-tc.x, vb[r][1], vb[r][0]), n12 = cq * fma(tc.x, vb[r][0], vb[r][1]);
            const double n21 = cq * fma(-tc.x, vb[r][3], vb[r][2]), n22 = cq * fma(tc.x, vb[r][2], vb[r][3]);
            if (np > 1) {
              const double t1 = dpp_f64<0x138>(n11), t2 = dpp_f64<0x138>(n21);      // top column of pair Q-1
              const double b1 = dpp_f64<0x138>(n12), b2 = dpp_f64<0x138>(n22);      // bottom column of pair Q-1
              const double c1 = dpp_f64<0x130>(n12), c2 = dpp_f64<0x130>(n22);      // bottom column of pair Q+1
              vb[r][0] = vQ == 0 ? n11 : (vQ == 1 ? b1 : t1);
              vb[r][2] = vQ == 0 ? n21 : (vQ == 1 ? b2 : t2);
              vb[r][1] = vQ == np - 1 ? n11 : c1;
              vb[r][3] = vQ == np - 1 ? n21 : c2;
            } else {
              vb[r][0] = n11; vb[r][1] = n12; vb[r][2] = n21; vb[r][3] = n22;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
          if (!itValid[u]) continue;
          const double2 tq = *reinterpret_cast<const double2 *>(csc + itCsQ[u]);      // (t, c0) of the column pair
          const double2 tp = *reinterpret_cast<const double2 *>(csc + itCsP[u]);      // ... of the row pair
          const double *src = Gc + itSrc[u];
          const double2 r0 = *reinterpret_cast<const double2 *>(src);
          double2 r1 = *reinterpret_cast<const double2 *>(src + ne);
          if (itDiag[u]) r1.x = r0.y;                           // lower element of a diagonal block = its mirror
          // R_P^T . blk . R_Q = cP cQ [[1, -tP], [tP, 1]] . blk . [[1, tQ], [-tQ, 1]]: the tangents act first (they need no
          // refinement), the product of the two cosines is formed meanwhile and multiplied in last
          const double a11 = fma(-tp.x, r1.x, r0.x), a12 = fma(-tp.x, r1.y, r0.y);
          const double a21 = fma(tp.x, r0.x, r1.x), a22 = fma(tp.x, r0.y, r1.y);
          const double b11 = fma(-tq.x, a12, a11), b12 = fma(tq.x, a11, a12);
          const double b21 = fma(-tq.x, a22, a21), b22 = fma(tq.x, a21, a22);
          const double c00 = tp.y * tq.y, corr = rot_corr(tp.x, tp.y) * rot_corr(tq.x, tq.y);
          double n11 = (b11 * c00) * corr, n12 = (b12 * c00) * corr, n21 = (b21 * c00) * corr, n22 = (b22 * c00) * corr;
          if (itDiag[u] && tq.x != 0.0) { n12 = 0.0; n21 = 0.0; }   // the annihilated element, exactly
          Gn[itD11[u]] = n11; Gn[itD12[u]] = n12;
          if (!itDiag[u]) Gn[itD21[u]] = n21;                   // (n21 of a diagonal block is n12's mirror)
          Gn[itD22[u]] = n22;
        }
        lds_barrier();
        double *tsw = Gc; Gc = Gn; Gn = tsw;
        cur ^= 1;
        last_big1 = max(last_big1, big_slot);
        ++round_idx;
  };
  if (n > 1) {
    kept2 = kept_scale(Gc);
    if (tid == 0) { k.sFlag[6] = 0; k.sFlag[7] = 0; }
    if (isParam) {                                   // rotations of the very first round
      const double2 top = *reinterpret_cast<const double2 *>(Gc + (2 * tid) * ne + 2 * tid);
      const RotT r = jacobi_rot_f32((float)top.x, (float)Gc[(2 * tid + 1) * ne + 2 * tid + 1], (float)top.y,
                                    fmaxf((float)kept2, kept_lo), (float)abs2, (float)p.svd_stop2);
      publish(k.dCS + (cur * np + tid) * 4, r, 0);
    }
    lds_barrier();
    // The iteration ends as soon as ne - 1 consecutive rounds -- any such window is a complete sweep over all pairs --
    // applied no big rotation: quadratic convergence then leaves off-diagonals of relative size ~svd_stop2, exactly the
    // guarantee of "a whole sweep without a big rotation", but the window need not start at a sweep boundary (it saves
    // about a third of a sweep per decomposition once the chain has settled).
    for (; sweeps < kJacobiMaxSweeps && !converged; ++sweeps) {
      for (int rnd = 0; rnd < ne - 1; ++rnd) {
        jacobi_round();
        // block-uniform, and told so: the flag word comes out of LDS (a vector register), and a loop exit the compiler has
        // to treat as divergent wraps every round in EXEC-mask bookkeeping
        if (round_idx - __builtin_amdgcn_readfirstlane(last_big1) >= ne - 1) { converged = 1; break; }
      }
      if (!converged) kept2 = kept_scale(Gc);
    }
  } else {
    converged = 1;
  }
  // the eigenvectors leave the registers: V[row][column position], as phase 9 reads them
  double *V = V0;
  if (vLaneOk) {
#pragma unroll
    for (int r = 0; r < kVR; ++r) {
      const int P = vP0 + r;
      if (r < vr && P < np) {
        *reinterpret_cast<double2 *>(V + (2 * P) * ne + 2 * vQ) = make_double2(vb[r][0], vb[r][1]);
        *reinterpret_cast<double2 *>(V + (2 * P + 1) * ne + 2 * vQ) = make_double2(vb[r][2], vb[r][3]);
      }
    }
  }
  if (p.stamps && tid == 0) t_c2 = __builtin_amdgcn_s_memtime();

  // ---- phase 8: eigenvalues = diag(G), descending order -------------------------------------------
  for (int j = tid; j < n; j += NT) k.dLam[j] = __builtin_amdgcn_ldexp(fmax(Gc[j * ne + j], 0.0), sc_exp);
  lds_barrier();
  if (use_chol) {
    // back from the eigenvectors u of G' = L^T L to those of G: v = L u / sqrt(lambda) (columns stay at their positions)
    for (int j = tid; j < n; j += NT) {                  // 1 / sqrt(lambda_j) once per column (dSq is free until phase 9)
      const double ls = Gc[j * ne + j];
      k.dSq[j] = ls > 1e-300 ? 1.0 / sqrt(ls) : 0.0;
    }
    lds_barrier();
    mm_lds(1, n, n, n, Lm, 0, 1, ne, V, 0, ne, 1, [&](int, int i, int j, double v) { Gn[i * ne + j] = v * k.dSq[j]; });
    V = Gn;
    lds_barrier();
  }
  for (int j = wave_u; j < n; j += NT >> 6) {           // one wave per entry: lane i votes "i sorts before j" (n <= 64)
    const int i = tid & 63;
    const double lj = k.dLam[j];
    const double li = i < n ? k.dLam[i] : 0.0;
    const int rank = __popcll(__ballot(i < n && ((li > lj) || (li == lj && i < j))));
    if (i == 0) {
      k.sOrd[rank] = j;
      if (p.dbg) p.dbg[4 * (size_t)Bs + rank] = sqrt(lj);
    }
  }
  if (tid == 0) {
    if (p.counters) {
      atomicAdd(p.counters, (unsigned long long)sweeps);                 // started sweeps (the last one may be partial)
      atomicAdd(p.counters + 1, 1ull);
      atomicAdd(p.counters + 2, (unsigned long long)round_idx);
      if (use_chol) atomicAdd(p.counters + 3, 1ull);
    }
    if (!converged) atomicOr(p.status, 2);
    if (p.dbg) {
      double *sc = p.dbg + 4 * (size_t)Bs + kDbgSigma;
      sc[3] = (double)sweeps;
      sc[4] = (double)n;
    }
  }
  lds_barrier();

  if (p.stamps && tid == 0) t_c2b = __builtin_amdgcn_s_memtime();
  // ---- adaptive truncation (not reference behaviour: the reference computes this index and never uses it,
  // Network_class.py:889-891): keep the fewest singular values whose cumulative share exceeds the threshold
  int mk = m, ob_s_h = p.ob_s_h, ob_s_d = p.ob_s_d, oa_s_d = p.oa_s_d, oa_s_g = p.oa_s_g;
  if (p.trunc_thr > 0.0) {
    if (tid == 0) {
      double tot = 0.0;
      for (int j = 0; j < n; ++j) tot += sqrt(k.dLam[k.sOrd[j]]);
      double cum = 0.0;
      int idx = 0;
      bool found = false;
      for (int j = 0; j < n && !found; ++j) {
        cum += sqrt(k.dLam[k.sOrd[j]]);
        if (cum / tot > p.trunc_thr) { idx = j; found = true; }      // np.argmax(cumsum(S) / S.sum() > threshold)
      }
      const int me = min(m, idx + 1);
      k.sFlag[2] = me;
      if (p.m_out) *p.m_out = me;
    }
    lds_barrier();
    mk = k.sFlag[2];
    if (!p.left_dir) { ob_s_h = D * mk; ob_s_d = mk; } else { oa_s_d = mk * L; oa_s_g = D * mk * L; }
  }
  // ---- phase 9: the two new cores -----------------------------------------------------------------
  // (persistent sweep: the batch-side workgroups of this launch extend their environments with the behind core -> agent scope)
  auto st_behind = [&](int off, float v) { if (p.persist) st_sc1(p.out_behind + off, v); else p.out_behind[off] = v; };
  const double lam_max = k.dLam[k.sOrd[0]];
  for (int sp = tid; sp < mk; sp += NT) {                 // sigma^(+-1/2) once per kept column (lam = sigma^2)
    const double lam = k.dLam[k.sOrd[sp]];
    const bool ok = lam > 1e-300 && lam > 1e-30 * lam_max;
    const double sq = ok ? sqrt(sqrt(lam)) : 0.0;
    k.dSq[sp] = sq;
    k.dSq[ne + sp] = ok ? 1.0 / sq : 0.0;
  }
  lds_barrier();
  // short-side factor: q_j * sigma_j^(1/2)   (rows kk over waves, kept columns over lanes: no divisions).  The same pass
  // gathers the kept eigenvectors, scaled by sigma_j^(-1/2), into a dense [n][mk] matrix for the long-side product below
  // (no per-element indirection through the order table there, no scaling in its epilogue)
  for (int kk = wave_u; kk < n; kk += NT >> 6)
   for (int sp = tid & 63; sp < mk; sp += 64) {
    const int j = k.sOrd[sp];
    const double vq = V[kk * ne + j];
    const float v = (float)(vq * k.dSq[sp]);
    k.dVs[kk * mk + sp] = vq * k.dSq[ne + sp];
    if (short_rows) {                       // kk = row index i = h_*D + dk  -> behind core
      k.sCb[kk * mk + sp] = v;
      if (p.persist) PL.Ad[kk * mk + sp] = vq * k.dSq[sp];
      st_behind((kk / D) * ob_s_h + (kk % D) * ob_s_d + sp * p.ob_s_m, v);
    } else {                                // kk = column index (dk1*g + g_)*L + l -> ahead core
      const int l = kk % L, q = kk / L;
      p.out_ahead[sp * p.oa_s_m + (q / g) * oa_s_d + (q % g) * oa_s_g + l] = v;
    }
   }
  lds_barrier();
  // long-side factor: W (q_j / sigma_j^(1/2)); columns s', inner index the short one.  When the short side is the behind core
  // (short_rows), the first product of the next behind norm environment, T2 = Nh . Cb, needs nothing the long-side product
  // writes: the two run back to back without a barrier, their tiles dealt to the waves as one list.
  const int DM = D * mk;
  auto store_T2 = [&](int, int i, int j, double v) { k.dT2[i * DM + j] = v; };
  if (short_rows) {
    // long index = ahead group x = (dk1, g_, l) = qq * L + l: the label is the batch, rows are qq = (dk1, g_)
    // (persistent sweep: the next step forms its merged tensor from T_{k+1} and this step's behind core, so the new label core
    // is only computed where somebody reads it -- at the last step of the launch)
    int slot = 0;
    if (!p.persist || p.write_ahead)
      slot = mm_lds(L, D * g, mk, n, k.fB, 1, L, c, k.dVs, 0, mk, 1,
           [&](int l, int qq, int sp, double acc) {
             const int dk1 = qq >= g ? 1 : 0;                     // D == 2
             p.out_ahead[__mul24(sp, p.oa_s_m) + dk1 * oa_s_d + __mul24(qq - dk1 * g, oa_s_g) + l] = (float)acc;
           });
    // T2[h_, (d, s'')] = sum_h' Nh[h_, h'] Cb[h', (d, s'')]
    if (p.Nh_new) mm_lds(1, h, DM, h, k.dNh, 0, h, 1, k.sCb, 0, DM, 1, store_T2, false, slot);
  } else {
    // long index = behind group x = (h_, dk) = h_ * D + dk: dk is the batch, rows are h_
    mm_lds(D, h, mk, n, k.fBp, c + 1, D * (c + 1), 1, k.dVs, 0, mk, 1,
           [&](int dk, int h_, int sp, double acc) {
             const float v = (float)acc;
             k.sCb[__mul24(h_ * D + dk, mk) + sp] = v;
             if (p.persist) PL.Ad[__mul24(h_ * D + dk, mk) + sp] = acc;
             st_behind(__mul24(h_, ob_s_h) + dk * ob_s_d + __mul24(sp, p.ob_s_m), v);
           });
  }
  lds_barrier();
  if (p.persist) {
    // The next step's projections are formed by the helper workgroups: they get the behind core before its rounding to float32
    // (the projection divides by sigma: rounding errors of A' would come back multiplied by sigma_max / sigma_j) and 1 / sigma, now,
    // so that the stores travel while the norm environment is formed; the batch-side workgroups extend their environments with
    // the float32 core in its slot.  Agent-scope stores; the flag follows at the end of the step.
    for (int e = tid; e < r * mk; e += NT) __hip_atomic_store(p.Apub + e, PL.Ad[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int sp = tid; sp < mk; sp += NT) {
      const double iq = k.dSq[ne + sp];
      __hip_atomic_store(p.Apub + r * mk + sp, iq * iq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  if (p.stamps && tid == 0) t_c2c = __builtin_amdgcn_s_memtime();
  // ---- phase 10: behind norm environment of the next step ------------------------------------------
  if (p.Nh_new) {
    if (!short_rows) {
      mm_lds(1, h, DM, h, k.dNh, 0, h, 1, k.sCb, 0, DM, 1, store_T2);
      lds_barrier();
    }
    // Nh_new[s', s''] = sum_{(h_, d)} Cb[(h_, d), s'] T2[(h_, d), s'']
    mm_lds(1, mk, mk, h * D, k.sCb, 0, 1, mk, k.dT2, 0, mk, 1,
           [&](int, int i, int j, double v) {
             p.Nh_new[i * mk + j] = v;
             if (p.persist) { PL.Nh[i * mk + j] = v; __hip_atomic_store(p.Apub + r * mk + mk + i * mk + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
           });
  }

  if (p.stamps && tid == 0) {
    // diagnostic stamps: shader cycles before / in / after the Jacobi loop and the 100 MHz real-time
    // counter, from which the host derives the clock the kernel ran at; they go to a buffer nothing
    // else reads
    const unsigned long long t_c3 = __builtin_amdgcn_s_memtime(), t_r3 = __builtin_amdgcn_s_memrealtime();
    p.stamps[0] = (double)(t_c1 - t_c0); p.stamps[1] = (double)(t_c2 - t_c1); p.stamps[2] = (double)(t_c3 - t_c2);
    p.stamps[3] = (double)(t_r3 - t_r0); p.stamps[4] = (double)sweeps; p.stamps[5] = (double)n;
    p.stamps[50] = (double)round_idx;                 // rounds actually run (the sliding window stops inside a sweep)
    for (int i = 0; i < 5; ++i) p.stamps[9 + i] = (double)(t_p[i] - (i ? t_p[i - 1] : t_c0));
    p.stamps[6] = (double)(t_c2b - t_c2); p.stamps[7] = (double)(t_c2c - t_c2b); p.stamps[8] = (double)(t_c3 - t_c2c);
  }

  // ---- phase 11: metrics of this step (var_hist, Network_class.py:739-750) --------------------------
  if (tid == 0 && p.metrics) {
    const double cnt = (double)ldtail(3);
    const double inv = cnt > 0 ? 1.0 / cnt : 0.0;
    p.metrics[0] = (float)((double)ldtail(0) * inv);
    p.metrics[1] = (float)((double)ldtail(1) * inv / (double)L);
    if (ldtail(2) != 0.f) atomicOr(p.status, 1);
  }
  if (p.persist) {
    // (the behind core, 1 / sigma and the behind norm environment left for the helper workgroups as their products finished: see
    // publish_core / the norm-environment product) every storing wave drains, the workgroup meets, one lane raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tid == 0) __hip_atomic_store(p.coreflag, p.coretoken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (p.done_flag) {
    // everything this workgroup wrote (the two cores, the norm environment, the metrics) out to where the side stream's kernels read
    // it, then the sequence number: what the end of the launch + an event would do, 6 us of this stream's time cheaper
    // (every wave's stores are in this XCD's L2 once its counter has drained; ONE wave then writes the L2 back)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave_u == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (tid == 0) __hip_atomic_store(p.done_flag, p.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return false;
}

// classic step: one narrow launch (workgroup 0 + optional reduce / slice helpers)
__global__ __launch_bounds__(kNarrowThreads) void narrow_step_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (p.fused && blockIdx.x > 0) { narrow_helper_block(p, smem_raw); return; }
  narrow_body(p, smem_raw);
}

void launch_narrow(const NarrowParams &p, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(narrow_step_kernel, dim3(p.fused ? 1 + p.wait_count : 1), dim3(kNarrowThreads), lds_bytes, st, p);
}

// pipelined step (wide_pipe_device.h): ONE launch per sweep step.  Block 0 updates and splits the merged tensor of step k,
// blocks 1..w.wg0-1 are its slice helpers (merged tensor and L2 term of step k: D*D slices, cut into row parts), blocks w.wg0.. are the batch-side workgroups
// that turn B_new(k) into f and the pre-gradient of step k+1 while block 0 runs the SVD.  With w.wg0 == 0 the launch
// carries batch-side workgroups only (start of a sweep, or after a classic step).
__global__ __launch_bounds__(kNarrowThreads) void step_pipe_kernel(NarrowParams p, WidePipeParams w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int blk = blockIdx.x;
  if (blk >= w.wg0) { wide_pipe_block(w, (float *)smem_raw); return; }
  if (blk > 0) { narrow_helper_block(p, smem_raw); return; }
  narrow_body(p, smem_raw);
}

void launch_step_pipe(const NarrowParams &p, const WidePipeParams &w, size_t lds_bytes, hipStream_t st) {
  const int grid = w.wg0 + (w.do_f || w.do_z || w.do_ext ? w.nwide : 0);
  hipLaunchKernelGGL(step_pipe_kernel, dim3(grid), dim3(kNarrowThreads), lds_bytes, st, p, w);
}

void launch_step_pipe_update(const NarrowParams &p, const WidePipeParams &w, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(step_pipe_kernel, dim3(w.wg0), dim3(kNarrowThreads), lds_bytes, st, p, w);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent sweep: ONE launch per sweep (single GPU, every step in the in-LDS regime, bond dimensions known in advance).
//   workgroup 0      update + SVD of step k, k = 0 .. n_steps-1 (narrow_body in its persistent mode)
//   workgroup 1      T_k = B_new(k-1) . A_{k+1} (the merged tensor of step k before the projection onto the new bond), formed
//                    beside the SVD of step k-1
//   workgroups 2..   the batch-side workgroups of the pipelined step (wide_pipe_block), each looping over the steps with its own
//                    samples: Z_0 from forward's f first, then per step: extend E with the new behind core, f from B_new(k), Z_{k+1}
// Against one launch per step this removes, from the critical path of every step: the launch itself, the fetch of a 900-byte
// argument block and of operands another launch wrote, and the slice helpers' two dependent product levels.
// All hand-offs follow the first row of the hand-off table of MI355X_MICROARCH.md (agent-scope stores, drained, barrier, one-lane
// flag; relaxed poll, barrier, agent-scope loads) on flags that only grow; every wait is bounded and gives up for the whole
// launch through one abort word, so the grid always drains.
// ------------------------------------------------------------------------------------------------------------------
// One helper workgroup of a persistent sweep, one call per step k.  Two parts:
//   (1) [beside the SVD of step k-1, once B_new(k-1) is stored]  its ROW slice of
//         T_k[i, d, (d', g), l]  = sum_s W[i, d, s, l] A_{k+1}(s, (d', g))       W = B_new(k-1) as [(h, d_{k-1}), d_k, s, l]; the label core at k == 0
//         TN_k[i, d, d', g', l]  = sum_g T_k[i, d, d', g, l] Ng[g, g']
//       in float64 (exact sums of float32 products), written to memory for everybody;
//   (2) [once the update workgroup has published A' = U sqrt(S), 1 / sigma and Nh of step k-1, every helper has finished (1) and the
//       pre-gradient Z_k is reduced]  its COLUMN slice of
//         dB_raw = A'^T Z_k,   B_k = diag(1 / sigma) A'^T T_k,   (Ln.B.Rn)_k = Nh^T diag(1 / sigma) A'^T TN_k
//       -> prepRaw / prepB / prepG, then its arrival.
// (2) of step k is the only part on the critical path of the sweep: three 2 x 2-tile products and one more, on LDS operands.
template <class HP>
__device__ __forceinline__ bool persist_helper_block(const HP &t, int hid, int nH, unsigned char *smem_raw, bool do_part2) {
  const int tid = threadIdx.x, NT = kNarrowThreads;
  const int D = kD, zr = t.zr, s = t.s, g = t.g, L = t.L, h = t.h;
  const int DG = D * g, RW = D * DG * L;
  __shared__ int sBad;
  auto give_up = [&](int bad) {            // called by thread 0
    if (bad == 1) { atomicOr(t.status, 4); __hip_atomic_store(t.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    sBad = bad;
  };
  if (!do_part2) {
    // ---------------- part 1: rows [r0, r1) of the zr * D rows (i, d) ----------------
    const int R1 = zr * D, per = (R1 + nH - 1) / nH, r0 = min(R1, hid * per), nr = min(R1, r0 + per) - r0;
    float *sW = (float *)smem_raw;                           // [nr][s][L]
    float *sP = sW + (((size_t)per * s * L + 3) & ~(size_t)3);   // [s][D g]
    double *sNg = (double *)(sP + (((size_t)s * DG + 3) & ~(size_t)3));   // [g][g]
    double *oT = sNg + (((size_t)g * g + 1) & ~(size_t)1);   // [nr][D g][L]
    double *oN = oT + (size_t)per * DG * L;                  // [nr][D][g][L]
    if (tid == 0) give_up(t.want ? spin_wait_ge(t.flag, t.want, t.abort_flag) : 0);
    lds_barrier();
    if (sBad) return true;
    if (nr > 0) {
      const int nW = nr * s * L;
      if (t.W) {
        const float *Wsrc = t.W + (size_t)r0 * s * L;
        for (int e = tid; e < nW; e += NT) sW[e] = ld_sc1(Wsrc + e);
      } else {                                               // k == 0: the label core, rows d (zr == 1)
        for (int e = tid; e < nW; e += NT) {
          const int l = e % L, q = e / L, s_ = q % s, d = r0 + q / s;
          sW[e] = t.lab.base[d * t.lab.s_d + s_ * t.lab.s_out + l];
        }
      }
      for (int e = tid; e < s * DG; e += NT) {
        const int g_ = e % g, q = e / g, d = q % D, s_ = q / D;
        sP[e] = t.pl.base[s_ * t.pl.s_in + d * t.pl.s_d + g_ * t.pl.s_out];
      }
      if (t.l2_flag) for (int e = tid; e < g * g; e += NT) sNg[e] = t.Ng ? t.Ng[e] : 1.0;
      lds_barrier();
      mm_lds(L, nr, DG, s, sW, 1, s * L, L, sP, 0, DG, 1, [&](int l, int row, int col, double v) { oT[(row * DG + col) * L + l] = v; });
      lds_barrier();
      if (t.l2_flag)      // rows (row, d'), inner index g, columns g'
        mm_lds(L, nr * D, g, g, oT, 1, g * L, L, sNg, 0, g, 1, [&](int l, int i, int j, double v) { oN[(i * g + j) * L + l] = v; });
      // the row slice is one contiguous block of T_k (and of TN_k)
      const int nT = nr * DG * L;
      double *Tdst = t.T + (size_t)r0 * DG * L, *Ndst = t.TN + (size_t)r0 * DG * L;
      for (int e = tid; e < nT; e += NT) __hip_atomic_store(Tdst + e, oT[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t.l2_flag) {
        lds_barrier();
        for (int e = tid; e < nT; e += NT) __hip_atomic_store(Ndst + e, oN[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tid == 0) __hip_atomic_fetch_add(t.tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  // ---------------- part 2: columns [c0, c0 + nc) of the RW columns ----------------
  // Order of the waits = order in which the operands become final: T_k / TN_k (beside the previous SVD), Z_k (shortly before or
  // after the previous step ends), A' / 1 / sigma / Nh (the previous step's last act).  Only the last load and the two products
  // are on the critical path of the sweep.
  const int cw = (RW + nH - 1) / nH, c0 = min(RW, hid * cw), nc = min(RW, c0 + cw) - c0;
  const int cw3 = 3 * cw;
  double *sA = (double *)smem_raw;                           // [zr][h]  A' (float64)
  double *sIv = sA + (size_t)zr * h;                         // [h]      1 / sigma
  double *sNh = sIv + ((h + 1) & ~1);                        // [h][h]
  double *sS = sNh + (((size_t)h * h + 1) & ~(size_t)1);     // [zr][3 cw]  (Z | T | TN) columns of this slice
  double *sP2 = sS + (size_t)zr * cw3;                       // [h][cw]  diag(1 / sigma) A'^T TN
  float *sZf = (float *)(sP2 + (size_t)h * cw);              // [zr][cw] Z columns of this slice
  float *sAff = sZf + (size_t)zr * cw;                       // [zr][h]  A' as stored (float32)
  double *hst = (t.stamps && hid == 0 && tid == 0) ? t.stamps : nullptr;
  auto rt = []() { return (double)(__builtin_amdgcn_s_memrealtime() & ((1ull << 40) - 1)); };
  if (tid == 0) { if (hst) hst[22] = rt(); give_up(spin_wait_ge(t.tcnt, (unsigned)nH, t.abort_flag)); if (hst) hst[30] = rt(); }
  lds_barrier();
  if (sBad) return true;
  for (int e = tid; e < zr * nc; e += NT) {
    const int i = e / nc, cc = e - i * nc;
    const size_t src = (size_t)i * RW + c0 + cc;
    sS[i * cw3 + cw + cc] = __hip_atomic_load(t.T + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sS[i * cw3 + 2 * cw + cc] = t.l2_flag ? __hip_atomic_load(t.TN + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  }
  lds_barrier();                                             // (tid 0 rewrites sBad)
  if (tid == 0) { give_up(spin_wait_ge(t.zready, t.zwant, t.abort_flag)); if (hst) hst[24] = rt(); }
  lds_barrier();
  if (sBad) return true;
  for (int e = tid; e < zr * nc; e += NT) {
    const int i = e / nc, cc = e - i * nc;
    sZf[i * cw + cc] = ld_sc1(t.Z + (size_t)i * RW + c0 + cc);
  }
  lds_barrier();
  if (tid == 0) { give_up(t.awant ? spin_wait_ge(t.aflag, t.awant, t.abort_flag) : 0); if (hst) hst[23] = rt(); }
  lds_barrier();
  if (sBad) return true;
  if (t.awant) {
    const double *pub = t.Apub;
    for (int e = tid; e < zr * h; e += NT) { const double a = __hip_atomic_load(pub + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sA[e] = a; sAff[e] = (float)a; }
    for (int e = tid; e < h; e += NT) sIv[e] = __hip_atomic_load(pub + zr * h + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t.l2_flag) for (int e = tid; e < h * h; e += NT) sNh[e] = __hip_atomic_load(pub + zr * h + h + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (tid == 0) {                                     // k == 0: zr == h == 1, the identity
    sA[0] = 1.0; sAff[0] = 1.f; sIv[0] = 1.0; sNh[0] = 1.0;
  }
  lds_barrier();
  if (hst) hst[25] = rt();
  if (nc > 0) {
    // [h][cw] = Af^T Z with the core as stored, on the float32 matrix pipe (the per-step path's contraction, number for number), and
    // [h][2 cw] = A'^T (T | TN) with the unrounded core; independent, their tiles dealt as one list
    mm_lds_f32(h, nc, zr, sAff, 1, h, sZf, cw, 1, [&](int i, int j, float v) { st_sc1(t.prepRaw + (size_t)i * RW + c0 + j, v); });
    const int slot = (((h + 15) >> 4) * ((nc + 15) >> 4)) & ((NT >> 6) - 1);
    mm_lds(1, h, t.l2_flag ? 2 * cw : cw, zr, sA, 0, 1, h, sS + cw, 0, cw3, 1,
           [&](int, int i, int j, double v) {
             if (j < cw) { if (j < nc) st_sc1(t.prepB + (size_t)i * RW + c0 + j, (float)(v * sIv[i])); }
             else sP2[i * cw + (j - cw)] = v * sIv[i];
           }, false, slot);
    if (t.l2_flag) {
      lds_barrier();
      // (Ln.B.Rn)[e_, c] = sum_a Nh[a, e_] P2[a, c]
      mm_lds(1, h, nc, h, sNh, 0, 1, h, sP2, 0, cw, 1,
             [&](int, int i, int j, double v) { __hip_atomic_store(t.prepG + (size_t)i * RW + c0 + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
    }
  }
  if (hst) hst[26] = rt();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0) __hip_atomic_fetch_add(t.pcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (hst) hst[27] = rt();
  return false;
}

__global__ __launch_bounds__(kNarrowThreads) void sweep_persist_kernel(const PersistStep *__restrict__ steps_g, int n_steps, int nH) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the host wrote the records before the launch and nothing in the kernel writes them: constant address space -> scalar loads
  // Every role works on a by-value copy of its record for the step (scalar registers, as the per-step kernels' arguments): fields
  // read on demand from memory would put scalar-load latencies into the inner loops.
  const int blk = blockIdx.x;
  if (blk == 0) {
#pragma nounroll
    for (int k = 0; k < n_steps; ++k) {
      NarrowParams n;
      __builtin_memcpy(&n, &steps_g[k].n, sizeof n);
      if (narrow_body(n, smem_raw)) break;
    }
  } else if (blk <= nH) {
    // T_0; then per step: the projections of step k, and T_{k+1} once B_new(k) is there -- one call site (the body is inlined once)
#pragma nounroll
    for (int ph = 0; ph < 2 * n_steps; ++ph) {
      PersistHelperParams t;
      __builtin_memcpy(&t, &steps_g[ph >> 1].t, sizeof t);
      lds_barrier();
      if (persist_helper_block(t, blk - 1, nH, smem_raw, (ph & 1) != 0)) break;
    }
  } else {
    // Z_0 from forward's f (record n_steps), then one iteration per step -- one call site
#pragma nounroll
    for (int it = 0; it <= n_steps; ++it) {
      WidePipeParams w;
      __builtin_memcpy(&w, &steps_g[it == 0 ? n_steps : it - 1].w, sizeof w);
      lds_barrier();                                       // the previous iteration's LDS arrays are dead
      if (wide_pipe_block(w, (float *)smem_raw)) break;
    }
  }
}

void launch_sweep_persist(const PersistStep *steps_dev, int n_steps, int n_helpers, int grid, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(sweep_persist_kernel, dim3(grid), dim3(kNarrowThreads), lds_bytes, st, steps_dev, n_steps, n_helpers);
}

// The same sweep as THREE launches, one per role, on three streams: the roles then get their own register allocation (inside one
// kernel the batch-side loops spill and lose their unrolling to the update workgroup's code, and run ~2.4x slower than in the
// per-step kernel).  The three grids communicate through the same flags; they need to be resident together, which holds when
// nothing serialises launches (88 workgroups on a 256-CU device) -- a dispatch-serialising profiler (rocprofv3 --pmc) makes the first
// grid wait in vain: its bounded polls time out and the sweep fails with TNML_ERR_STATE; use tnml_set_persistent(ctx, 1) (one
// kernel) or 0 (per-step launches) there.
__global__ __launch_bounds__(kNarrowThreads) void persist_update_kernel(const PersistStep *__restrict__ steps_g, int n_steps, int rec_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // The record of step k+1 is fetched into LDS while step k runs (two waves, at the start of the step) and becomes the by-value
  // argument of the next narrow_body from there: a fetch from memory between two steps is ~1.6 us on the critical path.
  constexpr int kWords = (int)(sizeof(NarrowParams) / sizeof(unsigned));
  static_assert(sizeof(NarrowParams) % sizeof(unsigned) == 0, "NarrowParams is copied word by word");
  unsigned *rec = reinterpret_cast<unsigned *>(smem_raw + rec_off);
  for (int e = threadIdx.x; e < kWords; e += kNarrowThreads) rec[e] = reinterpret_cast<const unsigned *>(&steps_g[0].n)[e];
  lds_barrier();
#pragma nounroll
  for (int k = 0; k < n_steps; ++k) {
    NarrowParams n;
    __builtin_memcpy(&n, rec, sizeof n);
    lds_barrier();                                         // everybody holds its copy: the buffer may take the next record
    if (k + 1 < n_steps)
      for (int e = threadIdx.x; e < kWords; e += kNarrowThreads) rec[e] = reinterpret_cast<const unsigned *>(&steps_g[k + 1].n)[e];
    if (narrow_body(n, smem_raw)) break;
  }
}
__global__ __launch_bounds__(kNarrowThreads) void persist_helper_kernel(const PersistStep *__restrict__ steps_g, int n_steps, int nH) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#pragma nounroll
  for (int ph = 0; ph < 2 * n_steps; ++ph) {
    PersistHelperParams t;
    __builtin_memcpy(&t, &steps_g[ph >> 1].t, sizeof t);
    lds_barrier();
    if (persist_helper_block(t, (int)blockIdx.x, nH, smem_raw, (ph & 1) != 0)) break;
  }
}
__global__ __launch_bounds__(kNarrowThreads) void persist_batch_kernel(const PersistStep *__restrict__ steps_g, int n_steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#pragma nounroll
  for (int it = 0; it <= n_steps; ++it) {
    WidePipeParams w;
    __builtin_memcpy(&w, &steps_g[it == 0 ? n_steps : it - 1].w, sizeof w);
    w.wg0 = 0;                                             // this grid holds batch-side workgroups only
    lds_barrier();
    if (wide_pipe_block(w, (float *)smem_raw)) break;
  }
}
void launch_sweep_persist_split(const PersistStep *steps_dev, int n_steps, int n_helpers, int n_wide, size_t lds_update, size_t lds_helper,
                                size_t lds_wide, int rec_off, hipStream_t st_update, hipStream_t st_helper, hipStream_t st_wide) {
  hipLaunchKernelGGL(persist_batch_kernel, dim3(n_wide), dim3(kNarrowThreads), lds_wide, st_wide, steps_dev, n_steps);
  hipLaunchKernelGGL(persist_helper_kernel, dim3(n_helpers), dim3(kNarrowThreads), lds_helper, st_helper, steps_dev, n_steps, n_helpers);
  hipLaunchKernelGGL(persist_update_kernel, dim3(1), dim3(kNarrowThreads), lds_update, st_update, steps_dev, n_steps, rec_off);
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
