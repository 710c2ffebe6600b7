// Internal declarations shared by the translation units of libtnml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tnml.h"

namespace tnml {

constexpr int kD = 2;           // feature dimension the kernels are specialised for
constexpr int kTS = 32;         // samples per workgroup in the wide step kernel (v1)
constexpr int kWideThreads = 1024; // 4 waves per SIMD: the v1 formulation is latency-bound
constexpr int kNarrowThreads = 1024;
constexpr int kMetricSlots = 4;    // per-slab tail: correct count, sum|y-fa|, non-finite count, pad
constexpr int kDbgScalars = 120 + 16 * 12;   // 5 scalars + 115 stamps + (experiment builds) 12 probe points x 16 waves
constexpr int kDbgSigma = 128;     // capture block: 4 tensors, then this many singular values, then scalars
constexpr double kCholThrSmall = 0.35;     // the same for matrices with n < 32
constexpr double kCholThrDefault = 0.22;   // off(G)/trace(G) above which the Cholesky step pays off (kernels_narrow.hip phase 6b)
constexpr double kSvdStop2Default = 1e-6;   // see jacobi_rot (jacobi_device.h) and tnml_set_svd_stop
constexpr int kCounterSlots = 16;   // device counters: [0..3] Jacobi statistics, [4..7] fused-launch timing diagnostics, [8..] spare

// A plain (label-free) core or the label core addressed in the sweep-relative frame.
//   plain:  A(in, d, out)      = base[in*s_in + d*s_d + out*s_out]
//   label:  A(h, d, s, l)      = base[h*s_in + d*s_d + s*s_out + l]   (label stride is 1)
struct CoreView {
  const float *base;
  int n_in, n_out;
  int s_in, s_d, s_out;
};

// One site of the forward environment chain (see env_chain_kernel).
struct ChainSite {
  int core_off;     // float offset of the core (into cores[] or the label core when is_label)
  int is_label;
  int n_in, n_out;  // n_in = dimension of the incoming environment, n_out of the produced one
  int s_in, s_d, s_out;
  int x_site;       // absolute site whose features are contracted
  long long env_out_off;  // float offset of the produced environment slot, -1 for the label site
};

// Everything the wide step kernel needs, in the sweep-relative frame.
struct WideParams {
  int b, b_pad, L;
  int h, g;            // behind / ahead bond of the merged tensor of THIS step
  int hp, gp;          // the same for the previous step (f part); gp == shared bond of this step
  int do_f;            // recompute f from the previous step's updated B
  int do_ext;          // behind environment has to be extended (k >= 1)
  int first_ext;       // extension starts from the scalar 1 (k == 1)
  int act_fn, loss_fn;
  float T;
  const float *x_km1, *x_k, *x_kp1;   // [b_pad][D]
  const float *Hprev;  // behind env of the previous step  [hp][b_pad]
  float *Hcur;         // behind env of this step          [h][b_pad]  (written when do_ext)
  const float *Gprev;  // ahead env of the previous step   [gp][b_pad]
  const float *Gcur;   // ahead env of this step           [g][b_pad]
  const float *Bprev;  // updated merged tensor of the previous step, relative layout
  CoreView ext_core;   // core of site t=k-1: A(hp, d, h)
  const int *y;        // [b_pad]
  float *f;            // [L][b_pad]   read (do_f == 0) or written (do_f == 1)
  float *slabs;        // [nblk][slab_stride]
  int slab_stride;
  int bsize;           // h*D*D*g*L
  double *stamps;      // diagnostic cycle stamps of workgroup 0 (nullptr normally)
};

struct NarrowParams {
  int L, D;
  int h, g, s, m;          // behind, ahead, shared bond before the step; kept rank m
  int bsize;               // h*D*D*g*L
  int l2_flag;
  float lr, wd;
  double inv_b_global;     // 1 / global batch
  const float *red;        // reduced slab: dB_raw (relative layout) + metric tail
  CoreView lab;            // label core of site t=k:   A(h, d, s, l)
  CoreView pl;             // plain core of site t=k+1: A(s, d, g)
  const double *Nh;        // behind norm env (h x h) or nullptr (== [[1]])
  const double *Ng;        // ahead  norm env (g x g) or nullptr
  float *Bnew;             // out: updated merged tensor, relative layout
  float *out_behind;       // out: new plain core of site t=k
  int ob_s_h, ob_s_d, ob_s_m;          // strides of (h, d, s') in out_behind
  float *out_ahead;        // out: new label core of site t=k+1
  int oa_s_m, oa_s_d, oa_s_g;          // strides of (s', d, g) in out_ahead (label stride 1)
  double *Nh_new;          // out: behind norm env of the next step (m x m)
  float *metrics;          // out: (accuracy, MAE) of this step
  double *dbg;             // debug block (see narrow kernel), may be nullptr
  double *stamps;          // cycle stamps (diagnostic), may be nullptr
  unsigned long long *counters;  // [0] += jacobi sweeps, [1] += SVDs, [2] += jacobi rounds, [3] Cholesky steps (always on)
  const float *Bdirect;    // if set: the merged tensor (relative layout) is given, the two cores are not read
  int stop_after_update;   // 1: return after B_new (standalone update_B / compute_L2_reg; needs dbg)
  // fused launch (single GPU, in-LDS path): workgroups 1.. of the same launch reduce the gradient slabs and
  // compute the merged tensor and the L2 term slice by slice; workgroup 0 waits for them on `sync`
  int fused;               // 1: gridDim.x = 1 + nred (+ D*D slice workgroups unless prep_ready)
  int prep_ready;          // prepB / prepG were produced by the preceding wide launch
  const float *slabs;      // [nslabs][slab_stride]
  int nslabs, slab_stride, nred;
  float *red_out;          // == red
  float *prepB;            // [bsize] merged tensor (float), written by the slice workgroups
  double *prepG;           // [bsize] Ln.B.Rn
  unsigned *sync;          // arrival counter, zero between launches
  double trunc_thr;        // > 0: adaptive truncation threshold on cumsum(S) / sum(S); m is then the cap
  int left_dir;            // direction (only read when trunc_thr > 0: the output strides follow the kept rank)
  int *m_out;              // device int receiving the kept rank (adaptive truncation), may be nullptr
  double chol_thr;         // > 0: one pivoted-Cholesky step before the Jacobi iteration when off(G) / trace(G) exceeds it
  double svd_stop2;        // Jacobi stops after a sweep whose rotations all had g^2 / scale^2 <= svd_stop2 (tnml_set_svd_stop)
  int *status;             // device status word: bit0 non-finite, bit1 jacobi not converged, bit2 helpers late, bit3 flag never seen
  int wait_count;          // fused / pipelined launch: helper arrivals workgroup 0 waits for on `sync`
  // pipelined step (wide_pipe_device.h): the raw gradient is A_{k-1}^T . Z_k, Z_k reduced by the previous launch
  int pipe;                // 1: `zred` replaces `red`
  int z_first;             // first step of a sweep: Z_0 is the gradient itself
  int z_rows;              // h_{k-1} * D
  int zsize;               // elements of Z_k (the metric tail follows)
  const float *zred;
  CoreView zcore;          // A_{k-1}(h_{k-1}, d, h_k)
  unsigned *flag;          // if set: B_new is stored with agent-scope stores and `token` is written here afterwards
  unsigned token;
  // persistent sweep (sweep_persist_kernel, kernels_narrow.hip): the workgroup loops over the steps of a sweep; merged tensor, L2 term
  // and raw gradient of step k come from the helper workgroups of the launch (projections with the behind core of step k-1)
  int persist;             // 1: called from sweep_persist_kernel (pipe == 1 as well)
  int write_ahead;         // the new label core is wanted (last step of the launch); otherwise its product is skipped
  int persist_off;         // byte offset of the persistent LDS region (PersistLds), Mcap its capacity per bond
  int Mcap;
  const float *prepRaw;    // raw gradient of this step [h][RW] (beside prepB / prepG), written by the helper workgroups
  const unsigned *pready;  // >= pwant: every helper workgroup has stored its slice of the three
  unsigned pwant;
  double *Apub;            // out: behind core before rounding [D h][m], 1 / sigma [m], next behind norm environment [m][m]
  unsigned *coreflag;      // set to coretoken once Apub and the float32 behind core are stored: helpers project, batch side extends
  unsigned coretoken;
  // two-stream communicator path without events (tnml_api.hip, split branch): the update workgroup waits for zpoll_flag to reach
  // zpoll_want before it touches zred (the side stream's all-reduce is followed by a one-thread kernel that stores it), and stores
  // done_val to done_flag behind an agent-scope release when everything it writes is out (the side stream's next batch launch sits
  // behind a one-wave kernel that waits for it)
  const unsigned *zpoll_flag; unsigned zpoll_want;
  unsigned *done_flag; unsigned done_val;
  unsigned *abort_flag;    // set by any workgroup whose wait timed out; every wait of the launch gives up once it is set
};

// LDS that survives from one step of a persistent sweep to the next (update workgroup only)
struct PersistLds { double *Nh, *Ad; };
__host__ __device__ inline PersistLds persist_lds(unsigned char *base, int Mcap) {
  PersistLds q;
  q.Nh = (double *)base;                       // [m][m]   behind norm environment of the next step
  q.Ad = q.Nh + (size_t)Mcap * Mcap;           // [D h][m] behind core of the step just finished (U sqrt(S)) before its rounding to float32
  return q;
}
inline size_t persist_lds_bytes(int Mcap) {
  return ((size_t)Mcap * Mcap + (size_t)kD * Mcap * Mcap) * sizeof(double) + 16;
}

struct NormChainSite {
  int core_off;
  int n_in, n_out, s_in, s_d, s_out;   // A(in, d, out): env over `in` -> env over `out`
  long long env_out_off;               // double offset of the produced norm environment
};

void launch_transpose_input(const float *X_bnd, float *X_nbd, int b, int b_pad, int N, hipStream_t st);
void launch_env_chain(const ChainSite *sites_dev, int n_sites, const float *cores, const float *labcore,
                      const float *X, float *env_base, float *f, int b, int b_pad, int L, int Mmax,
                      float *logmax_out, hipStream_t st, bool force_plain = false);
constexpr int kChainSamplesPerBlock = 16;
// slice-wise pre-computation of the merged tensor and its L2 term (small_gemm_device.h: prep_slice_block)
struct PrepParams {
  CoreView lab, pl;        // as NarrowParams
  const double *Nh, *Ng;
  int h, g, s, L, l2_flag;
  float *prepB;
  double *prepG;
  int nparts;              // row parts per slice (0 == 1): workgroup index = part * D * D + slice
};

// false: no kernel of this build fits the tile's operands into LDS.  *prep_done: the launch carried the D*D slice workgroups
// that fill prep.prepB / prep.prepG
bool launch_wide(const WideParams &p, int nblk, const PrepParams *prep, hipStream_t st, bool *prep_done);
void launch_f_only(const WideParams &p, int nblk, hipStream_t st);
void launch_reduce(const float *slabs, int nblk, int slab_stride, int n, float *red, hipStream_t st);
void launch_narrow(const NarrowParams &p, size_t lds_bytes, hipStream_t st);   // grid = p.fused ? 1 + p.wait_count : 1
// Large-tensor path of the same step (kernels_big.hip): HBM scratch, n = min(rows, cols) <= 128.
struct BigScratch {
  float *Bf;          // [bmax]   merged tensor
  double *T;          // [bmax]   Nh^T . B, or (factored form) NL = Nh^T . lab followed by PR = pl . Ng
  double *part;       // [kBigParts][3] block-partial sums, then {step factor, sum|B|, sum|dv|, L2 sum}
  double *gram;       // [8][128][128] partial Gram matrices
  double2 *rotlog;    // [(kJacobiMaxSweeps * 127 + 2)][64] rotations (c, s) in application order
  double *lam;        // [3][128] eigenvalues by position, sigma^(1/2), sigma^(-1/2) of the kept columns
  int *info;          // rounds applied, sweeps, converged, kept rank, then the eigenvalue order [128]
  double *VW;         // [(rows + cols)][n]  V, then W^T V
  float *Cb;          // [rows][m]  new behind core, contiguous
  double *T2;         // [rows][m]
  unsigned *prog;     // [0..6] progress / final word / time-out notes of the replay that rides in the Jacobi launch, [7] arrival
                      //        counter of the weight-decay kernel's blocks (kernels_big.hip)
};
constexpr int kBigMaxN = 128;
constexpr int kBigParts = 2048;
size_t big_jacobi_lds_bytes(int n);
// false: a launch of the path was illegal or (check) failed; big_launch_error() names it
// `front`: the raw gradient of a pipelined large-tensor step, red = A^T . Z, formed in the SAME launch as the merged tensor
// (neither needs the other); without a merged tensor to form it is launched alone, in front of the chain
struct BigFront {
  const float *Z; CoreView A; int ncols; float *red;
  // optional: the behind environment E_k and P'_k = E_k (x) x_k of the NEXT step's batch kernel in the same launch (big_ext_kernel's work)
  const float *ext_Eprev = nullptr, *ext_x_km1 = nullptr, *ext_x_k = nullptr; CoreView ext_A{}; int b_pad = 0;
  float *ext_Ecur = nullptr, *ext_Pk = nullptr;
  // optional: Z (and what the extension rewrites) belongs to the side stream until poll_flag has reached poll_want (big_signal_kernel);
  // wait_ev: the event to fall back on where the contraction is launched alone
  unsigned *poll_flag = nullptr; unsigned poll_want = 0; hipEvent_t wait_ev = nullptr;
  bool ext_acquire = false;      // the extension's inputs were written on the side stream too (behind the first pipelined step)
};
bool launch_narrow_big(const NarrowParams &p, const BigScratch &s, hipStream_t st, bool check, bool prep_only = false,
                       bool skip_prep = false, hipEvent_t after_update = nullptr, const BigFront *front = nullptr,
                       unsigned *sig_flag = nullptr, unsigned sig_val = 0);      // sig_flag: the Gram kernel stores sig_val there (instead of the event)
bool launch_big_signal(unsigned *flag, unsigned value, hipStream_t st);
bool launch_big_gate(const unsigned *flag, unsigned want, int *status, hipStream_t st);
// pipelined large-tensor step: behind environment + (h, d) operand of the pre-gradient; raw gradient = A^T . Z (kernels_big.hip)
bool launch_big_ext(const float *Eprev, const float *x_km1, const float *x_k, const CoreView &A, int b_pad, float *Ecur, float *Pk,
                    hipStream_t st);
bool launch_big_contract(const float *Zred, const CoreView &A, int ncols, float *red, hipStream_t st);
// LDS the tiled batch kernel asks for at these dimensions (0: the shape is beyond it)
size_t wide_tiled_lds_bytes(int L, int h, int hp, int gp, int g, bool ext);
void launch_wide_tiled(const WideParams &p, int nblk, size_t lds_bytes, hipStream_t st);
const char *big_launch_error();
size_t narrow_lds_bytes(int h, int g, int s, int L, int m);
void launch_norm_chain(const NormChainSite *sites_dev, int n_sites, const float *cores, double *env_base,
                       int Mmax, hipStream_t st);
void launch_scale(float *p, size_t n, float factor, hipStream_t st);
void launch_absmax(const float *f, int L, int b, int b_pad, float *out, hipStream_t st);
void launch_activation(const float *f, const int *y, int L, int b, int b_pad, int act_fn, int loss_fn,
                       float T, float *act_out, float *der_out, hipStream_t st);

}  // namespace tnml
