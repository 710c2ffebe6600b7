/*
 * tnml.h -- C ABI of libtnml_hip.so: the MI355X (gfx950) backend of the MPS two-site sweep
 * optimiser.
 *
 * The reference (francescovidaich964/TensorNetworkForML) is pure Python/NumPy and has no FFI:
 * its boundary for this path is the Python API of TensorNetwork/Network_class.py.  Each entry
 * point below names the reference method (file:line under /root/reference/TensorNetwork) whose
 * arithmetic it replaces; INTEGRATION.md shows the ctypes stub a maintainer of the reference
 * would add to Network_class.py to call it.
 *
 * Conventions
 *   - every function returns 0 on success and a negative tnml_status on failure;
 *     tnml_last_error() returns a thread-local, human readable message for the last failure;
 *   - plain pointers and sizes only; host pointers unless a name ends in _dev;
 *   - a tnml_ctx owns all device memory, one HIP stream and (optionally) one RCCL communicator;
 *     one host thread per context; calls are stream-ordered and return after enqueueing unless
 *     they hand data back to the host (those synchronise the stream);
 *   - all tensors are float32 on the device; batch-independent norm environments and the
 *     merged-tensor update/SVD run in float64 inside the kernels;
 *   - canonical layouts (identical to oracle/mps_oracle.py):
 *       bond[i]      dimension of the bond between site i and i+1, i = 0..N-2
 *       core i       [ml][D][mr]      ml = bond[i-1] (1 at i = 0), mr = bond[i] (1 at i = N-1)
 *       core l_pos   [ml][D][mr][L]   the label axis is last on the site that carries it
 *       cores_flat   the N cores above concatenated in site order
 *       X            [b][N][D]        as Network.forward receives it (Network_class.py:195)
 *       f, g         [L][b]
 *       env          [b][m]           on the host side of tnml_get_env
 *       B            [ml][D][D][mr][L]  merged two-site tensor (a, d, d', c, l)
 */
#ifndef TNML_H
#define TNML_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tnml_ctx tnml_ctx;

typedef enum {
  TNML_OK = 0,
  TNML_ERR_ARG = -1,       /* bad argument / shape mismatch (the reference's AssertionError)   */
  TNML_ERR_STATE = -2,     /* call not allowed in this state (the reference's Exception)       */
  TNML_ERR_HIP = -3,       /* HIP runtime failure                                               */
  TNML_ERR_NOGPU = -4,     /* no usable gfx950 device                                           */
  TNML_ERR_COMM = -5,      /* RCCL failure                                                      */
  TNML_ERR_SHAPE = -6,     /* reference truncation policy hits the reference's own ValueError   */
  TNML_ERR_NONFINITE = -7  /* non-finite values reached the SVD (reference: LinAlgError)        */
} tnml_status;

/* activation / loss / truncation selectors (Network_class.py:127-133, :894-945) */
enum { TNML_ACT_LINEAR = 0, TNML_ACT_SIGMOID = 1, TNML_ACT_SOFTMAX = 2 };
enum { TNML_LOSS_MSE = 0, TNML_LOSS_CROSS_ENTROPY = 1, TNML_LOSS_FULL_CROSS_ENT = 2 };
enum { TNML_TRUNC_REFERENCE = 0, TNML_TRUNC_FIXED = 1, TNML_TRUNC_ADAPTIVE = 2 };
enum { TNML_SIDE_LEFT = 0, TNML_SIDE_RIGHT = 1 };

/* what tnml_get_step_debug can hand back about the most recent sweep step */
enum {
  TNML_DBG_B = 0,        /* merged tensor before the update        [ml][D][D][mr][L]            */
  TNML_DBG_DB_RAW = 1,   /* bond gradient before weight decay      same shape                    */
  TNML_DBG_B_NEW = 2,    /* updated, un-truncated merged tensor    same shape                    */
  TNML_DBG_SIGMA = 3,    /* all singular values, descending        [min(rows, cols)]             */
  TNML_DBG_L2 = 4,       /* {L2 loss term, sum|B|, sum|dB|, jacobi sweeps, n, cycles before /
                            in / after the Jacobi loop, 100 MHz ticks of the whole kernel}  [9]  */
  TNML_DBG_L2_GRAD = 5   /* 2*wd*Ln.B.Rn (or wd*B)                 same shape as B               */
};

const char *tnml_last_error(void);
const char *tnml_version(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int tnml_device_count(void);

/* ---- life cycle ------------------------------------------------------------------------- */
/* Network.__init__ (Network_class.py:84-191) minus the random init, which stays on the host.
 * b_capacity: largest batch a later tnml_set_input may bring (buffers grow if exceeded). */
int tnml_create(tnml_ctx **out, int N, int D, int L, int Mmax, int b_capacity, int device);
int tnml_destroy(tnml_ctx *ctx);
int tnml_synchronize(tnml_ctx *ctx);

/* ---- multi-GPU: batch shards, one RCCL all-reduce of the bond gradient per step ---------- */
/* 128-byte RCCL unique id, created on rank 0 and handed to every rank by the caller. */
int tnml_comm_unique_id(void *uid128);
int tnml_comm_init(tnml_ctx *ctx, int rank, int nranks, const void *uid128);
/* With a communicator the pipelined step is launched in two parts -- update side on the context's stream, batch side on a second
 * stream followed by the all-reduce of the pre-gradient -- so that the exchange travels beside the SVD of the step instead of
 * behind it (on = 1, default).  on = 0: one fused launch per step with the all-reduce between launches (round 2). */
int tnml_set_comm_overlap(tnml_ctx *ctx, int on);
/* measurement: mean device time (us) of one all-reduce of n_floats floats on the exchange stream over `reps` back-to-back calls
 * (collective: every rank calls it; 0 without a communicator; needs one sweep before it) */
int tnml_comm_probe(tnml_ctx *ctx, int n_floats, int reps, double *us_per_allreduce);

/* ---- parameters ------------------------------------------------------------------------- */
/* replaces assignments to Network.As / Network.l_pos */
int tnml_set_cores(tnml_ctx *ctx, const float *cores_flat, size_t n_floats, const int32_t *bond,
                   int l_pos);
int tnml_cores_size(tnml_ctx *ctx, size_t *n_floats);
int tnml_get_cores(tnml_ctx *ctx, float *cores_flat, size_t capacity, int32_t *bond, int *l_pos);
/* every core *= factor: the calibration loop of Network.__init__ (Network_class.py:175-176) */
int tnml_scale_cores(tnml_ctx *ctx, double factor);

/* ---- batch ------------------------------------------------------------------------------ */
/* X [b][N][D] float32, y [b] int32 (may be NULL when only forward is wanted) */
int tnml_set_input(tnml_ctx *ctx, const float *X, const int32_t *y, int b);
/* A data loader that keeps several batches on the device: tnml_stage_batch copies X [b][N][D] and y [b] into device slot
 * `slot` (0..7, synchronous host -> device copy); tnml_select_batch makes a staged batch the resident one with device-side
 * work only (re-tiling to the site-major layout, label copy) and without waiting -- the counterpart of tnml_set_input for
 * inputs that are already in HBM.  No reference analogue (the reference's loaders hand NumPy arrays to forward,
 * Network_class.py:324-327). */
int tnml_stage_batch(tnml_ctx *ctx, int slot, const float *X, const int32_t *y, int b);
int tnml_select_batch(tnml_ctx *ctx, int slot);
/* labels of the resident batch alone (Network.sweep receives y after forward saw X,
 * Network_class.py:327-333) */
int tnml_set_labels(tnml_ctx *ctx, const int32_t *y, int b);

/* ---- hot path --------------------------------------------------------------------------- */
/* Network.forward (Network_class.py:195-258): builds the environment stack for the current
 * l_pos (0 -> right environments, N-1 -> left environments) and f.  f_out [L][b] may be NULL. */
int tnml_forward(tnml_ctx *ctx, float *f_out);
/* max |f| over the (global) batch after a forward: Network_class.py:169 */
int tnml_f_absmax(tnml_ctx *ctx, double *out);
/* log(max |f|) over the (global) batch with per-site renormalisation: the calibration of
 * Network.__init__ (Network_class.py:168-170) needs max|f| of the un-calibrated chain, which is
 * ~1e-66 at N = 784 and underflows float32; this variant is exact in any range and leaves the
 * environment stacks untouched. */
int tnml_forward_logabsmax(tnml_ctx *ctx, double *out);
/* the f the next sweep step starts from (Network.sweep's argument f, Network_class.py:384) */
int tnml_set_f(tnml_ctx *ctx, const float *f);
int tnml_get_f(tnml_ctx *ctx, float *f_out);

/* Network.sweep / sweep_step / update_B / tensor_svd / compute_L2_reg
 * (Network_class.py:384-436, 440-573, 577-763, 839-962, 966-1179): n_steps two-site steps in
 * the given direction, starting at the current l_pos.  A full sweep is n_steps = N-1 right
 * after tnml_forward.
 *   first_of_sweep  non-zero: reset the environment list grown by this direction (:426-429)
 *   metrics_out     [n_steps][2] = (accuracy, MAE) per step (var_hist, :739-750), or NULL
 *   f_out           [L][b] output recomputed from the last updated, un-truncated B (:494-523) */
int tnml_sweep(tnml_ctx *ctx, int left_dir, int n_steps, int first_of_sweep, float lr,
               float weight_dec, int l2_flag, int act_fn, int loss_fn, float T, int trunc_policy,
               float *metrics_out, float *f_out);

/* The three sub-steps the reference also exposes as methods, as standalone device calls (the same
 * kernels as tnml_sweep, run in "stop after the update" / "given merged tensor" modes).
 *   tnml_update_B   Network.update_B (Network_class.py:577-763): extends the behind environment,
 *                   returns the updated merged tensor of sites (p, p+1), p = l_pos (- 1 when left_dir);
 *                   B_canon [ml][D][D][mr][L] or NULL (= product of the two cores); cores, bonds and
 *                   l_pos are left untouched; metrics2 = (accuracy, MAE) or NULL.
 *   tnml_l2_term    Network.compute_L2_reg (:966-1179): loss = wd <B, Ln.B.Rn>, grad = 2 wd Ln.B.Rn.
 *   tnml_svd_split  Network.tensor_svd (:839-962): U sqrt(S) [rows][m] and sqrt(S) Vh [m][cols] of a
 *                   rows x cols matrix (both multiples of D, min <= 128), sigma[min(rows, cols)] or NULL. */
int tnml_update_B(tnml_ctx *ctx, const float *B_canon, int left_dir, float lr, float weight_dec, int l2_flag,
                  int act_fn, int loss_fn, float T, double *Bnew_canon, size_t capacity, float *metrics2);
int tnml_l2_term(tnml_ctx *ctx, const float *B_canon, int left_dir, float weight_dec, double *loss,
                 double *grad_canon, size_t capacity);
int tnml_svd_split(tnml_ctx *ctx, const float *mat, int rows, int cols, int m, float *US, float *SVh,
                   double *sigma);

/* Network.forward's return value for a batch that is NOT made resident (the validation loop of
 * Network.train, Network_class.py:339-346): X [b][N][D] -> f_out [L][b].  One chain towards the label
 * site, no environment is stored; the training batch, its environments and f stay as they are.  Same
 * l_pos restriction as tnml_forward. */
int tnml_predict(tnml_ctx *ctx, const float *X, int b, float *f_out);

/* Accuracy / speed of the in-kernel Jacobi SVD (no reference analogue: the reference calls LAPACK,
 * Network_class.py:887).  The iteration ends after a sweep in which every rotation had
 * g^2 <= stop2 * scale^2; the off-diagonals left behind are of relative size ~stop2.  Default 1e-6
 * (truncated product within ~6e-6 max|B| of LAPACK's); 1e-4 saves about one sweep in five and leaves ~2e-4;
 * 1e-8 costs one more and reaches ~1e-6.  Allowed range [1e-12, 1e-2]. */
int tnml_set_svd_stop(tnml_ctx *ctx, double stop2);

/* A sweep step is ONE launch by default: workgroup 0 updates and splits the merged tensor of step k while the other
 * workgroups of the same launch form f of step k and the batch-summed pre-gradient of step k+1.
 * on = 0 restores the classic sequence (batch kernel -> [reduction] -> update/SVD kernel), which is also what steps
 * whose merged tensor does not fit one workgroup's LDS take.  Results agree to float32 rounding.
 * on >= 2: number of 32-sample tiles a batch-side workgroup accumulates before it writes its partial pre-gradient on steps
 * whose SVD is long enough to hide that (short side >= 32); on = 1 means the default, 2. */
int tnml_set_step_pipeline(tnml_ctx *ctx, int on);

/* A FULL sweep (n_steps = N-1 right after tnml_forward) on a single GPU under the fixed or the reference truncation is ONE launch by
 * default: a persistent kernel whose update workgroup, helper workgroup and batch-side workgroups each loop over the N-1 steps and
 * hand their results to each other through flags in memory.  Sweeps it does not cover (partial sweeps, a
 * communicator, adaptive truncation, per-step capture, merged tensors beyond one workgroup's LDS) take one launch per step as
 * before.
 *   on = 1 (default)  the three roles in ONE launch
 *   on = 2            the three roles as three launches on three streams of the context, resident together (each role with its own
 *                     register allocation; measured equal to on = 1 within 2 %); a tool that SERIALISES launches (rocprofv3 --pmc)
 *                     keeps them from meeting: the bounded waits then time out and tnml_sweep fails with TNML_ERR_STATE
 *   on = 0            one launch per step everywhere */
int tnml_set_persistent(tnml_ctx *ctx, int on);

/* tnml_sweep enqueues every launch of its n_steps steps without waiting (2 - 14 launches per step).  A profiler that
 * intercepts dispatches (rocprofv3 --pmc serialises them and keeps per-dispatch state) can be overrun by tens of
 * thousands of queued launches; n_steps > 0 drains the stream every n_steps steps, 0 never.  Default 256: at most ~3600
 * dispatches outstanding on the large-tensor path, one host round trip per 14 ms of a C3 sweep (the persistent sweep is one
 * launch and never drains). */
int tnml_set_sync_interval(tnml_ctx *ctx, int n_steps);

/* The batch-independent part of a step (update_B's tail, compute_L2_reg, tensor_svd) runs in one
 * workgroup's LDS when the merged tensor fits (min(rows, cols) <= 64 and <= 160 KB of LDS: bond <= 32 at
 * two labels) and through HBM-resident kernels otherwise (min(rows, cols) <= 128: bond 50 with ten labels).
 * LIMIT: the Jacobi kernels take a short side of at most 128, i.e. bond dimension M <= 64 at D = 2; tnml_sweep
 * returns TNML_ERR_ARG at the first step beyond it (the reference itself has no such limit; its largest published
 * bond is 50).
 * force_large = 1 sends every step down the second path (tests, diagnostics); 0 restores the automatic
 * choice. */
int tnml_set_narrow_path(tnml_ctx *ctx, int force_large);

/* The hand-offs between the context's two streams (pipelined large-tensor step; update side / batch side + all-reduce of the communicator
 * path) are sequence numbers in memory by default: the consuming kernel, or a one-wave gate kernel in front of it, polls the word a
 * one-thread signal kernel (or the producing workgroup itself) stores -- an event dependency costs the stream that records or waits
 * 6-7 us even when it is satisfied.  Every waiting kernel is enqueued after the kernels it waits for and its wait is bounded
 * (TNML_ERR_STATE on a time-out), but a tool that lets only ONE kernel run at a time (rocprofv3 --pmc serialises dispatches) keeps the
 * producer from ever starting: on = 0 (or the environment variable TNML_EVENT_HANDOFFS=1 at tnml_create) uses events everywhere. */
int tnml_set_flag_handoffs(tnml_ctx *ctx, int on);

/* The forward environment chain (Network.forward, Network_class.py:227-255) runs on the matrix cores for bond dimensions
 * <= 32 (one wave per 16 samples) and as plain FMAs otherwise and in the renormalising calibration pass.  force_plain = 1
 * sends every chain down the plain-FMA kernel (tests, diagnostics); 0 restores the automatic choice. */
int tnml_set_chain_path(tnml_ctx *ctx, int force_plain);

/* TNML_TRUNC_ADAPTIVE (not reference behaviour): tensor_svd computes the cumulative share of the singular
 * values and the first index where it exceeds `threshold` (Network_class.py:889-891, default argument
 * 0.999) but never uses it.  Under this policy the kept rank is min(M, index + 1), decided on the device
 * from the full spectrum; tnml_sweep then synchronises once per step to learn the new bond dimension. */
int tnml_set_trunc_threshold(tnml_ctx *ctx, double threshold);

/* Network.apply_act_func / compute_loss_derivate on the device-resident f (:767-835);
 * act_out, lossder_out [L][b], either may be NULL.  input_is_activated != 0: f already went
 * through the activation (what compute_loss_derivate receives, :800), only the derivative runs. */
int tnml_activation(tnml_ctx *ctx, int act_fn, int loss_fn, float T, int input_is_activated,
                    float *act_out, float *lossder_out);

/* ---- inspection (API parity: Network.r_cum_contraction / l_cum_contraction) -------------- */
int tnml_get_env(tnml_ctx *ctx, int side, int site, float *out, size_t capacity, int *m);
/* bit 0: capture B, dB, B_new, sigma of every step into a debug block (tests only; off by default);
 * bit 1: in-kernel cycle stamps only; bit 2: read the launch status back after every kernel launch of a step
 * (a failed launch then names its kernel; launch geometry is validated before every launch regardless) */
int tnml_debug_enable(tnml_ctx *ctx, int on);
int tnml_get_step_debug(tnml_ctx *ctx, int what, double *out, size_t capacity, size_t *n);
int tnml_l_pos(tnml_ctx *ctx);
int tnml_batch(tnml_ctx *ctx);

/* ---- measurement ------------------------------------------------------------------------ */
/* HIP-event timing on the context's own stream (torch.cuda.Event would not see it) */
int tnml_timer_start(tnml_ctx *ctx);
/* phase boundary visible to a profiler: an empty kernel `tnml_phase_marker_kernel` of `id` workgroups of 64 threads (1 <= id <=
 * 1024) on the context's stream; bench.py brackets its warm-up / timed / resident / cold passes with it and
 * tools/rocprof_summary.py cuts kernel traces and counter passes to the window between two markers */
int tnml_marker(tnml_ctx *ctx, int id);
int tnml_timer_stop(tnml_ctx *ctx, double *elapsed_ms);
/* tnml_profile_enable(ctx, 1): accumulated per-kernel device time (ms) and launch counts since the last reset, measured
 * with HIP events around each launch (synchronises after every launch: slows the sweep; break-downs only)
 *   which: 0 forward chain, 1 batch-side kernel (classic wide kernel / prologue of the pipelined step), 2 reduce kernel,
 *          3 update+SVD kernel (classic narrow kernel / the single launch of a pipelined step)
 * tnml_profile_enable(ctx, 2): one HIP event pair per tnml_sweep call, nothing waits inside the timed region
 *   which: 4 -> ms between first and last launch of all sweeps since the reset, launches = kernel launches they made
 *          5 -> the same ms, launches = number of pipelined steps among them (single-launch steps and large-tensor steps that
 *               took their gradient from the pre-gradient the previous step's side stream left) */
int tnml_profile_enable(tnml_ctx *ctx, int on);
int tnml_profile_get(tnml_ctx *ctx, int which, double *ms, long long *launches);
int tnml_profile_reset(tnml_ctx *ctx);
/* Work done since the last tnml_profile_reset, computed from the dimensions of every step that ran:
 *   out8 = {sweep steps, algorithmic bytes of those steps (4 b (2h + g + 3D + 2L + 1) each: environments, features, f, labels),
 *           algorithmic flops (4 b D^2 h g L + 2 b D h^2 each), forward calls, algorithmic bytes of those forwards,
 *           kernel launches of the sweeps, pipelined steps among them (single-launch steps + large-tensor steps fed by Z),
 *           device ms inside the sweeps (0 unless tnml_profile_enable(ctx, 2) was on; read it through tnml_profile_get(4) first)} */
int tnml_get_counters(tnml_ctx *ctx, double *out8);
/* always-on counters of the SVD since the last reset: out3 = {Jacobi sweeps, number of SVDs, Jacobi rounds (one barrier each)} */
int tnml_svd_stats(tnml_ctx *ctx, int reset, double *out3);
/* the same with a capacity: out[0..capacity) of {sweeps, SVDs, rounds, SVDs that took the pivoted-Cholesky step}; entries beyond
 * the fourth are left untouched */
int tnml_svd_stats_ex(tnml_ctx *ctx, int reset, double *out, int capacity);

/* Host-side planning helper, exported so that CPU tests can check the bond bookkeeping without a
 * GPU: truncation rank kept by tensor_svd (Network_class.py:894-945) for a step on sites
 * (p, p+1).  Returns m >= 1, or TNML_ERR_SHAPE where the reference itself raises. */
int tnml_trunc_rank(int policy, int left_dir, int p, int N, int ml, int D, int mr, int L, int M);

#ifdef __cplusplus
}
#endif
#endif /* TNML_H */
