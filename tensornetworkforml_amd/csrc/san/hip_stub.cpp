// CPU stand-in for the HIP runtime and RCCL, for ONE purpose: running the HOST side of this library (tnml_api.hip and the launch
// wrappers of kernels_*.hip, compiled `--cuda-host-only -fsanitize=address,undefined`) on a machine without a GPU, so that the
// planning of whole sweeps -- strides, slot offsets, buffer sizing, the pipelined / classic / persistent state machine -- runs under
// AddressSanitizer and UBSan (GPU sanitizers are not available on this pool).  Nothing here computes: kernels are never executed.
// What it does instead:
//   * "device memory" comes from one reserved address range with unmapped gaps between allocations; every copy / memset the host
//     code issues is checked against the allocation registry;
//   * every launch is checked: grid, block and dynamic LDS against the gfx950 limits, and -- for the kernels that carry the sweep
//     (typed decoders below) -- every pointer of their argument blocks together with the extent the kernel will touch.
// A violation prints what and where and aborts.  Test infrastructure only; never linked into libtnml_hip.so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <cxxabi.h>
#include "../tnml_internal.h"
#include "../wide_pipe_device.h"

using namespace tnml;

namespace san {
static char *g_base = nullptr;
static size_t g_used = 0;
static constexpr size_t kArena = (size_t)1 << 38, kGap = (size_t)1 << 20;
// (function-local: kernels register themselves from static constructors that may run before this file's)
static std::map<uintptr_t, size_t> &alloc_map() { static auto *m = new std::map<uintptr_t, size_t>; return *m; }          // base -> bytes (live)
static std::map<const void *, std::string> &kernel_map() { static auto *m = new std::map<const void *, std::string>; return *m; }
static std::map<std::string, long> &launch_map() { static auto *m = new std::map<std::string, long>; return *m; }
#define g_alloc alloc_map()
#define g_kernels kernel_map()
#define g_launches launch_map()
static long g_checked_ptrs = 0, g_allreduces = 0;
static const char *g_ctx = "";

[[noreturn]] static void die(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "san-stub VIOLATION [%s]: ", g_ctx);
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  abort();
}
static bool in_arena(const void *p) { return g_base && (const char *)p >= g_base && (const char *)p < g_base + kArena; }
// [p, p + bytes) inside ONE live allocation
static void need(const void *p, size_t bytes, const char *what) {
  ++g_checked_ptrs;
  if (!p) die("%s: null pointer (%zu bytes wanted)", what, bytes);
  if (!in_arena(p)) die("%s: %p is not device memory", what, p);
  auto it = g_alloc.upper_bound((uintptr_t)p);
  if (it == g_alloc.begin()) die("%s: %p below every allocation", what, p);
  --it;
  const uintptr_t lo = it->first, hi = lo + it->second;
  if ((uintptr_t)p + bytes > hi) die("%s: [%p, +%zu) leaves its allocation [%p, +%zu) by %zu bytes", what, p, bytes, (void *)lo, it->second, (uintptr_t)p + bytes - hi);
}
static void opt(const void *p, size_t bytes, const char *what) { if (p) need(p, bytes, what); }
// every 8-byte word of an argument block that points into the device range must point into a live allocation
static void scan(const void *blk, size_t bytes, const char *what) {
  for (size_t o = 0; o + 8 <= bytes; o += 8) {
    void *v;
    memcpy(&v, (const char *)blk + o, 8);
    if (in_arena(v)) need(v, 1, what);
  }
}
static size_t view_extent(const CoreView &v, int tail) {     // elements reached through the strides (+ tail for the label index)
  return (size_t)(v.n_in - 1) * v.s_in + (size_t)(kD - 1) * v.s_d + (size_t)(v.n_out - 1) * v.s_out + tail;
}

static void check_wide(const WideParams &p, int nblk, bool f_only = false) {
  const size_t bp = p.b_pad;
  if (p.b < 1 || p.b > p.b_pad || p.b_pad % 64) die("WideParams: b %d b_pad %d", p.b, p.b_pad);
  scan(&p, sizeof p, "WideParams");
  if (f_only) {                            // f from the previous step's updated tensor: the "previous" operands only
    opt(p.Hprev, (size_t)p.hp * bp * 4, "WideParams.Hprev"); opt(p.Gprev, (size_t)p.gp * bp * 4, "WideParams.Gprev");
    need(p.x_km1, bp * kD * 4, "WideParams.x_km1"); need(p.x_k, bp * kD * 4, "WideParams.x_k");
    need(p.Bprev, (size_t)p.hp * kD * kD * p.gp * p.L * 4, "WideParams.Bprev"); need(p.f, (size_t)p.L * bp * 4, "WideParams.f");
    if ((size_t)nblk * kTS != bp) die("f_only grid %d x %d samples != b_pad %zu", nblk, kTS, bp);
    return;
  }
  opt(p.x_km1, bp * kD * 4, "WideParams.x_km1"); need(p.x_k, bp * kD * 4, "WideParams.x_k"); need(p.x_kp1, bp * kD * 4, "WideParams.x_kp1");
  if (p.do_ext && !p.first_ext) need(p.Hprev, (size_t)p.hp * bp * 4, "WideParams.Hprev");
  opt(p.Hcur, (size_t)p.h * bp * 4, "WideParams.Hcur");
  if (p.do_f) { opt(p.Gprev, (size_t)p.gp * bp * 4, "WideParams.Gprev"); need(p.Bprev, (size_t)p.hp * kD * kD * p.gp * p.L * 4, "WideParams.Bprev"); }
  opt(p.Gcur, (size_t)p.g * bp * 4, "WideParams.Gcur");
  if (p.do_ext) need(p.ext_core.base, view_extent(p.ext_core, 1) * 4, "WideParams.ext_core");
  need(p.y, bp * 4, "WideParams.y"); need(p.f, (size_t)p.L * bp * 4, "WideParams.f");
  if (p.slabs) {
    if (p.bsize != p.h * kD * kD * p.g * p.L) die("WideParams.bsize %d != h D D g L", p.bsize);
    if (p.slab_stride < p.bsize + kMetricSlots) die("WideParams.slab_stride %d < bsize + tail %d", p.slab_stride, p.bsize + kMetricSlots);
    need(p.slabs, (size_t)nblk * p.slab_stride * 4, "WideParams.slabs");
  }
}
struct BigExtArgs { const float *Eprev, *x_km1, *x_k; CoreView A; int b_pad; float *Ecur, *Pk; };     // kernels_big.hip
static void check_big_ext(const BigExtArgs &a, size_t shm) {
  const CoreView &A = a.A;
  const size_t bp = (size_t)a.b_pad;
  need(a.Eprev, (size_t)A.n_in * bp * 4, "big_ext: E_{k-1}");
  need(a.x_km1, bp * kD * 4, "big_ext: x_{k-1}"); need(a.x_k, bp * kD * 4, "big_ext: x_k");
  need(A.base, view_extent(A, 1) * 4, "big_ext: core");
  need(a.Ecur, (size_t)A.n_out * bp * 4, "big_ext: E_k"); need(a.Pk, (size_t)kD * A.n_out * bp * 4, "big_ext: P'_k");
  if (shm < (size_t)A.n_in * kD * ((A.n_out + 3) & ~3) * 4) die("big_ext: %zu bytes of LDS for a %d x %d x %d core", shm, A.n_in, kD, A.n_out);
}
static void check_narrow(const NarrowParams &p) {
  scan(&p, sizeof p, "NarrowParams");
  if (p.bsize != p.h * kD * kD * p.g * p.L) die("NarrowParams.bsize %d != h D D g L (%d %d %d)", p.bsize, p.h, p.g, p.L);
  const int rows = p.h * kD, cols = kD * p.g * p.L;     // (right sweep orientation; the kept rank bound is symmetric)
  if (p.m < 1 || p.m > (rows < cols ? rows : cols) * p.L) die("NarrowParams.m %d for a %d x %d tensor", p.m, rows, cols);
  if (!p.pipe) opt(p.red, (size_t)(p.bsize + kMetricSlots) * 4, "NarrowParams.red");
  if (!p.Bdirect) { need(p.lab.base, view_extent(p.lab, p.L) * 4, "NarrowParams.lab"); need(p.pl.base, view_extent(p.pl, 1) * 4, "NarrowParams.pl"); }
  else need(p.Bdirect, (size_t)p.bsize * 4, "NarrowParams.Bdirect");
  opt(p.Nh, (size_t)p.h * p.h * 8, "NarrowParams.Nh"); opt(p.Ng, (size_t)p.g * p.g * 8, "NarrowParams.Ng");
  need(p.Bnew, (size_t)p.bsize * 4, "NarrowParams.Bnew");
  if (!p.stop_after_update) {
    need(p.out_behind, ((size_t)(p.h - 1) * p.ob_s_h + (size_t)(kD - 1) * p.ob_s_d + (size_t)(p.m - 1) * p.ob_s_m + 1) * 4, "NarrowParams.out_behind");
    need(p.out_ahead, ((size_t)(p.m - 1) * p.oa_s_m + (size_t)(kD - 1) * p.oa_s_d + (size_t)(p.g - 1) * p.oa_s_g + p.L) * 4, "NarrowParams.out_ahead");
    opt(p.Nh_new, (size_t)p.m * p.m * 8, "NarrowParams.Nh_new");
  }
  if (p.zpoll_flag) need(p.zpoll_flag, 4, "NarrowParams.zpoll_flag");
  if (p.done_flag) need(p.done_flag, 4, "NarrowParams.done_flag");
  opt(p.metrics, 2 * 4, "NarrowParams.metrics"); opt(p.counters, 4 * 8, "NarrowParams.counters") /* null in compute_L2_reg alone: the kernel tests it */; need(p.status, 4, "NarrowParams.status");
  if (p.pipe) {
    need(p.zred, (size_t)(p.zsize + kMetricSlots) * 4, "NarrowParams.zred");
    // (persistent sweep: the helper workgroups contract Z with the published behind core; the update workgroup reads the metric tail)
    if (!p.z_first && !p.persist) need(p.zcore.base, view_extent(p.zcore, 1) * 4, "NarrowParams.zcore");
    opt(p.flag, 4, "NarrowParams.flag");
  }
  if (p.fused) {
    opt(p.slabs, (size_t)p.nslabs * p.slab_stride * 4, "NarrowParams.slabs");
    opt(p.prepB, (size_t)p.bsize * 4, "NarrowParams.prepB"); opt(p.prepG, (size_t)p.bsize * 8, "NarrowParams.prepG");
    need(p.sync, 4, "NarrowParams.sync");
  }
  if (p.persist) {
    need(p.prepB, (size_t)p.bsize * 4, "NarrowParams.prepB (persistent)"); need(p.prepG, (size_t)p.bsize * 8, "NarrowParams.prepG (persistent)");
    need(p.pready, 4, "NarrowParams.pready"); need(p.coreflag, 4, "NarrowParams.coreflag"); need(p.abort_flag, 4, "NarrowParams.abort_flag");
    need(p.Apub, ((size_t)kD * p.h * p.m + p.m + (size_t)p.m * p.m) * 8, "NarrowParams.Apub");
  }
}
static void check_pipe(const WidePipeParams &w) {
  scan(&w, sizeof w, "WidePipeParams");
  const size_t bp = w.b_pad;
  if (w.b < 1 || w.b > w.b_pad || w.b_pad % 64 || w.ntiles != w.b_pad / kTS) die("WidePipeParams: b %d b_pad %d ntiles %d", w.b, w.b_pad, w.ntiles);
  if (!w.first) {
    need(w.x_j, bp * kD * 4, "WidePipeParams.x_j");
    if (w.do_ext) {
      need(w.Ecur, (size_t)w.hj * bp * 4, "WidePipeParams.Ecur (written)");
      need(w.x_jm1, bp * kD * 4, "WidePipeParams.x_jm1");
      if (!w.first_ext) need(w.Eprev, (size_t)w.hprev * bp * 4, "WidePipeParams.Eprev");
      need(w.ext_core.base, view_extent(w.ext_core, 1) * 4, "WidePipeParams.ext_core");
    } else opt(w.Ecur, (size_t)w.hj * bp * 4, "WidePipeParams.Ecur (read)");
    if (w.do_f) { opt(w.Gj, (size_t)w.gj * bp * 4, "WidePipeParams.Gj"); need(w.Bnew, (size_t)w.hj * kD * kD * w.gj * w.L * 4, "WidePipeParams.Bnew"); }
  }
  need(w.x_jp1, bp * kD * 4, "WidePipeParams.x_jp1");
  need(w.y, bp * 4, "WidePipeParams.y"); need(w.f, (size_t)w.L * bp * 4, "WidePipeParams.f"); need(w.status, 4, "WidePipeParams.status");
  if (w.do_z) {
    need(w.x_jp2, bp * kD * 4, "WidePipeParams.x_jp2");
    opt(w.Gn, (size_t)w.gn * bp * 4, "WidePipeParams.Gn");
    const int nI = w.first ? 1 : w.hj * kD;
    if (w.zsize != nI * kD * kD * w.gn * w.L) die("WidePipeParams.zsize %d != nI D D gn L (%d %d %d)", w.zsize, nI, w.gn, w.L);
    if (w.slab_stride < w.zsize + kMetricSlots) die("WidePipeParams.slab_stride %d < zsize + tail", w.slab_stride);
    need(w.slabs, (size_t)w.nwide * w.slab_stride * 4, "WidePipeParams.slabs");
    if (!w.one_level) need(w.gslabs, (size_t)w.ngroups * w.slab_stride * 4, "WidePipeParams.gslabs");
    need(w.zred, (size_t)(w.zsize + kMetricSlots) * 4, "WidePipeParams.zred");
    need(w.gcnt, (size_t)(w.ngroups > 0 ? w.ngroups : 1) * 4, "WidePipeParams.gcnt"); need(w.tcnt, 4, "WidePipeParams.tcnt");
    if (w.nwide * w.tiles_per_wg < w.ntiles) die("WidePipeParams: %d workgroups x %d tiles < %d tiles", w.nwide, w.tiles_per_wg, w.ntiles);
  }
  if (w.wait_flag) need(w.flag, 4, "WidePipeParams.flag");
}
static void check_chain(const ChainSite *sites, int n, const float *cores, const float *lab, const float *X, float *env, float *f, int b_pad, int L) {
  need(sites, (size_t)n * sizeof(ChainSite), "chain table");
  for (int i = 0; i < n; ++i) {
    const ChainSite &c = sites[i];
    const size_t ext = (size_t)(c.n_in - 1) * c.s_in + (size_t)(kD - 1) * c.s_d + (size_t)(c.n_out - 1) * c.s_out + 1;
    need((c.is_label ? lab : cores) + c.core_off, ext * 4, "chain core");
    need(X + (size_t)c.x_site * b_pad * kD, (size_t)b_pad * kD * 4, "chain features");
    if (c.env_out_off >= 0) { if (env) need(env + c.env_out_off, (size_t)c.n_out * b_pad * 4, "chain environment slot"); }
    else need(f, (size_t)L * b_pad * 4, "chain f");
    if (i + 1 < n && sites[i + 1].n_in != c.n_out) die("chain: site %d produces %d, site %d takes %d", i, c.n_out, i + 1, sites[i + 1].n_in);
  }
}
}  // namespace san
using namespace san;

extern "C" {
// ---- registration and launch ----------------------------------------------------------------------------------------------------
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *dev_name, unsigned, void *, void *, void *, void *, int *) {
  int st = 0;
  char *dm = abi::__cxa_demangle(dev_name, nullptr, nullptr, &st);
  std::string nm = dm ? dm : dev_name;
  free(dm);
  for (size_t at; (at = nm.find("(anonymous namespace)::")) != std::string::npos;) nm.erase(at, 23);
  g_kernels[host_fn] = nm;
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
static thread_local struct { dim3 g, b; size_t shm; hipStream_t st; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t st) { t_cfg = {g, b, shm, st}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *shm, hipStream_t *st) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *st = t_cfg.st; return hipSuccess; }

hipError_t hipLaunchKernel(const void *fn, dim3 g, dim3 b, void **args, size_t shm, hipStream_t) {
  auto it = g_kernels.find(fn);
  const std::string name = it == g_kernels.end() ? "?" : it->second;
  g_ctx = name.c_str();
  const unsigned long long thr = (unsigned long long)b.x * b.y * b.z;
  if (g.x < 1 || g.y < 1 || g.z < 1 || g.y > 65535 || g.z > 65535 || thr < 1 || thr > 1024 || shm > 160 * 1024)
    die("illegal launch: grid (%u,%u,%u) block (%u,%u,%u) dynamic LDS %zu", g.x, g.y, g.z, b.x, b.y, b.z, shm);
  const std::string key = name.substr(0, name.find('('));
  g_launches[key]++;
  static const bool trace = getenv("SAN_TRACE") != nullptr;
  if (trace) fprintf(stderr, "launch %s grid %u block %u lds %zu\n", key.c_str(), g.x, b.x, shm);
  auto has = [&](const char *s) { return name.find(s) != std::string::npos; };
  if (has("sweep_persist") || has("persist_update_kernel") || has("persist_helper_kernel") || has("persist_batch_kernel")) {
    const PersistStep *st = *(const PersistStep **)args[0];
    const int n = *(int *)args[1];
    need(st, (size_t)(n + 1) * sizeof(PersistStep), "persistent step records");
    for (int k = 0; k < n; ++k) { check_narrow(st[k].n); check_pipe(st[k].w); scan(&st[k].t, sizeof st[k].t, "PersistHelperParams"); }
    check_pipe(st[n].w);                 // the prologue of the batch side
  } else if (has("step_pipe_kernel")) {
    const NarrowParams &n = *(const NarrowParams *)args[0];
    const WidePipeParams &w = *(const WidePipeParams *)args[1];
    if (n.bsize > 0) check_narrow(n);    // (a prologue launch carries no update)
    check_pipe(w);
    if (g.x > 256) die("step_pipe_kernel: %u workgroups cannot be co-resident on 256 CUs", g.x);
  } else if (has("narrow_step_kernel")) {
    check_narrow(*(const NarrowParams *)args[0]);
  } else if (has("wide_step") || has("f_only_kernel")) {
    // (the MFMA kernel's grid carries D*D slice workgroups behind the nblk sample workgroups)
    check_wide(*(const WideParams *)args[0], has("wide_step_mfma_kernel") ? *(int *)args[2] : (int)g.x, has("f_only_kernel"));
  } else if (has("env_chain_roles_kernel")) {
    check_chain(*(const ChainSite **)args[0], *(int *)args[1], *(const float **)args[2], *(const float **)args[3], *(const float **)args[4],
                *(float **)args[5], *(float **)args[6], *(int *)args[7], 1);
    if ((int)g.x * 16 != *(int *)args[7]) die("chain grid %u x 16 samples != b_pad %d", g.x, *(int *)args[7]);
  } else if (has("env_chain_kernel")) {
    check_chain(*(const ChainSite **)args[0], *(int *)args[1], *(const float **)args[2], *(const float **)args[3], *(const float **)args[4],
                has("<true>") ? nullptr : *(float **)args[5], *(float **)args[6], *(int *)args[8], *(int *)args[9]);
  } else if (has("big_merge_wd_kernel") || has("big_merge_wd_mfma_kernel")) { // (NarrowParams, merged tensor out, NL, PR, workspace, block partials): the factored chain's entry
    const NarrowParams &n = *(const NarrowParams *)args[0];
    check_narrow(n);
    need(*(float **)args[1], (size_t)n.bsize * 4, "large-tensor path: merged tensor");
    if (n.l2_flag) {
      need(*(const double **)args[2], (size_t)n.h * kD * n.s * n.L * 8, "large-tensor path: NL");
      need(*(const double **)args[3], (size_t)n.s * kD * n.g * 8, "large-tensor path: PR");
    }
    if (g.x > (unsigned)kBigParts) die("big_merge_wd_kernel: %u blocks leave partials, room for %d", g.x, kBigParts);
    need(*(double **)args[5], (size_t)g.x * 3 * 8, "large-tensor path: block partials");
  } else if (has("big_wd_kernel")) {       // (NarrowParams, merged tensor, Nh^T.B, workspace, block partials): the classic chain's entry
    const NarrowParams &n = *(const NarrowParams *)args[0];
    check_narrow(n);
    need(*(const float **)args[1], (size_t)n.bsize * 4, "large-tensor path: merged tensor");
    if (n.l2_flag) need(*(const double **)args[2], (size_t)n.bsize * 8, "large-tensor path: Nh^T.B");
    if (g.x > (unsigned)kBigParts) die("big_wd_kernel: %u blocks leave partials, room for %d", g.x, kBigParts);
    need(*(double **)args[4], (size_t)g.x * 3 * 8, "large-tensor path: block partials");
  } else if (has("big_nlpr_kernel")) {     // (NarrowParams, NL, PR)
    const NarrowParams &n = *(const NarrowParams *)args[0];
    check_narrow(n);
    need(*(double **)args[1], (size_t)n.h * kD * n.s * n.L * 8, "big_nlpr: NL");
    need(*(double **)args[2], (size_t)n.s * kD * n.g * 8, "big_nlpr: PR");
    if (*(double **)args[2] < *(double **)args[1] + (size_t)n.h * kD * n.s * n.L) die("big_nlpr: PR overlaps NL");
  } else if (has("big_gram_kernel")) {     // (B_new, n, len, si, sx, gram)
    const int n = *(int *)args[1], len = *(int *)args[2];
    need(*(const float **)args[0], (size_t)n * len * 4, "large-tensor path: B_new");
    need(*(double **)args[5], (size_t)8 * kBigMaxN * kBigMaxN * 8, "large-tensor path: partial Gram matrices");
    if (n > kBigMaxN || (n & 1)) die("big_gram_kernel: short side %d", n);
    if ((int)g.x * 16 < n || (int)g.y * 16 < n) die("big_gram_kernel: grid (%u, %u) of 16 x 16 tiles for n = %d", g.x, g.y, n);
    if (*(unsigned **)args[6]) need(*(unsigned **)args[6], 4, "big_gram: flag word of the side stream");
  } else if (has("big_ext_kernel")) {      // (BigExtArgs: E_{k-1}, x_{k-1}, x_k, core view, b_pad, E_k, P'_k)
    check_big_ext(*(const BigExtArgs *)args[0], shm);
    const BigExtArgs &a = *(const BigExtArgs *)args[0];
    if ((int)g.x * 64 != a.b_pad) die("big_ext_kernel: grid %u x 64 samples != b_pad %d", g.x, a.b_pad);
    if ((int)g.y * 16 < a.A.n_out) die("big_ext_kernel: %u workgroup rows of 16 for %d bond indices", g.y, a.A.n_out);
  } else if (has("big_contract_kernel")) { // (Z, core view, ncols, red)
    const CoreView &A = *(const CoreView *)args[1];
    const size_t nc = *(int *)args[2];
    need(*(const float **)args[0], ((size_t)A.n_in * kD * nc + kMetricSlots) * 4, "big_contract: Z");
    need(A.base, view_extent(A, 1) * 4, "big_contract: core");
    need(*(float **)args[3], ((size_t)A.n_out * nc + kMetricSlots) * 4, "big_contract: raw gradient");
    if (shm < (size_t)A.n_in * kD * 8 * 4) die("big_contract_kernel: %zu bytes of LDS", shm);
    if ((size_t)g.x * 64 < nc || (int)g.y * 8 < A.n_out) die("big_contract_kernel: grid (%u, %u) for %zu columns, %d rows", g.x, g.y, nc, A.n_out);
  } else if (has("big_front_kernel")) {    // (NarrowParams, NL, PR, Z, core view, ncols, red, tile split, BigExtArgs, poll flag, poll value, ext acquire)
    const NarrowParams &n = *(const NarrowParams *)args[0];
    check_narrow(n);
    const CoreView &A = *(const CoreView *)args[4];
    const size_t nc = *(int *)args[5];
    struct Tiles { int contract_blocks, ext_blocks; };
    const Tiles &ft = *(const Tiles *)args[7];
    if (n.l2_flag) {
      need(*(double **)args[1], (size_t)n.h * kD * n.s * n.L * 8, "big_front: NL");
      need(*(double **)args[2], (size_t)n.s * kD * n.g * 8, "big_front: PR");
      if (*(double **)args[2] < *(double **)args[1] + (size_t)n.h * kD * n.s * n.L) die("big_front: PR overlaps NL");
      if ((int)g.x <= ft.contract_blocks + ft.ext_blocks) die("big_front_kernel: grid %u leaves no workgroup for the L2 products", g.x);
    }
    need(*(const float **)args[3], ((size_t)A.n_in * kD * nc + kMetricSlots) * 4, "big_front: Z");
    need(A.base, view_extent(A, 1) * 4, "big_front: core");
    need(*(float **)args[6], ((size_t)A.n_out * nc + kMetricSlots) * 4, "big_front: raw gradient");
    if ((size_t)A.n_out * nc != (size_t)n.bsize) die("big_front_kernel: gradient %d x %zu, merged tensor %d", A.n_out, nc, n.bsize);
    if ((size_t)ft.contract_blocks * 4 < (size_t)((A.n_out + 15) / 16) * ((nc + 15) / 16)) die("big_front_kernel: %d workgroups for the contraction's tiles", ft.contract_blocks);
    if (*(const unsigned **)args[9]) need(*(const unsigned **)args[9], 4, "big_front: sequence number of the side stream");
    const BigExtArgs &ea = *(const BigExtArgs *)args[8];
    if (ft.ext_blocks) {
      check_big_ext(ea, (size_t)1 << 20);                      // (no LDS in this form)
      if (ea.b_pad % 64 || (size_t)ft.ext_blocks * 4 < (size_t)((ea.A.n_out + 15) / 16) * (ea.b_pad / 16)) die("big_front_kernel: %d workgroups for the extension's tiles, b_pad %d", ft.ext_blocks, ea.b_pad);
    }
  } else if (has("big_signal_kernel")) {   // (flag, value)
    need(*(unsigned **)args[0], 4, "big_signal: flag word");
  } else if (has("big_gate_kernel")) {     // (flag, want, status)
    need(*(const unsigned **)args[0], 4, "big_gate: flag word"); need(*(int **)args[2], 4, "big_gate: status");
  } else if (has("reduce_slabs")) {        // (slabs, nblk, slab_stride, n, red)
    const int nblk = *(int *)args[1], stride = *(int *)args[2], n = *(int *)args[3];
    if (n > stride) die("reduce_slabs: %d elements of a slab of %d", n, stride);
    need(*(const float **)args[0], (size_t)nblk * stride * 4, "slabs"); need(*(float **)args[4], (size_t)n * 4, "reduced slab");
  }
  g_ctx = "";
  return hipSuccess;
}

// ---- memory ----------------------------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t n) {
  if (!g_base) {
    g_base = (char *)mmap(nullptr, kArena, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_base == MAP_FAILED) { perror("mmap"); abort(); }
  }
  const size_t sz = (n + 4095) & ~(size_t)4095;
  if (g_used + sz + kGap > kArena) return hipErrorOutOfMemory;
  char *q = g_base + g_used + kGap;
  if (sz && mprotect(q, sz, PROT_READ | PROT_WRITE)) { perror("mprotect"); abort(); }
  g_used += sz + kGap;
  g_alloc[(uintptr_t)q] = n;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  auto it = g_alloc.find((uintptr_t)p);
  if (it == g_alloc.end()) die("hipFree(%p): not the base of a live allocation", p);
  const size_t sz = (it->second + 4095) & ~(size_t)4095;
  if (sz) { madvise(p, sz, MADV_DONTNEED); mprotect(p, sz, PROT_NONE); }     // a later use faults
  g_alloc.erase(it);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static void copy_checked(void *d, const void *s, size_t n, hipMemcpyKind k, const char *what) {
  g_ctx = what;
  if (n == 0) return;
  const bool dd = k == hipMemcpyHostToDevice || k == hipMemcpyDeviceToDevice || (k == hipMemcpyDefault && in_arena(d));
  const bool sd = k == hipMemcpyDeviceToHost || k == hipMemcpyDeviceToDevice || (k == hipMemcpyDefault && in_arena(s));
  if (dd) need(d, n, "copy destination"); else if (in_arena(d)) die("host destination %p is device memory", d);
  if (sd) need(s, n, "copy source"); else if (in_arena(s)) die("host source %p is device memory", s);
  memmove(d, s, n);                        // host sides are checked by AddressSanitizer
  g_ctx = "";
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { copy_checked(d, s, n, k, "hipMemcpy"); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { copy_checked(d, s, n, k, "hipMemcpyAsync"); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) {
  if (w > dp || w > sp) die("hipMemcpy2DAsync: width %zu exceeds a pitch (%zu, %zu)", w, dp, sp);
  for (size_t r = 0; r < h; ++r) copy_checked((char *)d + r * dp, (const char *)s + r * sp, w, k, "hipMemcpy2DAsync");
  return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) { g_ctx = "hipMemset"; if (n) { need(d, n, "memset"); memset(d, v, n); } g_ctx = ""; return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }

// ---- device, streams, events -----------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
  memset(p, 0, sizeof *p);
  strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
  strcpy(p->name, "sanitizer stand-in for MI355X");
  p->multiProcessorCount = 256;
  p->sharedMemPerBlock = 160 * 1024;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->totalGlobalMem = (size_t)288 << 30;
  p->warpSize = 64;
  p->maxThreadsPerBlock = 1024;
  return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error (stub)"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }

// ---- RCCL: a one-rank world --------------------------------------------------------------------------------------------------------
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId, int r) { if (n != 1 || r != 0) return ncclInvalidArgument; *c = (ncclComm_t)malloc(8); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t) { return "rccl (stub)"; }
ncclResult_t ncclAllReduce(const void *s, void *d, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t) {
  g_ctx = "ncclAllReduce";
  ++g_allreduces;
  const size_t el = t == ncclFloat ? 4 : (t == ncclDouble ? 8 : 4);
  need(s, count * el, "send buffer"); need(d, count * el, "receive buffer");
  if (s != d) memmove(d, s, count * el);
  g_ctx = "";
  return ncclSuccess;
}

// ---- report ----------------------------------------------------------------------------------------------------------------------
void san_stub_report(void) {
  long total = 0;
  for (auto &kv : g_launches) total += kv.second;
  printf("san-stub: %ld launches checked (%zu kernels), %ld pointer extents checked, %zu live allocations\n", total, g_launches.size(), g_checked_ptrs, g_alloc.size());
  for (auto &kv : g_launches) printf("    %-60s %ld\n", kv.first.c_str(), kv.second);
}
long san_stub_allreduces(void) { return g_allreduces; }
long san_stub_launches(const char *substr) {
  long n = 0;
  for (auto &kv : g_launches) if (kv.first.find(substr) != std::string::npos) n += kv.second;
  return n;
}
void san_stub_poke_int(void *dev, int v) { need(dev, 4, "poke"); memcpy(dev, &v, 4); }
}
