// Whole sweeps of the BASELINE shapes planned by the real host code of the library (tnml_api.hip + the launch wrappers, built
// --cuda-host-only with AddressSanitizer and UBSan) against the stand-in runtime of hip_stub.cpp, which checks every copy and every
// launch argument block against the allocation registry.  `make san` builds and runs it; tests/test_host_api.py runs `make san`.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../../include/tnml.h"

extern "C" void san_stub_report(void);
extern "C" long san_stub_launches(const char *substr);
extern "C" long san_stub_allreduces(void);

#define OK(call)                                                                              \
  do {                                                                                        \
    int rc_ = (call);                                                                         \
    if (rc_ != TNML_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, tnml_last_error()); exit(1); } \
  } while (0)

struct Cfg { const char *name; int N, M, b, L; int policy; int sweeps; };

static std::vector<int> start_bonds(int N, int M, int D, int L) {
  std::vector<int> bond(N - 1);
  for (int i = 0; i < N - 1; ++i) {                       // the reference's initial shapes: full bond M everywhere
    bond[i] = M;
  }
  (void)D; (void)L;
  return bond;
}

static void run(const Cfg &c, int mode) {
  const int D = 2;
  tnml_ctx *ctx = nullptr;
  OK(tnml_create(&ctx, c.N, D, c.L, c.M, c.b, 0));
  // cores: label on site 0, shapes (ml, D, mr[, L]) with the uniform initial bond
  std::vector<int> bond = start_bonds(c.N, c.M, D, c.L);
  size_t total = 0;
  std::vector<size_t> off(c.N + 1, 0);
  for (int i = 0; i < c.N; ++i) {
    const int ml = i == 0 ? 1 : bond[i - 1], mr = i == c.N - 1 ? 1 : bond[i];
    off[i] = total;
    total += (size_t)ml * D * mr * (i == 0 ? c.L : 1);
  }
  off[c.N] = total;
  std::vector<float> cores(total);
  for (size_t e = 0; e < total; ++e) cores[e] = 0.1f + 1e-3f * (float)(e % 97);
  OK(tnml_set_cores(ctx, cores.data(), total, bond.data(), 0));
  std::vector<float> X((size_t)c.b * c.N * D, 0.5f);
  std::vector<int> y(c.b);
  for (int s = 0; s < c.b; ++s) y[s] = s % c.L;
  OK(tnml_set_input(ctx, X.data(), y.data(), c.b));
  switch (mode) {
    case 0: break;                                          // default: persistent sweep
    case 1: OK(tnml_set_persistent(ctx, 0)); break;         // one launch per step
    case 2: OK(tnml_set_step_pipeline(ctx, 0)); break;      // classic launch sequence
    case 3: OK(tnml_set_narrow_path(ctx, 1)); OK(tnml_set_persistent(ctx, 0)); break;   // large-tensor path everywhere
    case 4: OK(tnml_set_persistent(ctx, 2)); break;         // persistent, one kernel per role
    case 5: case 6: {                                       // one-rank communicator: update side / batch side + all-reduce on two streams (5), fused (6)
      unsigned char uid[128];
      setenv("TNML_FORCE_COMM", "1", 1);
      OK(tnml_comm_unique_id(uid));
      OK(tnml_comm_init(ctx, 0, 1, uid));
      if (mode == 6) OK(tnml_set_comm_overlap(ctx, 0));
      break;
    }
  }
  std::vector<float> f((size_t)c.L * c.b), met((size_t)2 * (c.N - 1));
  double lm = 0;
  OK(tnml_forward_logabsmax(ctx, &lm));
  // staged batches (the bench's data path): two slots, one of them ragged, selected in turn
  OK(tnml_stage_batch(ctx, 0, X.data(), y.data(), c.b));
  OK(tnml_stage_batch(ctx, 1, X.data(), y.data(), c.b > 3 ? c.b - 3 : c.b));
  OK(tnml_stage_batch(ctx, 1, X.data(), y.data(), c.b));            // re-staging a slot replaces it
  if (tnml_select_batch(ctx, 5) == TNML_OK) { fprintf(stderr, "selecting an empty slot succeeded\n"); exit(1); }
  for (int sw = 0; sw < c.sweeps; ++sw) {
    OK(tnml_select_batch(ctx, sw & 1));
    OK(tnml_forward(ctx, f.data()));
    const int left = tnml_l_pos(ctx) == c.N - 1;
    // a sweep in two calls (the second continues mid-chain), as Network.sweep may be driven
    // (odd sweeps: one call, which the persistent path takes whole)
    const int n1 = (sw & 1) ? c.N - 1 : ((c.N - 1) / 3 > 0 ? (c.N - 1) / 3 : 1), n2 = c.N - 1 - n1;
    OK(tnml_sweep(ctx, left, n1, 1, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, c.policy, met.data(), n2 ? nullptr : f.data()));
    if (n2) OK(tnml_sweep(ctx, left, n2, 0, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, c.policy, met.data() + 2 * n1, f.data()));
    if (tnml_l_pos(ctx) != (left ? 0 : c.N - 1)) { fprintf(stderr, "%s: label at %d after a %s sweep\n", c.name, tnml_l_pos(ctx), left ? "left" : "right"); exit(1); }
  }
  // the network afterwards: cores come back with consistent bonds
  size_t need = 0;
  OK(tnml_cores_size(ctx, &need));
  std::vector<float> back(need);
  std::vector<int> bond2(c.N - 1);
  int lp = -1;
  OK(tnml_get_cores(ctx, back.data(), need, bond2.data(), &lp));
  for (int i = 0; i < c.N - 1; ++i)
    if (bond2[i] < 1 || bond2[i] > (c.M > D * c.L ? c.M : D * c.L)) { fprintf(stderr, "%s: bond %d = %d\n", c.name, i, bond2[i]); exit(1); }
  // prediction on a batch of another size (buffers grow), then teardown
  std::vector<float> Xp((size_t)(c.b / 2 + 3) * c.N * D, 0.25f), fp((size_t)c.L * (c.b / 2 + 3));
  OK(tnml_predict(ctx, Xp.data(), c.b / 2 + 3, fp.data()));
  OK(tnml_destroy(ctx));
  printf("planned %-28s mode %d: %d sweeps ok\n", c.name, mode, c.sweeps);
  fflush(stdout);
}

#define FAILS(call)                                                                           \
  do {                                                                                        \
    int rc_ = (call);                                                                         \
    if (rc_ == TNML_OK) { fprintf(stderr, "%s:%d %s succeeded but must fail\n", __FILE__, __LINE__, #call); exit(1); } \
  } while (0)

// The entry points beside tnml_sweep (the three sub-steps as standalone calls, activation, inspection, capture, the per-kernel
// timers) and the argument checks: every call plans its launches and copies through the same registry checks.
static void run_entry_points(int N, int M, int b, int L) {
  const int D = 2;
  tnml_ctx *ctx = nullptr;
  OK(tnml_create(&ctx, N, D, L, M, b, 0));
  std::vector<int> bond = start_bonds(N, M, D, L);
  size_t total = 0;
  for (int i = 0; i < N; ++i) total += (size_t)(i == 0 ? 1 : M) * D * (i == N - 1 ? 1 : M) * (i == 0 ? L : 1);
  std::vector<float> cores(total, 0.05f);
  OK(tnml_set_cores(ctx, cores.data(), total, bond.data(), 0));
  FAILS(tnml_set_cores(ctx, cores.data(), total - 1, bond.data(), 0));          // size does not match the bonds
  FAILS(tnml_set_cores(ctx, cores.data(), total, bond.data(), N));              // label site out of range
  std::vector<float> X((size_t)b * N * D, 0.5f), f((size_t)L * b), act((size_t)L * b), der((size_t)L * b);
  std::vector<int> y(b, 0);
  FAILS(tnml_forward(ctx, f.data()));                                           // no batch yet
  OK(tnml_set_input(ctx, X.data(), nullptr, b));                                // forward-only batch
  OK(tnml_forward(ctx, f.data()));
  FAILS(tnml_sweep(ctx, 0, 1, 1, 1e-3f, 0.f, 0, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 1.f, TNML_TRUNC_FIXED, nullptr, nullptr));  // no labels
  OK(tnml_set_labels(ctx, y.data(), b));
  FAILS(tnml_set_labels(ctx, y.data(), b + 1));
  double amax = 0;
  OK(tnml_f_absmax(ctx, &amax));
  OK(tnml_get_f(ctx, f.data()));
  OK(tnml_set_f(ctx, f.data()));
  OK(tnml_activation(ctx, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.5f, 0, act.data(), der.data()));
  OK(tnml_activation(ctx, TNML_ACT_LINEAR, TNML_LOSS_MSE, 1.f, 1, nullptr, der.data()));
  // update_B / compute_L2_reg on the product of the cores and on a given merged tensor, both directions where the label allows
  const int ml = 1, mr = N > 2 ? M : 1;
  const size_t nB = (size_t)ml * D * D * mr * L;
  std::vector<float> B(nB, 0.01f);
  std::vector<double> Bn(nB), grad(nB);
  float met2[2];
  double loss = 0;
  OK(tnml_update_B(ctx, nullptr, 0, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 1.f, Bn.data(), nB, met2));
  OK(tnml_update_B(ctx, B.data(), 0, 1e-3f, 0.f, 0, TNML_ACT_LINEAR, TNML_LOSS_MSE, 1.f, Bn.data(), nB, nullptr));
  FAILS(tnml_update_B(ctx, B.data(), 0, 1e-3f, 0.f, 0, TNML_ACT_LINEAR, TNML_LOSS_MSE, 1.f, Bn.data(), nB - 1, nullptr));    // capacity
  FAILS(tnml_update_B(ctx, B.data(), 1, 1e-3f, 0.f, 0, TNML_ACT_LINEAR, TNML_LOSS_MSE, 1.f, Bn.data(), nB, nullptr));        // label at 0 cannot move left
  OK(tnml_l2_term(ctx, B.data(), 0, 1e-3f, &loss, grad.data(), nB));
  FAILS(tnml_l2_term(ctx, nullptr, 0, 1e-3f, &loss, grad.data(), nB));
  FAILS(tnml_l2_term(ctx, B.data(), 0, 1e-3f, &loss, grad.data(), nB - 1));
  // tensor_svd alone: one-workgroup shapes, the HBM-resident path, the limits
  {
    const int shapes[][3] = {{4, 6, 2}, {2 * M, 2 * M * L, M}, {2 * M * L, 2 * M, M}, {2 * M, 2 * M, 2 * M}, {2, 2 * M * L, 1}};
    for (auto &sh : shapes) {
      std::vector<float> mat((size_t)sh[0] * sh[1], 0.1f), US((size_t)sh[0] * sh[2]), SV((size_t)sh[2] * sh[1]);
      std::vector<double> sig(sh[0] < sh[1] ? sh[0] : sh[1]);
      OK(tnml_svd_split(ctx, mat.data(), sh[0], sh[1], sh[2], US.data(), SV.data(), sig.data()));
      OK(tnml_svd_split(ctx, mat.data(), sh[0], sh[1], sh[2], US.data(), SV.data(), nullptr));
    }
    std::vector<float> mat(130 * 260, 0.1f), US(130 * 4), SV(4 * 260);
    FAILS(tnml_svd_split(ctx, mat.data(), 130, 260, 4, US.data(), SV.data(), nullptr));   // short side beyond 128
    FAILS(tnml_svd_split(ctx, mat.data(), 4, 6, 5, US.data(), SV.data(), nullptr));       // rank beyond the short side
    FAILS(tnml_svd_split(ctx, mat.data(), 4, 6, 0, US.data(), SV.data(), nullptr));
  }
  // a captured step and the per-kernel timers on each launch form
  for (int mode = 0; mode < 3; ++mode) {
    OK(tnml_set_step_pipeline(ctx, mode != 1));
    OK(tnml_set_narrow_path(ctx, mode == 2));
    OK(tnml_forward(ctx, nullptr));
    const int left = tnml_l_pos(ctx) == N - 1;
    if (mode == 2) OK(tnml_update_B(ctx, nullptr, left, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 1.f, Bn.data(), nB, met2));   // large-tensor path, factored form alone
    OK(tnml_debug_enable(ctx, 1));
    OK(tnml_sweep(ctx, left, 1, 1, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, TNML_TRUNC_FIXED, met2, f.data()));
    std::vector<double> cap((size_t)4 * M * M * L + 4096);
    size_t got = 0;
    for (int what = TNML_DBG_B; what <= TNML_DBG_L2_GRAD; ++what) OK(tnml_get_step_debug(ctx, what, cap.data(), cap.size(), &got));
    FAILS(tnml_get_step_debug(ctx, TNML_DBG_B, cap.data(), 1, &got));
    OK(tnml_debug_enable(ctx, 2));
    OK(tnml_sweep(ctx, left, 1, 0, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, TNML_TRUNC_FIXED, nullptr, nullptr));
    OK(tnml_debug_enable(ctx, 4));
    OK(tnml_profile_enable(ctx, 1));
    OK(tnml_sweep(ctx, left, N - 3 > 0 ? N - 3 : 0, 0, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, TNML_TRUNC_FIXED, nullptr, nullptr));
    OK(tnml_debug_enable(ctx, 0));
    OK(tnml_profile_enable(ctx, 2));
    FAILS(tnml_sweep(ctx, left, 1, 0, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, TNML_TRUNC_FIXED, nullptr, nullptr));   // past the chain end
    FAILS(tnml_sweep(ctx, !left, 1, 1, 1e-3f, 1e-3f, 1, TNML_ACT_SOFTMAX, TNML_LOSS_FULL_CROSS_ENT, 0.1f, 99, nullptr, nullptr));                // unknown policy
    double ms = 0, cnt[8], st[4];
    long long launches = 0;
    for (int which = 0; which <= 5; ++which) OK(tnml_profile_get(ctx, which, &ms, &launches));
    OK(tnml_get_counters(ctx, cnt));
    OK(tnml_svd_stats(ctx, 0, st));
    OK(tnml_svd_stats_ex(ctx, 1, st, 4));
    OK(tnml_profile_reset(ctx));
    OK(tnml_profile_enable(ctx, 0));
    // the environments a finished sweep leaves behind (Network.l_cum_contraction / r_cum_contraction)
    std::vector<float> env((size_t)M * 2 * L * b);
    int m = 0;
    const int side = tnml_l_pos(ctx) == N - 1 ? TNML_SIDE_LEFT : TNML_SIDE_RIGHT;
    OK(tnml_get_env(ctx, side, side == TNML_SIDE_LEFT ? 0 : N - 1, env.data(), env.size(), &m));
    FAILS(tnml_get_env(ctx, side, N + 3, env.data(), env.size(), &m));
    FAILS(tnml_get_env(ctx, side, side == TNML_SIDE_LEFT ? 0 : N - 1, env.data(), 1, &m));
  }
  OK(tnml_set_svd_stop(ctx, 1e-8));
  FAILS(tnml_set_svd_stop(ctx, 1.0));
  OK(tnml_set_trunc_threshold(ctx, 0.99));
  FAILS(tnml_set_trunc_threshold(ctx, 1.5));
  FAILS(tnml_marker(ctx, 0));
  OK(tnml_marker(ctx, 3));
  OK(tnml_scale_cores(ctx, 0.5));
  OK(tnml_destroy(ctx));
  printf("entry points N %d bond %d b %d L %d: ok\n", N, M, b, L);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  // BASELINE.json's single-GPU configurations at their true sizes, both directions (two sweeps), plus ragged small shapes
  std::vector<Cfg> cfgs = {
      {"c2 bond 10 b 1000", 784, 10, 1000, 2, TNML_TRUNC_FIXED, 2},
      {"c3 bond 20 b 5000", 784, 20, 5000, 2, TNML_TRUNC_FIXED, 2},
      {"c5 bond 50 L 10 b 5000", 784, 50, 5000, 10, TNML_TRUNC_FIXED, 2},
      {"c3 reference policy", 784, 20, 5000, 2, TNML_TRUNC_REFERENCE, 3},
      {"ragged N 25 bond 12 L 3", 25, 12, 77, 3, TNML_TRUNC_FIXED, 2},
      {"tiny N 3 bond 3", 3, 3, 9, 2, TNML_TRUNC_REFERENCE, 2},
      {"bond 64 L 3 (largest)", 18, 64, 200, 3, TNML_TRUNC_FIXED, 2},
  };
  for (size_t i = 0; i < cfgs.size(); ++i) {
    if (quick && cfgs[i].N == 784 && cfgs[i].M != 10) continue;
    run(cfgs[i], 0);
    if (cfgs[i].M <= 20) { run(cfgs[i], 1); run(cfgs[i], 2); run(cfgs[i], 4); run(cfgs[i], 5); run(cfgs[i], 6); }
    else if (cfgs[i].M == 50) run(cfgs[i], 5);
    if (cfgs[i].N < 100 || cfgs[i].M == 10) run(cfgs[i], 3);
  }
  run_entry_points(9, 6, 50, 2);
  run_entry_points(12, 20, 300, 3);
  run_entry_points(6, 50, 64, 10);
  run_entry_points(5, 64, 40, 2);
  san_stub_report();
  if (san_stub_launches("sweep_persist") < 1 || san_stub_launches("step_pipe_kernel") < 1 || san_stub_launches("big_jacobi") < 1 ||
      san_stub_launches("narrow_step_kernel") < 1) {
    fprintf(stderr, "a launch path was never taken\n");
    return 1;
  }
  if (san_stub_allreduces() < 5000) { fprintf(stderr, "the communicator path issued only %ld all-reduces\n", san_stub_allreduces()); return 1; }
  printf("communicator path: %ld all-reduces checked\n", san_stub_allreduces());
  printf("host planning under ASan + UBSan: ok\n");
  return 0;
}
