// Microbenchmark + self-check of the forward environment chain kernels (env_chain_kernel vs env_chain_mfma_kernel) at a BASELINE shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../tensornetworkforml_amd/csrc -I../../include chain_bench.hip -o chain_bench
//   ./chain_bench [M=20] [b=5000] [right_envs=0]
#include "../../tensornetworkforml_amd/csrc/kernels_wide.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace tnml;

int main(int argc, char **argv) {
  const int N = 784, D = 2, L = 2;
  const int M = argc > 1 ? atoi(argv[1]) : 20, b = argc > 2 ? atoi(argv[2]) : 5000, right = argc > 3 ? atoi(argv[3]) : 0;
  const int b_pad = (b + 63) / 64 * 64;
  std::vector<int> bond(N - 1);
  for (int i = 0; i < N - 1; ++i) { int e = std::min(i + 1, N - 1 - i); bond[i] = e >= 5 ? M : std::min(M, 1 << e); }
  const size_t core_stride = (size_t)M * D * M;
  std::vector<float> cores(N * core_stride, 0.f), lab((size_t)M * D * M * L, 0.f), X((size_t)N * b_pad * D);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (auto &v : X) v = 0.6f + 0.4f * rnd();
  std::vector<ChainSite> tab(N);
  for (int k = 0; k < N; ++k) {
    const int i = right ? N - 1 - k : k;
    const int ml = i == 0 ? 1 : bond[i - 1], mr = i == N - 1 ? 1 : bond[i];
    const bool islab = k == N - 1;
    float *dst = islab ? lab.data() : cores.data() + i * core_stride;
    const int cnt = ml * D * mr * (islab ? L : 1);
    for (int e = 0; e < cnt; ++e) dst[e] = rnd() * 4.2f / sqrtf((float)std::max(ml, mr));
    ChainSite cs{};
    cs.x_site = i; cs.is_label = islab; cs.core_off = islab ? 0 : (int)(i * core_stride);
    if (right) {
      cs.n_in = mr;
      if (!islab) { cs.n_out = ml; cs.s_in = 1; cs.s_d = mr; cs.s_out = D * mr; cs.env_out_off = (long long)i * M * b_pad; }
      else { cs.n_out = L; cs.s_in = L; cs.s_d = mr * L; cs.s_out = 1; cs.env_out_off = -1; }
    } else {
      cs.n_in = ml;
      if (!islab) { cs.n_out = mr; cs.s_in = D * mr; cs.s_d = mr; cs.s_out = 1; cs.env_out_off = (long long)i * M * b_pad; }
      else { cs.n_out = L; cs.s_in = D * L; cs.s_d = L; cs.s_out = 1; cs.env_out_off = -1; }
    }
    tab[k] = cs;
  }
  ChainSite *dtab; float *dc, *dl, *dX, *denv[2], *df[2];
  hipMalloc(&dtab, N * sizeof(ChainSite)); hipMalloc(&dc, cores.size() * 4); hipMalloc(&dl, lab.size() * 4 + 64); hipMalloc(&dX, X.size() * 4);
  for (int v = 0; v < 2; ++v) { hipMalloc(&denv[v], (size_t)N * M * b_pad * 4); hipMalloc(&df[v], (size_t)L * b_pad * 4); hipMemset(denv[v], 0, (size_t)N * M * b_pad * 4); }
  hipMemcpy(dtab, tab.data(), N * sizeof(ChainSite), hipMemcpyHostToDevice);
  hipMemcpy(dc, cores.data(), cores.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dl, lab.data(), lab.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int mo = std::max(M, L);
  const size_t lds = ((size_t)M * kD * mo + 2 * (size_t)mo * kChainTS + kChainTS * kD + 2 * kChainTS) * sizeof(float);
  auto run = [&](int variant, bool stores) {
    float *env = stores ? denv[variant] : nullptr;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int it = 0; it < 10; ++it) {
        if (variant == 0)
          hipLaunchKernelGGL(env_chain_kernel<false>, dim3(b_pad / kChainTS), dim3(kChainThreads), lds, 0, dtab, N, dc, dl, dX, env, df[0], b, b_pad, L, M, nullptr);
        else
          launch_env_chain(dtab, N, dc, dl, dX, env, df[1], b, b_pad, L, M, nullptr, 0);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s %s: %.3f ms per chain (%s)\n", variant ? "launch_env_chain (dispatch)" : "env_chain_kernel (plain FMA)", stores ? "stack stored" : "no stores   ", ms / 10, hipGetErrorString(hipGetLastError()));
  };
  run(0, true); run(1, true); run(0, false); run(1, false);
#ifdef TNML_CHAIN_STAMPS
  {
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chain_stamps), sizeof st);
    printf("roles kernel, workgroup 7, cycles per site over all launches so far (%d chains): computing wave waits %.0f | loader 0 waits for ring room %.0f, per own site; core -> staging (incl. its load wait) %.0f per own site\n",
           40 * 2 + 3, (double)st[8] / (83.0 * N), (double)st[9] / (83.0 * N / TNML_CHAIN_LOADERS), (double)st[10] / (83.0 * N / TNML_CHAIN_LOADERS));
  }
#endif
  std::vector<float> f0((size_t)L * b_pad), f1((size_t)L * b_pad);
  hipMemcpy(f0.data(), df[0], f0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(f1.data(), df[1], f1.size() * 4, hipMemcpyDeviceToHost);
  double mx = 0, md = 0;
  for (int l = 0; l < L; ++l) for (int s = 0; s < b; ++s) { mx = std::max(mx, (double)fabsf(f0[l * b_pad + s])); md = std::max(md, (double)fabsf(f0[l * b_pad + s] - f1[l * b_pad + s])); }
  printf("max|f| %.3e, max|f_fma - f_mfma| %.3e (rel %.2e)\n", mx, md, md / mx);
  std::vector<float> ea((size_t)M * b_pad), eb((size_t)M * b_pad);
  double worst = 0;
  for (int i : {0, 3, 100, 400, 782}) {
    hipMemcpy(ea.data(), denv[0] + (size_t)i * M * b_pad, ea.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(eb.data(), denv[1] + (size_t)i * M * b_pad, eb.size() * 4, hipMemcpyDeviceToHost);
    double m1 = 0, d1 = 0;
    for (size_t e = 0; e < ea.size(); ++e) { m1 = std::max(m1, (double)fabsf(ea[e])); d1 = std::max(d1, (double)fabsf(ea[e] - eb[e])); }
    if (m1 > 0) worst = std::max(worst, d1 / m1);
  }
  printf("environment stack, worst relative difference over 5 sites: %.2e\n", worst);
  return 0;
}
