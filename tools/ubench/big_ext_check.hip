// Self-check of big_ext_kernel / big_contract_kernel (kernels_big.hip) against a CPU loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../tensornetworkforml_amd/csrc -I../../include big_ext_check.hip -o big_ext_check
#include "../../tensornetworkforml_amd/csrc/kernels_big.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace tnml;
int main() {
  const int hp = 32, h = 50, b_pad = 5056, D = 2, ncols = 2000;
  std::vector<float> Ep((size_t)hp * b_pad), xm((size_t)b_pad * D), xk((size_t)b_pad * D), A((size_t)hp * D * h), Z((size_t)h * D * ncols + 4);
  srand(3);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (auto &v : Ep) v = rnd(); for (auto &v : xm) v = rnd(); for (auto &v : xk) v = rnd(); for (auto &v : A) v = rnd(); for (auto &v : Z) v = rnd();
  float *dEp, *dxm, *dxk, *dA, *dE, *dP, *dZ, *dR;
  hipMalloc(&dEp, Ep.size() * 4); hipMalloc(&dxm, xm.size() * 4); hipMalloc(&dxk, xk.size() * 4); hipMalloc(&dA, A.size() * 4);
  hipMalloc(&dE, (size_t)h * b_pad * 4); hipMalloc(&dP, (size_t)2 * h * b_pad * 4); hipMalloc(&dZ, Z.size() * 4); hipMalloc(&dR, ((size_t)h * ncols + 4) * 4);
  hipMemcpy(dEp, Ep.data(), Ep.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dxm, xm.data(), xm.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dxk, xk.data(), xk.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice);
  CoreView v{dA, hp, h, D * h, h, 1};
  bool ok = launch_big_ext(dEp, dxm, dxk, v, b_pad, dE, dP, 0);
  hipError_t e1 = hipDeviceSynchronize();
  printf("launch_big_ext %d %s %s\n", ok, hipGetErrorString(hipGetLastError()), hipGetErrorString(e1));
  std::vector<float> E((size_t)h * b_pad), P((size_t)2 * h * b_pad);
  hipMemcpy(E.data(), dE, E.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, mx = 0;
  for (int s = 0; s < b_pad; s += 97)
    for (int o = 0; o < h; ++o) {
      double acc = 0;
      for (int hh = 0; hh < hp; ++hh) for (int d = 0; d < D; ++d) acc += (double)Ep[(size_t)hh * b_pad + s] * xm[(size_t)s * D + d] * A[(size_t)hh * D * h + d * h + o];
      worst = std::max(worst, fabs(acc - E[(size_t)o * b_pad + s])); mx = std::max(mx, fabs(acc));
      worst = std::max(worst, fabs(acc * xk[(size_t)s * D + 1] - P[(size_t)(2 * o + 1) * b_pad + s]));
    }
  printf("big_ext: max |E| %.3e, worst difference %.3e\n", mx, worst);
  CoreView v2{dA, hp, h, D * h, h, 1};   // contraction with the same core: rows (hp, d) -> h
  std::vector<float> Z2((size_t)hp * D * ncols + 4);
  for (auto &x : Z2) x = rnd();
  hipMemcpy(dZ, Z2.data(), Z2.size() * 4, hipMemcpyHostToDevice);
  ok = launch_big_contract(dZ, v2, ncols, dR, 0);
  e1 = hipDeviceSynchronize();
  printf("launch_big_contract %d %s\n", ok, hipGetErrorString(e1));
  std::vector<float> R((size_t)h * ncols + 4);
  hipMemcpy(R.data(), dR, R.size() * 4, hipMemcpyDeviceToHost);
  worst = 0; mx = 0;
  for (int o = 0; o < h; o += 7) for (int c = 0; c < ncols; c += 13) {
    double acc = 0;
    for (int i = 0; i < hp * D; ++i) acc += (double)A[(size_t)(i >> 1) * D * h + (i & 1) * h + o] * Z2[(size_t)i * ncols + c];
    worst = std::max(worst, fabs(acc - R[(size_t)o * ncols + c])); mx = std::max(mx, fabs(acc));
  }
  printf("big_contract: max %.3e worst difference %.3e, tail %g %g (expect %g %g)\n", mx, worst, R[(size_t)h * ncols], R[(size_t)h * ncols + 3], Z2[(size_t)hp * D * ncols], Z2[(size_t)hp * D * ncols + 3]);
  return 0;
}
