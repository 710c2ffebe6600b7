#include <hip/hip_runtime.h>
template <int CTRL> __device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
  double v = threadIdx.x * 1.5;
  out[threadIdx.x] = dpp_f64<0x138>(v);          // wave_shr:1 -> lane i gets lane i-1
  out[64 + threadIdx.x] = dpp_f64<0x130>(v);     // wave_shl:1 -> lane i gets lane i+1
}
int main() {
  double* d; hipMalloc(&d, 128 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("shr1: lane0=%g lane1=%g lane16=%g lane63=%g | shl1: lane0=%g lane15=%g lane62=%g lane63=%g\n", h[0], h[1], h[16], h[63], h[64], h[79], h[126], h[127]);
  return 0;
}
