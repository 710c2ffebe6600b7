// Microbenchmark: cycles per v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, dependent chain vs
// independent accumulators, one wave and four waves per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dvec4 __attribute__((ext_vector_type(4)));
typedef float fvec4 __attribute__((ext_vector_type(4)));

template <int NACC, bool F64>
__global__ void k(unsigned long long *out, double *sink, int iters) {
  dvec4 d[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  fvec4 f[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  float af = (float)a, bf = (float)b;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) {
      if (F64) d[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d[u], 0, 0, 0);
      else f[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f[u], 0, 0, 0);
    }
  }
  double s = 0;
  for (int u = 0; u < 4; ++u) s += d[u][0] + d[u][1] + d[u][2] + d[u][3] + f[u][0] + f[u][1] + f[u][2] + f[u][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[threadIdx.x / 64] = t1 - t0;
  sink[threadIdx.x] = s;
}

template <int NACC, bool F64>
void run(const char *name, int threads) {
  unsigned long long *out; double *sink;
  hipMalloc(&out, 64 * sizeof(*out)); hipMalloc(&sink, 2048 * sizeof(double));
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, F64>), dim3(1), dim3(threads), 0, 0, out, sink, iters);
  hipDeviceSynchronize();
  unsigned long long h[64];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  printf("%-34s threads %4d : %.1f cycles per MFMA per wave (wave0), %.1f (last wave)\n", name, threads,
         (double)h[0] / (iters * NACC), (double)h[threads / 64 - 1] / (iters * NACC));
  hipFree(out); hipFree(sink);
}

int main() {
  for (int threads : {64, 256, 1024}) {
    run<1, true>("f64 16x16x4, 1 acc (dependent)", threads);
    run<2, true>("f64 16x16x4, 2 acc", threads);
    run<4, true>("f64 16x16x4, 4 acc", threads);
    run<1, false>("f32 16x16x4, 1 acc (dependent)", threads);
    run<4, false>("f32 16x16x4, 4 acc", threads);
  }
  return 0;
}
