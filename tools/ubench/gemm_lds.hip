// Microbenchmark: the step kernel's small float64 products (mm_lds / small_gemm_f64 of small_gemm_device.h) on LDS operands,
// one workgroup of 1024 threads.   hipcc --offload-arch=gfx950 -O3 -I../../tensornetworkforml_amd/csrc gemm_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "small_gemm_device.h"
#include "jacobi_device.h"
using namespace tnml;

__global__ __launch_bounds__(1024) void k(unsigned long long *out, double *sink, int n, int len, int variant) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *W = (float *)smem;                         // [n][len + 1]
  double *P = (double *)(smem + 32768);             // results
  const int tid = threadIdx.x;
  for (int e = tid; e < n * (len + 1); e += 1024) W[e] = 1.0f + 1e-3f * (e % 97);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (variant == 0) {          // Gram, two k halves, upper tiles
    mm_lds(2, n, n, len / 2, W, len / 2, len + 1, 1, W, len / 2, 1, len + 1, [&](int bt, int i, int j, double v) { P[(bt * n + i) * n + j] = v; }, true);
  } else if (variant == 1) {   // one k range, all tiles
    mm_lds(1, n, n, len, W, 0, len + 1, 1, W, 0, 1, len + 1, [&](int, int i, int j, double v) { P[i * n + j] = v; });
  } else if (variant == 2) {   // lambda form
    small_gemm_f64(1, n, n, len, [&](int, int i, int kk) { return (double)W[i * (len + 1) + kk]; },
                   [&](int, int kk, int j) { return (double)W[j * (len + 1) + kk]; }, [&](int, int i, int j, double v) { P[i * n + j] = v; });
  } else if (variant == 3) {   // 20 x 40, K = 20 (the T2 product)
    mm_lds(1, 20, 40, 20, W, 0, len + 1, 1, W, 0, 1, len + 1, [&](int, int i, int j, double v) { P[i * 40 + j] = v; });
  } else if (variant == 4) {   // nothing: the cost of the two stamps and the barrier
  } else if (variant == 5) {   // 20 x 40, K = 4: one MFMA per tile
    mm_lds(1, 20, 40, 4, W, 0, len + 1, 1, W, 0, 1, len + 1, [&](int, int i, int j, double v) { P[i * 40 + j] = v; });
  } else if (variant == 6) {   // 20 x 40, K = 20, float32 MFMA
    mm_lds_f32(20, 40, 20, W, len + 1, 1, W, 1, len + 1, [&](int i, int j, float v) { P[i * 40 + j] = (double)v; });
  } else {                     // 20 x 160, K = 40, float32 MFMA (the contraction)
    mm_lds_f32(20, 160, 40, W, len + 1, 1, W, 1, 21, [&](int i, int j, float v) { P[i * 160 + j] = (double)v; });
  }
  lds_barrier();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((tid & 63) == 0) out[tid >> 6] = t1 - t0;
  sink[tid] = P[tid];
}

int main() {
  unsigned long long *out; double *sink;
  hipMalloc(&out, 64 * sizeof(*out)); hipMalloc(&sink, 1024 * sizeof(double));
  const char *names[] = {"gram 40x40 K=80 as 2 halves, upper tiles (12 items)", "gram 40x40 K=80 all 9 tiles", "small_gemm_f64 40x40 K=80", "mm_lds 20x40 K=20", "empty (stamps + barrier)", "mm_lds 20x40 K=4", "mm_lds_f32 20x40 K=20", "mm_lds_f32 20x160 K=40"};
  for (int v = 0; v < 8; ++v) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 65536, 0, out, sink, 40, 80, v);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-55s: wave0 %llu cycles, max over waves %llu\n", names[v], h[0], [&] { unsigned long long m = 0; for (int i = 0; i < 16; ++i) m = h[i] > m ? h[i] : m; return m; }());
  }
  return 0;
}
