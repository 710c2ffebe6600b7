// Microbenchmark: the product chain of a slice helper (small_gemm_device.h: prep_slice_block) in isolation -- level 1 = three
// independent mm_lds calls back to back (B = lab . pl, X = Nh^T . lab, Y = pl . Ng), level 2 = X . Y -- on LDS operands of the
// C3 shape (h = 10 rows of 20, g = s = 20, L = 2), one workgroup of 1024 threads.  Inside the step kernel the two levels take
// 1.9 + 1.4 us; how long do they take alone?   hipcc --offload-arch=gfx950 -O3 -I../../tensornetworkforml_amd/csrc helper_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "small_gemm_device.h"
using namespace tnml;

__global__ __launch_bounds__(1024) void k(unsigned long long *out, double *sink, int nr, int h, int g, int s, int L, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, sL = s * L;
  double *dNh = (double *)smem, *dNg = dNh + h * h, *dX = dNg + g * g, *dY = dX + h * sL;
  float *sLab = (float *)(dY + s * g), *sPl = sLab + h * sL, *oB = sPl + s * g;
  double *oG = (double *)(oB + ((h * g * L + 3) & ~3));
  for (int e = tid; e < h * h; e += 1024) dNh[e] = 1.0 + 1e-3 * (e % 13);
  for (int e = tid; e < g * g; e += 1024) dNg[e] = 1.0 + 1e-3 * (e % 7);
  for (int e = tid; e < h * sL; e += 1024) sLab[e] = 1.0f + 1e-3f * (e % 11);
  for (int e = tid; e < s * g; e += 1024) sPl[e] = 1.0f + 1e-3f * (e % 5);
  __syncthreads();
  for (int rep = 0; rep < reps; ++rep) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int slot = mm_lds(L, nr, g, s, sLab, 1, sL, L, sPl, 0, g, 1, [&](int l, int i, int j, double v) { oB[(i * g + j) * L + l] = (float)v; });
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    slot = mm_lds(1, nr, sL, h, dNh, 0, 1, h, sLab, 0, sL, 1, [&](int, int i, int x, double v) { dX[i * sL + x] = v; }, false, slot);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    mm_lds(1, s, g, g, sPl, 0, g, 1, dNg, 0, g, 1, [&](int, int s_, int j, double v) { dY[s_ * g + j] = v; }, false, slot);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    mm_lds(L, nr, g, s, dX, 1, sL, L, dY, 0, g, 1, [&](int l, int i, int j, double v) { oG[(i * g + j) * L + l] = v; });
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t6 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { unsigned long long *o = out + rep * 8; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; }
  }
  sink[tid] = oG[tid % (nr * g * L)] + oB[tid % (nr * g * L)];
}

int main() {
  unsigned long long *dout, h[8 * 4];
  double *dsink;
  hipMalloc(&dout, sizeof h);
  hipMalloc(&dsink, 1024 * 8);
  for (int nr : {10, 20}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 64 * 1024, 0, dout, dsink, nr, 20, 20, 20, 2, 4);
    hipDeviceSynchronize();
    hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r)
      printf("rows %2d pass %d (wave 0, cycles): B %5llu | X %5llu | Y %5llu | barrier %5llu | level 2 %5llu | barrier %5llu\n", nr, r, h[r * 8], h[r * 8 + 1],
             h[r * 8 + 2], h[r * 8 + 3], h[r * 8 + 4], h[r * 8 + 5]);
  }
  return 0;
}
