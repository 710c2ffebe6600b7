// Microbenchmark of the primitives that set the length of the step kernel's critical path (one workgroup of 16 waves on one
// CU): workgroup barrier, dependent LDS reads, dependent float64 FMA / float32 transcendental chains, LDS write -> barrier ->
// read hand-off, integer division, global load latency.   hipcc --offload-arch=gfx950 -O3 prims.hip -o prims
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void k(unsigned long long *out, double *sink, const int *gbuf, int iters, int active_waves) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int *ilds = (int *)lds;
  for (int e = tid; e < 4096; e += 1024) ilds[e] = (e * 17 + 5) & 4095;       // pointer-chasing table
  for (int e = tid; e < 2048; e += 1024) lds[2048 + e] = 1.0 + e * 1e-6;
  __syncthreads();
  unsigned long long t[16];
  double acc = 0;
  // A: barrier
  t[0] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __syncthreads();
  t[1] = __builtin_amdgcn_s_memtime();
  // B: dependent LDS reads (b32), only `active_waves` waves issue them
  int idx = tid;
  if (wave < active_waves)
    for (int i = 0; i < iters; ++i) idx = ilds[idx];
  t[2] = __builtin_amdgcn_s_memtime();
  acc += idx;
  __syncthreads();
  // C: dependent f64 fma chain
  double x = 1.0 + tid * 1e-9;
  t[3] = __builtin_amdgcn_s_memtime();
  if (wave < active_waves)
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
  t[4] = __builtin_amdgcn_s_memtime();
  acc += x;
  __syncthreads();
  // D: dependent f32 sqrt -> rcp -> rsq chain (3 transcendentals + 1 add per iteration)
  float y = 1.5f + tid * 1e-3f;
  t[5] = __builtin_amdgcn_s_memtime();
  if (wave < active_waves)
    for (int i = 0; i < iters; ++i) y = __builtin_amdgcn_rsqf(__builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(y))) + 1.0f;
  t[6] = __builtin_amdgcn_s_memtime();
  acc += y;
  __syncthreads();
  // F: LDS write (b64) -> barrier -> read of another thread's value, repeated
  double v = tid;
  t[7] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    lds[2048 + tid] = v;
    __syncthreads();
    v = lds[2048 + ((tid + 65) & 1023)] + 1.0;
  }
  t[8] = __builtin_amdgcn_s_memtime();
  acc += v;
  __syncthreads();
  // G: integer division by a run-time value
  int z = 1000000007 - tid, dsr = gbuf[0] + 7;
  t[9] = __builtin_amdgcn_s_memtime();
  if (wave < active_waves)
    for (int i = 0; i < iters; ++i) z = z / dsr + 1000000007;
  t[10] = __builtin_amdgcn_s_memtime();
  acc += z;
  __syncthreads();
  // H: dependent global loads (L2-resident table of 4096 ints)
  int gi = tid;
  t[11] = __builtin_amdgcn_s_memtime();
  if (wave < active_waves)
    for (int i = 0; i < 64; ++i) gi = gbuf[gi & 4095];
  t[12] = __builtin_amdgcn_s_memtime();
  acc += gi;
  // I: f64 dependent chain through DPP moves (v_mov_dpp x2 + add)
  double w = 1.0 + lane;
  t[13] = __builtin_amdgcn_s_memtime();
  if (wave < active_waves)
    for (int i = 0; i < iters; ++i) {
      int lo = __double2loint(w), hi = __double2hiint(w);
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false);
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false);
      w += __hiloint2double(hi, lo) * 1e-9;
    }
  t[14] = __builtin_amdgcn_s_memtime();
  acc += w;
  if (lane == 0) for (int i = 0; i < 15; ++i) out[wave * 16 + i] = t[i];
  sink[tid] = acc;
}

int main() {
  unsigned long long *out; double *sink; int *gbuf;
  hipMalloc(&out, 16 * 16 * sizeof(*out)); hipMalloc(&sink, 1024 * sizeof(double)); hipMalloc(&gbuf, 4096 * sizeof(int));
  int hg[4096];
  for (int e = 0; e < 4096; ++e) hg[e] = (e * 33 + 7) & 4095;
  hg[0] = 3;
  hipMemcpy(gbuf, hg, sizeof hg, hipMemcpyHostToDevice);
  const int iters = 256;
  for (int aw : {1, 4, 16}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 32768, 0, out, sink, gbuf, iters, aw);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    auto d = [&](int a, int b, int n) { return (double)(h[b] - h[a]) / n; };
    printf("active waves %2d (of 16): barrier %.0f | dep LDS read %.0f | dep f64 fma %.0f | sqrt+rcp+rsq+add %.0f | LDS write->barrier->read %.0f | int div %.0f | dep global load %.0f | dpp f64 step %.0f  cycles\n",
           aw, d(0, 1, iters), d(1, 2, iters), d(3, 4, iters), d(5, 6, iters), d(7, 8, iters), d(9, 10, iters), d(11, 12, 64), d(13, 14, iters));
  }
  return 0;
}
