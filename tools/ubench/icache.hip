// Is once-through code instruction-fetch bound?  The step kernel is ~100 KB of mostly straight-line code executed once per
// launch by one workgroup; the instruction cache is 64 KB per pair of CUs.  This benchmark runs S bytes of straight-line
// VALU code (8-byte v_fma_f32, dependent chain per lane) three times inside one launch and reports the cycles of each
// pass: pass 0 of a launch is cold if the cache does not survive the launch boundary, passes 1-2 are warm if S fits.
//   hipcc --offload-arch=gfx950 -O3 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>

#if defined(NOP)   // 4-byte s_nop: ~1 cycle each, i.e. ~4 B/cycle of instruction demand -- exposes the fetch rate itself (sizes are halved)
#define I1 asm volatile("s_nop 0");
#define I8 I1 I1 I1 I1 I1 I1 I1 I1
#elif defined(INDEP)      // four independent chains: 4 cycles per instruction when warm, i.e. 2 B/cycle of instruction demand per wave
#define I1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define I2 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
#define I3 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
#define I4 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
#define I8 I1 I2 I3 I4 I1 I2 I3 I4
#else
#define I1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define I8 I1 I1 I1 I1 I1 I1 I1 I1
#endif
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64          // 4 KB
#define I1K I512 I512                                   // 8 KB
#define I2K I1K I1K                                     // 16 KB
#define I4K I2K I2K                                     // 32 KB
#define I8K I4K I4K                                     // 64 KB
#define I16K I8K I8K                                    // 128 KB

// a second flavour: independent short-latency instructions interleaved with an LDS round trip every 64 instructions (closer
// to the real kernel: the wave stalls regularly, which gives the fetcher time to run ahead -- if it does)
#define J64 I64 asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory"); a += d;
#define J512 J64 J64 J64 J64 J64 J64 J64 J64
#define J1K J512 J512
#define J2K J1K J1K
#define J4K J2K J2K
#define J8K J4K J4K

template <int KB, bool LDS>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink, float b, float c) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  float a = threadIdx.x * 1e-6f, d = 0.f, a2 = a + 1.f, a3 = a + 2.f, a4 = a + 3.f;
  const unsigned addr = threadIdx.x * 4;
  unsigned long long t[4];
#pragma nounroll
  for (int rep = 0; rep < 3; ++rep) {
    t[rep] = __builtin_amdgcn_s_memtime();
    if constexpr (!LDS) {
      if constexpr (KB == 8) { I1K } else if constexpr (KB == 16) { I2K } else if constexpr (KB == 32) { I4K }
      else if constexpr (KB == 64) { I8K } else { I16K }
    } else {
      if constexpr (KB == 8) { J1K } else if constexpr (KB == 16) { J2K } else if constexpr (KB == 32) { J4K } else { J8K }
    }
    __syncthreads();
  }
  t[3] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t[1] - t[0]; out[1] = t[2] - t[1]; out[2] = t[3] - t[2]; }
  sink[threadIdx.x] = a + d + a2 + a3 + a4;
}

template <int KB, bool LDS>
void run(int threads, unsigned long long *dout, float *dsink) {
  unsigned long long h[3];
  for (int launch = 0; launch < 3; ++launch) {
    hipLaunchKernelGGL((k<KB, LDS>), dim3(1), dim3(threads), 0, 0, dout, dsink, 1.0000001f, 1e-9f);
    hipDeviceSynchronize();
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    const double n = KB * 1024.0 / 8.0;
    printf("%3d KB %s %4d threads launch %d: pass0 %8llu  pass1 %8llu  pass2 %8llu cycles  (%.2f / %.2f / %.2f per instruction; cold fetch %.2f B/cycle)\n",
           KB, LDS ? "valu+lds" : "valu    ", threads, launch, h[0], h[1], h[2], h[0] / n, h[1] / n, h[2] / n, KB * 1024.0 / h[0]);
  }
}

int main() {
  unsigned long long *dout;
  float *dsink;
  hipMalloc(&dout, 64);
  hipMalloc(&dsink, 4096);
  for (int threads : {64, 1024}) {
    run<8, false>(threads, dout, dsink);
    run<16, false>(threads, dout, dsink);
    run<32, false>(threads, dout, dsink);
    run<64, false>(threads, dout, dsink);
    run<128, false>(threads, dout, dsink);
    run<8, true>(threads, dout, dsink);
    run<32, true>(threads, dout, dsink);
    run<64, true>(threads, dout, dsink);
  }
  return 0;
}
