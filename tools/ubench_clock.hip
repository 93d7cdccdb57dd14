// Calibration of s_memtime ticks against wall time (hipEvents) on gfx950: one wavefront runs a long
// dependent v_fma_f64 chain; prints ticks, nanoseconds and their ratio.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_clock tools/ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, double* sink, double seed, int reps) {
  unsigned long long t0, t1;
  double a = seed + threadIdx.x, b = 0.999999, c = 1e-9;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[0] = t1 - t0;
  sink[threadIdx.x] = a;
}
int main() {
  unsigned long long* out; double* sink;
  (void)hipMalloc(&out, 8); (void)hipMalloc(&sink, 512);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int reps : {1000, 20000, 100000}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 1.0, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 1.0, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
    printf("reps %d: %llu ticks, %.3f us wall -> %.1f ticks/us; %.3f ticks and %.3f ns per dependent fma\n", reps, t, ms * 1e3,
           t / (ms * 1e3), (double)t / (reps * 64.0), ms * 1e6 / (reps * 64.0));
  }
  return 0;
}
