// Does a wavefront with part of its lanes masked off issue fp64 arithmetic faster on gfx950?
// One wavefront; lanes >= `active` leave (EXEC narrows to the low lanes); the rest run eight
// independent v_fma_f64 chains (issue-bound) and one dependent chain.  s_memtime ticks per
// instruction for active = 64 / 32 / 16 lanes, alone on the SIMD and with `waves` wavefronts in
// the workgroup (4 = one per SIMD, 8 = two per SIMD).
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_exec tools/ubench_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, double* sink, double seed, int reps, int active, int dep) {
  const int lane = threadIdx.x & 63;
  if (lane >= active) return;
  unsigned long long t0, t1;
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + lane + i;
  const double b = 0.999999, c = 1e-9;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; r++) {
    if (dep) {
#pragma unroll
      for (int i = 0; i < 64; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
    } else {
#pragma unroll
      for (int i = 0; i < 64; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b), "v"(c));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  sink[threadIdx.x] = s;
}
int main() {
  unsigned long long* out; double* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 8 * 1024);
  const int reps = 2000;
  for (int waves : {1, 4, 8})
    for (int dep : {0, 1})
      for (int active : {64, 32, 16, 8}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, sink, 1.0, reps, active, dep);
        (void)hipDeviceSynchronize();
        unsigned long long t[8]; (void)hipMemcpy(t, out, 8 * waves, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int w = 0; w < waves; w++) mx = t[w] > mx ? t[w] : mx;
        printf("waves %d %s active %2d: %.2f ticks per fma (slowest wavefront)\n", waves,
               dep ? "dependent  " : "independent", active, (double)mx / (reps * 64.0));
      }
  return 0;
}
