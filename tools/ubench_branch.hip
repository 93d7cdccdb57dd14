// Cost of a TAKEN branch for a wavefront alone on its SIMD (gfx950): a loop of `body` independent
// v_fma_f64 closed by s_cbranch_scc1, against the same instructions straight-line; and the cost of
// a forward branch over a skipped block (s_cbranch_execz / vccz style: taken when it skips).
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_branch tools/ubench_branch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BODY>
__global__ void k_loop(unsigned long long* out, double* sink, double seed, int reps) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x + i;
  const double b = 0.999999, c = 1e-9;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < BODY; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b), "v"(c));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[0] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  sink[threadIdx.x] = s;
}
// forward branch: every `BODY` instructions a uniform condition skips 4 instructions (taken) or not
template <int BODY>
__global__ void k_skip(unsigned long long* out, double* sink, double seed, int reps, int skip) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x + i;
  const double b = 0.999999, c = 1e-9;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
#pragma unroll
      for (int i = 0; i < BODY; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b), "v"(c));
      asm volatile("s_cmp_eq_u32 %4, 1\n\ts_cbranch_scc1 1f\n\t"
                   "v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\t"
                   "v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n1:"
                   : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c), "s"(skip) : "scc");
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[0] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  sink[threadIdx.x] = s;
}
template <class F> double run(F f) {
  static unsigned long long* out = nullptr; static double* sink = nullptr;
  if (!out) { (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096); }
  f(out, sink); (void)hipDeviceSynchronize();
  f(out, sink); (void)hipDeviceSynchronize();
  unsigned long long t; (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
  return (double)t;
}
int main() {
  const int reps = 4000;
  double t8 = run([&](auto o, auto s) { hipLaunchKernelGGL(k_loop<8>, dim3(1), dim3(64), 0, 0, o, s, 1.0, reps * 8); });
  double t64 = run([&](auto o, auto s) { hipLaunchKernelGGL(k_loop<64>, dim3(1), dim3(64), 0, 0, o, s, 1.0, reps); });
  // same 64 * reps fma: t8 has 8 x the loop-closing branches (+ s_add / s_cmp each)
  printf("loop of 8 fma: %.1f ticks per trip; loop of 64 fma: %.1f per trip -> fma %.2f, taken backward branch + 2 salu %.1f ticks\n",
         t8 / (reps * 8.0), t64 / reps, (t64 / reps - t8 / (reps * 8.0)) / 56.0,
         t8 / (reps * 8.0) - 8.0 * (t64 / reps - t8 / (reps * 8.0)) / 56.0);
  for (int skip : {0, 1}) {
    double t = run([&](auto o, auto s) { hipLaunchKernelGGL(k_skip<16>, dim3(1), dim3(64), 0, 0, o, s, 1.0, reps, skip); });
    printf("forward branch every 16 fma, %s: %.1f ticks per block of 16 fma + branch%s\n", skip ? "TAKEN (skips 4 fma)" : "not taken (runs 4 fma)",
           t / (reps * 8.0), "");
  }
  return 0;
}
