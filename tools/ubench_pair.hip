// What exchanging one fp64 word between the two lanes of a lane PAIR costs a wavefront alone on
// its SIMD (gfx950) — the primitive of a two-lanes-per-problem form of the lane kernels (DESIGN.md
// §9).  The fp64 ALU takes DPP only as row_newbcast, so a pair exchange is two 32-bit moves:
//   a) 2 x v_mov_b32_dpp quad_perm:[1,0,3,2]            (swap inside lane pairs)
//   b) the same followed by the v_add_f64 that consumes it (partial sums: exchange + add)
//   c) 2 x ds_swizzle_b32 (swap, offset 0x041F) + wait    (through the LDS crossbar)
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_pair tools/ubench_pair.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
__global__ void k(unsigned long long* out, double* sink, double seed, int reps) {
  double a[8], p[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; p[i] = 0; }
  unsigned long long t0, t1;
  // a) eight independent exchanges per trip
  TICK(t0);
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "=&v"(((unsigned*)&p[i])[0]), "=&v"(((unsigned*)&p[i])[1])
                   : "v"(((unsigned*)&a[i])[0]), "v"(((unsigned*)&a[i])[1]));
  }
  TICK(t1);
  if (threadIdx.x == 0) out[0] = t1 - t0;
  // b) exchange + add, eight independent accumulators
  TICK(t0);
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "=&v"(((unsigned*)&p[i])[0]), "=&v"(((unsigned*)&p[i])[1])
                   : "v"(((unsigned*)&a[i])[0]), "v"(((unsigned*)&a[i])[1]));
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(p[i]));
    }
  }
  TICK(t1);
  if (threadIdx.x == 0) out[1] = t1 - t0;
  // c) ds_swizzle pair swap
  TICK(t0);
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      asm volatile("ds_swizzle_b32 %0, %2 offset:swizzle(SWAP,1)\n\t"
                   "ds_swizzle_b32 %1, %3 offset:swizzle(SWAP,1)"
                   : "=&v"(((unsigned*)&p[i])[0]), "=&v"(((unsigned*)&p[i])[1])
                   : "v"(((unsigned*)&a[i])[0]), "v"(((unsigned*)&a[i])[1]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(p[i]));
  }
  TICK(t1);
  if (threadIdx.x == 0) out[2] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i];
  sink[threadIdx.x] = s;
}
int main() {
  unsigned long long* out; double* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 512);
  const int reps = 4000;
  for (int w = 0; w < 2; w++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 1.0, reps); (void)hipDeviceSynchronize(); }
  unsigned long long t[3]; (void)hipMemcpy(t, out, 24, hipMemcpyDeviceToHost);
  printf("pair exchange of one fp64 word, wavefront alone on its SIMD (ticks = core cycles):\n");
  printf("  2 x v_mov_b32_dpp quad_perm            %.1f ticks per word\n", t[0] / (reps * 8.0));
  printf("  the same + v_add_f64 (partial sums)    %.1f ticks per word\n", t[1] / (reps * 8.0));
  printf("  2 x ds_swizzle_b32 + wait + v_add_f64  %.1f ticks per word (eight in flight)\n", t[2] / (reps * 8.0));
  return 0;
}
