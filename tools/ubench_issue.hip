// Micro-benchmark: what a wavefront ALONE on its SIMD pays per instruction (gfx950).
// One workgroup of 64 threads; s_memtime around unrolled instruction sequences.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

__global__ void k(unsigned long long* out, double* sink, double seed) {
  __shared__ double lds[1024];
  unsigned long long t0, t1;
  double a = seed + threadIdx.x, b = seed * 0.5, c = seed * 0.25, d = seed, e = 1.5, f = 2.5, g = 3.5, h = 4.5;
  int slot = 0;
  // 0: empty
  T0(); T1(); out[slot++] = t1 - t0;
  // 1: dependent v_fma_f64 chain
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  T1(); out[slot++] = t1 - t0;
  // 2: 4 independent v_fma_f64 chains
  T0();
#pragma unroll
  for (int i = 0; i < REP / 4; i++) {
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(e) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f) : "v"(b), "v"(c));
  }
  T1(); out[slot++] = t1 - t0;
  // 3: dependent v_add_f64
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  T1(); out[slot++] = t1 - t0;
  // 4: dependent v_rcp_f64
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_rcp_f64 %0, %0" : "+v"(g));
  T1(); out[slot++] = t1 - t0;
  // 5: independent v_rcp_f64 (4 chains)
  T0();
#pragma unroll
  for (int i = 0; i < REP / 4; i++) {
    asm volatile("v_rcp_f64 %0, %0" : "+v"(g));
    asm volatile("v_rcp_f64 %0, %0" : "+v"(h));
    asm volatile("v_rcp_f64 %0, %0" : "+v"(e));
    asm volatile("v_rcp_f64 %0, %0" : "+v"(f));
  }
  T1(); out[slot++] = t1 - t0;
  // 6: dependent v_mov_b32 (32-bit VALU)
  int x = threadIdx.x, y = 3;
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
  T1(); out[slot++] = t1 - t0;
  // 7: s_nop 0
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("s_nop 0");
  T1(); out[slot++] = t1 - t0;
  // 8: SALU dependent
  int sx = 1;
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sx));
  T1(); out[slot++] = t1 - t0;
  // 9: LDS round trip: write then dependent read, REP/8 times
  lds[threadIdx.x] = a;
  unsigned addr = threadIdx.x * 8;
  double v = a;
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
  }
  T1(); out[slot++] = t1 - t0;
  // 10: ds_read_b64 latency alone (dependent address chain not needed: wait each)
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("ds_read_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
  }
  T1(); out[slot++] = t1 - t0;
  // 11: 8 ds_read_b128 back to back then one wait (group-uniform addresses: 8 distinct)
  unsigned addr2 = (threadIdx.x >> 3) * 8 * 20;
  typedef float __attribute__((ext_vector_type(4))) f4;
  f4 q;
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("ds_read_b128 %0, %1\n\tds_read_b128 %0, %1 offset:16\n\tds_read_b128 %0, %1 offset:32\n\tds_read_b128 %0, %1 offset:48\n\t"
                 "ds_read_b128 %0, %1 offset:64\n\tds_read_b128 %0, %1 offset:80\n\tds_read_b128 %0, %1 offset:96\n\tds_read_b128 %0, %1 offset:112\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(q) : "v"(addr2) : "memory");
  }
  T1(); out[slot++] = t1 - t0;
  // 12: ds_write_b64, all 8 lanes of a group to the SAME address (8 distinct addresses per wave)
  unsigned addr3 = (threadIdx.x >> 3) * 8 * 20;
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8\n\tds_write_b64 %0, %1 offset:16\n\tds_write_b64 %0, %1 offset:24\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr3), "v"(v) : "memory");
  }
  T1(); out[slot++] = t1 - t0;
  // 13: same writes, only lane 0 of each group active
  T0();
  if ((threadIdx.x & 7) == 0) {
#pragma unroll
    for (int i = 0; i < REP / 8; i++) {
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8\n\tds_write_b64 %0, %1 offset:16\n\tds_write_b64 %0, %1 offset:24\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr3), "v"(v) : "memory");
    }
  }
  T1(); out[slot++] = t1 - t0;
  // 14: taken branches: REP/8 forward jumps over 64 instructions each
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("s_cbranch_execz 1f\n\ts_branch 2f\n1:\n\t.rept 200\n\tv_add_f64 %0, %0, %1\n\t.endr\n2:" : "+v"(a) : "v"(b));
  }
  T1(); out[slot++] = t1 - t0;
  // 15: v_cndmask dependent
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
  T1(); out[slot++] = t1 - t0;
  // 16: v_rndne_f64 + v_cvt_i32_f64 + v_ldexp_f64 dependent-ish
  int xi;
  T0();
#pragma unroll
  for (int i = 0; i < REP / 4; i++) {
    asm volatile("v_rndne_f64 %0, %0" : "+v"(a));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(xi) : "v"(a));
    asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(xi));
    asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  }
  T1(); out[slot++] = t1 - t0;
  // 17: v_mul_f64 dependent
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(e));
  T1(); out[slot++] = t1 - t0;
  // 18: v_fma_f32 dependent
  float fa = seed, fb = 0.5f, fc = 0.25f;
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
  T1(); out[slot++] = t1 - t0;
  sink[threadIdx.x] = a + d + e + f + g + h + x + sx + v + q.x + fa + xi;
}

int main() {
  unsigned long long* out;
  double* sink;
  (void)hipMalloc(&out, 64 * 8);
  (void)hipMalloc(&sink, 64 * 8);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 1.0000001);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(32);
  (void)hipMemcpy(h.data(), out, 32 * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"empty", "fma_f64 dependent", "fma_f64 4 chains", "add_f64 dependent", "rcp_f64 dependent",
                         "rcp_f64 4 chains", "add_u32 dependent", "s_nop 0", "s_add dependent", "LDS write->read round trip (x32)",
                         "ds_read_b64 + wait (x32)", "8 ds_read_b128 + wait (x32)", "4 ds_write_b64 same-addr groups + wait (x32)",
                         "4 ds_write_b64 lane0 of group + wait (x32)", "taken branch over 200 instr (x32)", "cndmask dependent",
                         "rndne/cvt/ldexp/max (x64 each)", "mul_f64 dependent", "fma_f32 dependent"};
  const int counts[] = {1, REP, REP, REP, REP, REP, REP, REP, REP, REP / 8, REP / 8, REP / 8, REP / 8, REP / 8, REP / 8, REP, REP, REP, REP};
  for (int i = 0; i < 19; i++)
    printf("%-48s total %8llu ticks   per item %8.2f (minus empty %6.2f)\n", names[i], h[i], (double)h[i] / counts[i],
           ((double)h[i] - (double)h[0]) / counts[i]);
  return 0;
}
