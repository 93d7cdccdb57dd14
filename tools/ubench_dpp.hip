// Micro-benchmark: fp64 DPP (row_newbcast) against LDS exchanges for a wavefront ALONE on its SIMD
// (gfx950).  The DP ALU supports DPP only as row_newbcast (v_fmac_f64_dpp, v_mov_b64_dpp), with
// row / bank write masks.  Question: what does a broadcast multiply-add cost against an LDS write ->
// barrier -> read round trip when eight lanes of a problem exchange columns?
//   hipcc --offload-arch=gfx950 -O2 -o tools/_diag/ubench_dpp tools/ubench_dpp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

__global__ void k(unsigned long long* out, double* sink, double seed) {
  __shared__ double lds[2048];
  unsigned long long t0, t1;
  double a = seed + threadIdx.x, b = seed * 0.5, c = seed * 0.25, d = seed, e = 1.5, f = 2.5, g = 3.5, h = 4.5;
  int slot = 0;
  // 0: empty
  T0(); T1(); out[slot++] = t1 - t0;
  // 1: dependent v_fma_f64 chain (baseline)
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  T1(); out[slot++] = t1 - t0;
  // 2: 4 independent v_fma_f64 chains (baseline)
  T0();
#pragma unroll
  for (int i = 0; i < REP / 4; i++) {
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(e) : "v"(b), "v"(c));
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f) : "v"(b), "v"(c));
  }
  T1(); out[slot++] = t1 - t0;
  // 3: dependent accumulator, v_fmac_f64_dpp row_newbcast (src0 constant register)
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));
  T1(); out[slot++] = t1 - t0;
  // 4: 4 independent accumulators, v_fmac_f64_dpp
  T0();
#pragma unroll
  for (int i = 0; i < REP / 4; i++) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(e) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(b), "v"(c));
  }
  T1(); out[slot++] = t1 - t0;
  // 5: the broadcast SOURCE is the result of the previous instruction (VALU write -> DPP read):
  //    a = fma(a, b, c); d += bcast(a) * c; alternating, chain through a only
  T0();
#pragma unroll
  for (int i = 0; i < REP / 2; i++) {
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(c));
  }
  T1(); out[slot++] = t1 - t0;
  // 6: full dependence through the broadcast: a = bcast(a) * c + a  (each step reads the previous
  //    result as the DPP source)
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
  T1(); out[slot++] = t1 - t0;
  // 7: v_mov_b64_dpp dependent (broadcast of the previous result)
  T0();
#pragma unroll
  for (int i = 0; i < REP; i++)
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(g));
  T1(); out[slot++] = t1 - t0;
  // 8: half-row pairs (eight-lane groups): two masked fmac per term, 4 independent accumulators
  T0();
#pragma unroll
  for (int i = 0; i < REP / 8; i++) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(a) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xc" : "+v"(d) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc" : "+v"(e) : "v"(b), "v"(c));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xc" : "+v"(f) : "v"(b), "v"(c));
  }
  T1(); out[slot++] = t1 - t0;
  // 9: the LDS form of one column exchange: 4 ds_write_b128, 12 ds_read_b128, 32 fma (REP/8 times)
  {
    typedef double __attribute__((ext_vector_type(2))) d2;
    const unsigned grp = threadIdx.x >> 3, col = threadIdx.x & 7;
    const unsigned wbase = (grp * 80 + col * 2) * 8;       // [row pair][column][2]
    const unsigned rbase = (grp * 80) * 8;
    const unsigned rbase2 = (grp * 80 + ((col >> 1) & 3) * 2) * 8;
    d2 w0 = {a, b}, w1 = {c, d}, w2 = {e, f}, w3 = {g, h};
    d2 s[12];
    T0();
#pragma unroll
    for (int i = 0; i < REP / 8; i++) {
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:128\n\tds_write_b128 %0, %3 offset:256\n\t"
                   "ds_write_b128 %0, %4 offset:384" : : "v"(wbase), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "memory");
      asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:16\n\tds_read_b128 %2, %12 offset:128\n\t"
                   "ds_read_b128 %3, %12 offset:144\n\tds_read_b128 %4, %12 offset:256\n\tds_read_b128 %5, %12 offset:272\n\t"
                   "ds_read_b128 %6, %12 offset:384\n\tds_read_b128 %7, %12 offset:400\n\t"
                   "ds_read_b128 %8, %13\n\tds_read_b128 %9, %13 offset:128\n\tds_read_b128 %10, %13 offset:256\n\t"
                   "ds_read_b128 %11, %13 offset:384\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]),
                     "=&v"(s[7]), "=&v"(s[8]), "=&v"(s[9]), "=&v"(s[10]), "=&v"(s[11])
                   : "v"(rbase), "v"(rbase2) : "memory");
      // column g of H: 8 rows x (c0 s0 + c1 s1 + own t1 + cdt sr)
      d2 hv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        d2 acc;
        acc.x = b * s[2 * r].x; acc.y = b * s[2 * r].y;
        acc.x = __builtin_fma(c, s[2 * r + 1].x, acc.x); acc.y = __builtin_fma(c, s[2 * r + 1].y, acc.y);
        const d2 own = r == 0 ? w0 : (r == 1 ? w1 : (r == 2 ? w2 : w3));
        acc.x = __builtin_fma(e, own.x, acc.x); acc.y = __builtin_fma(e, own.y, acc.y);
        acc.x = __builtin_fma(f, s[8 + r].x, acc.x); acc.y = __builtin_fma(f, s[8 + r].y, acc.y);
        hv[r] = acc;
      }
      w0 = hv[0]; w1 = hv[1]; w2 = hv[2]; w3 = hv[3];
    }
    T1(); out[slot++] = t1 - t0;
    a += w0.x + w1.y + w2.x + w3.y;
  }
  // 10: the DPP form of the same exchange: per row own mul + 2 x 2 masked fmac (columns 0, 1) + 4
  //     masked fmac (columns 2 / 3 to lanes 4 / 5): 9 instructions per row, 8 rows (REP/8 times)
  {
    double t[8] = {a, b, c, d, e, f, g, h};
    const double c0 = seed * 0.3, c1 = seed * 0.2, own = 1.0, cd2 = (threadIdx.x & 7) == 4 ? 1.0 : 0.0,
                 cd3 = (threadIdx.x & 7) == 5 ? 1.0 : 0.0;
    T0();
#pragma unroll
    for (int i = 0; i < REP / 8; i++) {
      double hh[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        double acc;
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(acc) : "v"(own), "v"(t[r]));
        asm volatile("s_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
                     "v_fmac_f64_dpp %0, %1, %3 row_newbcast:9 row_mask:0xf bank_mask:0xc\n\t"
                     "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0x2\n\t"
                     "v_fmac_f64_dpp %0, %1, %4 row_newbcast:10 row_mask:0xf bank_mask:0x8\n\t"
                     "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0x2\n\t"
                     "v_fmac_f64_dpp %0, %1, %5 row_newbcast:11 row_mask:0xf bank_mask:0x8"
                     : "+v"(acc) : "v"(t[r]), "v"(c0), "v"(c1), "v"(cd2), "v"(cd3));
        hh[r] = acc;
      }
#pragma unroll
      for (int r = 0; r < 8; r++) t[r] = hh[r];
    }
    T1(); out[slot++] = t1 - t0;
    a += t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7];
  }
  // 11: same, rows interleaved (the 8 rows' instructions alternate: no back-to-back dependence)
  {
    double t[8] = {a, b, c, d, e, f, g, h};
    const double c0 = seed * 0.3, c1 = seed * 0.2, own = 1.0, cd2 = (threadIdx.x & 7) == 4 ? 1.0 : 0.0,
                 cd3 = (threadIdx.x & 7) == 5 ? 1.0 : 0.0;
    T0();
#pragma unroll
    for (int i = 0; i < REP / 8; i++) {
      double hh[8];
#pragma unroll
      for (int r = 0; r < 8; r++) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(hh[r]) : "v"(own), "v"(t[r]));
#define RND(BC, BM, CO) \
      _Pragma("unroll") for (int r = 0; r < 8; r++) \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #BC " row_mask:0xf bank_mask:" #BM : "+v"(hh[r]) : "v"(t[r]), "v"(CO));
      RND(0, 0x3, c0) RND(8, 0xc, c0) RND(1, 0x3, c1) RND(9, 0xc, c1)
      RND(2, 0x2, cd2) RND(10, 0x8, cd2) RND(3, 0x2, cd3) RND(11, 0x8, cd3)
#undef RND
#pragma unroll
      for (int r = 0; r < 8; r++) t[r] = hh[r];
    }
    T1(); out[slot++] = t1 - t0;
    a += t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7];
  }
  // 12: ds_bpermute_b32 round trip (dependent): pull from another lane through the LDS crossbar
  {
    int x = threadIdx.x, addr = ((threadIdx.x & 56) | 3) * 4;
    T0();
#pragma unroll
    for (int i = 0; i < REP / 8; i++)
      asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(addr));
    T1(); out[slot++] = t1 - t0;
    a += x;
  }
  // 13: v_readlane -> s -> v_fma with SGPR operand (scalar broadcast of one lane), dependent
  {
    T0();
#pragma unroll
    for (int i = 0; i < REP / 4; i++) {
      unsigned lo, hi;
      asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3"
                   : "=s"(lo), "=s"(hi) : "v"((unsigned)__double2loint(a)), "v"((unsigned)__double2hiint(a)));
      const double sa = __hiloint2double(hi, lo);
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "s"(sa), "v"(c));
    }
    T1(); out[slot++] = t1 - t0;
  }
  sink[threadIdx.x] = a + d + e + f + g + h + lds[threadIdx.x];
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
  const char* names[] = {"empty", "fma_f64 dependent", "fma_f64 4 chains", "fmac_f64_dpp dependent acc", "fmac_f64_dpp 4 accs",
                         "fma -> fmac_dpp(src) (x128 pairs)", "fmac_dpp self-broadcast dependent", "mov_b64_dpp dependent",
                         "masked half-row fmac_dpp pairs, 4 accs (x256 instr)", "LDS exchange form of P2 (x32 steps)",
                         "DPP form of P2, row by row (x32 steps)", "DPP form of P2, interleaved (x32 steps)",
                         "ds_bpermute dependent (x32)", "readlane x2 + fma sgpr (x64)"};
  const int counts[] = {1, REP, REP, REP, REP, REP / 2, REP, REP, REP, REP / 8, REP / 8, REP / 8, REP / 8, REP / 4};
  for (int i = 0; i < 14; i++)
    printf("%-56s total %8llu ticks   per item %8.2f (minus empty %6.2f)\n", names[i], h[i], (double)h[i] / counts[i],
           ((double)h[i] - (double)h[0]) / counts[i]);
  return 0;
}
