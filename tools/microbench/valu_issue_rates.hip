// VALU issue rates on gfx950, measured: does a wave64 v_fma_f32 issue at the rate of v_fma_f64 (DESIGN.md section 1, item (ii) of the
// single-precision discussion) or twice as fast?  One workgroup of 64 * W threads per CU x 4 SIMDs, W = waves per SIMD x 4; every
// thread runs CH independent chains of ONE instruction kind (inline asm, nothing for the compiler to fold or pack), ITER times; s_memtime
// (shader cycles) around the loop, per wavefront; reported: cycles per instruction per SIMD (= wave cycles / instructions of the
// waves_per_simd waves that share it) and the flop/clk/SIMD that implies.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_rates valu_issue_rates.hip && ./valu_issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

enum { K_FMA_F64 = 0, K_FMA_F32, K_PK_FMA_F32, K_MUL_F64, K_ADD_F64, K_NKIND };
static const char* kname[] = {"v_fma_f64", "v_fma_f32", "v_pk_fma_f32", "v_mul_f64", "v_add_f64"};
static const double kflop[] = {2.0 * 64, 2.0 * 64, 4.0 * 64, 64.0, 64.0};      // flop per wave instruction

template <int KIND, int CH, bool DEP, int REP = 8>
__global__ void __launch_bounds__(1024) rate_kernel(long long* out, int iters, double seed) {
  double a[CH];
  float f[CH];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[CH];
  const double x = 1.0 + 1e-9 * seed, y = 1e-12 * (threadIdx.x + 1);
  const float xf = (float)x, yf = (float)y;
  const f2 xp = {xf, xf}, yp = {yf, yf};
  for (int c = 0; c < CH; c++) { a[c] = 0.5 + c; f[c] = 0.5f + c; p[c] = f2{0.5f + c, 1.5f + c}; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int cc = 0; cc < CH * REP; cc++) {      // REP x CH instructions per trip: the loop's own scalar instructions stay below 5 % of the stream
      const int c = cc % CH;
      const int s = DEP ? 0 : c;          // DEP: one dependent chain (latency); otherwise CH independent chains (issue rate)
      if (KIND == K_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[s]) : "v"(x), "v"(y));
      if (KIND == K_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[s]) : "v"(x));
      if (KIND == K_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[s]) : "v"(y));
      if (KIND == K_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[s]) : "v"(xf), "v"(yf));
      if (KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[s]) : "v"(xp), "v"(yp));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double sink = 0;
  for (int c = 0; c < CH; c++) sink += a[c] + f[c] + p[c].x + p[c].y;
  if (sink == 12345.678) out[0] = 0;      // keep the chains alive
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, bool DEP>
static void run(int waves_per_simd, int ncu, long long* d_out, FILE* fo) {
  constexpr int CH = 8;
  const int iters = 102400 / (CH * 8);
  const int block = 64 * 4 * waves_per_simd;
  const int nw = ncu * block / 64;
  hipLaunchKernelGGL((rate_kernel<KIND, CH, DEP>), dim3(ncu), dim3(block), 0, 0, d_out, iters, 1.0);   // warm-up (clocks)
  hipLaunchKernelGGL((rate_kernel<KIND, CH, DEP>), dim3(ncu), dim3(block), 0, 0, d_out, iters, 2.0);
  hipDeviceSynchronize();
  std::vector<long long> h(nw + 1);
  hipMemcpy(h.data(), d_out, sizeof(long long) * (nw + 1), hipMemcpyDeviceToHost);
  std::sort(h.begin() + 1, h.end());
  const double med = (double)h[1 + nw / 2];
  const double ninstr = (double)iters * CH * 8;
  const double cyc_per_instr_wave = med / ninstr;                       // one wave's view
  const double cyc_per_instr_simd = cyc_per_instr_wave / waves_per_simd; // the SIMD issues waves_per_simd such streams
  char line[512];
  snprintf(line, sizeof line, "%-14s %-11s waves/SIMD %d: %7.2f cycles per instruction and wave, %6.2f per instruction on the SIMD -> %6.1f flop/clk/SIMD (%6.1f TFLOP/s at 256 CUs x 4 SIMDs x 2.4 GHz)\n",
           kname[KIND], DEP ? "dependent" : "independent", waves_per_simd, cyc_per_instr_wave, cyc_per_instr_simd, kflop[KIND] / cyc_per_instr_simd,
           kflop[KIND] / cyc_per_instr_simd * 256 * 4 * 2.4e9 / 1e12);
  fputs(line, stdout);
  if (fo) fputs(line, fo);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  FILE* fo = argc > 1 ? fopen(argv[1], "w") : nullptr;
  char head[512];
  snprintf(head, sizeof head, "%s (%s), %d CUs; one workgroup per CU, 8 chains per thread, 64 instructions per loop trip, 102,400 instructions per wave, s_memtime (shader cycles), median over wavefronts\n",
           p.name, p.gcnArchName, p.multiProcessorCount);
  fputs(head, stdout);
  if (fo) fputs(head, fo);
  long long* d_out;
  hipMalloc(&d_out, sizeof(long long) * (1 + (size_t)p.multiProcessorCount * 16));
  for (int w : {1, 2, 4}) {
    run<K_FMA_F64, false>(w, p.multiProcessorCount, d_out, fo);
    run<K_FMA_F32, false>(w, p.multiProcessorCount, d_out, fo);
    run<K_PK_FMA_F32, false>(w, p.multiProcessorCount, d_out, fo);
    run<K_MUL_F64, false>(w, p.multiProcessorCount, d_out, fo);
    run<K_ADD_F64, false>(w, p.multiProcessorCount, d_out, fo);
  }
  for (int w : {1, 2}) {
    run<K_FMA_F64, true>(w, p.multiProcessorCount, d_out, fo);
    run<K_FMA_F32, true>(w, p.multiProcessorCount, d_out, fo);
    run<K_PK_FMA_F32, true>(w, p.multiProcessorCount, d_out, fo);
  }
  if (fo) fclose(fo);
  hipFree(d_out);
  return 0;
}
