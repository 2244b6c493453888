// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the step kernel is made of,
// on gfx950, at 1 / 2 / 4 / 8 wavefronts per SIMD, independent chains (throughput) and one dependent chain (latency).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kIters = 2000;

// 8 independent instances of one instruction per loop trip (INDEP) or a chain of 8 dependent ones (DEP)
#define BODY8(ASM_I) ASM_I(0) ASM_I(1) ASM_I(2) ASM_I(3) ASM_I(4) ASM_I(5) ASM_I(6) ASM_I(7)

template <int KIND, bool DEP>
__global__ void rate_kernel(unsigned long long* out, double seed) {
  double d[8];
  float f[8];
  int n[8];
  for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)d[i]; n[i] = (int)(seed * 100) + i + threadIdx.x; }
  const double c1 = seed * 0.999, c2 = 1e-9;
  const float g1 = (float)c1, g2 = 1e-9f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = DEP ? 0 : i;
      if constexpr (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(c1), "v"(c2));
      else if constexpr (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(c2));
      else if constexpr (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(c1));
      else if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(g1), "v"(g2));
      else if constexpr (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(g2));
      else if constexpr (KIND == 5) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[j]) : "v"(n[j]));
      else if constexpr (KIND == 6) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[j]) : "v"(n[j]));
      else if constexpr (KIND == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(f[j]));
      else if constexpr (KIND == 8) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j]));
      else if constexpr (KIND == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[j]));
      else if constexpr (KIND == 10) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[j]));
      else if constexpr (KIND == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[j]));
      else if constexpr (KIND == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
      else if constexpr (KIND == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[j]) : "v"(n[(j + 1) & 7]) : "vcc");
      else if constexpr (KIND == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[j]) : "v"(c1), "v"(c2));
      else if constexpr (KIND == 15) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[j]));
      else if constexpr (KIND == 16) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[j]) : "v"(d[j]));
      else if constexpr (KIND == 17) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(n[j]) : "v"(n[(j + 1) & 7]));
      else if constexpr (KIND == 18) asm volatile("v_cos_f32 %0, %0" : "+v"(f[j]));
      else if constexpr (KIND == 19) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[j]));
      else if constexpr (KIND == 20) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[j]) : "v"(n[(j + 1) & 7]), "v"(n[(j + 2) & 7]));
      else if constexpr (KIND == 21) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[j]) : "v"(n[j]), "v"(n[(j + 1) & 7]) : "vcc");
      else if constexpr (KIND == 22) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
      else if constexpr (KIND == 23) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[j]) : "v"(c1));
      else if constexpr (KIND == 24) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[j]) : "v"(c1), "v"(c2));
      else if constexpr (KIND == 25) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[j]) : "v"(c2));
      else if constexpr (KIND == 26) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[j]) : "v"(c1));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double acc = 0.0;
  for (int i = 0; i < 8; ++i) acc += d[i] + (double)f[i] + (double)n[i];
  if (acc == 1.2345e-300) out[1000] = 1;                         // keep everything alive
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

struct Kind { int id; const char* name; };

template <int KIND>
void run(const char* name, unsigned long long* dout) {
  // one CU worth of blocks is enough: rates are per SIMD.  waves per SIMD = block threads / 256 (one block, one CU)
  printf("%-16s", name);
  for (int dep = 0; dep < 2; ++dep) {
    for (int waves : {1, 2, 4, 8}) {
      // every CU gets the same load: 256 x (blocks per CU) blocks; 8 waves per SIMD = two 1024-thread blocks per CU
      const int threads = waves >= 4 ? 1024 : 256 * waves, per_cu = waves >= 4 ? waves / 4 : 1;
      const int grid = 256 * per_cu;
      static unsigned long long h[1024];
      for (int rep = 0; rep < 2; ++rep) {
        if (dep) hipLaunchKernelGGL((rate_kernel<KIND, true>), dim3(grid), dim3(threads), 0, 0, dout, 1.0000001);
        else hipLaunchKernelGGL((rate_kernel<KIND, false>), dim3(grid), dim3(threads), 0, 0, dout, 1.0000001);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(h, dout, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<unsigned long long> v(h, h + grid);
      std::sort(v.begin(), v.end());
      // cycles per instruction per SIMD: elapsed / (iters * 8 instructions * waves on that SIMD); median over the blocks
      const double cyc = (double)v[grid / 2] / (kIters * 8.0 * waves);
      printf("  %s w%d %5.2f", dep ? "dep" : "ind", waves, cyc);
    }
  }
  printf("\n");
}

int main() {
  unsigned long long* dout;
  CHECK(hipMalloc(&dout, 2048 * sizeof(unsigned long long)));
  {
    // tick calibration: s_memtime ticks of a long kernel against its wall time
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((rate_kernel<0, false>), dim3(256), dim3(1024), 0, 0, dout, 1.0000001);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((rate_kernel<0, false>), dim3(256), dim3(1024), 0, 0, dout, 1.0000001);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned long long t; CHECK(hipMemcpy(&t, dout, sizeof(t), hipMemcpyDeviceToHost));
    printf("calibration: %llu ticks inside a kernel of %.1f us wall -> >= %.0f MHz tick rate\n", t, ms * 1e3, (double)t / (ms * 1e3));
  }
  printf("cycles (s_memtime ticks) per wave64 instruction per SIMD; ind = 8 independent chains, dep = one dependent chain; wN = N waves per SIMD\n");
  run<0>("v_fma_f64", dout); run<24>("v_fmac_f64", dout); run<1>("v_add_f64", dout); run<2>("v_mul_f64", dout); run<23>("v_max_f64", dout);
  run<3>("v_fma_f32", dout); run<4>("v_add_f32", dout); run<14>("v_pk_fma_f32", dout); run<25>("v_pk_add_f32", dout); run<26>("v_pk_mul_f32", dout);
  run<5>("v_cvt_f64_i32", dout); run<6>("v_cvt_f32_i32", dout); run<7>("v_cvt_f64_f32", dout); run<8>("v_cvt_f32_f64", dout);
  run<15>("v_rndne_f64", dout); run<16>("v_cvt_i32_f64", dout);
  run<9>("v_rcp_f64", dout); run<10>("v_rsq_f64", dout); run<11>("v_rcp_f32", dout); run<18>("v_cos_f32", dout); run<19>("v_sqrt_f32", dout);
  run<12>("v_sub_u32", dout); run<13>("v_cndmask_b32", dout); run<17>("v_mov_b32_dpp", dout);
  run<20>("v_mad_i32_i24", dout); run<21>("v_mad_i64_i32", dout); run<22>("v_mul_lo_u32", dout);
  return 0;
}
