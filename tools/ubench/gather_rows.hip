// Micro-benchmark: what do N x K random row gathers from a small table (one lidar scan: 200 k points, 2.4-3.2 MB -- L2-resident)
// cost on gfx950, by row format (12-B rows: dwordx3, 16-B rows: dwordx4) and cache policy (plain / sc0 / nt / sc1 / sc0 sc1)?
// This is the floor of dc_features_fwd on a scan whose points are in no spatial order (RoomBoxDataset): every gather is its own line.
//   hipcc -O3 --offload-arch=gfx950 -o build/gather_rows tools/ubench/gather_rows.hip && build/gather_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef __amdgpu_buffer_rsrc_t BufRsrc;
constexpr int K = 10;

template <int AUX, int ROWB>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ out, int n) {
  BufRsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)n * ROWB, 0x00020000);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int row[K];
#pragma unroll
  for (int q = 0; q < K; ++q) row[q] = idx[(size_t)q * n + i];          // slot-major: coalesced
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    if (ROWB == 12) {
      auto v = __builtin_amdgcn_raw_buffer_load_b96(r, (unsigned)row[q] * 12u, 0, AUX);
      s += __int_as_float(v[0]) + __int_as_float(v[1]) + __int_as_float(v[2]);
    } else {
      auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)row[q] * 16u, 0, AUX);
      s += __int_as_float(v[0]) + __int_as_float(v[1]) + __int_as_float(v[2]);
    }
  }
  out[i] = s;
}

template <int AUX, int ROWB>
static float run(const float* x, const int* idx, float* out, int n, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gather_kernel<AUX, ROWB>), dim3((n + 255) / 256), dim3(256), 0, 0, x, idx, out, n);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<AUX, ROWB>), dim3((n + 255) / 256), dim3(256), 0, 0, x, idx, out, n);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200000, reps = 300;
  std::vector<float> hx((size_t)n * 4, 1.0f);
  float *x, *out; int* idx;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&out, (size_t)n * 4); hipMalloc(&idx, (size_t)n * K * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  std::mt19937 rng(1);
  for (int mode = 0; mode < 3; ++mode) {
    // 0: every gather a random row; 1: centre i gathers rows within +-32 of a random anchor shared by nobody (random centres, local
    // neighbours: an unordered scan); 2: rows within +-32 of i (a spatially ordered scan)
    std::vector<int> h((size_t)n * K);
    for (int i = 0; i < n; ++i) {
      const int anchor = mode == 2 ? i : (int)(rng() % n);
      for (int q = 0; q < K; ++q) {
        int j = mode == 0 ? (int)(rng() % n) : anchor + (int)(rng() % 65) - 32;
        h[(size_t)q * n + i] = std::min(std::max(j, 0), n - 1);
      }
    }
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("mode %d (%s), n = %d, K = %d, us per launch:\n", mode, mode == 0 ? "random rows" : mode == 1 ? "random anchors, local rows" : "ordered scan", n, K);
    printf("  12-B rows: plain %.2f  sc0 %.2f  nt %.2f  sc1 %.2f  sc0sc1 %.2f\n", run<0, 12>(x, idx, out, n, reps), run<1, 12>(x, idx, out, n, reps),
           run<2, 12>(x, idx, out, n, reps), run<16, 12>(x, idx, out, n, reps), run<17, 12>(x, idx, out, n, reps));
    printf("  16-B rows: plain %.2f  sc0 %.2f  nt %.2f  sc1 %.2f  sc0sc1 %.2f\n", run<0, 16>(x, idx, out, n, reps), run<1, 16>(x, idx, out, n, reps),
           run<2, 16>(x, idx, out, n, reps), run<16, 16>(x, idx, out, n, reps), run<17, 16>(x, idx, out, n, reps));
  }
  return 0;
}
