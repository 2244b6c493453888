// Micro-benchmark: rocPRIM radix_sort_pairs of n (u32 key, u32 value) pairs over `bits` key bits on gfx950, default configuration
// (merge sort below 1 Mi items) against one with the merge sort limit set to 0 (Onesweep at every size above one block).
//   hipcc -O3 --offload-arch=gfx950 -o build/sort_small tools/ubench/sort_small.hip && build/sort_small
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <random>
#include <rocprim/rocprim.hpp>

using Onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
template <unsigned BS, unsigned IPT, unsigned MinMp = (1u << 17) + 70000u>
using Merge = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, BS, IPT, 128, 128, 4, MinMp>, rocprim::default_config>;

template <class Cfg, class Key>
static float run(size_t n, unsigned bits, const Key* k, Key* ko, const uint32_t* v, uint32_t* vo, int reps) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs<Cfg>(nullptr, b, k, ko, v, vo, n, 0, bits, (hipStream_t)0);
  void* tmp; hipMalloc(&tmp, b);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) (void)rocprim::radix_sort_pairs<Cfg>(tmp, b, k, ko, v, vo, n, 0, bits, (hipStream_t)0);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) (void)rocprim::radix_sort_pairs<Cfg>(tmp, b, k, ko, v, vo, n, 0, bits, (hipStream_t)0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(tmp);
  return ms / reps * 1e3f;
}

int main() {
  std::mt19937_64 rng(1);
  for (size_t n : {20000ul, 50000ul, 200000ul, 430000ul, 1000000ul, 2000000ul}) {
    std::vector<uint32_t> k(n), v(n); std::vector<uint64_t> k64(n);
    for (size_t i = 0; i < n; ++i) { k64[i] = rng(); k[i] = (uint32_t)k64[i]; v[i] = (uint32_t)i; }
    uint32_t *dk, *dko, *dv, *dvo; uint64_t *dk64, *dk64o;
    hipMalloc(&dk, n * 4); hipMalloc(&dko, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dvo, n * 4); hipMalloc(&dk64, n * 8); hipMalloc(&dk64o, n * 8);
    hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk64, k64.data(), n * 8, hipMemcpyHostToDevice);
    for (unsigned bits : {16u, 24u, 30u, 32u})
      printf("n=%8zu u32 bits=%2u  default %7.1f us   onesweep %7.1f us\n", n, bits, run<rocprim::default_config>(n, bits, dk, dko, dv, dvo, 50),
             run<Onesweep>(n, bits, dk, dko, dv, dvo, 50));
    printf("n=%8zu u32 bits=30  merge sort, items per sorted block: 2048 %7.1f  4096 %7.1f  4096(1024x4) %7.1f  8192 %7.1f us; merge path from 0: default block %7.1f  4096 %7.1f\n", n,
           run<Merge<256, 8>>(n, 30, dk, dko, dv, dvo, 50), run<Merge<512, 8>>(n, 30, dk, dko, dv, dvo, 50), run<Merge<1024, 4>>(n, 30, dk, dko, dv, dvo, 50),
           run<Merge<1024, 8>>(n, 30, dk, dko, dv, dvo, 50), run<Merge<512, 2, 0>>(n, 30, dk, dko, dv, dvo, 50), run<Merge<512, 8, 0>>(n, 30, dk, dko, dv, dvo, 50));
    for (unsigned bits : {40u, 48u, 63u})
      printf("n=%8zu u64 bits=%2u  default %7.1f us   onesweep %7.1f us\n", n, bits, run<rocprim::default_config>(n, bits, dk64, dk64o, dv, dvo, 50),
             run<Onesweep>(n, bits, dk64, dk64o, dv, dvo, 50));
    hipFree(dk); hipFree(dko); hipFree(dv); hipFree(dvo); hipFree(dk64); hipFree(dk64o);
  }
  return 0;
}
