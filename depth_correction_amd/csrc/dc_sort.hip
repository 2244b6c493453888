// rocPRIM radix sorts and scans of the library, instantiated once (dc_sort.h).
#include "dc_sort.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

namespace dc {

size_t sort_pairs_bytes(size_t n, int key_bits) {
  size_t b = 0;
  const size_t ne = n > 0 ? n : 1;
  if (key_bits > 32)
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ne, 0, 64,
                                    (hipStream_t)0);
  else
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ne, 0, 32,
                                    (hipStream_t)0);
  return b;
}

hipError_t sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out,
                          size_t n, unsigned bit0, unsigned bit1, hipStream_t stream) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, (const uint32_t*)vals_in, (uint32_t*)vals_out, n, bit0, bit1, stream);
}

hipError_t sort_pairs_u32(void* tmp, size_t tmp_bytes, const void* keys_in, void* keys_out, const void* vals_in, void* vals_out,
                          size_t n, unsigned bit0, unsigned bit1, hipStream_t stream) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint32_t*)keys_in, (uint32_t*)keys_out, (const uint32_t*)vals_in,
                                   (uint32_t*)vals_out, n, bit0, bit1, stream);
}

size_t scan_bytes(size_t n) {
  size_t a = 0, b = 0;
  const size_t ne = n > 0 ? n : 1;
  (void)rocprim::inclusive_scan(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, ne, rocprim::plus<uint32_t>(), (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, ne, rocprim::plus<uint32_t>(), (hipStream_t)0);
  return a > b ? a : b;
}

hipError_t inclusive_scan_32(void* tmp, size_t tmp_bytes, const void* in, void* out, size_t n, hipStream_t stream) {
  return rocprim::inclusive_scan(tmp, tmp_bytes, (const uint32_t*)in, (uint32_t*)out, n, rocprim::plus<uint32_t>(), stream);
}

hipError_t exclusive_scan_32(void* tmp, size_t tmp_bytes, const void* in, void* out, size_t n, hipStream_t stream) {
  return rocprim::exclusive_scan(tmp, tmp_bytes, (const uint32_t*)in, (uint32_t*)out, 0u, n, rocprim::plus<uint32_t>(), stream);
}

}  // namespace dc
