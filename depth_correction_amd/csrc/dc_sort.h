// The library's radix sorts and prefix sums (rocPRIM) behind plain functions, compiled ONCE in dc_sort.hip.  Every translation unit
// that called rocprim::radix_sort_pairs / *_scan itself carried its own copies of those kernels -- six sort instantiations in three
// code objects, two thirds of an 18 MB library that is loaded (and, on a fresh machine, paged in) at the first call of a process.
// Values are 32-bit words whatever their signedness; int32 keys that are never negative sort as uint32.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace dc {

// bytes of temporary storage for n items (key width in bits: 32 or 64)
size_t sort_pairs_bytes(size_t n, int key_bits);
// stable sort of (key, value) pairs by the key bits [bit0, bit1)
hipError_t sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const void* vals_in, void* vals_out,
                          size_t n, unsigned bit0, unsigned bit1, hipStream_t stream);
hipError_t sort_pairs_u32(void* tmp, size_t tmp_bytes, const void* keys_in, void* keys_out, const void* vals_in, void* vals_out,
                          size_t n, unsigned bit0, unsigned bit1, hipStream_t stream);

// bytes of temporary storage of the prefix sums over n 32-bit items
size_t scan_bytes(size_t n);
// out[i] = in[0] + ... + in[i]  (32-bit integers; wrap-around like the unsigned sum)
hipError_t inclusive_scan_32(void* tmp, size_t tmp_bytes, const void* in, void* out, size_t n, hipStream_t stream);
// out[i] = in[0] + ... + in[i - 1], out[0] = 0
hipError_t exclusive_scan_32(void* tmp, size_t tmp_bytes, const void* in, void* out, size_t n, hipStream_t stream);

}  // namespace dc
