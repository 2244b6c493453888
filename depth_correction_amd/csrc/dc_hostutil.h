// Host-side helpers shared by the .hip translation units: workspace carving and HIP status propagation.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

namespace dc {

// Hands out 256-B aligned pieces of a caller-provided workspace; with a null base it only measures.
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  template <typename U> U* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    U* p = base ? (U*)(base + off) : nullptr;
    off += count * sizeof(U);
    return p;
  }
};

}  // namespace dc

#define DC_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return (int)e__; } while (0)
