// Device-side helpers shared by the kernels: typed loads/stores, wave64 / block reductions,
// XCD-aware block remap.  gfx950 only (wavefront = 64 lanes, 8 XCDs).
#pragma once
#include "dc_common.h"

namespace dc {

constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kWave = 64;
constexpr int kXcds = 8;
constexpr int kWavesPerBlock = kBlock / kWave;

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Give every XCD one contiguous
// range of the point array, so the neighbour gathers of spatially sorted points stay in that XCD's L2.
// Returns the logical block index, or -1 for the padding blocks of the last round.
__device__ __forceinline__ int64_t xcd_block(int64_t n_logical) {
  const int64_t per = (n_logical + kXcds - 1) / kXcds;
  const int64_t b = blockIdx.x;
  const int64_t logical = (b % kXcds) * per + b / kXcds;
  return logical < n_logical ? logical : -1;
}
// the same for block index b of a grid whose first blocks do something else (b = blockIdx.x - their number, a multiple of 8)
__device__ __forceinline__ int64_t xcd_block_of(int64_t b, int64_t n_logical) {
  const int64_t per = (n_logical + kXcds - 1) / kXcds;
  const int64_t logical = (b % kXcds) * per + b / kXcds;
  return logical < n_logical ? logical : -1;
}
inline int64_t xcd_grid(int64_t n_logical) { return ((n_logical + kXcds - 1) / kXcds) * kXcds; }

// ---- xyz rows: element stride 3 (API tensors) or 4 (padded internal layout) ------------------------
// Point formats: float, double, or q32 (fixed point, DC_Q32).  All loads return fp64 absolute coordinates.
struct q32 { int32_t v; };
// flag (optional device int): set when a coordinate does not fit the fixed-point range (or is NaN); see quantize()
struct QParams { double origin[3]; double scale; double inv_scale; int32_t* flag; };

template <typename T> struct Scalar { typedef T type; };          // element type of derived float outputs
template <> struct Scalar<q32> { typedef float type; };

template <typename T, int STRIDE> struct Row3;
template <typename T> struct Row3<T, 3> {
  static __device__ __forceinline__ void load(const T* p, int64_t i, double* o, const QParams&) {
    const T* q = p + i * 3;
    o[0] = (double)q[0]; o[1] = (double)q[1]; o[2] = (double)q[2];
  }
  static __device__ __forceinline__ void store(T* p, int64_t i, const double* v, const QParams&) {
    T* q = p + i * 3;
    q[0] = (T)v[0]; q[1] = (T)v[1]; q[2] = (T)v[2];
  }
};
template <> struct Row3<float, 4> {
  static __device__ __forceinline__ void load(const float* p, int64_t i, double* o, const QParams&) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    o[0] = (double)v.x; o[1] = (double)v.y; o[2] = (double)v.z;
  }
  static __device__ __forceinline__ void store(float* p, int64_t i, const double* v, const QParams&) {
    reinterpret_cast<float4*>(p)[i] = make_float4((float)v[0], (float)v[1], (float)v[2], 0.f);
  }
};
template <> struct Row3<double, 4> {
  static __device__ __forceinline__ void load(const double* p, int64_t i, double* o, const QParams&) {
    const double2* q = reinterpret_cast<const double2*>(p) + 2 * i;
    const double2 a = q[0], b = q[1];
    o[0] = a.x; o[1] = a.y; o[2] = b.x;
  }
  static __device__ __forceinline__ void store(double* p, int64_t i, const double* v, const QParams&) {
    double2* q = reinterpret_cast<double2*>(p) + 2 * i;
    q[0] = make_double2(v[0], v[1]);
    q[1] = make_double2(v[2], 0.0);
  }
};
// A coordinate outside the extent the format was sized for (poses moved far beyond the initial map) or a NaN cannot be
// represented: it is stored saturated / as INT32_MIN so that nothing faults, and `flag` is raised -- the evaluation's
// reduction turns the loss into NaN when it is (dc_sequence_eval), so the condition cannot pass unnoticed.
__device__ __forceinline__ int32_t quantize(double x, double origin, double inv_scale, int32_t* flag = nullptr) {
  double q = rint((x - origin) * inv_scale);
  if (!(fabs(q) <= 2147483520.0)) {                                                  // also true for NaN
    if (flag) atomicOr(flag, 1);
    q = q > 2147483520.0 ? 2147483520.0 : (q < -2147483520.0 ? -2147483520.0 : q);   // saturate, NaN -> below
  }
  return (q == q) ? (int32_t)q : (int32_t)0x80000000;
}
template <> struct Row3<q32, 4> {
  static __device__ __forceinline__ void load(const q32* p, int64_t i, double* o, const QParams& qp) {
    const int4 v = reinterpret_cast<const int4*>(p)[i];
    o[0] = qp.origin[0] + (double)v.x * qp.scale;
    o[1] = qp.origin[1] + (double)v.y * qp.scale;
    o[2] = qp.origin[2] + (double)v.z * qp.scale;
  }
  static __device__ __forceinline__ void store(q32* p, int64_t i, const double* v, const QParams& qp) {
    reinterpret_cast<int4*>(p)[i] = make_int4(quantize(v[0], qp.origin[0], qp.inv_scale, qp.flag),
                                              quantize(v[1], qp.origin[1], qp.inv_scale, qp.flag),
                                              quantize(v[2], qp.origin[2], qp.inv_scale, qp.flag), 0);
  }
};

// ---- backward record rows [N,8] = {cmean.xyz, c1, v0.xyz, c2} in the point format ----------------------
template <typename T> struct Rec8 {
  static __device__ __forceinline__ void store(T* rec, int64_t i, const double* m, double c1, const double* v, double c2,
                                               const QParams&) {
    T* r = rec + i * 8;
    r[0] = (T)m[0]; r[1] = (T)m[1]; r[2] = (T)m[2]; r[3] = (T)c1;
    r[4] = (T)v[0]; r[5] = (T)v[1]; r[6] = (T)v[2]; r[7] = (T)c2;
  }
  static __device__ __forceinline__ void load(const T* rec, int64_t i, double* m, double* c1, double* v, double* c2,
                                              const QParams&) {
    const T* r = rec + i * 8;
    m[0] = (double)r[0]; m[1] = (double)r[1]; m[2] = (double)r[2]; *c1 = (double)r[3];
    v[0] = (double)r[4]; v[1] = (double)r[5]; v[2] = (double)r[6]; *c2 = (double)r[7];
  }
};
template <> struct Rec8<float> {
  static __device__ __forceinline__ void store(float* rec, int64_t i, const double* m, double c1, const double* v,
                                               double c2, const QParams&) {
    float4* r = reinterpret_cast<float4*>(rec) + 2 * i;
    r[0] = make_float4((float)m[0], (float)m[1], (float)m[2], (float)c1);
    r[1] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)c2);
  }
  static __device__ __forceinline__ void load(const float* rec, int64_t i, double* m, double* c1, double* v, double* c2,
                                              const QParams&) {
    const float4* r = reinterpret_cast<const float4*>(rec) + 2 * i;
    const float4 a = r[0], b = r[1];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; *c1 = a.w; v[0] = b.x; v[1] = b.y; v[2] = b.z; *c2 = b.w;
  }
};
template <> struct Rec8<q32> {
  static __device__ __forceinline__ void store(q32* rec, int64_t i, const double* m, double c1, const double* v,
                                               double c2, const QParams& qp) {
    int4* r = reinterpret_cast<int4*>(rec) + 2 * i;
    r[0] = make_int4(quantize(m[0], qp.origin[0], qp.inv_scale), quantize(m[1], qp.origin[1], qp.inv_scale),
                     quantize(m[2], qp.origin[2], qp.inv_scale), __float_as_int((float)c1));
    reinterpret_cast<float4*>(r)[1] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)c2);
  }
  static __device__ __forceinline__ void load(const q32* rec, int64_t i, double* m, double* c1, double* v, double* c2,
                                              const QParams& qp) {
    const int4* r = reinterpret_cast<const int4*>(rec) + 2 * i;
    const int4 a = r[0];
    const float4 b = reinterpret_cast<const float4*>(r)[1];
    m[0] = qp.origin[0] + (double)a.x * qp.scale; m[1] = qp.origin[1] + (double)a.y * qp.scale;
    m[2] = qp.origin[2] + (double)a.z * qp.scale; *c1 = (double)__int_as_float(a.w);
    v[0] = b.x; v[1] = b.y; v[2] = b.z; *c2 = b.w;
  }
};

// ---- point-format policy for the hot kernels: raw rows, exact differences, quantised means ---------------
// Differences between a neighbour and its centre are formed on the RAW representation: for q32 an int32
// subtraction (exact) in units of `scale`, for float / double a subtraction of the widened values (exact for
// nearby fp32 points).  Moments are accumulated in those units and scaled once per point.
template <typename PT> struct Pt;
template <typename F> struct PtFloat {
  struct Raw { double v[3]; };
  template <int STRIDE> static __device__ __forceinline__ Raw load(const F* p, int64_t i, const QParams& qp) {
    Raw r; Row3<F, STRIDE>::load(p, i, r.v, qp); return r;
  }
  static __device__ __forceinline__ void delta(const Raw& a, const Raw& c, double* d) {
    d[0] = a.v[0] - c.v[0]; d[1] = a.v[1] - c.v[1]; d[2] = a.v[2] - c.v[2];
  }
  static __device__ __forceinline__ double unit(const QParams&) { return 1.0; }
  // centre + offset (offset in units) as a raw value
  static __device__ __forceinline__ Raw offset(const Raw& c, const double* off) {
    Raw r; r.v[0] = c.v[0] + off[0]; r.v[1] = c.v[1] + off[1]; r.v[2] = c.v[2] + off[2]; return r;
  }
};
template <> struct Pt<float> : PtFloat<float> {
  static constexpr int kRow16 = 1;                       // 16-B units per padded point row
  static __device__ __forceinline__ Raw from_row(const int4* row) {
    const int4 b = row[0];
    Raw r; r.v[0] = (double)__int_as_float(b.x); r.v[1] = (double)__int_as_float(b.y); r.v[2] = (double)__int_as_float(b.z);
    return r;
  }
};
template <> struct Pt<double> : PtFloat<double> {
  static constexpr int kRow16 = 2;
  static __device__ __forceinline__ Raw from_row(const int4* row) {
    const int4 a = row[0], b = row[1];
    Raw r;
    r.v[0] = __hiloint2double(a.y, a.x); r.v[1] = __hiloint2double(a.w, a.z); r.v[2] = __hiloint2double(b.y, b.x);
    return r;
  }
};
template <> struct Pt<q32> {
  struct Raw { int32_t v[3]; };
  static constexpr int kRow16 = 1;
  static __device__ __forceinline__ Raw from_row(const int4* row) {
    const int4 q = row[0];
    Raw r; r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; return r;
  }
  template <int STRIDE> static __device__ __forceinline__ Raw load(const q32* p, int64_t i, const QParams&) {
    const int4 q = reinterpret_cast<const int4*>(p)[i];
    Raw r; r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; return r;
  }
  static __device__ __forceinline__ void delta(const Raw& a, const Raw& c, double* d) {
    d[0] = (double)(a.v[0] - c.v[0]); d[1] = (double)(a.v[1] - c.v[1]); d[2] = (double)(a.v[2] - c.v[2]);
  }
  static __device__ __forceinline__ double unit(const QParams& qp) { return qp.scale; }
  // centre + offset, offset in grid steps.  The offset is a neighbourhood mean relative to its centre (|off| << 2^31);
  // v_cvt_i32_f64 saturates and maps NaN (empty neighbourhood: the record's coefficients are zero then) to 0.
  static __device__ __forceinline__ Raw offset(const Raw& c, const double* off) {
    Raw r;
#pragma unroll
    for (int a = 0; a < 3; ++a) r.v[a] = c.v[a] + (int32_t)rint(off[a]);
    return r;
  }
};

// ---- buffer-resource gathers: 32-bit byte offsets + hardware range check (an out-of-range row, e.g. index -1, reads 0)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ int4 buf_load16(BufRsrc r, uint32_t byte_off) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_int4((int)v[0], (int)v[1], (int)v[2], (int)v[3]);
}
template <typename PT> struct RowBytes { static constexpr uint32_t x = 16, rec = 32; };
template <> struct RowBytes<double> { static constexpr uint32_t x = 32, rec = 64; };

// rec rows from / to raw means
template <typename PT> struct RecRaw;
template <typename F> struct RecRawFloat {
  static __device__ __forceinline__ void store(F* rec, int64_t i, const typename Pt<F>::Raw& m, double c1, const double* v,
                                               double c2) {
    Rec8<F>::store(rec, i, m.v, c1, v, c2, QParams{});
  }
  static __device__ __forceinline__ void load(const F* rec, int64_t i, typename Pt<F>::Raw& m, double* c1, double* v, double* c2) {
    Rec8<F>::load(rec, i, m.v, c1, v, c2, QParams{});
  }
};
template <> struct RecRaw<float> : RecRawFloat<float> {
  static constexpr int kRow16 = 2;                       // 16-B units per record row
  static __device__ __forceinline__ void from_row(const int4* row, Pt<float>::Raw& m, double* c1, double* v, double* c2) {
    const int4 a = row[0], b = row[1];
    m.v[0] = (double)__int_as_float(a.x); m.v[1] = (double)__int_as_float(a.y); m.v[2] = (double)__int_as_float(a.z);
    *c1 = (double)__int_as_float(a.w);
    v[0] = (double)__int_as_float(b.x); v[1] = (double)__int_as_float(b.y); v[2] = (double)__int_as_float(b.z);
    *c2 = (double)__int_as_float(b.w);
  }
};
template <> struct RecRaw<double> : RecRawFloat<double> {
  static constexpr int kRow16 = 4;
  static __device__ __forceinline__ void from_row(const int4* row, Pt<double>::Raw& m, double* c1, double* v, double* c2) {
    const int4 a = row[0], b = row[1], c = row[2], d = row[3];
    m.v[0] = __hiloint2double(a.y, a.x); m.v[1] = __hiloint2double(a.w, a.z); m.v[2] = __hiloint2double(b.y, b.x);
    *c1 = __hiloint2double(b.w, b.z);
    v[0] = __hiloint2double(c.y, c.x); v[1] = __hiloint2double(c.w, c.z); v[2] = __hiloint2double(d.y, d.x);
    *c2 = __hiloint2double(d.w, d.z);
  }
};
template <> struct RecRaw<q32> {
  static constexpr int kRow16 = 2;
  static __device__ __forceinline__ void from_row(const int4* row, Pt<q32>::Raw& m, double* c1, double* v, double* c2) {
    const int4 a = row[0], b = row[1];
    m.v[0] = a.x; m.v[1] = a.y; m.v[2] = a.z; *c1 = (double)__int_as_float(a.w);
    v[0] = (double)__int_as_float(b.x); v[1] = (double)__int_as_float(b.y); v[2] = (double)__int_as_float(b.z);
    *c2 = (double)__int_as_float(b.w);
  }
  static __device__ __forceinline__ void store(q32* rec, int64_t i, const Pt<q32>::Raw& m, double c1, const double* v, double c2) {
    int4* r = reinterpret_cast<int4*>(rec) + 2 * i;
    r[0] = make_int4(m.v[0], m.v[1], m.v[2], __float_as_int((float)c1));
    reinterpret_cast<float4*>(r)[1] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)c2);
  }
  static __device__ __forceinline__ void load(const q32* rec, int64_t i, Pt<q32>::Raw& m, double* c1, double* v, double* c2) {
    const int4* r = reinterpret_cast<const int4*>(rec) + 2 * i;
    const int4 a = r[0];
    const float4 b = reinterpret_cast<const float4*>(r)[1];
    m.v[0] = a.x; m.v[1] = a.y; m.v[2] = a.z; *c1 = (double)__int_as_float(a.w);
    v[0] = b.x; v[1] = b.y; v[2] = b.z; *c2 = b.w;
  }
};

// ---- reductions ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Partial sums without LDS or barriers: lane 0 of every wavefront writes its sums to row 4 * block + wave of the
// accumulator-major workspace partials[NV][4 * gridDim.x]; the fixed-order final reduction reads the rows.
template <int NV>
__device__ __forceinline__ void wave_partials(const double* v, double* __restrict__ partials) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t rs = (int64_t)gridDim.x * kWavesPerBlock, row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const double s = wave_sum(v[q]);
    if (lane == 0) partials[q * rs + row] = s;
  }
}

// NV <= 8 sums per wavefront with fewer exchange steps than NV independent butterflies: the first log2(NP) steps also
// halve the number of values a lane carries (the partner takes over the other half), the remaining steps finish one value
// per lane.  NP + 5 - log2(NP)... exchanges instead of 6 NV (4 values: 7 instead of 24).  Afterwards lane l < NP holds the
// total of value bitrev(l); `value_of_lane` returns that index.  Fixed order => bitwise reproducible.
template <int NP> __device__ __forceinline__ int packed_value_of_lane(int lane) {
  int q = 0;
#pragma unroll
  for (int b = 1, r = NP >> 1; b < NP; b <<= 1, r >>= 1) q |= (lane & b) ? r : 0;
  return q;
}
template <int NP>
__device__ __forceinline__ double wave_sum_packed(double* v) {      // v[NP], NP a power of two <= 8; clobbers v
  const int lane = threadIdx.x & (kWave - 1);
  int width = NP;
#pragma unroll
  for (int bit = 1; bit < NP; bit <<= 1) {
    const bool up = (lane & bit) != 0;
    width >>= 1;
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) {
      if (k < width) {
        // this lane keeps the lower half's slot k if its bit is clear, the upper half's otherwise; the partner the other
        const double keep = up ? v[k + width] : v[k];
        const double give = up ? v[k] : v[k + width];
        v[k] = keep + __shfl_xor(give, bit, kWave);
      }
    }
  }
  double r = v[0];
#pragma unroll
  for (int off = NP; off < kWave; off <<= 1) r += __shfl_xor(r, off, kWave);
  return r;
}

// Sum `NV` per-thread doubles over the block; thread 0 receives the totals in `v`.
// `lds` must hold (kBlock / kWave) * NV doubles.  Fixed order => bitwise reproducible.
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* lds) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double s = wave_sum(v[q]);
    if (lane == 0) lds[wave * NV + q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double s = 0.0;
      for (int wv = 0; wv < kBlock / kWave; ++wv) s += lds[wv * NV + q];
      v[q] = s;
    }
  }
  __syncthreads();
}

// Same, for the first `n_used` (block-uniform) of NV values only: skips the shuffles of unused slots.
template <int NV>
__device__ __forceinline__ void block_sum_used(double* v, int n_used, double* lds) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (q < n_used) {
      double s = wave_sum(v[q]);
      if (lane == 0) lds[wave * NV + q] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if (q < n_used) {
        double s = 0.0;
        for (int wv = 0; wv < kBlock / kWave; ++wv) s += lds[wv * NV + q];
        v[q] = s;
      }
    }
  }
  __syncthreads();
}

// point / record rows through a buffer resource (row index may be -1: all-zero row)
template <typename PT>
__device__ __forceinline__ typename Pt<PT>::Raw buf_point(BufRsrc r, int32_t row) {
  int4 q[Pt<PT>::kRow16];
#pragma unroll
  for (int a = 0; a < Pt<PT>::kRow16; ++a) q[a] = buf_load16(r, (uint32_t)row * RowBytes<PT>::x + 16u * a);
  return Pt<PT>::from_row(q);
}
template <typename PT>
__device__ __forceinline__ void buf_record(BufRsrc r, int32_t row, typename Pt<PT>::Raw& m, double* c1, double* v, double* c2) {
  int4 q[RecRaw<PT>::kRow16];
#pragma unroll
  for (int a = 0; a < RecRaw<PT>::kRow16; ++a) q[a] = buf_load16(r, (uint32_t)row * RowBytes<PT>::rec + 16u * a);
  RecRaw<PT>::from_row(q, m, c1, v, c2);
}

}  // namespace dc
