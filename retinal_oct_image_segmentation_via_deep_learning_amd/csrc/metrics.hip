// Confusion-count reductions behind Metrics/Region_based_metrics.py and
// Metrics/ConfusionMatrix_based_metrics.py: ONE streaming pass over the two masks produces the six
// sums every metric of those files is built from (the reference makes 3-4 passes plus temporaries
// per metric).  HBM-bound: 16-B loads per lane, per-lane 64-bit counters, wave shuffle reduction,
// one atomic per wave and sum.
#include "common.h"

template <typename E, typename Acc>
__global__ void confusion_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t n, Acc* out) {
  constexpr int V = 16 / sizeof(E);
  Acc s[6] = {0, 0, 0, 0, 0, 0};
  const size_t nvec = n / V;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto one = [&](E t, E p) {
    // numpy semantics: products and (1 - y) are evaluated in the input dtype (integers wrap)
    const E nt = (E)((E)1 - t), np_ = (E)((E)1 - p);
    s[0] += (Acc)(E)(t * p); s[1] += (Acc)t; s[2] += (Acc)p;
    s[3] += (Acc)(E)(nt * np_); s[4] += (Acc)(E)(nt * p); s[5] += (Acc)(E)(t * np_);
  };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const VecT<E, V> a = reinterpret_cast<const VecT<E, V>*>(yt)[i];
    const VecT<E, V> b = reinterpret_cast<const VecT<E, V>*>(yp)[i];
#pragma unroll
    for (int j = 0; j < V; ++j) one(a.v[j], b.v[j]);
  }
  for (size_t i = nvec * V + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) one(yt[i], yp[i]);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    Acc v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[k], v);
  }
}

__global__ void zero_counts_kernel(unsigned long long* oi, double* of) {
  if (threadIdx.x < 6) { oi[threadIdx.x] = 0ull; of[threadIdx.x] = 0.0; }
}

extern "C" int oct_confusion_counts(const void* y_true, const void* y_pred, int elem, size_t n, int64_t* out_i,
                                    double* out_f, void* stream) {
  OCT_CHECK(out_i && out_f, "oct_confusion_counts: null output");
  OCT_CHECK(elem >= 0 && elem <= 7, "oct_confusion_counts: bad element type %d", elem);
  OCT_CHECK(n == 0 || (y_true && y_pred), "oct_confusion_counts: null input");
  OCT_CHECK((((uintptr_t)y_true | (uintptr_t)y_pred) & 15) == 0, "oct_confusion_counts: inputs must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  unsigned long long* oi = reinterpret_cast<unsigned long long*>(out_i);
  hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, s, oi, out_f);
  if (n > 0) {
    size_t b = (n + 256 * 64 - 1) / (256 * 64);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    const dim3 g((int)b), t(256);
    typedef unsigned long long u64;
    switch (elem) {
      case 0: hipLaunchKernelGGL((confusion_kernel<uint8_t, u64>), g, t, 0, s, (const uint8_t*)y_true, (const uint8_t*)y_pred, n, oi); break;
      case 1: hipLaunchKernelGGL((confusion_kernel<int32_t, u64>), g, t, 0, s, (const int32_t*)y_true, (const int32_t*)y_pred, n, oi); break;
      case 2: hipLaunchKernelGGL((confusion_kernel<int64_t, u64>), g, t, 0, s, (const int64_t*)y_true, (const int64_t*)y_pred, n, oi); break;
      case 3: hipLaunchKernelGGL((confusion_kernel<float, double>), g, t, 0, s, (const float*)y_true, (const float*)y_pred, n, out_f); break;
      case 4: hipLaunchKernelGGL((confusion_kernel<double, double>), g, t, 0, s, (const double*)y_true, (const double*)y_pred, n, out_f); break;
      case 5: hipLaunchKernelGGL((confusion_kernel<int8_t, u64>), g, t, 0, s, (const int8_t*)y_true, (const int8_t*)y_pred, n, oi); break;
      case 6: hipLaunchKernelGGL((confusion_kernel<int16_t, u64>), g, t, 0, s, (const int16_t*)y_true, (const int16_t*)y_pred, n, oi); break;
      default: hipLaunchKernelGGL((confusion_kernel<uint16_t, u64>), g, t, 0, s, (const uint16_t*)y_true, (const uint16_t*)y_pred, n, oi); break;
    }
  }
  return oct_check_launch("confusion_counts");
}

// ---------------------------------------------------------------------------------------------
// Metrics/PixelError_based_metrics.py:3-37  --  sum of (double(t) - double(p))^2 in one pass
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void sqdiff_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t n, double* out) {
  constexpr int V = 16 / sizeof(E);
  double s = 0.0;
  const size_t nvec = n / V, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const VecT<E, V> a = reinterpret_cast<const VecT<E, V>*>(yt)[i];
    const VecT<E, V> b = reinterpret_cast<const VecT<E, V>*>(yp)[i];
#pragma unroll
    for (int j = 0; j < V; ++j) { const double d = (double)a.v[j] - (double)b.v[j]; s = fma(d, d, s); }
  }
  for (size_t i = nvec * V + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double d = (double)yt[i] - (double)yp[i];
    s = fma(d, d, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// Metrics/Biomarker_based_metrics.py:3-21  --  sum over columns j of |sum_i t[i][j] - sum_i p[i][j]|
// with numpy's dtype semantics: column sums in uint64 (unsigned inputs: the difference WRAPS, exactly
// like the reference), int64 (signed and bool inputs) or the float type.
template <typename E, typename Acc, bool WRAP>
__global__ void column_absdiff_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t rows, size_t cols,
                                      double* out) {
  double s = 0.0;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (size_t)gridDim.x * blockDim.x) {
    Acc st = 0, sp = 0;
    for (size_t i = 0; i < rows; ++i) { st += (Acc)yt[i * cols + j]; sp += (Acc)yp[i * cols + j]; }
    if (WRAP) {
      s += (double)(unsigned long long)((unsigned long long)st - (unsigned long long)sp);
    } else {
      const Acc d = st - sp;
      s += (double)(d < 0 ? -d : d);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

__global__ void zero_f64_kernel(double* p) { if (threadIdx.x == 0) *p = 0.0; }

extern "C" int oct_sqdiff_sum(const void* y_true, const void* y_pred, int elem, size_t n, double* out, void* stream) {
  OCT_CHECK(out, "oct_sqdiff_sum: null output");
  OCT_CHECK(elem >= 0 && elem <= 7, "oct_sqdiff_sum: bad element type %d", elem);
  OCT_CHECK(n == 0 || (y_true && y_pred), "oct_sqdiff_sum: null input");
  OCT_CHECK((((uintptr_t)y_true | (uintptr_t)y_pred) & 15) == 0, "oct_sqdiff_sum: inputs must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(64), 0, s, out);
  if (n > 0) {
    size_t b = (n + 256 * 64 - 1) / (256 * 64);
    if (b > 2048) b = 2048;
    const dim3 g((int)b), t(256);
#define SQ(E) hipLaunchKernelGGL((sqdiff_kernel<E>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, n, out)
    switch (elem) {
      case 0: SQ(uint8_t); break; case 1: SQ(int32_t); break; case 2: SQ(int64_t); break; case 3: SQ(float); break;
      case 4: SQ(double); break; case 5: SQ(int8_t); break; case 6: SQ(int16_t); break; default: SQ(uint16_t); break;
    }
#undef SQ
  }
  return oct_check_launch("sqdiff_sum");
}

extern "C" int oct_column_absdiff_sum(const void* y_true, const void* y_pred, int elem, int unsigned_wrap, size_t rows,
                                      size_t cols, double* out, void* stream) {
  OCT_CHECK(out, "oct_column_absdiff_sum: null output");
  OCT_CHECK(elem >= 0 && elem <= 7, "oct_column_absdiff_sum: bad element type %d", elem);
  OCT_CHECK(rows * cols == 0 || (y_true && y_pred), "oct_column_absdiff_sum: null input");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(64), 0, s, out);
  if (rows * cols > 0) {
    size_t b = (cols + 255) / 256;
    if (b > 2048) b = 2048;
    const dim3 g((int)b), t(256);
    typedef unsigned long long u64;
    typedef long long i64;
#define CA(E, A, W) hipLaunchKernelGGL((column_absdiff_kernel<E, A, W>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, rows, cols, out)
    switch (elem) {
      case 0: if (unsigned_wrap) CA(uint8_t, u64, true); else CA(uint8_t, i64, false); break;
      case 1: CA(int32_t, i64, false); break;
      case 2: CA(int64_t, i64, false); break;
      case 3: CA(float, float, false); break;
      case 4: CA(double, double, false); break;
      case 5: CA(int8_t, i64, false); break;
      case 6: CA(int16_t, i64, false); break;
      default: if (unsigned_wrap) CA(uint16_t, u64, true); else CA(uint16_t, i64, false); break;
    }
#undef CA
  }
  return oct_check_launch("column_absdiff_sum");
}
