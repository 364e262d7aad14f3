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
