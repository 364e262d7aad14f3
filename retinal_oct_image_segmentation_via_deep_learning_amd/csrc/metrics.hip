// Confusion-count reductions behind Metrics/Region_based_metrics.py and
// Metrics/ConfusionMatrix_based_metrics.py: ONE streaming pass over the two masks produces the six
// sums every metric of those files is built from (the reference makes 3-4 passes plus temporaries
// per metric).  HBM-bound: 16-B loads per lane, per-lane 64-bit counters, wave shuffle reduction,
// one atomic per wave and sum.
#include "common.h"

// wave -> workgroup -> ONE atomic per sum and workgroup.  Every workgroup of a metric kernel adds to the same few
// words, and same-address atomics serialise at the memory side (~12 ns each): one per wave (8 k waves) cost 0.1-0.4 ms
// on streams that take 10-70 us, so all kernels of this file are launched with <= 512 workgroups of 256 threads and
// reduce through LDS first.
template <int K, typename Acc>
__device__ __forceinline__ void block_sum_atomic(Acc (&s)[K], Acc* out, int valid) {
  __shared__ Acc red[4][K];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    Acc v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[wave][k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < valid; k += blockDim.x) atomicAdd(&out[k], red[0][k] + red[1][k] + red[2][k] + red[3][k]);
}

template <typename E, typename Acc>
__global__ void __launch_bounds__(256) confusion_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t n, Acc* out) {
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
  const VecT<E, V>* vt = reinterpret_cast<const VecT<E, V>*>(yt);
  const VecT<E, V>* vp = reinterpret_cast<const VecT<E, V>*>(yp);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {   // four chunk pairs in flight per lane
    VecT<E, V> a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = vt[i + u * stride]; b[u] = vp[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) one(a[u].v[j], b[u].v[j]);
  }
  for (; i < nvec; i += stride) {
    const VecT<E, V> a = vt[i], b = vp[i];
#pragma unroll
    for (int j = 0; j < V; ++j) one(a.v[j], b.v[j]);
  }
  for (size_t k = nvec * V + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) one(yt[k], yp[k]);
  block_sum_atomic<6, Acc>(s, out, 6);
}

// uint8 / bool masks -- the evaluation case (32 x 512 x 1024 masks = 33.5 MB per pass).  A 16-byte chunk pair whose
// bytes are all 0 or 1 (a binary mask, what a segmentation metric is fed) is reduced with three v_dot4_u32_u8 per
// word -- sum(t*p), sum(t), sum(p) -- and (1-t)(1-p), (1-t)p, t(1-p) follow from those three and the pixel count,
// because no product can wrap; any other chunk takes the general per-element path with numpy's wrap-around
// semantics.  ~1.5 vector instructions per pixel instead of ~25: the kernel becomes HBM-bound (it was issue-bound at
// 0.1 TB/s).  Four chunk pairs per thread are in flight.
__device__ __forceinline__ unsigned dot4_u8(unsigned a, unsigned b, unsigned c) {
#if __has_builtin(__builtin_amdgcn_udot4)
  return __builtin_amdgcn_udot4(a, b, c, false);
#else
  return c + (a & 255u) * (b & 255u) + ((a >> 8) & 255u) * ((b >> 8) & 255u) + ((a >> 16) & 255u) * ((b >> 16) & 255u) + (a >> 24) * (b >> 24);
#endif
}
__global__ void __launch_bounds__(256) confusion_u8_kernel(const uint8_t* __restrict__ yt, const uint8_t* __restrict__ yp, size_t n,
                                                           unsigned long long* out) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  unsigned long long s[6] = {0, 0, 0, 0, 0, 0};   // general chunks, numpy semantics
  unsigned btp = 0, bt = 0, bp = 0, bn = 0;       // binary chunks (a thread sees < 2^32 pixels)
  auto one = [&](uint8_t t, uint8_t p) {
    const uint8_t nt = (uint8_t)(1 - t), np_ = (uint8_t)(1 - p);
    s[0] += (uint8_t)(t * p); s[1] += t; s[2] += p;
    s[3] += (uint8_t)(nt * np_); s[4] += (uint8_t)(nt * p); s[5] += (uint8_t)(t * np_);
  };
  auto chunk = [&](const u32x4 a, const u32x4 b) {
    const unsigned hi = (a[0] | a[1] | a[2] | a[3] | b[0] | b[1] | b[2] | b[3]) & 0xFEFEFEFEu;
    if (hi == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        btp = dot4_u8(a[k], b[k], btp); bt = dot4_u8(a[k], 0x01010101u, bt); bp = dot4_u8(b[k], 0x01010101u, bp);
      }
      bn += 16;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) one((uint8_t)(a[k] >> (8 * j)), (uint8_t)(b[k] >> (8 * j)));
    }
  };
  const size_t nvec = n / 16, stride = (size_t)gridDim.x * blockDim.x;
  const u32x4* vt = reinterpret_cast<const u32x4*>(yt);
  const u32x4* vp = reinterpret_cast<const u32x4*>(yp);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    u32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = __builtin_nontemporal_load(vt + i + u * stride); b[u] = __builtin_nontemporal_load(vp + i + u * stride); }
#pragma unroll
    for (int u = 0; u < 4; ++u) chunk(a[u], b[u]);
  }
  for (; i < nvec; i += stride) chunk(vt[i], vp[i]);
  for (size_t j = nvec * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) one(yt[j], yp[j]);
  // binary chunks: tn = n - t - p + tp, fp = p - tp, fn = t - tp
  s[0] += btp; s[1] += bt; s[2] += bp;
  s[3] += (unsigned long long)bn - bt - bp + btp; s[4] += bp - btp; s[5] += bt - btp;
  block_sum_atomic<6, unsigned long long>(s, out, 6);
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
    if (b > 512) b = 512;   // two workgroups per CU, 3 k same-address atomics in all
    if (b < 1) b = 1;
    const dim3 g((int)b), t(256);
    typedef unsigned long long u64;
    switch (elem) {
      case 0: {   // two workgroups per CU: the streaming sweet spot measured on the BN kernels
        size_t bb = (n / 16 + 256 * 4 - 1) / (256 * 4);
        if (bb > 256) bb = 256;   // one workgroup per CU: 1.5 k same-address atomics in all
        if (bb < 1) bb = 1;
        hipLaunchKernelGGL(confusion_u8_kernel, dim3((int)bb), t, 0, s, (const uint8_t*)y_true, (const uint8_t*)y_pred, n, oi);
        break;
      }
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
// Per-class confusion counts of two CLASS MAPS in one pass: for every class c the six sums of the binary
// masks (t == c), (p == c) -- i.e. what the reference's formulas (Region_based_metrics.py:3-61,
// ConfusionMatrix_based_metrics.py:4-63) see when a caller evaluates class c one-vs-rest, for all classes at
// once instead of one upload + 3-4 passes per class and metric.  Per-lane 32-bit counters [3][16] in
// registers (a lane sees < 2^32 pixels), flushed with a wave + workgroup reduction and one 64-bit atomic per
// workgroup and sum (block_sum_atomic).
// ---------------------------------------------------------------------------------------------
#define OCT_METRIC_MAX_CLASSES 16
// C = the class count rounded up to 4 / 8 / 16: the kernel is issue-bound (~5 vector ops per pixel and class), so an
// 8-class map costs half of a 16-class one.
template <typename E, int C>
__global__ void __launch_bounds__(256) class_confusion_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t n,
                                                              unsigned long long* raw /* [3][16] */) {
  constexpr int V = 16 / sizeof(E), CM = OCT_METRIC_MAX_CLASSES;
  unsigned tp[C], ct[C], cp[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { tp[c] = 0; ct[c] = 0; cp[c] = 0; }
  auto one = [&](E t, E p) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const bool a = t == (E)c, b = p == (E)c;
      ct[c] += a ? 1u : 0u; cp[c] += b ? 1u : 0u; tp[c] += (a && b) ? 1u : 0u;
    }
  };
  const size_t nvec = n / V, stride = (size_t)gridDim.x * blockDim.x;
  const VecT<E, V>* vt = reinterpret_cast<const VecT<E, V>*>(yt);
  const VecT<E, V>* vp = reinterpret_cast<const VecT<E, V>*>(yp);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < nvec; i += 2 * stride) {   // two chunk pairs in flight per lane
    const VecT<E, V> a0 = vt[i], b0 = vp[i], a1 = vt[i + stride], b1 = vp[i + stride];
#pragma unroll
    for (int j = 0; j < V; ++j) one(a0.v[j], b0.v[j]);
#pragma unroll
    for (int j = 0; j < V; ++j) one(a1.v[j], b1.v[j]);
  }
  for (; i < nvec; i += stride) {
    const VecT<E, V> a = vt[i], b = vp[i];
#pragma unroll
    for (int j = 0; j < V; ++j) one(a.v[j], b.v[j]);
  }
  for (size_t k = nvec * V + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) one(yt[k], yp[k]);
  unsigned long long s[C];   // raw layout: [tp | t | p][16]; classes beyond C stay zero
#pragma unroll
  for (int c = 0; c < C; ++c) s[c] = tp[c];
  block_sum_atomic<C, unsigned long long>(s, raw, C);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) s[c] = ct[c];
  block_sum_atomic<C, unsigned long long>(s, raw + CM, C);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) s[c] = cp[c];
  block_sum_atomic<C, unsigned long long>(s, raw + 2 * CM, C);
}
__global__ void class_confusion_zero_kernel(unsigned long long* raw) { if (threadIdx.x < 3 * OCT_METRIC_MAX_CLASSES) raw[threadIdx.x] = 0ull; }
// out[c] = { tp, t, p, tn, fp, fn } of the one-vs-rest masks of class c
__global__ void class_confusion_finish_kernel(const unsigned long long* raw, int classes, unsigned long long n, long long* out) {
  const int c = threadIdx.x;
  if (c >= classes) return;
  const long long tp = (long long)raw[c], t = (long long)raw[OCT_METRIC_MAX_CLASSES + c], p = (long long)raw[2 * OCT_METRIC_MAX_CLASSES + c];
  long long* o = out + 6 * c;
  o[0] = tp; o[1] = t; o[2] = p; o[3] = (long long)n - t - p + tp; o[4] = p - tp; o[5] = t - tp;
}

extern "C" int oct_class_confusion_counts(const void* y_true, const void* y_pred, int elem, size_t n, int classes,
                                          int64_t* out, uint64_t* scratch, void* stream) {
  OCT_CHECK(out && scratch, "oct_class_confusion_counts: null output / scratch");
  OCT_CHECK(classes >= 1 && classes <= OCT_METRIC_MAX_CLASSES, "oct_class_confusion_counts: classes must be 1..%d", OCT_METRIC_MAX_CLASSES);
  OCT_CHECK(elem == 0 || elem == 1 || elem == 2 || elem == 5 || elem == 6 || elem == 7,
            "oct_class_confusion_counts: class maps are integer arrays (element type %d)", elem);
  OCT_CHECK(n == 0 || (y_true && y_pred), "oct_class_confusion_counts: null input");
  OCT_CHECK((((uintptr_t)y_true | (uintptr_t)y_pred) & 15) == 0, "oct_class_confusion_counts: inputs must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  unsigned long long* raw = reinterpret_cast<unsigned long long*>(scratch);
  hipLaunchKernelGGL(class_confusion_zero_kernel, dim3(1), dim3(64), 0, s, raw);
  if (n > 0) {
    size_t b = (n + 256 * 64 - 1) / (256 * 64);
    if (b > 1024) b = 1024;   // issue-bound (~80 vector ops per pixel): four workgroups per CU
    const dim3 g((int)b), t(256);
#define CC(E)                                                                                                              \
  do {                                                                                                                       \
    if (classes <= 4) hipLaunchKernelGGL((class_confusion_kernel<E, 4>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, n, raw);       \
    else if (classes <= 8) hipLaunchKernelGGL((class_confusion_kernel<E, 8>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, n, raw);  \
    else hipLaunchKernelGGL((class_confusion_kernel<E, 16>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, n, raw);        \
  } while (0)
    switch (elem) {
      case 0: CC(uint8_t); break; case 1: CC(int32_t); break; case 2: CC(int64_t); break;
      case 5: CC(int8_t); break; case 6: CC(int16_t); break; default: CC(uint16_t); break;
    }
#undef CC
  }
  hipLaunchKernelGGL(class_confusion_finish_kernel, dim3(1), dim3(64), 0, s, raw, classes, (unsigned long long)n,
                     reinterpret_cast<long long*>(out));
  return oct_check_launch("class_confusion_counts");
}

// ---------------------------------------------------------------------------------------------
// Metrics/PixelError_based_metrics.py:3-37  --  sum of (double(t) - double(p))^2 in one pass
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) sqdiff_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t n, double* out) {
  constexpr int V = 16 / sizeof(E);
  double s[1] = {0.0};
  const size_t nvec = n / V, stride = (size_t)gridDim.x * blockDim.x;
  const VecT<E, V>* vt = reinterpret_cast<const VecT<E, V>*>(yt);
  const VecT<E, V>* vp = reinterpret_cast<const VecT<E, V>*>(yp);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {   // four chunk pairs in flight per lane
    VecT<E, V> a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = vt[i + u * stride]; b[u] = vp[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) { const double d = (double)a[u].v[j] - (double)b[u].v[j]; s[0] = fma(d, d, s[0]); }
  }
  for (; i < nvec; i += stride) {
    const VecT<E, V> a = vt[i], b = vp[i];
#pragma unroll
    for (int j = 0; j < V; ++j) { const double d = (double)a.v[j] - (double)b.v[j]; s[0] = fma(d, d, s[0]); }
  }
  for (size_t k = nvec * V + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const double d = (double)yt[k] - (double)yp[k];
    s[0] = fma(d, d, s[0]);
  }
  block_sum_atomic<1, double>(s, out, 1);
}

// Metrics/Biomarker_based_metrics.py:3-21  --  sum over columns j of |sum_i t[i][j] - sum_i p[i][j]|
// with numpy's dtype semantics: column sums in uint64 (unsigned inputs: the difference WRAPS, exactly
// like the reference), int64 (signed and bool inputs) or the float type.
template <typename E, typename Acc, bool WRAP>
__global__ void __launch_bounds__(256) column_absdiff_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t rows, size_t cols,
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
  double r[1] = {s};
  block_sum_atomic<1, double>(r, out, 1);
}

// the same with 16-byte loads: a lane owns V = 16/sizeof(E) adjacent columns (cols % V == 0, 16-byte aligned bases)
// and keeps four rows in flight; the scalar kernel above reads one element per lane and row (64 B per wave load).
template <typename E, typename Acc, bool WRAP>
__global__ void __launch_bounds__(256) column_absdiff_vec_kernel(const E* __restrict__ yt, const E* __restrict__ yp, size_t rows,
                                                                 size_t cols, double* out) {
  constexpr int V = 16 / sizeof(E);
  const size_t cv = cols / V;
  const VecT<E, V>* vt = reinterpret_cast<const VecT<E, V>*>(yt);
  const VecT<E, V>* vp = reinterpret_cast<const VecT<E, V>*>(yp);
  double s = 0.0;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < cv; j += (size_t)gridDim.x * blockDim.x) {
    Acc st[V], sp[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { st[k] = 0; sp[k] = 0; }
    size_t i = 0;
    for (; i + 4 <= rows; i += 4) {
      VecT<E, V> a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = vt[(i + u) * cv + j]; b[u] = vp[(i + u) * cv + j]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < V; ++k) { st[k] += (Acc)a[u].v[k]; sp[k] += (Acc)b[u].v[k]; }
    }
    for (; i < rows; ++i) {
      const VecT<E, V> a = vt[i * cv + j], b = vp[i * cv + j];
#pragma unroll
      for (int k = 0; k < V; ++k) { st[k] += (Acc)a.v[k]; sp[k] += (Acc)b.v[k]; }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {   // column order within the lane, like the scalar kernel's per-column terms
      if (WRAP) {
        s += (double)(unsigned long long)((unsigned long long)st[k] - (unsigned long long)sp[k]);
      } else {
        const Acc d = st[k] - sp[k];
        s += (double)(d < 0 ? -d : d);
      }
    }
  }
  double r[1] = {s};
  block_sum_atomic<1, double>(r, out, 1);
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
    if (b > 512) b = 512;
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
    typedef unsigned long long u64;
    typedef long long i64;
    static const int esz[8] = {1, 4, 8, 4, 8, 1, 2, 2};
    const size_t v = 16 / esz[elem];
    const bool vec = cols % v == 0 && (((uintptr_t)y_true | (uintptr_t)y_pred) & 15) == 0;
    size_t b = ((vec ? cols / v : cols) + 255) / 256;
    if (b > 512) b = 512;
    const dim3 g((int)b), t(256);
#define CA(E, A, W)                                                                                                              \
  do {                                                                                                                           \
    if (vec) hipLaunchKernelGGL((column_absdiff_vec_kernel<E, A, W>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, rows, cols, out); \
    else hipLaunchKernelGGL((column_absdiff_kernel<E, A, W>), g, t, 0, s, (const E*)y_true, (const E*)y_pred, rows, cols, out);  \
  } while (0)
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
