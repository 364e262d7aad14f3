// Element-wise building blocks of the reference's other U-Net families:
// residual blocks (SD_Layer_Net/common.py:6-25), bilinear x-s up-sampling with align_corners=True
// (common.py:31, MGUNet_2021.py:79,98), k x k max-pooling on materialised activations, the k4s4
// transposed convolution's depth-to-space step (MGUNet_2021.py:95) and the attention gate's
// broadcast product (common.py:85-91).  All are HBM-bound streaming kernels over NHWC tensors: a
// thread owns one 8-channel vector (16 B of bf16) when c % 8 == 0, one element otherwise.
#include "common.h"

#define BK_THREADS 256
#define BK_MAX_BLOCKS 4096

static inline int bk_vec(int c) { return (c % 8 == 0) ? 8 : 1; }
static inline int bk_blocks(size_t work) {
  size_t b = (work + BK_THREADS - 1) / BK_THREADS;
  if (b > BK_MAX_BLOCKS) b = BK_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
#define BK_DISPATCH(NAME, ...)                                                                 \
  do {                                                                                         \
    if (dtype == OCT_DT_BF16) { if (v == 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 1); }       \
    else if (dtype == OCT_DT_F32) { if (v == 8) LAUNCH(float, 8); else LAUNCH(float, 1); }     \
    else OCT_CHECK(false, NAME ": bad dtype");                                                 \
  } while (0)

__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == OCT_ACT_RELU) return fmaxf(z, 0.f);
  if (act == OCT_ACT_SIGMOID) return 1.f / (1.f + __expf(-z));
  return z;
}

// ---------------------------------------------------------------------------------------------
// out = act(scale[c] * y + shift[c] (+ res))
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void affine_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const T* __restrict__ res,
                                      const float* __restrict__ res_shift, int act, T* __restrict__ out, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; const size_t off = (i / G) * c + g * V;
    float sc[V], sh[V], v[V], r[V];
    load_vec<float, V>(scale + g * V, sc); load_vec<float, V>(shift + g * V, sh);
    load_vec<T, V>(y + off, v);
    if (res) {
      load_vec<T, V>(res + off, r);
      if (res_shift) {   // the residual's deferred bias: (r + b) in the storage type first, as if it had been materialised
        float rb[V];
        load_vec<float, V>(res_shift + g * V, rb);
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = to_f32(from_f32<T>(r[j] + rb[j]));
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = act_apply(fmaf(v[j], sc[j], sh[j]) + (res ? r[j] : 0.f), act);
    store_vec<T, V>(out + off, v);
  }
}
extern "C" int oct_affine_res_act_fwd(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                                      const float* res_shift, int act, void* out, size_t npix, int c, void* stream) {
  OCT_CHECK(y && scale && shift && out && npix > 0 && c > 0, "oct_affine_act_fwd: bad args");
  OCT_CHECK(res || !res_shift, "oct_affine_act_fwd: res_shift without a residual");
  OCT_CHECK(act >= OCT_ACT_NONE && act <= OCT_ACT_SIGMOID, "oct_affine_act_fwd: bad activation %d", act);
  const int v = bk_vec(c);
  const int blocks = bk_blocks(npix * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((affine_act_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, \
                                        (const T*)y, scale, shift, (const T*)res, res_shift, act, (T*)out, npix, c)
  BK_DISPATCH("oct_affine_act_fwd");
#undef LAUNCH
  return oct_check_launch("affine_act_fwd");
}
extern "C" int oct_affine_act_fwd(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                                  int act, void* out, size_t npix, int c, void* stream) {
  return oct_affine_res_act_fwd(dtype, y, scale, shift, res, nullptr, act, out, npix, c, stream);
}

// dz = dout * act'(out) expressed through the stored output: relu -> [out > 0], sigmoid -> out (1 - out)
template <typename T, int V>
__global__ void act_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, int act, T* __restrict__ dz,
                               size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float d[V], o[V];
    load_vec<T, V>(dout + i * V, d); load_vec<T, V>(out + i * V, o);
#pragma unroll
    for (int j = 0; j < V; ++j) d[j] = act == OCT_ACT_RELU ? (o[j] > 0.f ? d[j] : 0.f) : d[j] * o[j] * (1.f - o[j]);
    store_vec<T, V>(dz + i * V, d);
  }
}
extern "C" int oct_act_bwd(int dtype, const void* dout, const void* out, int act, void* dz, size_t n, void* stream) {
  OCT_CHECK(dout && out && dz && n > 0, "oct_act_bwd: bad args");
  OCT_CHECK(act == OCT_ACT_RELU || act == OCT_ACT_SIGMOID, "oct_act_bwd: bad activation %d", act);
  const int v = (n % 8 == 0) ? 8 : 1;
  const int blocks = bk_blocks(n / v);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((act_bwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)dout, \
                                        (const T*)out, act, (T*)dz, n / V)
  BK_DISPATCH("oct_act_bwd");
#undef LAUNCH
  return oct_check_launch("act_bwd");
}

// ---------------------------------------------------------------------------------------------
// k x k / stride k max-pooling of a materialised activation (torch.nn.MaxPool2d(k)); backward
// routes the gradient to the first maximum in row-major window order, like torch.
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void maxpool_fwd_kernel(const T* __restrict__ a, T* __restrict__ out, int n, int h, int w, int c, int k) {
  const int G = c / V;
  const int ho = h / k, wo = w / k;   // torch's floor mode: trailing rows / columns that no whole window covers are ignored
  const size_t total = (size_t)n * ho * wo * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    float m[V];
#pragma unroll
    for (int j = 0; j < V; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        float v[V];
        load_vec<T, V>(a + (((size_t)img * h + yo * k + dy) * w + xo * k + dx) * c + g * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    store_vec<T, V>(out + (((size_t)img * ho + yo) * wo + xo) * c + g * V, m);
  }
}
template <typename T, int V>
__global__ void maxpool_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dout, T* __restrict__ da, int n,
                                   int h, int w, int c, int k) {
  const int G = c / V;
  const int ho = h / k, wo = w / k;
  const size_t total = (size_t)n * ho * wo * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    float m[V], d[V]; int arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { m[j] = -INFINITY; arg[j] = 0; }
    for (int q = 0; q < k * k; ++q) {
      float v[V];
      load_vec<T, V>(a + (((size_t)img * h + yo * k + q / k) * w + xo * k + q % k) * c + g * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) if (v[j] > m[j]) { m[j] = v[j]; arg[j] = q; }
    }
    load_vec<T, V>(dout + (((size_t)img * ho + yo) * wo + xo) * c + g * V, d);
    for (int q = 0; q < k * k; ++q) {
      float v[V];
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = arg[j] == q ? d[j] : 0.f;
      store_vec<T, V>(da + (((size_t)img * h + yo * k + q / k) * w + xo * k + q % k) * c + g * V, v);
    }
  }
}
extern "C" int oct_maxpool_fwd(int dtype, const void* a, void* out, int n, int h, int w, int c, int k, void* stream) {
  OCT_CHECK(a && out && n > 0 && h > 0 && w > 0 && c > 0 && k >= 1, "oct_maxpool_fwd: bad args");
  OCT_CHECK(h >= k && w >= k, "oct_maxpool_fwd: %dx%d is smaller than the window %d", h, w, k);
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * (h / k) * (w / k) * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((maxpool_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)a, \
                                        (T*)out, n, h, w, c, k)
  BK_DISPATCH("oct_maxpool_fwd");
#undef LAUNCH
  return oct_check_launch("maxpool_fwd");
}
extern "C" int oct_maxpool_bwd(int dtype, const void* a, const void* dout, void* da, int n, int h, int w, int c, int k,
                               void* stream) {
  OCT_CHECK(a && dout && da && n > 0 && h > 0 && w > 0 && c > 0 && k >= 1, "oct_maxpool_bwd: bad args");
  OCT_CHECK(h >= k && w >= k, "oct_maxpool_bwd: %dx%d is smaller than the window %d", h, w, k);
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * (h / k) * (w / k) * (c / v));
  hipStream_t s = (hipStream_t)stream;
  if (h % k || w % k) {   // floor mode: the rows / columns outside every window get no gradient
    const size_t esz = dtype == OCT_DT_BF16 ? 2 : 4;
    if (hipMemsetAsync(da, 0, (size_t)n * h * w * c * esz, s) != hipSuccess) { oct_set_error("oct_maxpool_bwd: memset failed"); return OCT_E_LAUNCH; }
  }
#define LAUNCH(T, V) hipLaunchKernelGGL((maxpool_bwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)a, \
                                        (const T*)dout, (T*)da, n, h, w, c, k)
  BK_DISPATCH("oct_maxpool_bwd");
#undef LAUNCH
  return oct_check_launch("maxpool_bwd");
}

// ---------------------------------------------------------------------------------------------
// bilinear up-sampling by an integer factor, align_corners=True:  src = dst * (in-1)/(out-1),
// i0 = floor(src), lambda = src - i0, i1 = min(i0+1, in-1)  (torch upsample_bilinear2d).
// Backward is the transposed gather: an input pixel collects every output whose two taps touch it,
// recomputing the forward's own (i0, lambda) so both directions agree to the last bit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(int o, float r, int in, int& i0, int& i1, float& l1) {
  const float s = r * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}
template <typename T, int V>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int n, int h, int w, int c, int ho, int wo,
                                    float ry, float rx) {
  const int G = c / V;
  const size_t total = (size_t)n * ho * wo * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    int y0, y1, x0, x1; float ly, lx;
    bil_src(yo, ry, h, y0, y1, ly); bil_src(xo, rx, w, x0, x1, lx);
    float a[V], b[V], cc[V], d[V];
    const T* base = x + (size_t)img * h * w * c + g * V;
    load_vec<T, V>(base + ((size_t)y0 * w + x0) * c, a); load_vec<T, V>(base + ((size_t)y0 * w + x1) * c, b);
    load_vec<T, V>(base + ((size_t)y1 * w + x0) * c, cc); load_vec<T, V>(base + ((size_t)y1 * w + x1) * c, d);
#pragma unroll
    for (int j = 0; j < V; ++j)
      a[j] = (1.f - ly) * ((1.f - lx) * a[j] + lx * b[j]) + ly * ((1.f - lx) * cc[j] + lx * d[j]);
    store_vec<T, V>(out + (((size_t)img * ho + yo) * wo + xo) * c + g * V, a);
  }
}
template <typename T, int V>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int n, int h, int w, int c, int ho, int wo,
                                    float ry, float rx) {
  const int G = c / V;
  const size_t total = (size_t)n * h * w * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xi = p % w; p /= w; const int yi = p % h; const int img = p / h;
    // outputs that can touch input row yi have src = r*yo in (yi-1, yi+1); one row of slack either
    // side covers the rounding of the division, the exact test below uses the forward's own taps
    // (ratio 0: a one-pixel input or output axis -- every output reads input 0, scan them all)
    const int ylo = ry > 0.f ? max(0, (int)floorf((float)(yi - 1) / ry) - 1) : 0;
    const int yhi = ry > 0.f ? min(ho - 1, (int)ceilf((float)(yi + 1) / ry) + 1) : ho - 1;
    const int xlo = rx > 0.f ? max(0, (int)floorf((float)(xi - 1) / rx) - 1) : 0;
    const int xhi = rx > 0.f ? min(wo - 1, (int)ceilf((float)(xi + 1) / rx) + 1) : wo - 1;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      int y0, y1; float ly;
      bil_src(yo, ry, h, y0, y1, ly);
      const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
      if (wy == 0.f && !(y0 == yi || y1 == yi)) continue;
      for (int xo = xlo; xo <= xhi; ++xo) {
        int x0, x1; float lx;
        bil_src(xo, rx, w, x0, x1, lx);
        if (!(x0 == xi || x1 == xi)) continue;
        const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
        float d[V];
        load_vec<T, V>(dout + (((size_t)img * ho + yo) * wo + xo) * c + g * V, d);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(wy * wx, d[j], acc[j]);
      }
    }
    store_vec<T, V>(dx + (((size_t)img * h + yi) * w + xi) * c + g * V, acc);
  }
}
static inline float bil_ratio(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
extern "C" int oct_bilinear_resize_fwd(int dtype, const void* x, void* out, int n, int h, int w, int c, int ho, int wo,
                                       void* stream) {
  OCT_CHECK(x && out && n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0, "oct_bilinear_resize_fwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * ho * wo * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((bilinear_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)x, \
                                        (T*)out, n, h, w, c, ho, wo, bil_ratio(h, ho), bil_ratio(w, wo))
  BK_DISPATCH("oct_bilinear_resize_fwd");
#undef LAUNCH
  return oct_check_launch("bilinear_resize_fwd");
}
extern "C" int oct_bilinear_resize_bwd(int dtype, const void* dout, void* dx, int n, int h, int w, int c, int ho, int wo,
                                       void* stream) {
  OCT_CHECK(dout && dx && n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0, "oct_bilinear_resize_bwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * h * w * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((bilinear_bwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)dout, \
                                        (T*)dx, n, h, w, c, ho, wo, bil_ratio(h, ho), bil_ratio(w, wo))
  BK_DISPATCH("oct_bilinear_resize_bwd");
#undef LAUNCH
  return oct_check_launch("bilinear_resize_bwd");
}
extern "C" int oct_bilinear_up_fwd(int dtype, const void* x, void* out, int n, int h, int w, int c, int factor,
                                   void* stream) {
  OCT_CHECK(factor >= 1, "oct_bilinear_up_fwd: bad args");
  return oct_bilinear_resize_fwd(dtype, x, out, n, h, w, c, h * factor, w * factor, stream);
}
extern "C" int oct_bilinear_up_bwd(int dtype, const void* dout, void* dx, int n, int h, int w, int c, int factor,
                                   void* stream) {
  OCT_CHECK(factor >= 1, "oct_bilinear_up_bwd: bad args");
  return oct_bilinear_resize_bwd(dtype, dout, dx, n, h, w, c, h * factor, w * factor, stream);
}

// ---------------------------------------------------------------------------------------------
// depth-to-space / space-to-depth with block s: in[n,h,w,(dy*s+dx)*cout+co] <-> out[n,h*s+dy,w*s+dx,co]
// (the scatter step of ConvTranspose2d(kernel=s, stride=s), bias added on the way out)
// ---------------------------------------------------------------------------------------------
template <typename T, int V, bool TO_SPACE>
__global__ void d2s_kernel(const T* __restrict__ in, const float* __restrict__ bias, T* __restrict__ out, int n, int h,
                           int w, int cout, int s) {
  const int G = cout / V;
  const size_t total = (size_t)n * h * s * w * s * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % (w * s); p /= (w * s); const int yo = p % (h * s); const int img = p / (h * s);
    const size_t deep = ((((size_t)img * h + yo / s) * w + xo / s) * s * s + (yo % s) * s + xo % s) * cout + g * V;
    const size_t wide = (((size_t)img * h * s + yo) * w * s + xo) * cout + g * V;
    float v[V];
    if (TO_SPACE) {
      load_vec<T, V>(in + deep, v);
      if (bias) {
        float b[V];
        load_vec<float, V>(bias + g * V, b);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += b[j];
      }
      store_vec<T, V>(out + wide, v);
    } else {
      load_vec<T, V>(in + wide, v);
      store_vec<T, V>(out + deep, v);
    }
  }
}
extern "C" int oct_depth_to_space(int dtype, const void* in, const float* bias, void* out, int n, int h, int w, int cout,
                                  int s, void* stream) {
  OCT_CHECK(in && out && n > 0 && h > 0 && w > 0 && cout > 0 && s >= 1, "oct_depth_to_space: bad args");
  const int v = bk_vec(cout);
  const int blocks = bk_blocks((size_t)n * h * s * w * s * (cout / v));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((d2s_kernel<T, V, true>), dim3(blocks), dim3(BK_THREADS), 0, st, (const T*)in, \
                                        bias, (T*)out, n, h, w, cout, s)
  BK_DISPATCH("oct_depth_to_space");
#undef LAUNCH
  return oct_check_launch("depth_to_space");
}
extern "C" int oct_space_to_depth(int dtype, const void* in, void* out, int n, int h, int w, int cout, int s,
                                  void* stream) {
  OCT_CHECK(in && out && n > 0 && h > 0 && w > 0 && cout > 0 && s >= 1, "oct_space_to_depth: bad args");
  const int v = bk_vec(cout);
  const int blocks = bk_blocks((size_t)n * h * s * w * s * (cout / v));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((d2s_kernel<T, V, false>), dim3(blocks), dim3(BK_THREADS), 0, st, (const T*)in, \
                                        (const float*)nullptr, (T*)out, n, h, w, cout, s)
  BK_DISPATCH("oct_space_to_depth");
#undef LAUNCH
  return oct_check_launch("space_to_depth");
}

// ---------------------------------------------------------------------------------------------
// attention gate product  out[pix, c] = x[pix, c] * p[pix]   (SD_Layer_Net/common.py:91)
// backward: dx = dout * p,  dp[pix] = sum_c dout * x  -- one wave-level reduction per pixel
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void gate_fwd_kernel(const T* __restrict__ x, const T* __restrict__ p, T* __restrict__ out, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / G; const size_t off = pix * c + (i % G) * V;
    float v[V];
    load_vec<T, V>(x + off, v);
    const float pv = to_f32(p[pix]);
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] *= pv;
    store_vec<T, V>(out + off, v);
  }
}
// backward: a pixel's c channels are spread over LPP = c/8 consecutive lanes (8-channel vectors, 16-B
// accesses contiguous across lanes); dp[pix] = sum_c dout*x is a segmented shuffle reduction over those
// lanes.  Requires c % 8 == 0 and c/8 a power of two <= 64; other channel counts take the scalar kernel.
template <typename T>
__global__ void gate_bwd_vec_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ p,
                                    T* __restrict__ dx, T* __restrict__ dp, size_t npix, int c) {
  const int lpp = c >> 3;
  const size_t total = npix * lpp;
  // the grid-stride keeps whole pixels inside one wave: blockDim (256) is a multiple of lpp
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((total + 63) / 64) * 64; i += (size_t)gridDim.x * blockDim.x) {
    const bool live = i < total;
    const size_t pix = live ? i / lpp : 0;
    const size_t off = pix * c + (live ? (i % lpp) * 8 : 0);
    float d[8], xv[8];
    load_vec<T, 8>(dout + off, d);
    load_vec<T, 8>(x + off, xv);
    const float pv = to_f32(p[pix]);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc = fmaf(d[j], xv[j], acc); d[j] *= pv; }
    if (!live) acc = 0.f;
    for (int o = lpp >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (live) {
      store_vec<T, 8>(dx + off, d);
      if ((i % lpp) == 0) dp[pix] = from_f32<T>(acc);
    }
  }
}
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ p,
                                T* __restrict__ dx, T* __restrict__ dp, size_t npix, int c) {
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const float pv = to_f32(p[pix]);
    float acc = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      const float d = to_f32(dout[pix * c + ch]);
      acc = fmaf(d, to_f32(x[pix * c + ch]), acc);
      dx[pix * c + ch] = from_f32<T>(d * pv);
    }
    dp[pix] = from_f32<T>(acc);
  }
}
extern "C" int oct_gate_fwd(int dtype, const void* x, const void* p, void* out, size_t npix, int c, void* stream) {
  OCT_CHECK(x && p && out && npix > 0 && c > 0, "oct_gate_fwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks(npix * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((gate_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)x, \
                                        (const T*)p, (T*)out, npix, c)
  BK_DISPATCH("oct_gate_fwd");
#undef LAUNCH
  return oct_check_launch("gate_fwd");
}
extern "C" int oct_gate_bwd(int dtype, const void* dout, const void* x, const void* p, void* dx, void* dp, size_t npix,
                            int c, void* stream) {
  OCT_CHECK(dout && x && p && dx && dp && npix > 0 && c > 0, "oct_gate_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int lpp = c >> 3;
  if ((c & 7) == 0 && lpp <= 64 && (lpp & (lpp - 1)) == 0) {
    const int vb = bk_blocks(npix * lpp);
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL(gate_bwd_vec_kernel<bf16_t>, dim3(vb), dim3(BK_THREADS), 0, s, (const bf16_t*)dout,
                         (const bf16_t*)x, (const bf16_t*)p, (bf16_t*)dx, (bf16_t*)dp, npix, c);
    else if (dtype == OCT_DT_F32)
      hipLaunchKernelGGL(gate_bwd_vec_kernel<float>, dim3(vb), dim3(BK_THREADS), 0, s, (const float*)dout,
                         (const float*)x, (const float*)p, (float*)dx, (float*)dp, npix, c);
    else
      OCT_CHECK(false, "oct_gate_bwd: bad dtype");
    return oct_check_launch("gate_bwd");
  }
  const int blocks = bk_blocks(npix);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL(gate_bwd_kernel<bf16_t>, dim3(blocks), dim3(BK_THREADS), 0, s, (const bf16_t*)dout,
                       (const bf16_t*)x, (const bf16_t*)p, (bf16_t*)dx, (bf16_t*)dp, npix, c);
  else if (dtype == OCT_DT_F32)
    hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(blocks), dim3(BK_THREADS), 0, s, (const float*)dout,
                       (const float*)x, (const float*)p, (float*)dx, (float*)dp, npix, c);
  else
    OCT_CHECK(false, "oct_gate_bwd: bad dtype");
  return oct_check_launch("gate_bwd");
}


// ---------------------------------------------------------------------------------------------
// ReLayNet_2017.py:133-201 building blocks: BatchNorm + PReLU (one learnable slope, nn.PReLU() default),
// MaxPool2d(return_indices=True) and MaxUnpool2d.
// ---------------------------------------------------------------------------------------------
// out = z > 0 ? z : alpha*z with z = y*scale[c] + shift[c]   (BasicBlock.forward :164-168)
template <typename T, int V>
__global__ void affine_prelu_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ alpha, T* __restrict__ out, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  const float al = alpha[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; const size_t off = (i / G) * c + g * V;
    float sc[V], sh[V], v[V];
    load_vec<float, V>(scale + g * V, sc); load_vec<float, V>(shift + g * V, sh);
    load_vec<T, V>(y + off, v);
#pragma unroll
    for (int j = 0; j < V; ++j) { const float z = fmaf(v[j], sc[j], sh[j]); v[j] = z > 0.f ? z : al * z; }
    store_vec<T, V>(out + off, v);
  }
}
extern "C" int oct_affine_prelu_fwd(int dtype, const void* y, const float* scale, const float* shift, const float* alpha,
                                    void* out, size_t npix, int c, void* stream) {
  OCT_CHECK(y && scale && shift && alpha && out && npix > 0 && c > 0, "oct_affine_prelu_fwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks(npix * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((affine_prelu_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)y, \
                                        scale, shift, alpha, (T*)out, npix, c)
  BK_DISPATCH("oct_affine_prelu_fwd");
#undef LAUNCH
  return oct_check_launch("affine_prelu_fwd");
}
// dz = dout * (z > 0 ? 1 : alpha); dalpha += sum dout * z * [z <= 0]   (ATen's prelu backward: the slope branch at z == 0);
// z is recomputed from the stored raw conv output.  dalpha: one fp32 atomic per wave.
template <typename T, int V>
__global__ void affine_prelu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ y, const float* __restrict__ scale,
                                        const float* __restrict__ shift, const float* __restrict__ alpha, T* __restrict__ dz,
                                        float* __restrict__ dalpha, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  const float al = alpha[0];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; const size_t off = (i / G) * c + g * V;
    float sc[V], sh[V], v[V], d[V];
    load_vec<float, V>(scale + g * V, sc); load_vec<float, V>(shift + g * V, sh);
    load_vec<T, V>(y + off, v); load_vec<T, V>(dout + off, d);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float z = fmaf(v[j], sc[j], sh[j]);
      if (z > 0.f) { /* identity branch */ } else { acc = fmaf(d[j], z, acc); d[j] *= al; }
    }
    store_vec<T, V>(dz + off, d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(dalpha, acc);
}
extern "C" int oct_affine_prelu_bwd(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                                    const float* alpha, void* dz, float* dalpha, size_t npix, int c, void* stream) {
  OCT_CHECK(dout && y && scale && shift && alpha && dz && dalpha && npix > 0 && c > 0, "oct_affine_prelu_bwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks(npix * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((affine_prelu_bwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)dout, \
                                        (const T*)y, scale, shift, alpha, (T*)dz, dalpha, npix, c)
  BK_DISPATCH("oct_affine_prelu_bwd");
#undef LAUNCH
  return oct_check_launch("affine_prelu_bwd");
}

// MaxPool2d(k, stride k, return_indices=True) (EncoderBlock :174-179): pooled value + torch's index convention,
// iy*W + ix inside the (n, c) plane of the INPUT; first maximum in row-major window order.  idx is NHWC like out.
template <typename T, int V>
__global__ void maxpool_idx_fwd_kernel(const T* __restrict__ a, T* __restrict__ out, long long* __restrict__ idx, int n, int ho,
                                       int wo, int c, int k) {
  const int G = c / V;
  const size_t total = (size_t)n * ho * wo * G;
  const int w = wo * k, h = ho * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    float m[V]; int arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { m[j] = -INFINITY; arg[j] = (yo * k) * w + xo * k; }
    for (int q = 0; q < k * k; ++q) {
      const int iy = yo * k + q / k, ix = xo * k + q % k;
      float v[V];
      load_vec<T, V>(a + (((size_t)img * h + iy) * w + ix) * c + g * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) if (v[j] > m[j]) { m[j] = v[j]; arg[j] = iy * w + ix; }
    }
    const size_t o = (((size_t)img * ho + yo) * wo + xo) * c + g * V;
    store_vec<T, V>(out + o, m);
#pragma unroll
    for (int j = 0; j < V; ++j) idx[o + j] = arg[j];
  }
}
extern "C" int oct_maxpool_idx_fwd(int dtype, const void* a, void* out, int64_t* idx, int n, int h, int w, int c, int k,
                                   void* stream) {
  OCT_CHECK(a && out && idx && n > 0 && c > 0 && k > 0 && h > 0 && w > 0 && h % k == 0 && w % k == 0, "oct_maxpool_idx_fwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * (h / k) * (w / k) * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((maxpool_idx_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)a, \
                                        (T*)out, (long long*)idx, n, h / k, w / k, c, k)
  BK_DISPATCH("oct_maxpool_idx_fwd");
#undef LAUNCH
  return oct_check_launch("maxpool_idx_fwd");
}
// scatter (MaxUnpool2d forward :185-188, and max-pool backward): out[n, idx, c] = v[n, p, c], zero elsewhere.  `out` must be
// zero-filled by the caller; an index outside [0, h*w) is skipped.  The pooled grid may be any (hp, wp): indices decide.
// gather (MaxUnpool2d backward, max-pool forward re-read): v[n, p, c] = x[n, idx, c].
template <typename T, bool SCATTER>
__global__ void index_move_kernel(const T* __restrict__ src, const long long* __restrict__ idx, T* __restrict__ dst, int n,
                                  size_t npool, size_t plane, int c) {
  const size_t total = (size_t)n * npool * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = i % c; const size_t img = i / ((size_t)npool * c);
    const long long q = idx[i];
    if (q < 0 || (size_t)q >= plane) { if (!SCATTER) dst[i] = from_f32<T>(0.f); continue; }
    const size_t big = (img * plane + (size_t)q) * c + ch;
    if (SCATTER) dst[big] = src[i]; else dst[i] = src[big];
  }
}
extern "C" int oct_index_scatter(int dtype, const void* v, const int64_t* idx, void* out, int n, size_t npool, size_t plane,
                                 int c, void* stream) {
  OCT_CHECK(v && idx && out && n > 0 && npool > 0 && plane > 0 && c > 0, "oct_index_scatter: bad args");
  const int blocks = bk_blocks((size_t)n * npool * c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OCT_DT_BF16) hipLaunchKernelGGL((index_move_kernel<bf16_t, true>), dim3(blocks), dim3(BK_THREADS), 0, s, (const bf16_t*)v, (const long long*)idx, (bf16_t*)out, n, npool, plane, c);
  else if (dtype == OCT_DT_F32) hipLaunchKernelGGL((index_move_kernel<float, true>), dim3(blocks), dim3(BK_THREADS), 0, s, (const float*)v, (const long long*)idx, (float*)out, n, npool, plane, c);
  else OCT_CHECK(false, "oct_index_scatter: bad dtype");
  return oct_check_launch("index_scatter");
}
extern "C" int oct_index_gather(int dtype, const void* x, const int64_t* idx, void* v, int n, size_t npool, size_t plane,
                                int c, void* stream) {
  OCT_CHECK(x && idx && v && n > 0 && npool > 0 && plane > 0 && c > 0, "oct_index_gather: bad args");
  const int blocks = bk_blocks((size_t)n * npool * c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OCT_DT_BF16) hipLaunchKernelGGL((index_move_kernel<bf16_t, false>), dim3(blocks), dim3(BK_THREADS), 0, s, (const bf16_t*)x, (const long long*)idx, (bf16_t*)v, n, npool, plane, c);
  else if (dtype == OCT_DT_F32) hipLaunchKernelGGL((index_move_kernel<float, false>), dim3(blocks), dim3(BK_THREADS), 0, s, (const float*)x, (const long long*)idx, (float*)v, n, npool, plane, c);
  else OCT_CHECK(false, "oct_index_gather: bad dtype");
  return oct_check_launch("index_gather");
}

// Pool / un-pool by WINDOW CODE (round 3; ReLayNet's encoder -> decoder path inside the network, where the indices never leave the
// library): code[n, yo, xo, c] (one byte) = (iy - yo*k)*k + (ix - xo*k) of the winner -- what torch's int64 plane index says, for an
// index that lies inside its own window (true of everything MaxPool2d returns).  The int64 form cost 8 B per pooled element in each
// of its four uses and a scattered write into a tensor the caller had to zero first (fill + scatter: two passes over the big
// tensor); by code the un-pooling is DENSE -- a thread owns a window, writes the value at its code and zeros at the other k*k - 1
// positions, V channels at a time -- and needs no fill.  scatter = MaxUnpool2d forward = max-pool backward; gather = MaxUnpool2d
// backward.  Same winners as oct_maxpool_idx_fwd (first maximum in row-major window order).
template <typename T, int V>
__global__ void maxpool_code_fwd_kernel(const T* __restrict__ a, T* __restrict__ out, unsigned char* __restrict__ code, int n, int ho,
                                        int wo, int c, int k) {
  const int G = c / V;
  const size_t total = (size_t)n * ho * wo * G;
  const int w = wo * k, h = ho * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    float m[V]; unsigned char arg[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { m[j] = -INFINITY; arg[j] = 0; }
    for (int q = 0; q < k * k; ++q) {
      const int iy = yo * k + q / k, ix = xo * k + q % k;
      float v[V];
      load_vec<T, V>(a + (((size_t)img * h + iy) * w + ix) * c + g * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) if (v[j] > m[j]) { m[j] = v[j]; arg[j] = (unsigned char)q; }
    }
    const size_t o = (((size_t)img * ho + yo) * wo + xo) * c + g * V;
    store_vec<T, V>(out + o, m);
#pragma unroll
    for (int j = 0; j < V; ++j) code[o + j] = arg[j];
  }
}
template <typename T, int V, bool SCATTER>
__global__ void window_move_kernel(const T* __restrict__ src, const unsigned char* __restrict__ code, T* __restrict__ dst, int n, int ho,
                                   int wo, int c, int k) {
  const int G = c / V;
  const size_t total = (size_t)n * ho * wo * G;
  const int w = wo * k, h = ho * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t p = i / G;
    const int xo = p % wo; p /= wo; const int yo = p % ho; const int img = p / ho;
    const size_t o = (((size_t)img * ho + yo) * wo + xo) * c + g * V;
    unsigned char cd[V];
#pragma unroll
    for (int j = 0; j < V; ++j) cd[j] = code[o + j];
    float v[V];
    if (SCATTER) load_vec<T, V>(src + o, v);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
    }
    for (int q = 0; q < k * k; ++q) {
      const size_t big = (((size_t)img * h + yo * k + q / k) * w + xo * k + q % k) * c + g * V;
      float t[V];
      if (SCATTER) {
#pragma unroll
        for (int j = 0; j < V; ++j) t[j] = cd[j] == q ? v[j] : 0.f;
        store_vec<T, V>(dst + big, t);
      } else {
        load_vec<T, V>(src + big, t);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = cd[j] == q ? t[j] : v[j];
      }
    }
    if (!SCATTER) store_vec<T, V>(dst + o, v);
  }
}
extern "C" int oct_maxpool_code_fwd(int dtype, const void* a, void* out, unsigned char* code, int n, int h, int w, int c, int k,
                                    void* stream) {
  OCT_CHECK(a && out && code && n > 0 && c > 0 && k > 0 && k <= 15 && h > 0 && w > 0 && h % k == 0 && w % k == 0, "oct_maxpool_code_fwd: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * (h / k) * (w / k) * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((maxpool_code_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)a, \
                                        (T*)out, code, n, h / k, w / k, c, k)
  BK_DISPATCH("oct_maxpool_code_fwd");
#undef LAUNCH
  return oct_check_launch("maxpool_code_fwd");
}
extern "C" int oct_window_scatter(int dtype, const void* v_, const unsigned char* code, void* out, int n, int hp, int wp, int c, int k,
                                  void* stream) {
  OCT_CHECK(v_ && code && out && n > 0 && hp > 0 && wp > 0 && c > 0 && k > 0 && k <= 15, "oct_window_scatter: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * hp * wp * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((window_move_kernel<T, V, true>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)v_, code, \
                                        (T*)out, n, hp, wp, c, k)
  BK_DISPATCH("oct_window_scatter");
#undef LAUNCH
  return oct_check_launch("window_scatter");
}
extern "C" int oct_window_gather(int dtype, const void* x, const unsigned char* code, void* v_, int n, int hp, int wp, int c, int k,
                                 void* stream) {
  OCT_CHECK(x && code && v_ && n > 0 && hp > 0 && wp > 0 && c > 0 && k > 0 && k <= 15, "oct_window_gather: bad args");
  const int v = bk_vec(c);
  const int blocks = bk_blocks((size_t)n * hp * wp * (c / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((window_move_kernel<T, V, false>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)x, code, \
                                        (T*)v_, n, hp, wp, c, k)
  BK_DISPATCH("oct_window_gather");
#undef LAUNCH
  return oct_check_launch("window_gather");
}

// ---------------------------------------------------------------------------------------------
// MaxPool3d(2) = 2x2 pooling inside every slice (oct_bn_relu_pool_fwd) followed by THIS pairwise maximum over
// consecutive slices; the first maximum in torch's (d, h, w) scanning order is "slice 0 unless slice 1 is strictly
// larger", so the backward routing decomposes the same way: oct_depth_pool_bwd, then the 2-D routing of
// oct_dact_bn_reduce.  p2: (nvol, 2*dout, m) with m = (h/2)*(w/2)*c contiguous elements per slice.
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void depth_pool_fwd_kernel(const T* __restrict__ p2, T* __restrict__ out, size_t nslab, size_t mvec) {
  const size_t total = nslab * mvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t slab = i / mvec, e = i - slab * mvec;
    float a[V], b[V];
    load_vec<T, V>(p2 + ((2 * slab) * mvec + e) * V, a);
    load_vec<T, V>(p2 + ((2 * slab + 1) * mvec + e) * V, b);
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = fmaxf(a[j], b[j]);
    store_vec<T, V>(out + i * V, a);
  }
}
template <typename T, int V>
__global__ void depth_pool_bwd_kernel(const T* __restrict__ p2, const T* __restrict__ dout, T* __restrict__ dp2, size_t nslab,
                                      size_t mvec) {
  const size_t total = nslab * mvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t slab = i / mvec, e = i - slab * mvec;
    float a[V], b[V], d[V], d0[V], d1[V];
    load_vec<T, V>(p2 + ((2 * slab) * mvec + e) * V, a);
    load_vec<T, V>(p2 + ((2 * slab + 1) * mvec + e) * V, b);
    load_vec<T, V>(dout + i * V, d);
#pragma unroll
    for (int j = 0; j < V; ++j) { const bool second = b[j] > a[j]; d0[j] = second ? 0.f : d[j]; d1[j] = second ? d[j] : 0.f; }
    store_vec<T, V>(dp2 + ((2 * slab) * mvec + e) * V, d0);
    store_vec<T, V>(dp2 + ((2 * slab + 1) * mvec + e) * V, d1);
  }
}
extern "C" int oct_depth_pool_fwd(int dtype, const void* p2, void* out, size_t nslab, size_t m, void* stream) {
  OCT_CHECK(p2 && out && nslab > 0 && m > 0, "oct_depth_pool_fwd: bad args");
  const int v = (m % 8 == 0) ? 8 : 1;
  const int blocks = bk_blocks(nslab * (m / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((depth_pool_fwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)p2, (T*)out, nslab, m / V)
  BK_DISPATCH("oct_depth_pool_fwd");
#undef LAUNCH
  return oct_check_launch("depth_pool_fwd");
}
extern "C" int oct_depth_pool_bwd(int dtype, const void* p2, const void* dout, void* dp2, size_t nslab, size_t m, void* stream) {
  OCT_CHECK(p2 && dout && dp2 && nslab > 0 && m > 0, "oct_depth_pool_bwd: bad args");
  const int v = (m % 8 == 0) ? 8 : 1;
  const int blocks = bk_blocks(nslab * (m / v));
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, V) hipLaunchKernelGGL((depth_pool_bwd_kernel<T, V>), dim3(blocks), dim3(BK_THREADS), 0, s, (const T*)p2, \
                                        (const T*)dout, (T*)dp2, nslab, m / V)
  BK_DISPATCH("oct_depth_pool_bwd");
#undef LAUNCH
  return oct_check_launch("depth_pool_bwd");
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution with K <= 4 output channels -- Attention_block's psi (common.py:79-83: Conv2d(F_int, 1, 1)) and the
// SD_Layer_Net heads Conv_1x1 (unet.py:38,113: Conv2d(64, output_ch, 1)):
//   y[pix][k] = sum_c x[pix][c] * w[k][c],  dx[pix][c] = sum_k dy[pix][k] * w[k][c],  dw[k][c] = sum_pix dy[pix][k] * x[pix][c].
// On the MFMA kernels that is a GEMM with 28-31 of 32 output rows (or K lanes) padded: 3.8 + 1.9 ms per cfg4 step for what
// are three streaming passes over the input tensor each.  G = c/8 lanes share a pixel (16-B loads / stores), a dot
// product closes with log2(G) shuffles.  bf16 mode rounds the weight to bf16 first, like the packed MFMA operand it
// replaces; sums are fp32.
// ---------------------------------------------------------------------------------------------
#define RD_MAX_BLOCKS 512
// K = 5 .. 12 (the classifier heads of the layer networks: ReLayNet_2017.py:118-126 Conv2d(64, 10, 1), MGUNet_2021's 11 classes)
// with c <= 128: on the generic MFMA kernels (Cout not a multiple of 32) the three passes of ReLayNet's head took 0.7-0.8 ms
// each at 496 x 768 x 16, 2.3 of 21.5 ms per step, for 0.13 ms of memory traffic.
#define RD_MAX_K 12
#define RD_WIDE_C 128   /* K > 4: the weight-gradient kernel folds its waves through K * c floats of LDS */
static inline bool rowdot_shape_ok(int c, int k) {
  const int g = c / 8;
  return c % 8 == 0 && g >= 1 && g <= 64 && (g & (g - 1)) == 0 && k >= 1 && k <= RD_MAX_K && (k <= 4 || c <= RD_WIDE_C);
}
static inline int rowdot_grid(size_t npix, int c) {
  const size_t ppb = BK_THREADS / (c / 8);
  size_t b = (npix + 4 * ppb - 1) / (4 * ppb);
  if (b > RD_MAX_BLOCKS) b = RD_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" int oct_rowdot_ok(int c, int k) { return rowdot_shape_ok(c, k) ? 1 : 0; }
extern "C" int oct_rowdot_blocks(size_t npix, int c) { return rowdot_shape_ok(c, 1) ? rowdot_grid(npix, c) : 0; }

template <typename T> __device__ __forceinline__ float rd_round(float v) { return to_f32(from_f32<T>(v)); }

template <typename T, int K>
__global__ void __launch_bounds__(BK_THREADS) rowdot_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                                float* __restrict__ stats, size_t npix, int c) {
  const int G = c / 8, gi = threadIdx.x % G, slot = threadIdx.x / G, ppb = BK_THREADS / G;
  float wv[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[k][j] = rd_round<T>(w[(size_t)k * c + gi * 8 + j]);
  float s1[K], s2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  const size_t stride = (size_t)gridDim.x * ppb;
  auto one = [&](const float (&xv)[8], size_t pix) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(xv[j], wv[k][j], d);
      for (int o = 1; o < G; o <<= 1) d += __shfl_xor(d, o);
      if (gi == 0) { y[pix * K + k] = from_f32<T>(d); s1[k] += d; s2[k] = fmaf(d, d, s2[k]); }
    }
  };
  size_t pix = (size_t)blockIdx.x * ppb + slot;
  for (; pix + 3 * stride < npix; pix += 4 * stride) {   // four pixels in flight per lane
    float xv[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_vec_nt<T, 8>(x + (pix + u * stride) * c + gi * 8, xv[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(xv[u], pix + u * stride);
  }
  for (; pix < npix; pix += stride) {
    float xv[8];
    load_vec<T, 8>(x + pix * c + gi * 8, xv);
    one(xv, pix);
  }
  if (stats) {   // one row [2][K] per workgroup, the layout oct_bn_finalize reads
    __shared__ float red[2][K][BK_THREADS / 64];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float a = s1[k], b = s2[k];
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if ((threadIdx.x & 63) == 0) { red[0][k][threadIdx.x >> 6] = a; red[1][k][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * K) {
      const int st = threadIdx.x / K, k = threadIdx.x % K;
      stats[((size_t)blockIdx.x * 2 + st) * K + k] = red[st][k][0] + red[st][k][1] + red[st][k][2] + red[st][k][3];
    }
  }
}

// Forward for K = 5 .. 12 classes in bf16 without statistics (the class heads) on the matrix pipe: per 16 pixels one
// v_mfma_f32_16x16x32_bf16 per 32 input channels, D[row = class][col = pixel].  The B operand is the NHWC tensor as it lies in
// memory (lane (pixel l & 15, k quarter l >> 4) loads the 16 B of channels 8 * (l >> 4) ..+7 of its pixel), the A operand the
// bf16-rounded filter with rows >= K zero.  The shuffle kernel above closes every one of the K dot products with log2(c / 8)
// cross-lane steps and lets one lane in c / 8 store 2-byte values: at K = 10, c = 64 it took 1.12 ms for ReLayNet's head at
// 496 x 768 x 16 (0.9 GB of traffic), slower than the padded GEMM it replaced.
template <int K>
__global__ void __launch_bounds__(256) rowdot_fwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                              bf16_t* __restrict__ y, size_t npix, int c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int nkc = c >> 5;   // 32-channel steps: 1, 2 or 4 (checked by the host)
  bf16x8 afr[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      afr[kc][j] = (kc < nkc && r16 < K) ? (bf16_t)w[(size_t)r16 * c + kc * 32 + kq * 8 + j] : (bf16_t)0.0f;
  const size_t ngroups = (npix + 15) >> 4, gstride = (size_t)gridDim.x * 4;
  for (size_t g = (size_t)blockIdx.x * 4 + wave; g < ngroups; g += gstride) {
    const size_t pix = g * 16 + r16;
    const bool in = pix < npix;
    const bf16_t* xp = x + (in ? pix : npix - 1) * c + kq * 8;
    bf16x8 b[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      if (kc < nkc) b[kc] = *reinterpret_cast<const bf16x8*>(xp + kc * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      if (kc < nkc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], b[kc], acc, 0, 0, 0);
    if (!in) continue;
    bf16_t* const yp = y + pix * K + 4 * kq;   // classes 4 * kq ..+3 of this pixel
    if constexpr ((K & 1) == 0) {              // even K: the pixel's row is 4-byte aligned, pairs leave as one dword
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 4; e += 2)
        if (4 * kq + e < K) {
          bf16x2 v; v[0] = (bf16_t)acc[e]; v[1] = (bf16_t)acc[e + 1];
          *reinterpret_cast<bf16x2*>(yp + e) = v;
        }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * kq + e < K) yp[e] = (bf16_t)acc[e];
    }
  }
}

template <typename T, int K>
__global__ void __launch_bounds__(BK_THREADS) rowdot_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                     T* __restrict__ dx, size_t npix, int c) {
  const int G = c / 8, gi = threadIdx.x % G, slot = threadIdx.x / G, ppb = BK_THREADS / G;
  float wv[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[k][j] = rd_round<T>(w[(size_t)k * c + gi * 8 + j]);
  const size_t stride = (size_t)gridDim.x * ppb;
  for (size_t pix = (size_t)blockIdx.x * ppb + slot; pix < npix; pix += stride) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float g = to_f32(dy[pix * K + k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(g, wv[k][j], o[j]);
    }
    store_vec_nt<T, 8>(dx + pix * c + gi * 8, o);
  }
}

// per-workgroup partial rows part[block][K*c] (plain stores), summed in a fixed order by rowdot_bwd_weight_sum_kernel:
// the weight gradient is bit-reproducible from run to run.  bias (may be NULL): part_b[block][K] = sum_pix dy[pix][k].
template <typename T, int K>
__global__ void __launch_bounds__(BK_THREADS) rowdot_bwd_weight_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                       float* __restrict__ part, size_t npix, int c,
                                                                       float* __restrict__ part_b) {
  const int G = c / 8, gi = threadIdx.x % G, slot = threadIdx.x / G, ppb = BK_THREADS / G;
  float bacc[K];   // bias gradient sum_pix dy[pix][k] (part_b != NULL): the lane with gi == 0 of every pixel slot carries it
#pragma unroll
  for (int k = 0; k < K; ++k) bacc[k] = 0.f;
  float acc[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  const size_t stride = (size_t)gridDim.x * ppb;
  size_t pix = (size_t)blockIdx.x * ppb + slot;
  for (; pix + 3 * stride < npix; pix += 4 * stride) {
    float xv[4][8], g[4][K];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load_vec_nt<T, 8>(x + (pix + u * stride) * c + gi * 8, xv[u]);
#pragma unroll
      for (int k = 0; k < K; ++k) g[u][k] = to_f32(dy[(pix + u * stride) * K + k]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        bacc[k] += g[u][k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(g[u][k], xv[u][j], acc[k][j]);
      }
  }
  for (; pix < npix; pix += stride) {
    float xv[8];
    load_vec<T, 8>(x + pix * c + gi * 8, xv);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float g = to_f32(dy[pix * K + k]);
      bacc[k] += g;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(g, xv[j], acc[k][j]);
    }
  }
  if (part_b) {   // one LDS atomic per (pixel slot, k), then one row [K] per workgroup (summed in order by the caller's second pass)
    __shared__ float sb[K];
    if (threadIdx.x < K) sb[threadIdx.x] = 0.f;
    __syncthreads();
    if (gi == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) atomicAdd(&sb[k], bacc[k]);
    }
    __syncthreads();
    if (threadIdx.x < K) part_b[(size_t)blockIdx.x * K + threadIdx.x] = sb[threadIdx.x];
  }
  // lanes gi, gi + G, gi + 2G ... of a wave own the same channels: fold them, then the four waves through LDS
  __shared__ float red[BK_THREADS / 64][K * (K > 4 ? RD_WIDE_C : 512)];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[k][j];
      for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if ((threadIdx.x & 63) < G) red[threadIdx.x >> 6][k * c + gi * 8 + j] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < K * c; i += BK_THREADS) part[(size_t)blockIdx.x * (K * c) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}
// 16 columns x 16 row-slices per workgroup; slice s adds rows s, s + 16, ... in order, the slices are folded in order:
// a fixed summation tree (reproducible), 32 dependent loads per thread instead of 512
__global__ void __launch_bounds__(256) rowdot_bwd_weight_sum_kernel(const float* __restrict__ part, int nblk, int c,
                                                                    float* __restrict__ dw, int accumulate) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, i = blockIdx.x * 16 + cl;
  float v = 0.f;
  if (i < c)
    for (int b = sl; b < nblk; b += 16) v += part[(size_t)b * c + i];
  red[sl][cl] = v;
  __syncthreads();
  if (sl == 0 && i < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

#define RD_LAUNCH(KERNEL, T, ...)                                                                                  \
  do {                                                                                                             \
    if (k == 1) hipLaunchKernelGGL((KERNEL<T, 1>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);               \
    else if (k == 2) hipLaunchKernelGGL((KERNEL<T, 2>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 3) hipLaunchKernelGGL((KERNEL<T, 3>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 4) hipLaunchKernelGGL((KERNEL<T, 4>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 5) hipLaunchKernelGGL((KERNEL<T, 5>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 6) hipLaunchKernelGGL((KERNEL<T, 6>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 7) hipLaunchKernelGGL((KERNEL<T, 7>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 8) hipLaunchKernelGGL((KERNEL<T, 8>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 9) hipLaunchKernelGGL((KERNEL<T, 9>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);          \
    else if (k == 10) hipLaunchKernelGGL((KERNEL<T, 10>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);        \
    else if (k == 11) hipLaunchKernelGGL((KERNEL<T, 11>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERNEL<T, 12>), dim3(grid), dim3(BK_THREADS), 0, s, __VA_ARGS__);                     \
  } while (0)

extern "C" int oct_rowdot_fwd(int dtype, const void* x, const float* w, void* y, float* stats, size_t npix, int c, int k, void* stream) {
  OCT_CHECK(x && w && y && npix > 0, "oct_rowdot_fwd: bad args");
  OCT_CHECK(rowdot_shape_ok(c, k), "oct_rowdot_fwd: c = %d must be 8 * 2^j <= 512 and k = %d in 1..12 (k > 4: c <= 128; ask oct_rowdot_ok)", c, k);
  const int grid = rowdot_grid(npix, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OCT_DT_BF16 && k > 4 && !stats && (c % 32) == 0) {   // class heads: the matrix-pipe kernel
    size_t gb = ((npix + 15) / 16 + 3) / 4;
    if (gb > 2048) gb = 2048;   // eight waves per SIMD
#define RDM(KK) hipLaunchKernelGGL((rowdot_fwd_mfma_kernel<KK>), dim3((int)gb), dim3(256), 0, s, (const bf16_t*)x, w, (bf16_t*)y, npix, c)
    switch (k) { case 5: RDM(5); break; case 6: RDM(6); break; case 7: RDM(7); break; case 8: RDM(8); break; case 9: RDM(9); break;
                 case 10: RDM(10); break; case 11: RDM(11); break; default: RDM(12); break; }
#undef RDM
    return oct_check_launch("rowdot_fwd_mfma");
  }
  if (dtype == OCT_DT_BF16) RD_LAUNCH(rowdot_fwd_kernel, bf16_t, (const bf16_t*)x, w, (bf16_t*)y, stats, npix, c);
  else if (dtype == OCT_DT_F32) RD_LAUNCH(rowdot_fwd_kernel, float, (const float*)x, w, (float*)y, stats, npix, c);
  else OCT_CHECK(false, "oct_rowdot_fwd: bad dtype");
  return oct_check_launch("rowdot_fwd");
}
extern "C" int oct_rowdot_bwd_data(int dtype, const void* dy, const float* w, void* dx, size_t npix, int c, int k, void* stream) {
  OCT_CHECK(dy && w && dx && npix > 0, "oct_rowdot_bwd_data: bad args");
  OCT_CHECK(rowdot_shape_ok(c, k), "oct_rowdot_bwd_data: c = %d must be 8 * 2^j <= 512 and k = %d in 1..12 (k > 4: c <= 128)", c, k);
  const int grid = rowdot_grid(npix, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OCT_DT_BF16) RD_LAUNCH(rowdot_bwd_data_kernel, bf16_t, (const bf16_t*)dy, w, (bf16_t*)dx, npix, c);
  else if (dtype == OCT_DT_F32) RD_LAUNCH(rowdot_bwd_data_kernel, float, (const float*)dy, w, (float*)dx, npix, c);
  else OCT_CHECK(false, "oct_rowdot_bwd_data: bad dtype");
  return oct_check_launch("rowdot_bwd_data");
}
extern "C" int oct_rowdot_bwd_weight(int dtype, const void* dy, const void* x, float* dw, float* partials, size_t npix, int c,
                                     int k, int accumulate, void* stream) {
  OCT_CHECK(dy && x && dw && partials && npix > 0, "oct_rowdot_bwd_weight: bad args");
  OCT_CHECK(rowdot_shape_ok(c, k), "oct_rowdot_bwd_weight: c = %d must be 8 * 2^j <= 512 and k = %d in 1..12 (k > 4: c <= 128)", c, k);
  const int grid = rowdot_grid(npix, c);
  hipStream_t s = (hipStream_t)stream;
  float* const part_b = nullptr;
  if (dtype == OCT_DT_BF16) RD_LAUNCH(rowdot_bwd_weight_kernel, bf16_t, (const bf16_t*)dy, (const bf16_t*)x, partials, npix, c, part_b);
  else if (dtype == OCT_DT_F32) RD_LAUNCH(rowdot_bwd_weight_kernel, float, (const float*)dy, (const float*)x, partials, npix, c, part_b);
  else OCT_CHECK(false, "oct_rowdot_bwd_weight: bad dtype");
  hipLaunchKernelGGL(rowdot_bwd_weight_sum_kernel, dim3((k * c + 15) / 16), dim3(256), 0, s, partials, grid, k * c, dw, accumulate);
  return oct_check_launch("rowdot_bwd_weight");
}
// The same with the bias gradient dbias[k] (+)= sum_pix dy[pix][k] from the same pass over dy (partials: [oct_rowdot_blocks][k*c + k]).
// A class head's bias gradient was a pass of its own (oct_channel_sum over a 10-channel tensor: 0.34 ms at ReLayNet's size).
extern "C" int oct_rowdot_bwd_weight_bias(int dtype, const void* dy, const void* x, float* dw, float* dbias, float* partials,
                                          size_t npix, int c, int k, int accumulate, void* stream) {
  OCT_CHECK(dy && x && dw && dbias && partials && npix > 0, "oct_rowdot_bwd_weight_bias: bad args");
  OCT_CHECK(rowdot_shape_ok(c, k), "oct_rowdot_bwd_weight_bias: c = %d must be 8 * 2^j <= 512 and k = %d in 1..12 (k > 4: c <= 128)", c, k);
  const int grid = rowdot_grid(npix, c);
  hipStream_t s = (hipStream_t)stream;
  float* const part_b = partials + (size_t)grid * k * c;
  if (dtype == OCT_DT_BF16) RD_LAUNCH(rowdot_bwd_weight_kernel, bf16_t, (const bf16_t*)dy, (const bf16_t*)x, partials, npix, c, part_b);
  else if (dtype == OCT_DT_F32) RD_LAUNCH(rowdot_bwd_weight_kernel, float, (const float*)dy, (const float*)x, partials, npix, c, part_b);
  else OCT_CHECK(false, "oct_rowdot_bwd_weight_bias: bad dtype");
  hipLaunchKernelGGL(rowdot_bwd_weight_sum_kernel, dim3((k * c + 15) / 16), dim3(256), 0, s, partials, grid, k * c, dw, accumulate);
  hipLaunchKernelGGL(rowdot_bwd_weight_sum_kernel, dim3((k + 15) / 16), dim3(256), 0, s, part_b, grid, k, dbias, accumulate);
  return oct_check_launch("rowdot_bwd_weight_bias");
}
#undef RD_LAUNCH
