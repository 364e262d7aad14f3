// BatchNorm (training mode) statistics / backward, ReLU, 2x2 max-pool and layout helpers.
// All of these are HBM-bound streaming kernels over NHWC tensors: every thread owns one
// 8-channel vector lane (16 B of bf16) so loads/stores are fully coalesced, per-channel sums
// live in registers across a grid-stride loop and are combined once per workgroup through LDS.
#include "common.h"

#define EW_THREADS 256
// Grid cap of the streaming kernels: two workgroups per CU.  Measured on bn_bwd_apply (6 B/element over 1-GB
// tensors): 256 / 512 / 1024 / 2048 / 4096 workgroups of one 16-B group per thread and iteration ->
// 5.1 / 3.5 / 3.6 / 4.1 / 4.1 ms per step; more groups in flight per thread shift the optimum to fewer
// workgroups (4 per thread: 3.5 ms at 256, 4.5 at 2048).  About 4 MB in flight chip-wide is the sweet spot;
// beyond it the extra streams cost DRAM page locality.
#define EW_MAX_BLOCKS 512

// ---------------------------------------------------------------------------------------------
// bn_finalize: partials [nblocks][2][c] -> statistics and fused affine coefficients
// ---------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ partials, int nblocks, int c, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* mean_out,
                                   float* invstd_out, float* scale, float* shift, const float* __restrict__ conv_bias) {
  const int ch = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    s1 += (double)partials[((size_t)b * 2 + 0) * c + ch];
    s2 += (double)partials[((size_t)b * 2 + 1) * c + ch];
  }
  __shared__ double red[2][EW_THREADS / 64];
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < EW_THREADS / 64; ++i) { s1 += red[0][i]; s2 += red[1][i]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;  // biased
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma[ch], bt = beta[ch];
    const float sc = (float)((double)g * invstd);
    mean_out[ch] = (float)mean;
    invstd_out[ch] = (float)invstd;
    scale[ch] = sc;
    shift[ch] = (float)((double)bt - mean * (double)g * invstd);
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
      // a conv bias in front of a train-mode BN cancels in the output; it only shifts the running mean
      const double mb = mean + (conv_bias ? (double)conv_bias[ch] : 0.0);
      running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mb);
      running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
    }
  }
}

extern "C" int oct_bn_finalize(const float* partials, int nblocks, int c, double count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* invstd, float* scale, float* shift,
                               const float* conv_bias, void* stream) {
  OCT_CHECK(partials && gamma && beta && mean && invstd && scale && shift, "oct_bn_finalize: null pointer");
  OCT_CHECK(nblocks > 0 && c > 0 && count > 0, "oct_bn_finalize: bad sizes");
  OCT_CHECK((running_mean == nullptr) == (running_var == nullptr), "oct_bn_finalize: running stats mismatch");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(EW_THREADS), 0, as_stream(stream), partials, nblocks, c, count,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, conv_bias);
  return oct_check_launch("bn_finalize");
}

__global__ void bn_eval_coeffs_kernel(int c, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, const float* conv_bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c) {
    const float sc = gamma[i] / sqrtf(rv[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - (rm[i] - (conv_bias ? conv_bias[i] : 0.f)) * sc;   // z = sc*(y + b - rm) + beta
  }
}
extern "C" int oct_bn_eval_coeffs(int c, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift,
                                  const float* conv_bias, void* stream) {
  OCT_CHECK(c > 0 && gamma && beta && running_mean && running_var && scale && shift, "oct_bn_eval_coeffs: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, as_stream(stream), c, gamma, beta,
                     running_mean, running_var, eps, scale, shift, conv_bias);
  return oct_check_launch("bn_eval_coeffs");
}

// ---------------------------------------------------------------------------------------------
// helpers for the vector-lane mapping: thread t of the launch owns channel group (t % G) and
// walks items t / G, t / G + stride, ...  (V channels per group; G = C / V)
// ---------------------------------------------------------------------------------------------
static inline int vec_width(int c) { return (c % 8 == 0) ? 8 : 1; }
static inline bool lane_mapping_ok(int c, int v) { return EW_THREADS % (c / v) == 0; }
static inline int ew_blocks(size_t items, int groups, size_t cap = EW_MAX_BLOCKS) {
  const size_t work = items * (size_t)groups;
  size_t b = (work + EW_THREADS - 1) / EW_THREADS;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&v)[V]) { load_vec<T, V>(p, v); }

// ---------------------------------------------------------------------------------------------
// forward: pooled = maxpool2x2(relu(y*scale+shift))
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void bn_relu_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                        const float* __restrict__ shift, T* __restrict__ out, int n, int h, int w,
                                        int c) {
  const int G = c / V;
  const int ho = h >> 1, wo = w >> 1;
  const size_t total = (size_t)n * ho * wo * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; size_t r = i / G;
    const int xo = r % wo; r /= wo;
    const int yo = r % ho; const int img = r / ho;
    float sc[V], sh[V], m[V];
    ldv<float, V>(scale + g * V, sc); ldv<float, V>(shift + g * V, sh);
#pragma unroll
    for (int j = 0; j < V; ++j) m[j] = 0.f;  // relu output is >= 0, so 0 is the identity of the max
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t pix = ((size_t)img * h + (2 * yo + (q >> 1))) * w + (2 * xo + (q & 1));
      float v[V];
      ldv<T, V>(y + pix * c + g * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) m[j] = fmaxf(m[j], fmaf(v[j], sc[j], sh[j]));
    }
    store_vec<T, V>(out + (((size_t)img * ho + yo) * wo + xo) * c + g * V, m);
  }
}

extern "C" int oct_bn_relu_pool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* pooled,
                                    int n, int h, int w, int c, void* stream) {
  OCT_CHECK(y && scale && shift && pooled, "oct_bn_relu_pool_fwd: null pointer");
  OCT_CHECK(n > 0 && h > 0 && w > 0 && c > 0 && (h % 2 == 0) && (w % 2 == 0), "oct_bn_relu_pool_fwd: bad shape");
  const int v = vec_width(c);
  const int blocks = ew_blocks((size_t)n * (h / 2) * (w / 2), c / v);
  hipStream_t s = as_stream(stream);
#define LAUNCH(T, V) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<T, V>), dim3(blocks), dim3(EW_THREADS), 0, s, \
                                        (const T*)y, scale, shift, (T*)pooled, n, h, w, c)
  if (dtype == OCT_DT_BF16) { if (v == 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 1); }
  else if (dtype == OCT_DT_F32) { if (v == 8) LAUNCH(float, 8); else LAUNCH(float, 1); }
  else OCT_CHECK(false, "oct_bn_relu_pool_fwd: bad dtype");
#undef LAUNCH
  return oct_check_launch("bn_relu_pool_fwd");
}

template <typename T, int V>
__global__ void bn_relu_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, T* __restrict__ out, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i % G; const size_t pix = i / G;
    float sc[V], sh[V], v[V];
    ldv<float, V>(scale + g * V, sc); ldv<float, V>(shift + g * V, sh);
    ldv<T, V>(y + pix * c + g * V, v);
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
    store_vec<T, V>(out + pix * c + g * V, v);
  }
}
extern "C" int oct_bn_relu_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                               size_t npix, int c, void* stream) {
  OCT_CHECK(y && scale && shift && out && npix > 0 && c > 0, "oct_bn_relu_fwd: bad args");
  const int v = vec_width(c);
  const int blocks = ew_blocks(npix, c / v);
  hipStream_t s = as_stream(stream);
#define LAUNCH(T, V) hipLaunchKernelGGL((bn_relu_fwd_kernel<T, V>), dim3(blocks), dim3(EW_THREADS), 0, s, (const T*)y, \
                                        scale, shift, (T*)out, npix, c)
  if (dtype == OCT_DT_BF16) { if (v == 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 1); }
  else if (dtype == OCT_DT_F32) { if (v == 8) LAUNCH(float, 8); else LAUNCH(float, 1); }
  else OCT_CHECK(false, "oct_bn_relu_fwd: bad dtype");
#undef LAUNCH
  return oct_check_launch("bn_relu_fwd");
}

// ---------------------------------------------------------------------------------------------
// backward through ReLU (+ max-pool routing) with the BatchNorm reductions fused:
//   g = (da + route(dpool)) * [z > 0],  partial sums of g and g*xhat per workgroup
// REG: per-thread channel group is loop invariant -> register accumulators (needs 256 % G == 0);
// otherwise LDS atomics (tiny shapes only).
// ---------------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void block_reduce_store(float (&s1)[V], float (&s2)[V], int g, int G, int c,
                                                   float* partials, bool reg_path, float* lds /* [2][c] */) {
  const int tid = threadIdx.x;
  if (reg_path) {
    // lds layout [EW_THREADS][2*V]; threads with equal (tid % G) share a channel group
    __shared__ float red[EW_THREADS * 2 * V > 4096 ? 4096 : EW_THREADS * 2 * V];
    float* my = red + tid * 2 * V;
#pragma unroll
    for (int j = 0; j < V; ++j) { my[j] = s1[j]; my[V + j] = s2[j]; }
    __syncthreads();
    for (int i = tid; i < 2 * c; i += EW_THREADS) {
      const int st = i / c, ch = i - st * c;
      const int gg = ch / V, j = ch - gg * V;
      float acc = 0.f;
      for (int t = gg; t < EW_THREADS; t += G) acc += red[t * 2 * V + st * V + j];
      partials[((size_t)blockIdx.x * 2 + st) * c + ch] = acc;
    }
  } else {
    __syncthreads();
    for (int i = tid; i < 2 * c; i += EW_THREADS) partials[((size_t)blockIdx.x * 2 + i / c) * c + (i % c)] = lds[i];
  }
}

template <typename T, int V, bool POOL>
__global__ void dact_bn_reduce_kernel(const T* da, const T* __restrict__ dpool, const T* __restrict__ y,
                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                      const float* __restrict__ mean, const float* __restrict__ invstd, T* g_out,
                                      float* __restrict__ partials, int n, int h, int w, int c, int reg_path) {
  extern __shared__ float lds_acc[];  // [2][c] for the atomic path
  const int G = c / V;
  if (!reg_path) {
    for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) lds_acc[i] = 0.f;
    __syncthreads();
  }
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int hi = POOL ? (h >> 1) : h, wi = POOL ? (w >> 1) : w;  // item grid
  const size_t total = (size_t)n * hi * wi * G;
  const size_t start = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int gfix = start % G;
  for (size_t i = start; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = reg_path ? gfix : (int)(i % G);
    size_t r = i / G;
    const int xi = r % wi; r /= wi;
    const int yi = r % hi; const int img = r / hi;
    float sc[V], sh[V], mu[V], is[V];
    ldv<float, V>(scale + g * V, sc); ldv<float, V>(shift + g * V, sh);
    ldv<float, V>(mean + g * V, mu); ldv<float, V>(invstd + g * V, is);
    float a1[V], a2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    if (POOL) {
      float yv[4][V], z[4][V], dp[V];
      int arg[V];
      float best[V];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t pix = ((size_t)img * h + (2 * yi + (q >> 1))) * w + (2 * xi + (q & 1));
        ldv<T, V>(y + pix * c + g * V, yv[q]);
#pragma unroll
        for (int j = 0; j < V; ++j) z[q][j] = fmaf(yv[q][j], sc[j], sh[j]);
      }
      ldv<T, V>(dpool + (((size_t)img * hi + yi) * wi + xi) * c + g * V, dp);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        best[j] = fmaxf(z[0][j], 0.f); arg[j] = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const float a = fmaxf(z[q][j], 0.f);
          if (a > best[j]) { best[j] = a; arg[j] = q; }  // strict >: the first maximum wins (ATen)
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t pix = ((size_t)img * h + (2 * yi + (q >> 1))) * w + (2 * xi + (q & 1));
        float dv[V], gv[V];
        if (da) ldv<T, V>(da + pix * c + g * V, dv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float d = da ? dv[j] : 0.f;
          if (arg[j] == q) d += dp[j];
          gv[j] = z[q][j] > 0.f ? d : 0.f;
          // statistics of the value as stored (rounded to T), so that dy is consistent with g
          const float gr = to_f32(from_f32<T>(gv[j]));
          a1[j] += gr;
          a2[j] = fmaf(gr, (yv[q][j] - mu[j]) * is[j], a2[j]);
        }
        store_vec<T, V>(g_out + pix * c + g * V, gv);
      }
    } else {
      const size_t pix = ((size_t)img * h + yi) * w + xi;
      float yv[V], dv[V], gv[V];
      ldv<T, V>(y + pix * c + g * V, yv);
      ldv<T, V>(da + pix * c + g * V, dv);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float z = fmaf(yv[j], sc[j], sh[j]);
        gv[j] = z > 0.f ? dv[j] : 0.f;
        const float gr = to_f32(from_f32<T>(gv[j]));
        a1[j] = gr;
        a2[j] = gr * ((yv[j] - mu[j]) * is[j]);
      }
      if (g_out) store_vec<T, V>(g_out + pix * c + g * V, gv);  // reduce-only mode: bn_bwd_apply re-derives the mask
    }
    if (reg_path) {
#pragma unroll
      for (int j = 0; j < V; ++j) { s1[j] += a1[j]; s2[j] += a2[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) { atomicAdd(&lds_acc[g * V + j], a1[j]); atomicAdd(&lds_acc[c + g * V + j], a2[j]); }
    }
  }
  block_reduce_store<V>(s1, s2, gfix, G, c, partials, reg_path != 0, lds_acc);
}

// Pooled variant with fully coalesced traffic: one thread per (full-resolution column x, 8-channel
// group) handles the two rows of its 2x2 window; the horizontal neighbour's activations come from
// lane ^ G by shuffle, so every global access of a wave is one contiguous run (the window-per-thread
// mapping above reads every other pixel per instruction and measured 1.6 TB/s).
// APPLY = false: the reduction pass; g_out may be NULL (reduce only: the masked gradient is not written -- the apply pass
// below re-derives it).  APPLY = true (`coef` = k[3][c]): the same routing and mask, then dy = k0*g + k1*y + k2 with g rounded
// to the activation dtype first, i.e. bit for bit what bn_bwd_apply computes from a stored g -- but the stored g (2 B written,
// 2 B read back per element) never exists: 11 instead of 12.5 bytes per element for the pooled layers.
template <typename T, bool APPLY = false>
__global__ void dact_pool_coalesced_kernel(const T* da, const T* __restrict__ dpool, const T* __restrict__ y,
                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                           T* g_out, float* __restrict__ partials, int n, int h, int w, int c,
                                           const float* __restrict__ coef = nullptr) {
  constexpr int V = 8;
  const int G = c / V;  // power of two, <= 32: lane ^ G is the same wave's neighbour column
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int ho = h >> 1, wo = w >> 1;
  const size_t total = (size_t)n * ho * w * G;
  const size_t start = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = start % G;
  float sc[V], sh[V], mu[V], is[V];
  load_vec<float, V>(scale + g * V, sc); load_vec<float, V>(shift + g * V, sh);
  if (APPLY) {   // mu / is carry k1 / k2, k0 its own registers
    load_vec<float, V>(coef + c + g * V, mu); load_vec<float, V>(coef + 2 * c + g * V, is);
  } else {
    load_vec<float, V>(mean + g * V, mu); load_vec<float, V>(invstd + g * V, is);
  }
  float k0[V];
  if (APPLY) load_vec<float, V>(coef + g * V, k0);
  // 32-bit index arithmetic (the host routes tensors of 2^31 items and more to the general kernel): the five
  // 64-bit divisions this loop head used to carry are ~500 instructions per 100 B of payload
  const unsigned gshift = 31 - __builtin_clz((unsigned)G);   // G is a power of two
  for (unsigned i = (unsigned)start; i < (unsigned)total; i += gridDim.x * blockDim.x) {
    unsigned r = i >> gshift;
    const int x = r % (unsigned)w; r /= (unsigned)w;
    const int yi = r % (unsigned)ho; const int img = r / (unsigned)ho;
    const bool odd = (x & 1) != 0;
    const size_t p0 = ((size_t)img * h + 2 * yi) * w + x, p1 = p0 + w;
    float y0[V], y1[V], d0[V], d1[V], dp[V];
    load_vec_nt<T, V>(y + p0 * c + g * V, y0);
    load_vec_nt<T, V>(y + p1 * c + g * V, y1);
    if (da) { load_vec_nt<T, V>(da + p0 * c + g * V, d0); load_vec_nt<T, V>(da + p1 * c + g * V, d1); }
    load_vec<T, V>(dpool + (((size_t)img * ho + yi) * wo + (x >> 1)) * c + g * V, dp);
    float g0[V], g1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float z0 = fmaf(y0[j], sc[j], sh[j]), z1 = fmaf(y1[j], sc[j], sh[j]);
      const float a0 = fmaxf(z0, 0.f), a1 = fmaxf(z1, 0.f);
      const float n0 = __shfl_xor(a0, G), n1 = __shfl_xor(a1, G);
      // window order of ATen: (row0,even) (row0,odd) (row1,even) (row1,odd); first maximum wins
      const float q0 = odd ? n0 : a0, q1 = odd ? a0 : n0, q2 = odd ? n1 : a1, q3 = odd ? a1 : n1;
      int arg = 0; float best = q0;
      if (q1 > best) { best = q1; arg = 1; }
      if (q2 > best) { best = q2; arg = 2; }
      if (q3 > best) { best = q3; arg = 3; }
      const int m0 = odd ? 1 : 0, m1 = odd ? 3 : 2;
      float e0 = da ? d0[j] : 0.f, e1 = da ? d1[j] : 0.f;
      if (arg == m0) e0 += dp[j];
      if (arg == m1) e1 += dp[j];
      g0[j] = z0 > 0.f ? e0 : 0.f;
      g1[j] = z1 > 0.f ? e1 : 0.f;
      const float r0 = to_f32(from_f32<T>(g0[j])), r1 = to_f32(from_f32<T>(g1[j]));
      if (APPLY) {   // the arithmetic of bn_bwd_apply on the stored (rounded) g
        g0[j] = fmaf(k0[j], r0, fmaf(mu[j], y0[j], is[j]));
        g1[j] = fmaf(k0[j], r1, fmaf(mu[j], y1[j], is[j]));
      } else {
        s1[j] += r0 + r1;
        s2[j] = fmaf(r0, (y0[j] - mu[j]) * is[j], s2[j]);
        s2[j] = fmaf(r1, (y1[j] - mu[j]) * is[j], s2[j]);
      }
    }
    if (APPLY || g_out) {
      store_vec_nt<T, V>(g_out + p0 * c + g * V, g0);
      store_vec_nt<T, V>(g_out + p1 * c + g * V, g1);
    }
  }
  if (!APPLY) block_reduce_store<V>(s1, s2, g, G, c, partials, true, nullptr);
}

// Un-pooled layers with 8-channel groups that divide the workgroup: the NHWC tensor is a flat array of
// 16-B groups, thread t always owns channel group t % G (the grid stride is a multiple of G), so there is
// no index arithmetic at all, the BatchNorm coefficients are loaded ONCE, and U groups per tensor are in
// flight per thread.  The general kernel above re-derives (image, row, column, group) with 64-bit divisions
// and re-loads eight coefficient vectors per 32 B of payload: it ran at 3.0 TB/s (reduce-only) where this
// shape of access reaches the copy rate.
template <typename T, int U>
__global__ void __launch_bounds__(EW_THREADS) dact_bn_reduce_flat_kernel(
    const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd, T* g_out,
    float* __restrict__ partials, size_t total, int c) {
  constexpr int V = 8;
  const int G = c / V, gi = threadIdx.x % G;
  float sc[V], sh[V], mu[V], is[V], s1[V], s2[V];
  ldv<float, V>(scale + gi * V, sc); ldv<float, V>(shift + gi * V, sh);
  ldv<float, V>(mean + gi * V, mu); ldv<float, V>(invstd + gi * V, is);
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const size_t stride = (size_t)gridDim.x * EW_THREADS;
  size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
  auto one = [&](const float (&yv)[V], const float (&dv)[V], size_t at) {
    float gv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float z = fmaf(yv[j], sc[j], sh[j]);
      gv[j] = z > 0.f ? dv[j] : 0.f;
      const float gr = to_f32(from_f32<T>(gv[j]));   // statistics of the value as stored
      s1[j] += gr;
      s2[j] += gr * ((yv[j] - mu[j]) * is[j]);
    }
    if (g_out) store_vec_nt<T, V>(g_out + at * V, gv);   // reduce-only mode: bn_bwd_apply re-derives the mask
  };
  for (; i + (U - 1) * stride < total; i += U * stride) {
    float yv[U][V], dv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) { load_vec_nt<T, V>(y + (i + u * stride) * V, yv[u]); load_vec_nt<T, V>(da + (i + u * stride) * V, dv[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) one(yv[u], dv[u], i + u * stride);
  }
  for (; i < total; i += stride) {
    float yv[V], dv[V];
    ldv<T, V>(y + i * V, yv); ldv<T, V>(da + i * V, dv);
    one(yv, dv, i);
  }
  block_reduce_store<V>(s1, s2, gi, G, c, partials, true, nullptr);
}

// PReLU in place of the ReLU mask (ReLayNet's BasicBlock, ReLayNet_2017.py:164-168: conv -> BN -> nn.PReLU()): dz = dA * (z > 0 ? 1 :
// alpha) is re-derived in BOTH BatchNorm-backward passes instead of being written by a pass of its own (oct_affine_prelu_bwd: read dA,
// y, write dz -- three of the eight tensor passes of the layer's backward, 2.1 of ReLayNet's 18.4 ms per step).  The value that
// enters the sums and the apply is rounded to the storage type exactly where the separate pass stored it, so the results are
// bit-identical to the three-kernel flow; d(alpha) = sum dA * z * [z <= 0] rides on the reduction (one atomic per wave, as before).
template <typename T, int U>
__global__ void __launch_bounds__(EW_THREADS) dact_bn_reduce_prelu_flat_kernel(
    const T* __restrict__ da, const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ alpha, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ partials, float* __restrict__ dalpha, size_t total, int c) {
  constexpr int V = 8;
  const int G = c / V, gi = threadIdx.x % G;
  float sc[V], sh[V], mu[V], is[V], s1[V], s2[V];
  ldv<float, V>(scale + gi * V, sc); ldv<float, V>(shift + gi * V, sh);
  ldv<float, V>(mean + gi * V, mu); ldv<float, V>(invstd + gi * V, is);
  const float al = alpha[0];
  float sa = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const size_t stride = (size_t)gridDim.x * EW_THREADS;
  size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
  auto one = [&](const float (&yv)[V], const float (&dv)[V]) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float z = fmaf(yv[j], sc[j], sh[j]);
      float gv = dv[j];
      if (!(z > 0.f)) { sa = fmaf(dv[j], z, sa); gv = to_f32(from_f32<T>(dv[j] * al)); }   // the slope branch at z == 0 (ATen)
      s1[j] += gv;
      s2[j] += gv * ((yv[j] - mu[j]) * is[j]);
    }
  };
  for (; i + (U - 1) * stride < total; i += U * stride) {
    float yv[U][V], dv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) { load_vec_nt<T, V>(y + (i + u * stride) * V, yv[u]); load_vec_nt<T, V>(da + (i + u * stride) * V, dv[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) one(yv[u], dv[u]);
  }
  for (; i < total; i += stride) {
    float yv[V], dv[V];
    ldv<T, V>(y + i * V, yv); ldv<T, V>(da + i * V, dv);
    one(yv, dv);
  }
  sa = wave_sum(sa);
  if ((threadIdx.x & 63) == 0) atomicAdd(dalpha, sa);
  block_reduce_store<V>(s1, s2, gi, G, c, partials, true, nullptr);
}
template <typename T, int U>
__global__ void __launch_bounds__(EW_THREADS) bn_bwd_apply_prelu_flat_kernel(T* dst, const T* g, const T* __restrict__ y,
                                                                            const float* __restrict__ coef,
                                                                            const float* __restrict__ scale,
                                                                            const float* __restrict__ shift,
                                                                            const float* __restrict__ alpha, size_t total, int c) {
  constexpr int V = 8;
  const int G = c / V, gi = threadIdx.x % G;
  float k0[V], k1[V], k2[V], sc[V], sh[V];
  ldv<float, V>(coef + gi * V, k0); ldv<float, V>(coef + c + gi * V, k1); ldv<float, V>(coef + 2 * c + gi * V, k2);
  ldv<float, V>(scale + gi * V, sc); ldv<float, V>(shift + gi * V, sh);
  const float al = alpha[0];
  const size_t stride = (size_t)gridDim.x * EW_THREADS;
  size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
  auto one = [&](float (&gv)[V], const float (&yv)[V], size_t at) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (!(fmaf(yv[j], sc[j], sh[j]) > 0.f)) gv[j] = to_f32(from_f32<T>(gv[j] * al));
      gv[j] = fmaf(k0[j], gv[j], fmaf(k1[j], yv[j], k2[j]));
    }
    store_vec_nt<T, V>(dst + at * V, gv);
  };
  for (; i + (U - 1) * stride < total; i += U * stride) {
    float gv[U][V], yv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) { load_vec_nt<T, V>(g + (i + u * stride) * V, gv[u]); load_vec_nt<T, V>(y + (i + u * stride) * V, yv[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) one(gv[u], yv[u], i + u * stride);
  }
  for (; i < total; i += stride) {
    float gv[V], yv[V];
    ldv<T, V>(g + i * V, gv); ldv<T, V>(y + i * V, yv);
    one(gv, yv, i);
  }
}
// 1 when the two entry points below take this channel count (else: oct_affine_prelu_bwd + the plain passes)
extern "C" int oct_prelu_bn_fused_ok(int dtype, int c) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_PRELU_FUSED"); on = (e && e[0] == '0') ? 0 : 1; }
  return (on && (dtype == OCT_DT_BF16 || dtype == OCT_DT_F32) && c > 0 && vec_width(c) == 8 && lane_mapping_ok(c, 8)) ? 1 : 0;
}
// partials: [oct_dact_bn_reduce_blocks(n, h, w, c, 0)][2][c]; dalpha: float[1], zeroed by the caller
extern "C" int oct_dact_bn_reduce_prelu(int dtype, const void* da, const void* y, const float* scale, const float* shift,
                                        const float* alpha, const float* mean, const float* invstd, float* partials,
                                        float* dalpha, int n, int h, int w, int c, void* stream) {
  OCT_CHECK(da && y && scale && shift && alpha && mean && invstd && partials && dalpha, "oct_dact_bn_reduce_prelu: null pointer");
  OCT_CHECK(n > 0 && h > 0 && w > 0 && oct_prelu_bn_fused_ok(dtype, c), "oct_dact_bn_reduce_prelu: shape / dtype not supported (oct_prelu_bn_fused_ok)");
  const int blocks = oct_dact_bn_reduce_blocks(n, h, w, c, 0);
  const size_t total = (size_t)n * h * w * (c / 8);
  hipStream_t s = as_stream(stream);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL((dact_bn_reduce_prelu_flat_kernel<bf16_t, 4>), dim3(blocks), dim3(EW_THREADS), 0, s, (const bf16_t*)da,
                       (const bf16_t*)y, scale, shift, alpha, mean, invstd, partials, dalpha, total, c);
  else
    hipLaunchKernelGGL((dact_bn_reduce_prelu_flat_kernel<float, 2>), dim3(blocks), dim3(EW_THREADS), 0, s, (const float*)da,
                       (const float*)y, scale, shift, alpha, mean, invstd, partials, dalpha, total, c);
  return oct_check_launch("dact_bn_reduce_prelu");
}
extern "C" int oct_bn_bwd_apply_prelu_to(int dtype, void* dst, const void* da, const void* y, const float* coef, const float* scale,
                                         const float* shift, const float* alpha, size_t npix, int c, void* stream) {
  OCT_CHECK(dst && da && y && coef && scale && shift && alpha && npix > 0, "oct_bn_bwd_apply_prelu_to: bad args");
  OCT_CHECK(oct_prelu_bn_fused_ok(dtype, c), "oct_bn_bwd_apply_prelu_to: shape / dtype not supported (oct_prelu_bn_fused_ok)");
  const int blocks = ew_blocks(npix, c / 8);
  const size_t total = npix * (size_t)(c / 8);
  hipStream_t s = as_stream(stream);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_prelu_flat_kernel<bf16_t, 1>), dim3(blocks), dim3(EW_THREADS), 0, s, (bf16_t*)dst,
                       (const bf16_t*)da, (const bf16_t*)y, coef, scale, shift, alpha, total, c);
  else
    hipLaunchKernelGGL((bn_bwd_apply_prelu_flat_kernel<float, 1>), dim3(blocks), dim3(EW_THREADS), 0, s, (float*)dst,
                       (const float*)da, (const float*)y, coef, scale, shift, alpha, total, c);
  return oct_check_launch("bn_bwd_apply_prelu");
}

static bool pool_coalesced_ok(int n, int h, int w, int c) {
  return vec_width(c) == 8 && c <= 256 && ((c / 8) & (c / 8 - 1)) == 0 && (size_t)n * (h / 2) * w * (c / 8) < (1u << 31);
}
// 1 when the pooled BatchNorm backward can run as reduce-only pass (oct_dact_bn_reduce with g = NULL) + oct_bn_bwd_apply_pool
extern "C" int oct_bn_bwd_apply_pool_ok(int dtype, int n, int h, int w, int c) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_POOL_APPLY"); on = (e && e[0] == '0') ? 0 : 1; }
  return (on && (dtype == OCT_DT_BF16 || dtype == OCT_DT_F32) && n > 0 && h > 1 && w > 1 && (h % 2) == 0 && (w % 2) == 0 &&
          pool_coalesced_ok(n, h, w, c)) ? 1 : 0;
}
// dy = k0*g + k1*y + k2 with g = [relu(bn(y)) > 0] * (da + dpool routed to the first maximum of its 2x2 window) re-derived
// on the fly (see dact_pool_coalesced_kernel<T, true>); dy may alias da.
extern "C" int oct_bn_bwd_apply_pool(int dtype, const void* da, const void* dpool, const void* y, const float* scale,
                                     const float* shift, const float* coef, void* dy, int n, int h, int w, int c, void* stream) {
  OCT_CHECK(dpool && y && scale && shift && coef && dy, "oct_bn_bwd_apply_pool: null pointer");
  OCT_CHECK(oct_bn_bwd_apply_pool_ok(dtype, n, h, w, c), "oct_bn_bwd_apply_pool: shape not eligible (oct_bn_bwd_apply_pool_ok)");
  const int blocks = ew_blocks((size_t)n * (h / 2) * w, c / 8, 2048);
  hipStream_t s = as_stream(stream);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL((dact_pool_coalesced_kernel<bf16_t, true>), dim3(blocks), dim3(EW_THREADS), 0, s, (const bf16_t*)da,
                       (const bf16_t*)dpool, (const bf16_t*)y, scale, shift, nullptr, nullptr, (bf16_t*)dy, nullptr, n, h, w, c, coef);
  else
    hipLaunchKernelGGL((dact_pool_coalesced_kernel<float, true>), dim3(blocks), dim3(EW_THREADS), 0, s, (const float*)da,
                       (const float*)dpool, (const float*)y, scale, shift, nullptr, nullptr, (float*)dy, nullptr, n, h, w, c, coef);
  return oct_check_launch("bn_bwd_apply_pool");
}

extern "C" int oct_dact_bn_reduce_blocks(int n, int h, int w, int c, int has_pool) {
  const int v = vec_width(c);
  if (has_pool && v == 8 && c <= 256 && ((c / 8) & (c / 8 - 1)) == 0 && (size_t)n * (h / 2) * w * (c / 8) < (1u << 31))  // coalesced pooled kernel: item = (row pair, x)
    return ew_blocks((size_t)n * (h / 2) * w, c / v, 2048);   // this kernel (five streams, shuffles) measured best at 2048: 1.11 vs 1.35 ms at 512
  const size_t items = has_pool ? (size_t)n * (h / 2) * (w / 2) : (size_t)n * h * w;
  return ew_blocks(items, c / v);
}

extern "C" int oct_dact_bn_reduce(int dtype, const void* da, const void* dpool, const void* y, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, void* g,
                                  float* partials, int n, int h, int w, int c, void* stream) {
  OCT_CHECK(y && scale && shift && mean && invstd && partials, "oct_dact_bn_reduce: null pointer");
  OCT_CHECK(g || !dpool || oct_bn_bwd_apply_pool_ok(dtype, n, h, w, c),
            "oct_dact_bn_reduce: the pooled variant must write g (reduce-only needs oct_bn_bwd_apply_pool_ok)");
  OCT_CHECK(da || dpool, "oct_dact_bn_reduce: need da or dpool");
  OCT_CHECK(n > 0 && h > 0 && w > 0 && c > 0, "oct_dact_bn_reduce: bad shape");
  OCT_CHECK(!dpool || ((h % 2 == 0) && (w % 2 == 0)), "oct_dact_bn_reduce: pooled layer needs even h, w");
  const int v = vec_width(c);
  const int reg = lane_mapping_ok(c, v) ? 1 : 0;
  const int blocks = oct_dact_bn_reduce_blocks(n, h, w, c, dpool != nullptr);
  const size_t lds = (size_t)2 * c * sizeof(float);
  hipStream_t s = as_stream(stream);
  if (dpool && v == 8 && c <= 256 && ((c / 8) & (c / 8 - 1)) == 0 && (size_t)n * (h / 2) * w * (c / 8) < (1u << 31)) {
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL(dact_pool_coalesced_kernel<bf16_t>, dim3(blocks), dim3(EW_THREADS), 0, s, (const bf16_t*)da,
                         (const bf16_t*)dpool, (const bf16_t*)y, scale, shift, mean, invstd, (bf16_t*)g, partials, n, h, w, c);
    else if (dtype == OCT_DT_F32)
      hipLaunchKernelGGL(dact_pool_coalesced_kernel<float>, dim3(blocks), dim3(EW_THREADS), 0, s, (const float*)da,
                         (const float*)dpool, (const float*)y, scale, shift, mean, invstd, (float*)g, partials, n, h, w, c);
    else
      OCT_CHECK(false, "oct_dact_bn_reduce: bad dtype");
    return oct_check_launch("dact_pool_coalesced");
  }
  if (!dpool && v == 8 && reg) {
    const size_t total = (size_t)n * h * w * (c / 8);
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL((dact_bn_reduce_flat_kernel<bf16_t, 4>), dim3(blocks), dim3(EW_THREADS), 0, s, (const bf16_t*)da,
                         (const bf16_t*)y, scale, shift, mean, invstd, (bf16_t*)g, partials, total, c);
    else if (dtype == OCT_DT_F32)
      hipLaunchKernelGGL((dact_bn_reduce_flat_kernel<float, 2>), dim3(blocks), dim3(EW_THREADS), 0, s, (const float*)da,
                         (const float*)y, scale, shift, mean, invstd, (float*)g, partials, total, c);
    else
      OCT_CHECK(false, "oct_dact_bn_reduce: bad dtype");
    return oct_check_launch("dact_bn_reduce_flat");
  }
#define LAUNCH(T, V, P) hipLaunchKernelGGL((dact_bn_reduce_kernel<T, V, P>), dim3(blocks), dim3(EW_THREADS), lds, s, \
                                           (const T*)da, (const T*)dpool, (const T*)y, scale, shift, mean, invstd, \
                                           (T*)g, partials, n, h, w, c, reg)
#define DISPATCH(T)                                                  \
  do {                                                               \
    if (dpool) { if (v == 8) LAUNCH(T, 8, true); else LAUNCH(T, 1, true); } \
    else { if (v == 8) LAUNCH(T, 8, false); else LAUNCH(T, 1, false); }     \
  } while (0)
  if (dtype == OCT_DT_BF16) DISPATCH(bf16_t);
  else if (dtype == OCT_DT_F32) DISPATCH(float);
  else OCT_CHECK(false, "oct_dact_bn_reduce: bad dtype");
#undef DISPATCH
#undef LAUNCH
  return oct_check_launch("dact_bn_reduce");
}

// ---------------------------------------------------------------------------------------------
// bn_bwd_finalize: partials -> dgamma, dbeta, coefficients of dy = k0*g + k1*y + k2
//   dy = gamma*invstd * (g - sum(g)/N - xhat * sum(g*xhat)/N),  xhat = (y - mean)*invstd
// ---------------------------------------------------------------------------------------------
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblocks, int c, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* dgamma, float* dbeta, float* coef,
                                       int accumulate) {
  const int ch = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    s1 += (double)partials[((size_t)b * 2 + 0) * c + ch];
    s2 += (double)partials[((size_t)b * 2 + 1) * c + ch];
  }
  __shared__ double red[2][EW_THREADS / 64];
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < EW_THREADS / 64; ++i) { s1 += red[0][i]; s2 += red[1][i]; }
    const double a = (double)gamma[ch] * (double)invstd[ch];
    const double mg = s1 / count, mgx = s2 / count;
    const double k1 = -a * (double)invstd[ch] * mgx;
    coef[ch] = (float)a;
    coef[c + ch] = (float)k1;
    coef[2 * c + ch] = (float)(-a * mg - k1 * (double)mean[ch]);
    if (accumulate) { dgamma[ch] += (float)s2; dbeta[ch] += (float)s1; }
    else { dgamma[ch] = (float)s2; dbeta[ch] = (float)s1; }
  }
}
extern "C" int oct_bn_bwd_finalize(const float* partials, int nblocks, int c, double count, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                                   int accumulate, void* stream) {
  OCT_CHECK(partials && gamma && mean && invstd && dgamma && dbeta && coef, "oct_bn_bwd_finalize: null pointer");
  OCT_CHECK(nblocks > 0 && c > 0 && count > 0, "oct_bn_bwd_finalize: bad sizes");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(EW_THREADS), 0, as_stream(stream), partials, nblocks, c,
                     count, gamma, mean, invstd, dgamma, dbeta, coef, accumulate);
  return oct_check_launch("bn_bwd_finalize");
}

// masked != 0: `g` holds dA (gradient w.r.t. the activation); the ReLU mask [y*scale+shift > 0] is
// re-derived here, so the reduction pass before it never has to write a masked copy.
template <typename T, int V>
__global__ void bn_bwd_apply_kernel(T* dst, const T* g, const T* __restrict__ y, const float* __restrict__ coef,
                                    const float* __restrict__ scale, const float* __restrict__ shift, size_t npix, int c) {
  const int G = c / V;
  const size_t total = npix * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int gi = i % G; const size_t pix = i / G;
    float k0[V], k1[V], k2[V], gv[V], yv[V];
    ldv<float, V>(coef + gi * V, k0); ldv<float, V>(coef + c + gi * V, k1); ldv<float, V>(coef + 2 * c + gi * V, k2);
    ldv<T, V>(g + pix * c + gi * V, gv);
    ldv<T, V>(y + pix * c + gi * V, yv);
    if (scale) {
      float sc[V], sh[V];
      ldv<float, V>(scale + gi * V, sc); ldv<float, V>(shift + gi * V, sh);
#pragma unroll
      for (int j = 0; j < V; ++j) gv[j] = fmaf(yv[j], sc[j], sh[j]) > 0.f ? gv[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) gv[j] = fmaf(k0[j], gv[j], fmaf(k1[j], yv[j], k2[j]));
    store_vec<T, V>(dst + pix * c + gi * V, gv);
  }
}
// flat variant (see dact_bn_reduce_flat_kernel): coefficients once per thread, no index arithmetic
template <typename T, int U>
__global__ void __launch_bounds__(EW_THREADS) bn_bwd_apply_flat_kernel(T* dst, const T* g, const T* __restrict__ y,
                                                                      const float* __restrict__ coef,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, size_t total, int c) {
  constexpr int V = 8;
  const int G = c / V, gi = threadIdx.x % G;
  float k0[V], k1[V], k2[V], sc[V], sh[V];
  ldv<float, V>(coef + gi * V, k0); ldv<float, V>(coef + c + gi * V, k1); ldv<float, V>(coef + 2 * c + gi * V, k2);
  const bool masked = scale != nullptr;
  if (masked) { ldv<float, V>(scale + gi * V, sc); ldv<float, V>(shift + gi * V, sh); }
  const size_t stride = (size_t)gridDim.x * EW_THREADS;
  size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
  auto one = [&](float (&gv)[V], const float (&yv)[V], size_t at) {
    if (masked) {
#pragma unroll
      for (int j = 0; j < V; ++j) gv[j] = fmaf(yv[j], sc[j], sh[j]) > 0.f ? gv[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) gv[j] = fmaf(k0[j], gv[j], fmaf(k1[j], yv[j], k2[j]));
    store_vec_nt<T, V>(dst + at * V, gv);
  };
  for (; i + (U - 1) * stride < total; i += U * stride) {
    float gv[U][V], yv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) { load_vec_nt<T, V>(g + (i + u * stride) * V, gv[u]); load_vec_nt<T, V>(y + (i + u * stride) * V, yv[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u) one(gv[u], yv[u], i + u * stride);
  }
  for (; i < total; i += stride) {
    float gv[V], yv[V];
    ldv<T, V>(g + i * V, gv); ldv<T, V>(y + i * V, yv);
    one(gv, yv, i);
  }
}
extern "C" int oct_bn_bwd_apply_to(int dtype, void* dst, const void* g, const void* y, const float* coef, const float* scale,
                                   const float* shift, size_t npix, int c, void* stream) {
  OCT_CHECK(dst && g && y && coef && npix > 0 && c > 0, "oct_bn_bwd_apply: bad args");
  OCT_CHECK((scale == nullptr) == (shift == nullptr), "oct_bn_bwd_apply: scale/shift mismatch");
  const int v = vec_width(c);
  const int blocks = ew_blocks(npix, c / v);
  hipStream_t s = as_stream(stream);
  if (v == 8 && lane_mapping_ok(c, 8)) {
    const size_t total = npix * (size_t)(c / 8);
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL((bn_bwd_apply_flat_kernel<bf16_t, 1>), dim3(blocks), dim3(EW_THREADS), 0, s, (bf16_t*)dst,
                         (const bf16_t*)g, (const bf16_t*)y, coef, scale, shift, total, c);
    else if (dtype == OCT_DT_F32)
      hipLaunchKernelGGL((bn_bwd_apply_flat_kernel<float, 1>), dim3(blocks), dim3(EW_THREADS), 0, s, (float*)dst,
                         (const float*)g, (const float*)y, coef, scale, shift, total, c);
    else
      OCT_CHECK(false, "oct_bn_bwd_apply: bad dtype");
    return oct_check_launch("bn_bwd_apply_flat");
  }
#define LAUNCH(T, V) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V>), dim3(blocks), dim3(EW_THREADS), 0, s, (T*)dst, \
                                        (const T*)g, (const T*)y, coef, scale, shift, npix, c)
  if (dtype == OCT_DT_BF16) { if (v == 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 1); }
  else if (dtype == OCT_DT_F32) { if (v == 8) LAUNCH(float, 8); else LAUNCH(float, 1); }
  else OCT_CHECK(false, "oct_bn_bwd_apply: bad dtype");
#undef LAUNCH
  return oct_check_launch("bn_bwd_apply");
}
extern "C" int oct_bn_bwd_apply(int dtype, void* g, const void* y, const float* coef, const float* scale,
                                const float* shift, size_t npix, int c, void* stream) {
  return oct_bn_bwd_apply_to(dtype, g, g, y, coef, scale, shift, npix, c, stream);
}

// ---------------------------------------------------------------------------------------------
// per-channel sum over pixels (bias gradient of the transposed convolution)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void channel_sum_kernel(const T* __restrict__ x, float* out, size_t npix, int c) {
  extern __shared__ float acc[];  // [c]
  for (int i = threadIdx.x; i < c; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const size_t total = npix * c;
  // consecutive threads read consecutive elements; per-thread channel = i % c
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t start = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (stride % c == 0) {
    float s = 0.f;
    for (size_t i = start; i < total; i += stride) s += to_f32(x[i]);
    atomicAdd(&acc[start % c], s);
  } else {
    for (size_t i = start; i < total; i += stride) atomicAdd(&acc[i % c], to_f32(x[i]));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(&out[i], acc[i]);
}
// c % 8 == 0 with c/8 dividing the workgroup: a lane owns one 8-channel group for the whole kernel (16-B loads, four
// in flight), folds with the lanes that share its group, one LDS row per wave, one atomic per channel and workgroup.
// The element-per-thread kernel above read 2 B per lane and instruction: 1.2 TB/s on a 0.8 GB tensor.
template <typename T>
__global__ void __launch_bounds__(EW_THREADS) channel_sum_flat_kernel(const T* __restrict__ x, float* out, size_t total /* groups */, int c) {
  constexpr int V = 8;
  const int G = c / V, gi = threadIdx.x % G;
  __shared__ float red[EW_THREADS / 64][512];
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  const size_t stride = (size_t)gridDim.x * EW_THREADS;
  size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
  for (; i + 3 * stride < total; i += 4 * stride) {
    float v[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_vec_nt<T, V>(x + (i + u * stride) * V, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += v[u][j];
  }
  for (; i < total; i += stride) {
    float v[V];
    ldv<T, V>(x + i * V, v);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < V; ++j) {   // G <= 64 (c <= 512): the lanes gi, gi + G, ... of a wave share the group
    float v = acc[j];
    for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o);
    if (lane < G) red[wave][gi * V + j] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < c; k += EW_THREADS) {
    float v = 0.f;
    for (int w_ = 0; w_ < EW_THREADS / 64; ++w_) v += red[w_][k];
    atomicAdd(&out[k], v);
  }
}
__global__ void zero_f32_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
extern "C" int oct_channel_sum(int dtype, const void* x, float* out, size_t npix, int c, int accumulate, void* stream) {
  OCT_CHECK(x && out && npix > 0 && c > 0, "oct_channel_sum: bad args");
  hipStream_t s = as_stream(stream);
  if (!accumulate) hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, s, out, (size_t)c);
  size_t b = (npix * c + 256 * 64 - 1) / (256 * 64);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  if (c % 8 == 0 && c <= 512 && ((c / 8) & (c / 8 - 1)) == 0 && lane_mapping_ok(c, 8) && (((uintptr_t)x) & 15) == 0) {
    const size_t total = npix * (size_t)(c / 8);
    size_t fb = (total + 4 * EW_THREADS - 1) / (4 * EW_THREADS);
    if (fb > 512) fb = 512;
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL(channel_sum_flat_kernel<bf16_t>, dim3((int)fb), dim3(EW_THREADS), 0, s, (const bf16_t*)x, out, total, c);
    else if (dtype == OCT_DT_F32)
      hipLaunchKernelGGL(channel_sum_flat_kernel<float>, dim3((int)fb), dim3(EW_THREADS), 0, s, (const float*)x, out, total, c);
    else
      OCT_CHECK(false, "oct_channel_sum: bad dtype");
    return oct_check_launch("channel_sum_flat");
  }
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3((int)b), dim3(256), c * sizeof(float), s, (const bf16_t*)x, out, npix, c);
  else if (dtype == OCT_DT_F32)
    hipLaunchKernelGGL(channel_sum_kernel<float>, dim3((int)b), dim3(256), c * sizeof(float), s, (const float*)x, out, npix, c);
  else
    OCT_CHECK(false, "oct_channel_sum: bad dtype");
  return oct_check_launch("channel_sum");
}

// ---------------------------------------------------------------------------------------------
// layout conversion of the network input / debugging export
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int n, int c, int h, int w) {
  const size_t hw = (size_t)h * w, total = (size_t)n * c * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes the OUTPUT (n, pix, ch) so stores are coalesced
    const int ch = i % c; size_t r = i / c; const size_t pix = r % hw; const int img = r / hw;
    out[i] = from_f32<T>(x[((size_t)img * c + ch) * hw + pix]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ out, int n, int c, int h, int w) {
  const size_t hw = (size_t)h * w, total = (size_t)n * c * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % hw; size_t r = i / hw; const int ch = r % c; const int img = r / c;
    out[i] = to_f32(x[((size_t)img * hw + pix) * c + ch]);
  }
}
extern "C" int oct_nchw_to_nhwc(int dtype, const float* x, void* out, int n, int c, int h, int w, void* stream) {
  OCT_CHECK(x && out && n > 0 && c > 0 && h > 0 && w > 0, "oct_nchw_to_nhwc: bad args");
  const size_t total = (size_t)n * c * h * w;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), x, (bf16_t*)out, n, c, h, w);
  else if (dtype == OCT_DT_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), x, (float*)out, n, c, h, w);
  else
    OCT_CHECK(false, "oct_nchw_to_nhwc: bad dtype");
  return oct_check_launch("nchw_to_nhwc");
}
extern "C" int oct_nhwc_to_nchw(int dtype, const void* x, float* out, int n, int c, int h, int w, void* stream) {
  OCT_CHECK(x && out && n > 0 && c > 0 && h > 0 && w > 0, "oct_nhwc_to_nchw: bad args");
  const size_t total = (size_t)n * c * h * w;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)x, out, n, c, h, w);
  else if (dtype == OCT_DT_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)x, out, n, c, h, w);
  else
    OCT_CHECK(false, "oct_nhwc_to_nchw: bad dtype");
  return oct_check_launch("nhwc_to_nchw");
}

// ---------------------------------------------------------------------------------------------
// SGD with momentum over a flat fp32 parameter buffer
// ---------------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n,
                           float lr, float momentum, float wd, float gscale, int first) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gr = g[i] * gscale;
    const float pv = p[i];
    if (wd != 0.f) gr = fmaf(wd, pv, gr);
    if (momentum != 0.f) {
      const float b = first ? gr : fmaf(momentum, buf[i], gr);
      buf[i] = b;
      gr = b;
    }
    p[i] = pv - lr * gr;
  }
}
extern "C" int oct_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                            float weight_decay, float grad_scale, int first, void* stream) {
  OCT_CHECK(p && g && n > 0, "oct_sgd_step: bad args");
  OCT_CHECK(momentum == 0.f || buf, "oct_sgd_step: momentum needs a buffer");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, momentum,
                     weight_decay, grad_scale, first);
  return oct_check_launch("sgd_step");
}
