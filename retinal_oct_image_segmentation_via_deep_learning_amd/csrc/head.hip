// Segmentation head: 1x1 conv + channel softmax (+ arg-max), fused with the per-pixel
// cross-entropy / soft-Dice loss reductions, and the matching d(loss)/d(logits) kernel.
// One thread per pixel: the 64-B NHWC feature row of a pixel is read with 16-B loads, the
// class weights sit in LDS (broadcast reads), probabilities go out NCHW so that consecutive
// lanes write consecutive addresses.  Loss sums are reduced wave -> workgroup -> fp64 partials.
#include "common.h"

#include <stdlib.h>

#define HEAD_THREADS 256
int oct_head_backward_mfma(const OctHeadDesc* d, const void* y, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* w, const float* b, const int64_t* target,
                           const float* dice_coef, float w_ce, void* da, float* partials, float* dbias, float* dweight,
                           double* loss_partials, int grid, void* stream);
#define HEAD_MAX_FEAT 128

struct HeadParams {
  const void* y; const float* scale; const float* shift; const float* w; const float* b;
  const int64_t* target; float* probs; int64_t* argmax; float* logits; double* loss_partials;
  const float* dice_coef; const float* dprobs; void* dlogits; float w_ce;
  int n, h, wd, feat, classes;
};

template <typename T, int CMAX>
__device__ __forceinline__ void head_logits(const HeadParams& p, size_t pix, const float* sw, const float* sb,
                                            const float* ssc, const float* ssh, float (&l)[CMAX]) {
  const T* y = reinterpret_cast<const T*>(p.y) + pix * p.feat;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) l[c] = (c < p.classes) ? sb[c] : 0.f;
  if ((p.feat & 7) == 0) {
    for (int f0 = 0; f0 < p.feat; f0 += 8) {
      float a[8];
      load_vec<T, 8>(y + f0, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaxf(fmaf(a[j], ssc[f0 + j], ssh[f0 + j]), 0.f);
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.classes) {
#pragma unroll
          for (int j = 0; j < 8; ++j) l[c] = fmaf(sw[c * p.feat + f0 + j], a[j], l[c]);
        }
    }
  } else {
    for (int f = 0; f < p.feat; ++f) {
      const float a = fmaxf(fmaf(to_f32(y[f]), ssc[f], ssh[f]), 0.f);
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.classes) l[c] = fmaf(sw[c * p.feat + f], a, l[c]);
    }
  }
}

template <int CMAX>
__device__ __forceinline__ void softmax_c(int classes, const float (&l)[CMAX], float (&pr)[CMAX], float& m, float& lse) {
  m = l[0];
#pragma unroll
  for (int c = 1; c < CMAX; ++c)
    if (c < classes) m = fmaxf(m, l[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    pr[c] = (c < classes) ? expf(l[c] - m) : 0.f;
    s += pr[c];
  }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) pr[c] *= inv;
  lse = logf(s);
}

__device__ __forceinline__ void head_load_consts(const HeadParams& p, float* sw, float* sb, float* ssc, float* ssh) {
  for (int i = threadIdx.x; i < p.classes * p.feat; i += blockDim.x) sw[i] = p.w[i];
  for (int i = threadIdx.x; i < p.classes; i += blockDim.x) sb[i] = p.b[i];
  for (int i = threadIdx.x; i < p.feat; i += blockDim.x) { ssc[i] = p.scale[i]; ssh[i] = p.shift[i]; }
  __syncthreads();
}

template <typename T, int CMAX>
__global__ void __launch_bounds__(HEAD_THREADS) head_fwd_kernel(const HeadParams p) {
  __shared__ float sw[OCT_MAX_CLASSES * HEAD_MAX_FEAT];
  __shared__ float sb[OCT_MAX_CLASSES], ssc[HEAD_MAX_FEAT], ssh[HEAD_MAX_FEAT];
  __shared__ double red[HEAD_THREADS / 64][OCT_HEAD_LOSS_SLOTS];
  head_load_consts(p, sw, sb, ssc, ssh);
  const size_t hw = (size_t)p.h * p.wd, npix = (size_t)p.n * hw;
  float ce = 0.f, si[CMAX], sp[CMAX], sy[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { si[c] = 0.f; sp[c] = 0.f; sy[c] = 0.f; }
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    float l[CMAX], pr[CMAX], m, lse;
    head_logits<T, CMAX>(p, pix, sw, sb, ssc, ssh, l);
    softmax_c<CMAX>(p.classes, l, pr, m, lse);
    const size_t img = pix / hw, off = pix - img * hw;
    if (p.logits) {
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.classes) p.logits[(img * p.classes + c) * hw + off] = l[c];
    }
    if (p.probs) {
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.classes) p.probs[(img * p.classes + c) * hw + off] = pr[c];
    }
    if (p.argmax) {
      int best = 0; float bv = pr[0];
#pragma unroll
      for (int c = 1; c < CMAX; ++c)
        if (c < p.classes && pr[c] > bv) { bv = pr[c]; best = c; }  // first maximum wins (torch.argmax)
      p.argmax[pix] = best;
    }
    if (p.target) {
      const int t = (int)p.target[pix];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const bool hit = (c == t);
        if (hit) ce -= (l[c] - m - lse);
        si[c] += hit ? pr[c] : 0.f;
        sp[c] += pr[c];
        sy[c] += hit ? 1.f : 0.f;
      }
      // torch's nll_loss asserts on a target outside [0, classes); here the loss comes out as NaN (no sync needed)
      if ((unsigned)t >= (unsigned)p.classes) ce = __builtin_nanf("");
    }
  }
  if (p.loss_partials) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v = wave_sum((double)ce);
    if (lane == 0) { red[wave][0] = v; red[wave][1] = 0.0; }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const double a = wave_sum((double)si[c]), b = wave_sum((double)sp[c]), d = wave_sum((double)sy[c]);
      if (lane == 0) {
        red[wave][2 + c] = a; red[wave][2 + OCT_MAX_CLASSES + c] = b; red[wave][2 + 2 * OCT_MAX_CLASSES + c] = d;
      }
    }
    if (lane == 0)
      for (int c = CMAX; c < OCT_MAX_CLASSES; ++c) {
        red[wave][2 + c] = 0.0; red[wave][2 + OCT_MAX_CLASSES + c] = 0.0; red[wave][2 + 2 * OCT_MAX_CLASSES + c] = 0.0;
      }
    __syncthreads();
    for (int i = threadIdx.x; i < OCT_HEAD_LOSS_SLOTS; i += blockDim.x) {
      double s = 0.0;
      for (int wv = 0; wv < HEAD_THREADS / 64; ++wv) s += red[wv][i];
      p.loss_partials[(size_t)blockIdx.x * OCT_HEAD_LOSS_SLOTS + i] = s;
    }
  }
}

// loss_out[0..2] = total, ce, dice.  dice_coef[0][c] = A_c, dice_coef[1][c] = B_c with
// d(w_dice*dice)/dp_c = A_c*y_c + B_c
__global__ void __launch_bounds__(1024) head_loss_finalize_kernel(const double* __restrict__ partials, int nblocks, int classes, double npix,
                                          float w_ce, float w_dice, float eps, float* loss_out, float* dice_coef) {
  __shared__ double tot[OCT_HEAD_LOSS_SLOTS];
  __shared__ double part[16][64];
  {
    // 1024 threads: slot = tid % 64, sixteenth of the block range = tid / 64
    const int slot = threadIdx.x & 63, seg = threadIdx.x >> 6;
    // eight independent partial sums per thread: the block range is walked with eight loads in flight instead of one
    // dependent load per ~2 us round trip (the kernel was 56 us for 4096 rows: profiles/r03_bench_kernel_stats.csv)
    double sv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (slot < OCT_HEAD_LOSS_SLOTS) {
      int b = seg;
      for (; b + 7 * 16 < nblocks; b += 8 * 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] += partials[(size_t)(b + 16 * u) * OCT_HEAD_LOSS_SLOTS + slot];
      }
      for (; b < nblocks; b += 16) sv[0] += partials[(size_t)b * OCT_HEAD_LOSS_SLOTS + slot];
    }
    part[seg][slot] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    __syncthreads();
    if (threadIdx.x < OCT_HEAD_LOSS_SLOTS) {
      double a = 0.0;
      for (int k = 0; k < 16; ++k) a += part[k][slot];
      tot[threadIdx.x] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ce = tot[0] / npix;
    double dsum = 0.0;
    for (int c = 0; c < classes; ++c) {
      const double I = tot[2 + c], P = tot[2 + OCT_MAX_CLASSES + c], Y = tot[2 + 2 * OCT_MAX_CLASSES + c];
      const double den = P + Y + (double)eps, num = 2.0 * I + (double)eps;
      dsum += num / den;
      dice_coef[c] = (float)(-(double)w_dice / classes * 2.0 / den);
      dice_coef[OCT_MAX_CLASSES + c] = (float)((double)w_dice / classes * num / (den * den));
    }
    const double dice = 1.0 - dsum / classes;
    loss_out[0] = (float)((double)w_ce * ce + (double)w_dice * dice);
    loss_out[1] = (float)ce;
    loss_out[2] = (float)dice;
  }
}

// d(loss)/d(logits) as an NHWC tensor of the activation dtype
template <typename T, int CMAX>
__global__ void __launch_bounds__(HEAD_THREADS) head_dlogits_kernel(const HeadParams p) {
  __shared__ float sw[OCT_MAX_CLASSES * HEAD_MAX_FEAT];
  __shared__ float sb[OCT_MAX_CLASSES], ssc[HEAD_MAX_FEAT], ssh[HEAD_MAX_FEAT];
  __shared__ float sdc[2 * OCT_MAX_CLASSES];
  if (p.dice_coef) for (int i = threadIdx.x; i < 2 * OCT_MAX_CLASSES; i += blockDim.x) sdc[i] = p.dice_coef[i];
  head_load_consts(p, sw, sb, ssc, ssh);
  const size_t hw = (size_t)p.h * p.wd, npix = (size_t)p.n * hw;
  const float inv_n = 1.f / (float)npix;
  T* out = reinterpret_cast<T*>(p.dlogits);
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    float l[CMAX], pr[CMAX], dp[CMAX], m, lse;
    head_logits<T, CMAX>(p, pix, sw, sb, ssc, ssh, l);
    softmax_c<CMAX>(p.classes, l, pr, m, lse);
    float dl[CMAX];
    if (p.dprobs) {
      const size_t img = pix / hw, off = pix - img * hw;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        dp[c] = (c < p.classes) ? p.dprobs[(img * p.classes + c) * hw + off] : 0.f;
        dot = fmaf(pr[c], dp[c], dot);
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) dl[c] = pr[c] * (dp[c] - dot);
    } else {
      const int t = (int)p.target[pix];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        dp[c] = 0.f;
        if (p.dice_coef && c < p.classes) dp[c] = (c == t ? sdc[c] : 0.f) + sdc[OCT_MAX_CLASSES + c];
        dot = fmaf(pr[c], dp[c], dot);
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        dl[c] = p.w_ce * (pr[c] - (c == t ? 1.f : 0.f)) * inv_n + pr[c] * (dp[c] - dot);
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < p.classes) out[pix * p.classes + c] = from_f32<T>(dl[c]);
  }
}

// Fused head backward for feat == 32: d(loss)/d(logits), the 1x1 data gradient dA = W^T dlogits, the
// bias and weight gradients, and the BatchNorm-backward partial sums of the layer feeding the head --
// one pass over y instead of (dlogits kernel + 1x1 implicit GEMM + 1x1 wgrad + ReLU/BN reduction).
//
// Four lanes share a pixel, each owning 8 of the 32 features: the 64-B feature row and the 64-B dA
// row of a pixel are one 16-B access per lane (a wave instruction moves 1 KB contiguously), the
// lane's 8 x classes slice of W lives in registers for the whole kernel (no LDS traffic in the
// loop), logits are completed with two quad-permute adds, and the next pixel's row and label are
// loaded before the current one is processed.
__device__ __forceinline__ float quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

template <typename T, int CMAX, bool DW>
__global__ void __launch_bounds__(HEAD_THREADS) head_bwd_fused_kernel(const HeadParams p, const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd, T* __restrict__ da_out,
                                                                       float* __restrict__ partials, float* __restrict__ dbias,
                                                                       float* __restrict__ dweight, double* __restrict__ loss_partials) {
  constexpr int F = 32, G = 4, PPB = HEAD_THREADS / G;   // pixels per block per iteration
  __shared__ float red[HEAD_THREADS / 64][2 * F + OCT_MAX_CLASSES];
  __shared__ double redce[HEAD_THREADS / 64];
  float ce = 0.f;   // cross-entropy sum of this lane's pixels (g == 0 lanes only; training step without a forward head pass)
  __shared__ float sdw[DW ? CMAX * F : 1];
  const int g = threadIdx.x & 3, slot = threadIdx.x >> 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float w[CMAX][8], bias[CMAX], sc[8], sh[8], mu[8], is[8], dcA[CMAX], dcB[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    bias[c] = c < p.classes ? p.b[c] : 0.f;
    dcA[c] = (p.dice_coef && c < p.classes) ? p.dice_coef[c] : 0.f;
    dcB[c] = (p.dice_coef && c < p.classes) ? p.dice_coef[OCT_MAX_CLASSES + c] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[c][j] = c < p.classes ? p.w[c * F + g * 8 + j] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = p.scale[g * 8 + j]; sh[j] = p.shift[g * 8 + j]; mu[j] = mean[g * 8 + j]; is[j] = invstd[g * 8 + j];
  }
  if (DW) {
    for (int i = threadIdx.x; i < CMAX * F; i += HEAD_THREADS) sdw[i] = 0.f;
    __syncthreads();
  }
  const size_t hw = (size_t)p.h * p.wd, npix = (size_t)p.n * hw;
  const float inv_n = 1.f / (float)npix;
  float s1[8], s2[8], sdb[CMAX], dwa[DW ? CMAX : 1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    sdb[c] = 0.f;
    if (DW) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dwa[c][j] = 0.f;
    }
  }
  T* dl_out = reinterpret_cast<T*>(p.dlogits);
  const T* ybase = reinterpret_cast<const T*>(p.y);
  const size_t stride = (size_t)gridDim.x * PPB;
  size_t pix = (size_t)blockIdx.x * PPB + slot;
  // software pipeline: the row and label of the next pixel are in flight while this one is processed
  float yn[8];
  int tn = 0;
  {
    const size_t q = pix < npix ? pix : 0;
    load_vec<T, 8>(ybase + q * F + g * 8, yn);
    if (p.target) tn = (int)p.target[q];
  }
  for (; pix < npix; pix += stride) {
    float yv[8], z[8], a[8], l[CMAX], pr[CMAX], dl[CMAX], m, lse;
    const int t = tn;
#pragma unroll
    for (int j = 0; j < 8; ++j) yv[j] = yn[j];
    {
      const size_t q = pix + stride < npix ? pix + stride : pix;   // last iteration: harmless re-read
      load_vec<T, 8>(ybase + q * F + g * 8, yn);
      if (p.target) tn = (int)p.target[q];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[j] = fmaf(yv[j], sc[j], sh[j]); a[j] = fmaxf(z[j], 0.f); }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(w[c][j], a[j], s);
      s += quad_xor1(s);
      s += quad_xor2(s);
      l[c] = s + bias[c];
    }
    softmax_c<CMAX>(p.classes, l, pr, m, lse);
    if (p.dprobs) {
      const size_t img = pix / hw, off = pix - img * hw;
      float dp[CMAX], dot = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        dp[c] = (c < p.classes) ? p.dprobs[(img * p.classes + c) * hw + off] : 0.f;
        dot = fmaf(pr[c], dp[c], dot);
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) dl[c] = pr[c] * (dp[c] - dot);
    } else {
      float dp[CMAX], dot = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        dp[c] = (c == t ? dcA[c] : 0.f) + dcB[c];
        dot = fmaf(pr[c], dp[c], dot);
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) dl[c] = p.w_ce * (pr[c] - (c == t ? 1.f : 0.f)) * inv_n + pr[c] * (dp[c] - dot);
      if (loss_partials && g == 0) {
        float lt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) lt = (c == t) ? l[c] : lt;
        ce -= lt - m - lse;
        if ((unsigned)t >= (unsigned)p.classes) ce = __builtin_nanf("");   // out-of-range target -> NaN loss
      }
    }
    // the rest of the backward sees dlogits as stored (activation dtype), like the unfused path
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      dl[c] = (c < p.classes) ? to_f32(from_f32<T>(dl[c])) : 0.f;
      if (g == 0) sdb[c] += dl[c];
    }
    if (dl_out && g == 0) {
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < p.classes) dl_out[pix * p.classes + c] = from_f32<T>(dl[c]);
    }
    float dav[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) d = fmaf(w[c][j], dl[c], d);
      dav[j] = d;
      const float dr = to_f32(from_f32<T>(d));   // dA as stored
      const float gm = z[j] > 0.f ? dr : 0.f;
      s1[j] += gm;
      s2[j] = fmaf(gm, (yv[j] - mu[j]) * is[j], s2[j]);
    }
    store_vec<T, 8>(da_out + pix * F + g * 8, dav);
    if (DW) {
      // dW[c][f] = sum_pix dlogits[c] * a[f], with a as the weight-gradient GEMM would read it (dtype-rounded)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ar = to_f32(from_f32<T>(a[j]));
#pragma unroll
        for (int c = 0; c < CMAX; ++c) dwa[c][j] = fmaf(dl[c], ar, dwa[c][j]);
      }
    }
  }
  // lanes with equal g hold the same feature group: fold the 16 pixel slots of a wave together
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float a = s1[j], b = s2[j];
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane < G) { red[wave][g * 8 + j] = a; red[wave][F + g * 8 + j] = b; }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const float a = wave_sum(sdb[c]);
    if (lane == 0) red[wave][2 * F + c] = a;
  }
  if (DW) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = dwa[c][j];
#pragma unroll
        for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o);
        if (lane < G) atomicAdd(&sdw[c * F + g * 8 + j], v);
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * F + CMAX; i += blockDim.x) {
    float s = 0.f;
    for (int wv = 0; wv < HEAD_THREADS / 64; ++wv) s += red[wv][i];
    if (i < 2 * F) partials[(size_t)blockIdx.x * 2 * F + i] = s;            // [block][2][F]
    else if (i - 2 * F < p.classes) atomicAdd(&dbias[i - 2 * F], s);
  }
  if (DW) {
    for (int i = threadIdx.x; i < p.classes * F; i += blockDim.x) atomicAdd(&dweight[i], sdw[i]);
  }
  if (loss_partials) {   // same row layout as head_fwd_kernel: slot 0 = CE sum, the Dice slots stay zero
    const double v = wave_sum((double)ce);
    if (lane == 0) redce[wave] = v;
    __syncthreads();
    for (int i = threadIdx.x; i < OCT_HEAD_LOSS_SLOTS; i += blockDim.x) {
      double s = 0.0;
      if (i == 0) for (int wv = 0; wv < HEAD_THREADS / 64; ++wv) s += redce[wv];
      loss_partials[(size_t)blockIdx.x * OCT_HEAD_LOSS_SLOTS + i] = s;
    }
  }
}

static int head_grid(const OctHeadDesc* d) {
  const size_t npix = (size_t)d->n * d->h * d->w;
  size_t b = (npix + HEAD_THREADS - 1) / HEAD_THREADS;
  if (b > 2048) b = 2048;
  return (int)b;
}
extern "C" int oct_head_blocks(const OctHeadDesc* d) { return d ? head_grid(d) : 0; }

static int head_check(const OctHeadDesc* d, const char* who) {
  OCT_CHECK(d, "%s: null descriptor", who);
  OCT_CHECK(d->dtype == OCT_DT_BF16 || d->dtype == OCT_DT_F32, "%s: bad dtype %d", who, d->dtype);
  OCT_CHECK(d->n > 0 && d->h > 0 && d->w > 0, "%s: bad shape", who);
  OCT_CHECK(d->feat > 0 && d->feat <= HEAD_MAX_FEAT, "%s: feat %d not in [1,%d]", who, d->feat, HEAD_MAX_FEAT);
  OCT_CHECK(d->classes > 0 && d->classes <= OCT_MAX_CLASSES, "%s: classes %d not in [1,%d]", who, d->classes, OCT_MAX_CLASSES);
  return OCT_OK;
}

#define HEAD_DISPATCH(KERNEL, d, grid, s, p)                                                           \
  do {                                                                                                 \
    const int cm = (d)->classes <= 2 ? 2 : (d)->classes <= 4 ? 4 : (d)->classes <= 8 ? 8 : 16;         \
    if ((d)->dtype == OCT_DT_BF16) {                                                                   \
      if (cm == 2) hipLaunchKernelGGL((KERNEL<bf16_t, 2>), dim3(grid), dim3(HEAD_THREADS), 0, s, p);   \
      else if (cm == 4) hipLaunchKernelGGL((KERNEL<bf16_t, 4>), dim3(grid), dim3(HEAD_THREADS), 0, s, p); \
      else if (cm == 8) hipLaunchKernelGGL((KERNEL<bf16_t, 8>), dim3(grid), dim3(HEAD_THREADS), 0, s, p); \
      else hipLaunchKernelGGL((KERNEL<bf16_t, 16>), dim3(grid), dim3(HEAD_THREADS), 0, s, p);          \
    } else {                                                                                           \
      if (cm == 2) hipLaunchKernelGGL((KERNEL<float, 2>), dim3(grid), dim3(HEAD_THREADS), 0, s, p);    \
      else if (cm == 4) hipLaunchKernelGGL((KERNEL<float, 4>), dim3(grid), dim3(HEAD_THREADS), 0, s, p); \
      else if (cm == 8) hipLaunchKernelGGL((KERNEL<float, 8>), dim3(grid), dim3(HEAD_THREADS), 0, s, p); \
      else hipLaunchKernelGGL((KERNEL<float, 16>), dim3(grid), dim3(HEAD_THREADS), 0, s, p);           \
    }                                                                                                  \
  } while (0)

extern "C" int oct_head_forward(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                                const float* w, const float* b, const int64_t* target, float* probs,
                                int64_t* argmax, float* logits, double* loss_partials, void* stream) {
  int rc = head_check(d, "oct_head_forward");
  if (rc) return rc;
  OCT_CHECK(y && scale && shift && w && b, "oct_head_forward: null pointer");
  OCT_CHECK(!(loss_partials && !target), "oct_head_forward: loss partials need a target");
  HeadParams p = {};
  p.y = y; p.scale = scale; p.shift = shift; p.w = w; p.b = b; p.target = target; p.probs = probs;
  p.argmax = argmax; p.logits = logits; p.loss_partials = target ? loss_partials : nullptr;
  p.n = d->n; p.h = d->h; p.wd = d->w; p.feat = d->feat; p.classes = d->classes;
  if (!loss_partials) p.target = nullptr;
  const int grid = head_grid(d);
  hipStream_t s = as_stream(stream);
  HEAD_DISPATCH(head_fwd_kernel, d, grid, s, p);
  return oct_check_launch("head_fwd");
}

extern "C" int oct_head_loss_finalize(const OctHeadDesc* d, const double* loss_partials, int nblocks, float w_ce,
                                      float w_dice, float dice_eps, float* loss_out, float* dice_coef, void* stream) {
  int rc = head_check(d, "oct_head_loss_finalize");
  if (rc) return rc;
  OCT_CHECK(loss_partials && loss_out && dice_coef && nblocks > 0, "oct_head_loss_finalize: bad args");
  hipLaunchKernelGGL(head_loss_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), loss_partials, nblocks,
                     d->classes, (double)d->n * d->h * d->w, w_ce, w_dice, dice_eps, loss_out, dice_coef);
  return oct_check_launch("head_loss_finalize");
}

extern "C" int oct_head_dlogits(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                                const float* w, const float* b, const int64_t* target, const float* dice_coef,
                                float w_ce, const float* dprobs, void* dlogits, void* stream) {
  int rc = head_check(d, "oct_head_dlogits");
  if (rc) return rc;
  OCT_CHECK(y && scale && shift && w && b && dlogits, "oct_head_dlogits: null pointer");
  OCT_CHECK(target || dprobs, "oct_head_dlogits: need a target or dprobs");
  HeadParams p = {};
  p.y = y; p.scale = scale; p.shift = shift; p.w = w; p.b = b; p.target = target; p.dice_coef = dice_coef;
  p.dprobs = dprobs; p.dlogits = dlogits; p.w_ce = w_ce;
  p.n = d->n; p.h = d->h; p.wd = d->w; p.feat = d->feat; p.classes = d->classes;
  const int grid = head_grid(d);
  hipStream_t s = as_stream(stream);
  HEAD_DISPATCH(head_dlogits_kernel, d, grid, s, p);
  return oct_check_launch("head_dlogits");
}

extern "C" int oct_head_backward_fused(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* w, const float* b,
                                       const int64_t* target, const float* dice_coef, float w_ce, const float* dprobs,
                                       void* dlogits, void* da, float* partials, float* dbias, float* dweight,
                                       double* loss_partials, void* stream) {
  int rc = head_check(d, "oct_head_backward_fused");
  if (rc) return rc;
  OCT_CHECK(d->feat == 32, "oct_head_backward_fused: only feat == 32 is fused (got %d); use oct_head_dlogits", d->feat);
  OCT_CHECK(y && scale && shift && mean && invstd && w && b && da && partials && dbias, "oct_head_backward_fused: null pointer");
  OCT_CHECK(target || dprobs, "oct_head_backward_fused: need a target or dprobs");
  HeadParams p = {};
  p.y = y; p.scale = scale; p.shift = shift; p.w = w; p.b = b; p.target = target; p.dice_coef = dice_coef;
  p.dprobs = dprobs; p.dlogits = dlogits; p.w_ce = w_ce;
  p.n = d->n; p.h = d->h; p.wd = d->w; p.feat = d->feat; p.classes = d->classes;
  OCT_CHECK(!dweight || d->classes <= 8, "oct_head_backward_fused: the fused weight gradient needs classes <= 8 (got %d)", d->classes);
  OCT_CHECK(dlogits || dweight, "oct_head_backward_fused: without dlogits the weight gradient must be fused (dweight)");
  OCT_CHECK(!loss_partials || target, "oct_head_backward_fused: loss partials need a target");
  {
    // matrix-pipe formulation (head_mfma.hip): bf16, loss from labels, fused dW, dlogits not requested
    static int mfma_on = -1;
    if (mfma_on < 0) { const char* e = getenv("OCT_HEAD_MFMA"); mfma_on = (e && e[0] == '0') ? 0 : 1; }
    if (mfma_on && d->dtype == OCT_DT_BF16 && d->classes <= 8 && target && !dprobs && dweight && !dlogits)
      return oct_head_backward_mfma(d, y, scale, shift, mean, invstd, w, b, target, dice_coef, w_ce, da, partials, dbias,
                                    dweight, loss_partials, head_grid(d), stream);
  }
  const int grid = head_grid(d);
  hipStream_t s = as_stream(stream);
  const int cm = d->classes <= 2 ? 2 : d->classes <= 4 ? 4 : d->classes <= 8 ? 8 : 16;
#define LAUNCH(T, C, W) hipLaunchKernelGGL((head_bwd_fused_kernel<T, C, W>), dim3(grid), dim3(HEAD_THREADS), 0, s, p, mean, \
                                           invstd, (T*)da, partials, dbias, dweight, loss_partials)
#define BYCLS(T, W)                                                                                   \
  do {                                                                                                \
    if (cm == 2) LAUNCH(T, 2, W); else if (cm == 4) LAUNCH(T, 4, W); else LAUNCH(T, 8, W);            \
  } while (0)
  if (cm == 16) { if (d->dtype == OCT_DT_BF16) LAUNCH(bf16_t, 16, false); else LAUNCH(float, 16, false); }
  else if (d->dtype == OCT_DT_BF16) { if (dweight) BYCLS(bf16_t, true); else BYCLS(bf16_t, false); }
  else { if (dweight) BYCLS(float, true); else BYCLS(float, false); }
#undef BYCLS
#undef LAUNCH
  return oct_check_launch("head_bwd_fused");
}
