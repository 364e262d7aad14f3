// Fused head backward on the matrix pipe (bf16, 32 head features, <= 8 classes, loss from labels).
//
// head.hip's VALU formulation is bound by vector-instruction throughput (about 900 wave64
// instructions per 16 pixels).  Here a wave owns 32 pixels per iteration and both small GEMMs run as
// 32x32x16 MFMAs in the convolution kernels' operand layout (pixel on the lane):
//   logits[class][pixel]  = W  [class x feat] . a[feat x pixel]     2 k16 steps  (A = W rows, padded to 32)
//   dA    [feat ][pixel]  = W^T[feat x class] . dl[class x pixel]   1 k16 step   (classes padded to 16)
// W enters as a bf16 hi + lo pair (two MFMAs per step), i.e. with ~16 mantissa bits; the activation a
// is the bf16 tensor the weight-gradient kernels would read as well.  Everything per-pixel (softmax,
// d(loss)/d(logits), cross-entropy, BatchNorm partial sums, dW = dl x a) stays in registers:
//   * accumulator layout: lane (r, hh) holds classes 4hh..4hh+3 of pixel r -> softmax needs two
//     exchanges with lane r + 32, the dl fragment four more;
//   * dA comes out in the accumulator's feature order; one exchange of four packed registers turns it
//     into the feature order of the loaded y fragments (8hh..8hh+7 and 16+8hh..), in which the masked
//     sums are taken and two 16-B stores per lane write the NHWC rows;
//   * dW[class][feat] = sum_pixel dl . a contracts over the PIXEL index, which sits on the lanes: the wave
//     writes its a tile and dl tile as [pixel][64 B] blocks into a private LDS region and reads both
//     back with ds_read_b64_tr_b16 (the weight-gradient kernels' transposed fragment read), two more
//     MFMAs per 32 pixels into ONE persistent accumulator -- instead of 256 vector FMAs and 128
//     accumulator registers per pixel pair.
// About 330 vector instructions per 32 pixels: the kernel becomes a streaming pass (64 B read + 64 B
// written per pixel).
#include "common.h"

#define HM_THREADS 256

struct HeadMfmaParams {
  const bf16_t* y; const float* scale; const float* shift; const float* mean; const float* invstd;
  const float* w; const float* b; const int64_t* target; const float* dice_coef;
  bf16_t* da; float* partials; float* dbias; float* dweight; double* loss_partials;
  float w_ce; int classes; unsigned npix;
};

typedef unsigned int hm_u32x4 __attribute__((ext_vector_type(4)));
typedef short hm_s16x4 __attribute__((ext_vector_type(4)));

// MFMA operand fragment with k = 8 consecutive pixels per lane from a [pixel][64 B] LDS block (see wgrad2.hip)
__device__ __forceinline__ bf16x8 hm_tr_frag(const unsigned char* base_lo) {
  typedef __attribute__((address_space(3))) hm_s16x4 lds_s16x4;
  const hm_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
  const hm_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo + 4 * 64));  // pixels +4
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ unsigned hm_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float hm_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hm_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float hm_other(float v) { return __shfl_xor(v, 32); }            // lane r <-> lane r + 32
__device__ __forceinline__ unsigned hm_other(unsigned v) { return (unsigned)__shfl_xor((int)v, 32); }

__global__ void __launch_bounds__(HM_THREADS, 2) head_bwd_mfma_kernel(const HeadMfmaParams p) {
  typedef Mma<bf16_t> M;
  typedef M::Frag Frag;
  constexpr int F = 32;
  __shared__ float red[HM_THREADS / 64][2 * F + 8];
  __shared__ float sdw[8 * F];
  __shared__ __attribute__((aligned(16))) unsigned char tiles[HM_THREADS / 64][2][32 * 64];   // per wave: a, dl as [pixel][64 B]
  __shared__ float sdc[2 * OCT_MAX_CLASSES];
  __shared__ double redce[HM_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ncls = p.classes;
  for (int i = tid; i < 8 * F; i += HM_THREADS) sdw[i] = 0.f;
  for (int i = tid; i < (int)sizeof(tiles) / 4; i += HM_THREADS) reinterpret_cast<unsigned*>(&tiles[0][0][0])[i] = 0u;   // dl channels 8..31 stay zero
  if (p.dice_coef) for (int i = tid; i < 2 * OCT_MAX_CLASSES; i += HM_THREADS) sdc[i] = p.dice_coef[i];
  __syncthreads();

  // ---- loop-invariant operands ----------------------------------------------------------------------
  // A fragment of a 32 x 16 block: lane (row = lane & 31, hh) holds k = 8 hh + j
  Frag wl_hi[2], wl_lo[2];   // logits: rows = classes (zero beyond), k = features 16 s + ...
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = r < ncls ? p.w[r * F + 16 * s + 8 * hh + j] : 0.f;
      const bf16_t h = (bf16_t)v;
      wl_hi[s][j] = h;
      wl_lo[s][j] = (bf16_t)(v - (float)h);
    }
  Frag wt_hi, wt_lo;         // dA: rows = features, k = classes (zero beyond)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * hh + j;
    const float v = c < ncls ? p.w[c * F + r] : 0.f;
    const bf16_t h = (bf16_t)v;
    wt_hi[j] = h;
    wt_lo[j] = (bf16_t)(v - (float)h);
  }
  // this lane's 16 features in the order of the loaded fragments: f(q, j) = 16 q + 8 hh + j
  float sc[16], sh[16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[8 * q + j] = p.scale[16 * q + 8 * hh + j]; sh[8 * q + j] = p.shift[16 * q + 8 * hh + j]; }
  float bias4[4];            // this lane's classes 4 hh + i
#pragma unroll
  for (int i = 0; i < 4; ++i) bias4[i] = (4 * hh + i) < ncls ? p.b[4 * hh + i] : 0.f;

  float s1[16], s2[16], sdb[4], ce = 0.f;
  f32x16 accw;               // dW[class = (i & 3) + 8 (i >> 2) + 4 hh][feat = lane & 31], all pixels of this wave
#pragma unroll
  for (int k = 0; k < 16; ++k) { s1[k] = 0.f; s2[k] = 0.f; accw[k] = 0.f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) sdb[i] = 0.f;
  unsigned char* const a_tile = &tiles[wave][0][0];
  unsigned char* const d_tile = &tiles[wave][1][0];
  const int g4 = lane >> 4, li = lane & 15;
  const int tr_off = (8 * (g4 >> 1) + (li >> 2)) * 64 + (16 * (g4 & 1) + 4 * (li & 3)) * 2;
  const float inv_n = 1.f / (float)p.npix;

  // ---- grid-stride over groups of 32 pixels, the next group's rows in flight ----------------------------
  const unsigned ngroups = (p.npix + 31) / 32;
  const unsigned gstride = gridDim.x * (HM_THREADS / 64);
  unsigned grp = blockIdx.x * (HM_THREADS / 64) + wave;
  hm_u32x4 yn0, yn1;
  int tn;
  {
    const unsigned px = grp < ngroups ? min(grp * 32 + r, p.npix - 1) : 0u;
    yn0 = *reinterpret_cast<const hm_u32x4*>(p.y + (size_t)px * F + 8 * hh);
    yn1 = *reinterpret_cast<const hm_u32x4*>(p.y + (size_t)px * F + 16 + 8 * hh);
    tn = (int)p.target[px];
  }
  for (; grp < ngroups; grp += gstride) {
    const unsigned pix = grp * 32 + r;
    const bool valid = pix < p.npix;
    const hm_u32x4 y0 = yn0, y1 = yn1;
    const int t = tn;
    {
      const unsigned g2 = grp + gstride < ngroups ? grp + gstride : grp;
      const unsigned px = min(g2 * 32 + r, p.npix - 1);
      yn0 = *reinterpret_cast<const hm_u32x4*>(p.y + (size_t)px * F + 8 * hh);
      yn1 = *reinterpret_cast<const hm_u32x4*>(p.y + (size_t)px * F + 16 + 8 * hh);
      tn = (int)p.target[px];
    }
    // activation a = relu(y * scale + shift) as bf16 B fragments; z > 0 <=> a > 0 except at exact zeros
    float yv[16], av[16];
    unsigned apk[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      yv[2 * j] = hm_lo(y0[j]); yv[2 * j + 1] = hm_hi(y0[j]);
      yv[8 + 2 * j] = hm_lo(y1[j]); yv[8 + 2 * j + 1] = hm_hi(y1[j]);
    }
    unsigned zmask = 0;      // bit k: pre-activation of feature k is positive
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float z = fmaf(yv[k], sc[k], sh[k]);
      zmask |= z > 0.f ? (1u << k) : 0u;
      av[k] = fmaxf(z, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) apk[j] = hm_pack(av[2 * j], av[2 * j + 1]);
    const hm_u32x4 a0 = {apk[0], apk[1], apk[2], apk[3]}, a1 = {apk[4], apk[5], apk[6], apk[7]};
    const Frag xa0 = __builtin_bit_cast(Frag, a0), xa1 = __builtin_bit_cast(Frag, a1);

    // logits
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    M::mma(acc, wl_hi[0], xa0); M::mma(acc, wl_lo[0], xa0);
    M::mma(acc, wl_hi[1], xa1); M::mma(acc, wl_lo[1], xa1);
    float l[4], pr[4], dl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = (4 * hh + i) < ncls ? acc[i] + bias4[i] : -INFINITY;
    float m = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
    m = fmaxf(m, hm_other(m));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { pr[i] = (4 * hh + i) < ncls ? expf(l[i] - m) : 0.f; s += pr[i]; }
    s += hm_other(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 4; ++i) pr[i] *= inv;
    // d(loss)/d(logits) of this lane's four classes
#pragma unroll
    for (int i = 0; i < 4; ++i) dl[i] = p.w_ce * (pr[i] - ((4 * hh + i) == t ? 1.f : 0.f)) * inv_n;
    if (p.dice_coef) {   // wave-uniform
      float dp[4], dot = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * hh + i;
        dp[i] = c < ncls ? (c == t ? sdc[c] : 0.f) + sdc[OCT_MAX_CLASSES + c] : 0.f;
        dot = fmaf(pr[i], dp[i], dot);
      }
      dot += hm_other(dot);
#pragma unroll
      for (int i = 0; i < 4; ++i) dl[i] = fmaf(pr[i], dp[i] - dot, dl[i]);
    }
    if (p.loss_partials) {
      float lt = 0.f;
      bool mine = false;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((4 * hh + i) == t) { lt = l[i]; mine = true; }
      if (mine && valid) ce -= lt - m - logf(s);
      if (valid && (unsigned)t >= (unsigned)ncls) ce = __builtin_nanf("");   // out-of-range target -> NaN loss
    }
    unsigned dpk[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dl[i] = (valid && (4 * hh + i) < ncls) ? (float)(bf16_t)dl[i] : 0.f;   // as the unfused path would store it
      sdb[i] += dl[i];
    }
    dpk[0] = hm_pack(dl[0], dl[1]); dpk[1] = hm_pack(dl[2], dl[3]);
    // dl as a B fragment (k = class): lanes hh = 0 carry classes 0..7 (their own four + the partner's), hh = 1 zeros
    const unsigned o0 = hm_other(dpk[0]), o1 = hm_other(dpk[1]);
    const hm_u32x4 dfr = {hh ? 0u : dpk[0], hh ? 0u : dpk[1], hh ? 0u : o0, hh ? 0u : o1};
    const Frag xd = __builtin_bit_cast(Frag, dfr);
    f32x16 acd;
#pragma unroll
    for (int i = 0; i < 16; ++i) acd[i] = 0.f;
    M::mma(acd, wt_hi, xd); M::mma(acd, wt_lo, xd);
    // accumulator order: register i = feature (i & 3) + 8 (i >> 2) + 4 hh.  Packed, register pair g holds
    // features 8 g + 4 hh .. + 3; the y order wants, for hh = 0: {0-3, 4-7 | 16-19, 20-23}, i.e. pairs
    // g = 0 (own), g = 0 of the partner, g = 2 (own), g = 2 of the partner; for hh = 1: {8-11 (partner's
    // g = 1), 12-15 (own g = 1) | 24-27 (partner's g = 3), 28-31 (own g = 3)}.
    unsigned cpk[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      cpk[2 * g] = hm_pack(acd[4 * g], acd[4 * g + 1]);
      cpk[2 * g + 1] = hm_pack(acd[4 * g + 2], acd[4 * g + 3]);
    }
    // each lane sends the two pairs its partner needs: hh = 0 sends g = 1, 3; hh = 1 sends g = 0, 2
    const unsigned sA0 = hh ? cpk[0] : cpk[2], sA1 = hh ? cpk[1] : cpk[3];
    const unsigned sB0 = hh ? cpk[4] : cpk[6], sB1 = hh ? cpk[5] : cpk[7];
    const unsigned rA0 = hm_other(sA0), rA1 = hm_other(sA1), rB0 = hm_other(sB0), rB1 = hm_other(sB1);
    unsigned dq[8];          // dA in the y order: dq[0..3] = features 8 hh .. 8 hh + 7, dq[4..7] = 16 + 8 hh ..
    if (hh == 0) {
      dq[0] = cpk[0]; dq[1] = cpk[1]; dq[2] = rA0; dq[3] = rA1;
      dq[4] = cpk[4]; dq[5] = cpk[5]; dq[6] = rB0; dq[7] = rB1;
    } else {
      dq[0] = rA0; dq[1] = rA1; dq[2] = cpk[2]; dq[3] = cpk[3];
      dq[4] = rB0; dq[5] = rB1; dq[6] = cpk[6]; dq[7] = cpk[7];
    }
    if (valid) {
      const hm_u32x4 o0v = {dq[0], dq[1], dq[2], dq[3]}, o1v = {dq[4], dq[5], dq[6], dq[7]};
      *reinterpret_cast<hm_u32x4*>(p.da + (size_t)pix * F + 8 * hh) = o0v;
      *reinterpret_cast<hm_u32x4*>(p.da + (size_t)pix * F + 16 + 8 * hh) = o1v;
    }
    // BatchNorm-backward partial sums of the masked gradient, dA as stored; sum g*y converted at the end
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float d = (k & 1) ? hm_hi(dq[k >> 1]) : hm_lo(dq[k >> 1]);
      const float gm = (valid && ((zmask >> k) & 1u)) ? d : 0.f;
      s1[k] += gm;
      s2[k] = fmaf(gm, yv[k], s2[k]);
    }
    // dW[c][f] += sum over these 32 pixels dl[c] * a[f], both as stored (bf16): tiles to LDS, transposed reads
    *reinterpret_cast<hm_u32x4*>(a_tile + r * 64 + 16 * hh) = a0;
    *reinterpret_cast<hm_u32x4*>(a_tile + r * 64 + 32 + 16 * hh) = a1;
    if (hh == 0) *reinterpret_cast<hm_u32x4*>(d_tile + r * 64) = dfr;
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16)
      M::mma(accw, hm_tr_frag(d_tile + k16 * 16 * 64 + tr_off), hm_tr_frag(a_tile + k16 * 16 * 64 + tr_off));
  }

  // ---- fold the 32 pixel lanes of each half-wave, then the waves of the workgroup -------------------------
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float a = s1[k], b = s2[k];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (r == 0) {
      const int f = 16 * (k >> 3) + 8 * hh + (k & 7);
      red[wave][f] = a; red[wave][F + f] = b;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = sdb[i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (r == 0) red[wave][2 * F + 4 * hh + i] = a;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = (i & 3) + 8 * (i >> 2) + 4 * hh;
    if (c < ncls) atomicAdd(&sdw[c * F + r], accw[i]);
  }
  if (p.loss_partials) {
    const double v = wave_sum((double)ce);
    if (lane == 0) redce[wave] = v;
  }
  __syncthreads();
  for (int i = tid; i < 2 * F + 8; i += HM_THREADS) {
    float s = 0.f;
    for (int wv = 0; wv < HM_THREADS / 64; ++wv) s += red[wv][i];
    if (i < F) p.partials[(size_t)blockIdx.x * 2 * F + i] = s;                       // sum g
    else if (i < 2 * F) {                                                           // sum g*xhat = (sum g*y - mean sum g) invstd
      float sg = 0.f;
      for (int wv = 0; wv < HM_THREADS / 64; ++wv) sg += red[wv][i - F];
      p.partials[(size_t)blockIdx.x * 2 * F + i] = (s - p.mean[i - F] * sg) * p.invstd[i - F];
    } else if (i - 2 * F < ncls) atomicAdd(&p.dbias[i - 2 * F], s);
  }
  for (int i = tid; i < ncls * F; i += HM_THREADS) atomicAdd(&p.dweight[i], sdw[i]);
  if (p.loss_partials) {
    for (int i = tid; i < OCT_HEAD_LOSS_SLOTS; i += HM_THREADS) {
      double s = 0.0;
      if (i == 0) for (int wv = 0; wv < HM_THREADS / 64; ++wv) s += redce[wv];
      p.loss_partials[(size_t)blockIdx.x * OCT_HEAD_LOSS_SLOTS + i] = s;
    }
  }
}

// launched by oct_head_backward_fused (head.hip) when eligible; `grid` = oct_head_blocks
int oct_head_backward_mfma(const OctHeadDesc* d, const void* y, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* w, const float* b, const int64_t* target,
                           const float* dice_coef, float w_ce, void* da, float* partials, float* dbias, float* dweight,
                           double* loss_partials, int grid, void* stream) {
  HeadMfmaParams p;
  p.y = (const bf16_t*)y; p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.w = w; p.b = b;
  p.target = target; p.dice_coef = dice_coef; p.da = (bf16_t*)da; p.partials = partials; p.dbias = dbias;
  p.dweight = dweight; p.loss_partials = loss_partials; p.w_ce = w_ce; p.classes = d->classes;
  p.npix = (unsigned)((size_t)d->n * d->h * d->w);
  hipLaunchKernelGGL(head_bwd_mfma_kernel, dim3(grid), dim3(HM_THREADS), 0, as_stream(stream), p);
  return oct_check_launch("head_bwd_mfma");
}
