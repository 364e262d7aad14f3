// First layer of the U-Net (Conv2d(1 -> F, 3x3), YNet_2022.py:578 with in_channels = 1): K = 9 is
// far too shallow for MFMA -- this is a bandwidth stencil.  Direct VALU kernels:
//   fprop : one thread per (pixel, 8-channel group), the 72 filter taps of the group in registers,
//           16-B NHWC stores contiguous across lanes, BatchNorm partial sums in registers;
//   wgrad : the same mapping, 72 register accumulators, one LDS + atomic reduction per workgroup.
// Both keep the loads of the next pixel in flight while the current one is processed.
// bf16 only; other dtypes / channel counts use the generic implicit-GEMM kernels.
#include "common.h"
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f1_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}

// Position of a pixel as counters (x, y, slice): a thread walks pixels pix, pix + stride, ... and advances the counters by the
// decomposition of `stride` instead of dividing -- q % w, (q / w) % h (and (q / plane) % depth three times per voxel in
// the 3-D kernels) were ~45 of the ~220 vector instructions per pixel of these VALU-bound kernels.
struct F1Pos {
  int x, y, z;
};
struct F1Step {
  int dx, dy, dz, w, h, d;
  __device__ __forceinline__ F1Step(unsigned stride, int w_, int h_, int d_) : w(w_), h(h_), d(d_ > 0 ? d_ : 1) {
    dx = (int)(stride % (unsigned)w_);
    const unsigned r = stride / (unsigned)w_;
    dy = (int)(r % (unsigned)h_);
    dz = (int)((r / (unsigned)h_) % (unsigned)d);
  }
  __device__ __forceinline__ F1Pos at(unsigned q) const {
    F1Pos p;
    p.x = (int)(q % (unsigned)w);
    const unsigned r = q / (unsigned)w;
    p.y = (int)(r % (unsigned)h);
    p.z = (int)((r / (unsigned)h) % (unsigned)d);
    return p;
  }
  __device__ __forceinline__ F1Pos next(F1Pos p) const {
    p.x += dx; int c = p.x >= w ? 1 : 0; p.x -= c ? w : 0;
    p.y += dy + c; c = p.y >= h ? 1 : 0; p.y -= c ? h : 0;
    p.z += dz + c; p.z -= p.z >= d ? d : 0;
    return p;
  }
};

// the 3x3 neighbourhood of pixel q = (.., yy, xx) of a single-channel image, zero outside; loads are unconditional
// (clamped to q itself) so that they can be issued an iteration ahead without divergent control flow
__device__ __forceinline__ void f1_taps(const bf16_t* __restrict__ x, unsigned q, int xx, int yy, int h, int w, float (&v)[9]) {
  const bool r0 = yy > 0, r2 = yy + 1 < h, c0 = xx > 0, c2 = xx + 1 < w;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool ok = (dy < 0 ? r0 : (dy > 0 ? r2 : true)) && (dx < 0 ? c0 : (dx > 0 ? c2 : true));
    const float val = (float)x[ok ? (int)q + dy * w + dx : (int)q];
    v[t] = ok ? val : 0.f;
  }
}

// the same neighbourhood one slice up / down (volumes of `depth` slices, n = N*D images): zero outside the volume
__device__ __forceinline__ void f1_taps_z(const bf16_t* __restrict__ x, unsigned q, F1Pos pos, int h, int w, int depth, int shift,
                                          float (&v)[9]) {
  const unsigned plane = (unsigned)h * (unsigned)w;
  const int dz = pos.z + shift;
  const bool zok = dz >= 0 && dz < depth;
  const unsigned qz = zok ? q + shift * (int)plane : q;
  f1_taps(x, qz, pos.x, pos.y, h, w, v);
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = zok ? v[t] : 0.f;
}

// Conv3d(1 -> F, 3x3x3) of the volumetric U-Net's first layer: the same mapping with the 27 x 8 filter taps of the
// group in registers (216 VGPRs: one wave per SIMD -- the kernel is bound by its 16-B stores, not by occupancy)
template <int F>
__global__ void __launch_bounds__(256) first_fprop3d_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                            bf16_t* __restrict__ y, float* __restrict__ stats, int n,
                                                            int h, int w, int depth) {
  // wp: OCT_PACK_CONV3D_FPROP of the (F,1,3,3,3) filter (kch = 3, nk16 = 1): A[row = co][k = kd] of tap (kh,kw) sits at
  // ((co/32 * 9 + tap) * 512 + (co%32) * 8 + kd)
  constexpr int G = F / 8, PPB = 256 / G;
  __shared__ float red[4][2][F];
  const int g = threadIdx.x % G, slot = threadIdx.x / G;
  float wv[27][8];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = g * 8 + j;
        wv[kd * 9 + t][j] = (float)wp[((co >> 5) * 9 + t) * 512 + (co & 31) * 8 + kd];
      }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const unsigned npix = (unsigned)n * h * w, stride = gridDim.x * PPB;
  const F1Step step(stride, w, h, depth);
  F1Pos pos = step.at(blockIdx.x * PPB + slot);
  for (unsigned pix = blockIdx.x * PPB + slot; pix < npix; pix += stride, pos = step.next(pos)) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      float v[9];
      f1_taps_z(x, pix, pos, h, w, depth, kd - 1, v);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv[kd * 9 + t][j], v[t], acc[j]);
    }
    const u32x4 o = {f1_pack(acc[0], acc[1]), f1_pack(acc[2], acc[3]), f1_pack(acc[4], acc[5]), f1_pack(acc[6], acc[7])};
    *reinterpret_cast<u32x4*>(y + (size_t)pix * F + g * 8) = o;
    if (stats) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
    }
  }
  if (stats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = s1[j], b = s2[j];
#pragma unroll
      for (int o = 32; o >= G; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (lane < G) { red[wave][0][g * 8 + j] = a; red[wave][1][g * 8 + j] = b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * F; i += 256) {
      const int st = i / F, c = i % F;
      stats[((size_t)blockIdx.x * 2 + st) * F + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
    }
  }
}

// fprop: a thread owns (pixel, 8-channel group): its 72 filter taps stay in registers for the whole
// kernel, consecutive lanes store consecutive 16-B chunks (1 KB per wave instruction), the taps of
// the next pixel are loaded while the current one is computed.
template <int F>
__global__ void __launch_bounds__(256) first_fprop_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                          bf16_t* __restrict__ y, float* __restrict__ stats, int n,
                                                          int h, int w) {
  // wp: the MFMA fragment-order packing of the (F,1,3,3) filter (OCT_PACK_CONV_FPROP, nk16 = 1):
  // A[row = co][k = 0] sits at ((co/32 * 9 + tap) * 512 + (co%32) * 8)
  constexpr int G = F / 8, PPB = 256 / G;
  __shared__ float red[4][2][F];
  const int g = threadIdx.x % G, slot = threadIdx.x / G;
  float wv[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = g * 8 + j;
      wv[t][j] = (float)wp[((co >> 5) * 9 + t) * 512 + (co & 31) * 8];
    }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const unsigned npix = (unsigned)n * h * w, stride = gridDim.x * PPB;
  unsigned pix = blockIdx.x * PPB + slot;
  const F1Step step(stride, w, h, 1);
  F1Pos pos = step.at(pix < npix ? pix : 0u);
  float vn[9];
  f1_taps(x, pix < npix ? pix : 0u, pos.x, pos.y, h, w, vn);
  for (; pix < npix; pix += stride) {
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = vn[t];
    {
      const bool more = pix + stride < npix;
      const F1Pos pn = step.next(pos);
      pos.x = more ? pn.x : pos.x; pos.y = more ? pn.y : pos.y;
      f1_taps(x, more ? pix + stride : pix, pos.x, pos.y, h, w, vn);
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv[t][j], v[t], acc[j]);
    const u32x4 o = {f1_pack(acc[0], acc[1]), f1_pack(acc[2], acc[3]), f1_pack(acc[4], acc[5]), f1_pack(acc[6], acc[7])};
    *reinterpret_cast<u32x4*>(y + (size_t)pix * F + g * 8) = o;
    if (stats) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
    }
  }
  if (stats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = s1[j], b = s2[j];
#pragma unroll
      for (int o = 32; o >= G; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (lane < G) { red[wave][0][g * 8 + j] = a; red[wave][1][g * 8 + j] = b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * F; i += 256) {
      const int st = i / F, c = i % F;
      stats[((size_t)blockIdx.x * 2 + st) * F + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
    }
  }
}

// fprop on the matrix pipe (W % 32 == 0): a wave owns 32 consecutive pixels of one image row.  The stencil kernels above are
// bound by vector-instruction throughput (~160 instructions per pixel and 8-channel group: 72 FMAs, nine 2-byte loads with
// their bounds logic; the 3-D one 216 FMAs and 27 loads), not by their 64 B per pixel of output.  Here the K = 9 (27) taps
// are one (two) k16 steps of v_mfma_f32_32x32x16_bf16 padded with zeros: per 32 pixels a lane loads 8 (16) input values --
// row and slice validity are wave-uniform, only the two end lanes of a row test x -- packs them into the B fragment and one
// (two) MFMA per 32 output channels replace 9,216 (27,648) FMAs; the epilogue is igemm2's (bf16 pack, wave-private LDS
// transpose, 16-byte NHWC stores, BatchNorm sums in registers).
// KD = 7: not a depth count but ReLayNet's 7x3 kernel (ReLayNet_2017.py:155-160, padding (3, 1)): 21 taps T = kh*3 + kw, two k16
// steps, tap rows -3 .. +3 (f1_ntap / f1_dy / f1_pady below; the packed filter is [cout/32][21][512], see oct_pack_weights_kk)
template <int KD> constexpr int f1_ntap() { return KD == 7 ? 21 : 9 * KD; }
template <int KD> constexpr int f1_pady() { return KD == 7 ? 3 : 1; }
template <int F, int KD>
__global__ void __launch_bounds__(256) first_fprop_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                               bf16_t* __restrict__ y, float* __restrict__ stats, int n,
                                                               int h, int w, int depth) {
  constexpr int NTAP = f1_ntap<KD>(), PADY = f1_pady<KD>(), KS = (NTAP + 15) / 16, NB = F / 32, NV = 8 * KS;
  typedef Mma<bf16_t> M;
  __shared__ __attribute__((aligned(16))) unsigned char scr[4][32 * 80];
  __shared__ float red[4][2][F];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  // A fragments (filters): lane = (output channel r of the block, k half hh), k = tap index T = kd*9 + kh*3 + kw
  bf16x8 afr[NB][KS];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int T = 16 * ks + 8 * hh + j;
        const int kd = T / 9, t = T - 9 * kd;
        if (KD == 7) afr[cb][ks][j] = T < NTAP ? wp[(cb * NTAP + T) * 512 + r * 8] : (bf16_t)0.0f;
        else afr[cb][ks][j] = T < NTAP ? wp[(cb * 9 + t) * 512 + r * 8 + (KD == 3 ? kd : 0)] : (bf16_t)0.0f;
      }
  // this lane's taps: element offset from the pixel and the (dz, dy, dx) class of each
  int toff[NV];
  unsigned tcls[NV];   // bits 0-1 dz+1, 2-4 dy+PADY, 5-6 dx+1, bit 7 dead (zero padding of K)
  const int plane = h * w;
#pragma unroll
  for (int jj = 0; jj < NV; ++jj) {
    const int T = 16 * (jj / 8) + 8 * hh + (jj % 8);
    const int kd = T / 9, t = T - 9 * kd;
    const int dz = KD == 3 ? kd - 1 : 0, dy = KD == 7 ? T / 3 - 3 : t / 3 - 1, dx = (KD == 7 ? T : t) % 3 - 1;
    toff[jj] = dz * plane + dy * w + dx;
    tcls[jj] = T < NTAP ? (unsigned)((dz + 1) | ((dy + PADY) << 2) | ((dx + 1) << 5)) : 128u;
  }
  float s1[NB][16], s2[NB][16];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[cb][i] = 0.f; s2[cb][i] = 0.f; }

  const unsigned ngroups = ((unsigned)n * h * w) >> 5, gstride = gridDim.x * 4;
  auto gather = [&](unsigned g, unsigned short (&v)[NV]) {
    const unsigned q0 = g << 5;
    const unsigned row = q0 / (unsigned)w;
    const int x0 = (int)(q0 - row * (unsigned)w), yy = (int)(row % (unsigned)h);
    const int zz = KD == 3 ? (int)((row / (unsigned)h) % (unsigned)depth) : 0;
    // validity masks, bit c = class c valid: slice and row wave-uniform, column per lane
    const unsigned zm = KD == 3 ? ((zz > 0 ? 1u : 0u) | 2u | (zz + 1 < depth ? 4u : 0u)) : 2u;
    unsigned ym = 0;   // bit c: row yy + c - PADY lies in the image
#pragma unroll
    for (int c = 0; c <= 2 * PADY; ++c) ym |= (yy + c - PADY >= 0 && yy + c - PADY < h) ? (1u << c) : 0u;
    const int xx = x0 + r;
    const unsigned xm = (xx > 0 ? 1u : 0u) | 2u | (xx + 1 < w ? 4u : 0u);
    const unsigned q = q0 + (unsigned)r;
#pragma unroll
    for (int jj = 0; jj < NV; ++jj) {
      const unsigned c = tcls[jj];
      const bool ok = (c & 128u) == 0 && ((zm >> (c & 3u)) & (ym >> ((c >> 2) & 7u)) & (xm >> ((c >> 5) & 3u)) & 1u) != 0;
      const unsigned short val = reinterpret_cast<const unsigned short*>(x)[ok ? (int)q + toff[jj] : (int)q];
      v[jj] = ok ? val : (unsigned short)0;
    }
  };
  unsigned g = blockIdx.x * 4 + wave;
  unsigned short vn[NV];
  gather(g < ngroups ? g : 0u, vn);
  unsigned char* const sc = scr[wave];
  for (; g < ngroups; g += gstride) {
    bf16x8 bfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      u16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = vn[8 * ks + j];
      bfr[ks] = __builtin_bit_cast(bf16x8, t);
    }
    gather(g + gstride < ngroups ? g + gstride : g, vn);   // next group's taps in flight under this one's epilogue
    const size_t q0 = (size_t)g << 5;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      f32x16 acc;
      M::mma0(acc, afr[cb][0], bfr[0]);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks) M::mma(acc, afr[cb][ks], bfr[ks]);
      // D[row = channel][col = pixel]: lane = pixel r, registers = channels (i&3) + 8*(i>>2) + 4*hh of the block
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const u32x2 pk = {f1_pack(acc[4 * gq], acc[4 * gq + 1]), f1_pack(acc[4 * gq + 2], acc[4 * gq + 3])};
        *reinterpret_cast<u32x2*>(sc + r * 80 + (8 * gq + 4 * hh) * 2) = pk;
      }
      if (stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { s1[cb][i] += acc[i]; s2[cb][i] = fmaf(acc[i], acc[i], s2[cb][i]); }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int chunk = lane + 64 * k;   // 128 chunks of 16 B: pixel = chunk / 4, part = chunk % 4
        const u32x4 tv = *reinterpret_cast<const u32x4*>(sc + (chunk >> 2) * 80 + (chunk & 3) * 16);
        *reinterpret_cast<u32x4*>(y + (q0 + (chunk >> 2)) * F + cb * 32 + (chunk & 3) * 8) = tv;
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const float t1 = reduce32_scatter16(s1[cb], lane);
      const float t2 = reduce32_scatter16(s2[cb], lane);
      if ((lane & 1) == 0) {
        const int reg = scatter16_reg_of_lane(lane);
        const int cl = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        red[wave][0][cl] = t1;
        red[wave][1][cl] = t2;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * F; i += 256) {
      const int st = i / F, c = i % F;
      stats[((size_t)blockIdx.x * 2 + st) * F + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
    }
  }
}

// wgrad: the same (pixel, 8-channel group) mapping, 72 register accumulators, one LDS + atomic
// reduction per workgroup; dY (and y for the fused BN-backward apply) of the next pixel are in flight
// while the current one is accumulated.
// Z3: one depth tap of the Conv3d(1 -> F) weight gradient -- the input taps come from slice d + zshift (compile-time switch:
// the 2-D kernel keeps its instruction stream)
template <int F, bool Z3 = false>
__global__ void __launch_bounds__(256) first_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ dwp, int n, int h, int w,
                                                          const bf16_t* __restrict__ yraw, const float* __restrict__ coef,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int part_mode, int depth = 0, int zshift = 0) {
  constexpr int G = F / 8, PPB = 256 / G;
  __shared__ float sacc[4][9 * F];   // one row per wave: summed in wave order, so the block result does not depend on timing
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const int g = threadIdx.x % G, slot = threadIdx.x / G;
  // fused BN-backward apply: dy = k0*[z>0]*dA + k1*y + k2 (per-channel constants of this thread's group)
  float k0[8], k1[8], k2[8], sc[8], sh[8];
  if (coef) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      k0[j] = coef[g * 8 + j]; k1[j] = coef[F + g * 8 + j]; k2[j] = coef[2 * F + g * 8 + j];
      sc[j] = scale[g * 8 + j]; sh[j] = shift[g * 8 + j];
    }
  }
  const unsigned npix = (unsigned)n * h * w, stride = gridDim.x * PPB;
  unsigned pix = blockIdx.x * PPB + slot;
  const bf16_t* ysrc = coef ? yraw : dy;     // one unconditional load either way
  u32x4 dn, yn;
  float vn[9];
  const F1Step step(stride, w, h, Z3 ? depth : 1);
  F1Pos pos = step.at(pix < npix ? pix : 0u);
  {
    const unsigned q = pix < npix ? pix : 0u;
    dn = *reinterpret_cast<const u32x4*>(dy + (size_t)q * F + g * 8);
    yn = *reinterpret_cast<const u32x4*>(ysrc + (size_t)q * F + g * 8);
    if (Z3) f1_taps_z(x, q, pos, h, w, depth, zshift, vn); else f1_taps(x, q, pos.x, pos.y, h, w, vn);
  }
  for (; pix < npix; pix += stride) {
    const u32x4 d = dn, yq = yn;
    float xv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) xv[t] = vn[t];
    {
      const bool more = pix + stride < npix;
      const unsigned q = more ? pix + stride : pix;
      const F1Pos pn = step.next(pos);
      pos.x = more ? pn.x : pos.x; pos.y = more ? pn.y : pos.y; pos.z = more ? pn.z : pos.z;
      dn = *reinterpret_cast<const u32x4*>(dy + (size_t)q * F + g * 8);
      yn = *reinterpret_cast<const u32x4*>(ysrc + (size_t)q * F + g * 8);
      if (Z3) f1_taps_z(x, q, pos, h, w, depth, zshift, vn); else f1_taps(x, q, pos.x, pos.y, h, w, vn);
    }
    float dv[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dv[2 * j] = __uint_as_float(d[j] << 16); dv[2 * j + 1] = __uint_as_float(d[j] & 0xffff0000u); }
    if (coef) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float yv = (j & 1) ? __uint_as_float(yq[j >> 1] & 0xffff0000u) : __uint_as_float(yq[j >> 1] << 16);
        const float gm = fmaf(yv, sc[j], sh[j]) > 0.f ? dv[j] : 0.f;
        // rounded to the activation dtype, exactly what the unfused apply pass would have stored
        dv[j] = (float)(bf16_t)fmaf(k0[j], gm, fmaf(k1[j], yv, k2[j]));
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = fmaf(dv[j], xv[t], acc[t][j]);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // threads with equal g across the wave: reduce with shuffles over the lanes sharing g first
      float v = acc[t][j];
#pragma unroll
      for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o);
      if ((threadIdx.x & 63) < G) sacc[threadIdx.x >> 6][t * F + g * 8 + j] = v;   // lanes < G hold distinct channel groups
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * F; i += 256) {   // dwp[tap][co][ci = 0]
    const float v = (sacc[0][i] + sacc[1][i]) + (sacc[2][i] + sacc[3][i]);
    if (part_mode) dwp[(size_t)blockIdx.x * 9 * F + i] = v;   // partials mode: slab per workgroup, summed in order by the unpack pass
    else atomicAdd(&dwp[i], v);
  }
}

// wgrad on the matrix pipe (2-D, W % 32 == 0): dW[co][T] = sum over pixels dY[px][co] * x[px + tap T] as
// D[32 co][32 T] += A[co][16 px] * B[16 px][T]: per 32 pixels of a row a wave brings dY (with the fused BatchNorm-backward
// apply, rounded to bf16 exactly as the unfused pass would store it) into a wave-private [pixel][64 B] LDS tile, reads it back
// transposed (ds_read_b64_tr_b16, as wgrad2) and multiplies by the tap matrix -- lane T gathers the 8 consecutive input
// pixels of its tap per k16 step (lanes T >= 9 and rows outside the image contribute zeros).  Replaces 72 FMAs and nine
// bounds-checked loads per pixel and 8-channel group.
typedef short f1_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 f1_tr_frag(const unsigned char* base_lo) {
  typedef __attribute__((address_space(3))) f1_s16x4 lds_s16x4;
  const f1_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
  const f1_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo + 4 * 64));  // pixels +4
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// KD = 3: Conv3d(1 -> F, 3x3x3) -- all 27 taps T = kd*9 + kh*3 + kw in ONE launch (dwp[T][co] = the [3][9][cout] slab of the
// three per-depth-tap launches of the stencil kernel, which read dY three times)
template <int F, int KD>
__global__ void __launch_bounds__(256) first_wgrad_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               float* __restrict__ dwp, int n, int h, int w,
                                                               const bf16_t* __restrict__ yraw, const float* __restrict__ coef,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int part_mode, int depth) {
  constexpr int NTAP = f1_ntap<KD>();   // KD = 7: the 7x3 kernel (see first_fprop_mfma_kernel), dwp = [21][cout]
  constexpr int NB = F / 32, CPP = F / 8, NCH = (32 * CPP) / 64;   // 16-B chunks per pixel / per lane and tensor
  typedef Mma<bf16_t> M;
  __shared__ __attribute__((aligned(16))) unsigned char tile[4][NB][32 * 64];
  __shared__ float sacc[4][NTAP * F];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = lane & 31, khalf = lane >> 5;
  const int g4 = lane >> 4, li = lane & 15;
  const int tr_off = (8 * (g4 >> 1) + (li >> 2)) * 64 + (16 * (g4 & 1) + 4 * (li & 3)) * 2;   // see wgrad2.hip
  // this lane's chunks of the dY tile: chunk c = lane + 64k -> pixel c / CPP, 8-channel group c % CPP (the same for every k)
  const int part = lane % CPP;
  float k0[8], k1[8], k2[8], sc[8], sh[8];
  if (coef) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      k0[j] = coef[part * 8 + j]; k1[j] = coef[F + part * 8 + j]; k2[j] = coef[2 * F + part * 8 + j];
      sc[j] = scale[part * 8 + j]; sh[j] = shift[part * 8 + j];
    }
  }
  const bf16_t* ysrc = coef ? yraw : dy;
  // tap of this lane
  const int t9 = T % 9;
  const int tdz = (KD == 3 && T < NTAP) ? T / 9 - 1 : 0, tdy = T < NTAP ? (KD == 7 ? T / 3 - 3 : t9 / 3 - 1) : 0,
            tdx = T < NTAP ? (KD == 7 ? T % 3 : t9 % 3) - 1 : 0;
  const int toff = tdz * h * w + tdy * w + tdx;
  f32x16 acc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;

  const unsigned npix = (unsigned)n * h * w;
  const unsigned ngroups = npix >> 5, gstride = gridDim.x * 4;
  u32x4 dn[NCH], yn[NCH];
  unsigned short vn[16];
  unsigned vmask = 0, e0 = 0, e31 = 0;
  auto fetch = [&](unsigned g) {
    const unsigned q0 = g << 5;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const unsigned c = lane + 64 * k;
      const size_t off = ((size_t)q0 + c / CPP) * F + part * 8;
      dn[k] = *reinterpret_cast<const u32x4*>(dy + off);
      yn[k] = *reinterpret_cast<const u32x4*>(ysrc + off);
    }
    const unsigned row = q0 / (unsigned)w;
    const int x0 = (int)(q0 - row * (unsigned)w), yy = (int)(row % (unsigned)h);
    const int zz = KD == 3 ? (int)((row / (unsigned)h) % (unsigned)depth) : 0;
    const bool ok = T < NTAP && yy + tdy >= 0 && yy + tdy < h &&
                    (tdz < 0 ? zz > 0 : (tdz > 0 ? zz + 1 < depth : true));
    vmask = ok ? 0xffffffffu : 0u;
    // the only pixels of a 32-pixel row segment whose x neighbour can leave the image: the first (dx = -1) and the last (dx = +1)
    e0 = (tdx < 0 && x0 == 0 && khalf == 0) ? 0xffff0000u : 0xffffffffu;          // element j = 0 of k16 step 0
    e31 = (tdx > 0 && x0 + 32 == w && khalf == 1) ? 0x0000ffffu : 0xffffffffu;     // element j = 7 of k16 step 1
    const long long base = (long long)q0 + 8 * khalf + (ok ? toff : 0);
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      long long idx = base + 16 * (jj / 8) + (jj % 8);
      if (jj == 0) idx = idx < 0 ? 0 : idx;                                   // x[-1] at the very first pixel
      if (jj == 15) idx = idx > (long long)npix - 1 ? (long long)npix - 1 : idx;   // one past the last
      vn[jj] = xs[idx];
    }
  };
  unsigned g = blockIdx.x * 4 + wave;
  fetch(g < ngroups ? g : 0u);
  for (; g < ngroups; g += gstride) {
    // ---- dY tile of this group into the wave's LDS tile ----
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      u32x4 d = dn[k];
      if (coef) {
        const u32x4 yq = yn[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dv[2], yv[2];
          dv[0] = __uint_as_float(d[e] << 16); dv[1] = __uint_as_float(d[e] & 0xffff0000u);
          yv[0] = __uint_as_float(yq[e] << 16); yv[1] = __uint_as_float(yq[e] & 0xffff0000u);
          float o[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int j = 2 * e + u;
            const float gm = fmaf(yv[u], sc[j], sh[j]) > 0.f ? dv[u] : 0.f;
            o[u] = fmaf(k0[j], gm, fmaf(k1[j], yv[u], k2[j]));   // rounded to bf16 by the pack: what the unfused pass stores
          }
          d[e] = f1_pack(o[0], o[1]);
        }
      }
      const unsigned c = lane + 64 * k;
      const int px = c / CPP;
      *reinterpret_cast<u32x4*>(&tile[wave][part / 4][px * 64 + (part % 4) * 16]) = d;
    }
    // ---- tap matrix fragments ----
    bf16x8 bfr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = ((unsigned)vn[8 * ks + 2 * e] | ((unsigned)vn[8 * ks + 2 * e + 1] << 16)) & vmask;
      if (ks == 0) pk[0] &= e0; else pk[3] &= e31;
      bfr[ks] = __builtin_bit_cast(bf16x8, pk);
    }
    fetch(g + gstride < ngroups ? g + gstride : g);   // next group's loads in flight under the MFMAs
    asm volatile("" ::: "memory");                    // the transposed reads below must follow the tile writes above
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 afr = f1_tr_frag(&tile[wave][cb][0] + tr_off + ks * 16 * 64);
        M::mma(acc[cb], afr, bfr[ks]);
      }
    asm volatile("" ::: "memory");
  }
  // D[row = co][col = T]: lane = tap T, registers = channels (i&3) + 8*(i>>2) + 4*khalf of the block
  if (T < NTAP) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[wave][T * F + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * khalf] = acc[cb][i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NTAP * F; i += 256) {   // dwp[tap][co][ci = 0], waves summed in a fixed order
    const float v = (sacc[0][i] + sacc[1][i]) + (sacc[2][i] + sacc[3][i]);
    if (part_mode) dwp[(size_t)blockIdx.x * NTAP * F + i] = v;
    else atomicAdd(&dwp[i], v);
  }
}

static bool f1_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_DISABLE_V2"); on = (e && e[0] == '1') ? 0 : 1; }
  return on == 1;
}
static bool first_ok(int dtype, int c0, int c1, int cout, int taps) {
  return f1_enabled() && dtype == OCT_DT_BF16 && c0 == 1 && c1 == 0 && taps == 9 && (cout == 16 || cout == 32 || cout == 64);
}
static bool first_mfma_enabled() {
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("OCT_FIRST_MFMA"); use_mfma = (e && e[0] == '0') ? 0 : 1; }
  return use_mfma == 1;
}
// ReLayNet's first layer, Conv2d(1 -> F, 7x3): the matrix-pipe kernels only (W % 32 == 0, F = 32 or 64)
static bool first73_ok(int dtype, int c0, int c1, int cout, int taps, int kh, int kw, int w, int depth, size_t npix) {
  return f1_enabled() && first_mfma_enabled() && dtype == OCT_DT_BF16 && c0 == 1 && c1 == 0 && taps == 21 && kh == 7 && kw == 3 &&
         (cout == 32 || cout == 64) && (w % 32) == 0 && depth == 0 && npix < (1ull << 31);
}
static int first_grid(const OctConvDesc* d) {
  const size_t work = (size_t)d->n * d->h * d->w * (d->cout / 8);   // a thread per (pixel, 8-channel group)
  size_t b = (work + 255) / 256;
  if (b > 4096) b = 4096;
  return (int)b;
}

int oct_first_stat_rows(const OctConvDesc* d) {
  const bool k73 = first73_ok(d->dtype, d->c0, d->c1, d->cout, d->taps, d->kh, d->kw, d->w, d->depth, (size_t)d->n * d->h * d->w);
  if (d->kh == 7 && !k73) return -1;
  if ((!k73 && !first_ok(d->dtype, d->c0, d->c1, d->cout, d->taps)) || d->in_mode || d->out_mode || d->xform0 || d->split) return -1;
  return first_grid(d);
}

int oct_first_fprop(const OctConvDesc* d, const OctConvArgs* a, void* stream) {
  if (oct_first_stat_rows(d) < 0) return 0;
  const int grid = first_grid(d);
  hipStream_t s = as_stream(stream);
  float* st = d->want_stats ? a->stat_partials : nullptr;
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("OCT_FIRST_MFMA"); use_mfma = (e && e[0] == '0') ? 0 : 1; }
  if (use_mfma && (d->w % 32) == 0 && d->cout >= 32 && (size_t)d->n * d->h * d->w < (1ull << 31)) {
    // matrix-pipe kernel: a wave per 32 consecutive pixels of a row
#define LAUNCHM(F, KD) hipLaunchKernelGGL((first_fprop_mfma_kernel<F, KD>), dim3(grid), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                          (const bf16_t*)a->wpacked, (bf16_t*)a->y0, st, d->n, d->h, d->w, d->depth)
    if (d->kh == 7) { if (d->cout == 32) LAUNCHM(32, 7); else LAUNCHM(64, 7); }
    else if (d->depth > 0) { if (d->cout == 32) LAUNCHM(32, 3); else LAUNCHM(64, 3); }
    else { if (d->cout == 32) LAUNCHM(32, 1); else LAUNCHM(64, 1); }
#undef LAUNCHM
    int rcm = oct_check_launch("first_fprop_mfma");
    return rcm ? rcm : 1;
  }
  if (d->depth > 0) {
    if ((size_t)d->n * d->h * d->w >= (1ull << 32)) return 0;   // 32-bit voxel arithmetic
#define LAUNCH3(F) hipLaunchKernelGGL(first_fprop3d_kernel<F>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                      (const bf16_t*)a->wpacked, (bf16_t*)a->y0, st, d->n, d->h, d->w, d->depth)
    if (d->cout == 16) LAUNCH3(16); else if (d->cout == 32) LAUNCH3(32); else LAUNCH3(64);
#undef LAUNCH3
    int rc3 = oct_check_launch("first_fprop3d");
    return rc3 ? rc3 : 1;
  }
#define LAUNCH(F) hipLaunchKernelGGL(first_fprop_kernel<F>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                     (const bf16_t*)a->wpacked,                                      (bf16_t*)a->y0, st, d->n, d->h, d->w)
  if (d->cout == 16) LAUNCH(16); else if (d->cout == 32) LAUNCH(32); else LAUNCH(64);
#undef LAUNCH
  int rc = oct_check_launch("first_fprop");
  return rc ? rc : 1;
}

// Host query behind OctWgradArgs.dy_coef: whether oct_conv_wgrad can apply the BatchNorm backward on load
// for this descriptor (today: the direct first-layer kernel; OCT_DISABLE_V2=1 switches it off with the
// other pipelined kernels, and the caller then materialises dY with oct_bn_bwd_apply).
extern "C" int oct_conv_wgrad_fused_apply_ok(const OctWgradDesc* d) {
  if (!d) return 0;
  if (!first_ok(d->dtype, d->c0, d->c1, d->cout, d->taps) || d->xform0 || d->dy_mode) return 0;
  return (size_t)d->n * d->h * d->w < (1u << 31) ? 1 : 0;
}

// Host query: 1 when oct_conv_wgrad accepts in_img_shift = OCT_IMG_SHIFT_ALL for this descriptor (all three depth taps of a
// Conv3d(1 -> F) weight gradient in one launch, dwp = [3][9][cout]); else the caller launches once per depth tap.
extern "C" int oct_conv_wgrad_all_depth_taps_ok(const OctWgradDesc* d) {
  if (!d || d->depth <= 0 || d->partials) return 0;
  if (!first_ok(d->dtype, d->c0, d->c1, d->cout, d->taps) || d->xform0 || d->dy_mode) return 0;
  const char* e = getenv("OCT_FIRST_MFMA");
  if (e && e[0] == '0') return 0;
  return ((d->w % 32) == 0 && d->cout >= 32 && (size_t)d->n * d->h * d->w < (1u << 31)) ? 1 : 0;
}

int oct_first_wgrad(const OctWgradDesc* d, const OctWgradArgs* a, void* stream, int* query) {
  const bool k73 = first73_ok(d->dtype, d->c0, d->c1, d->cout, d->taps, d->kh, d->kw, d->w, d->depth, (size_t)d->n * d->h * d->w);
  if (d->kh == 7 && (!k73 || (!query && a->dy_coef))) return 0;
  if ((!k73 && !first_ok(d->dtype, d->c0, d->c1, d->cout, d->taps)) || d->xform0 || d->dy_mode) return 0;
  if (!query && a->dbias) return 0;   // no bias-gradient path in the direct kernel: the MFMA kernels take it
  if (!query && a->dy_coef && (!a->dy_y || !a->dy_scale || !a->dy_shift)) { oct_set_error("oct_conv_wgrad: fused apply needs y, scale, shift"); return OCT_E_INVALID; }
  const size_t total = (size_t)d->n * d->h * d->w * (d->cout / 8);
  if ((size_t)d->n * d->h * d->w >= (1u << 31)) return 0;   // 32-bit pixel arithmetic in the kernel
  size_t b = (total + 255) / 256;
  if (b > 512) b = 512;   // two workgroups per CU: 0.54 ms against 0.62 at 4096 (fprop, write-dominated, prefers 2048-4096)
  if (query) { *query = (int)b; return 1; }
  const int part_mode = d->partials ? 1 : 0;
  hipStream_t s = as_stream(stream);
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("OCT_FIRST_MFMA"); use_mfma = (e && e[0] == '0') ? 0 : 1; }
  const bool all_taps = d->depth > 0 && d->in_img_shift == OCT_IMG_SHIFT_ALL;
  if (use_mfma && (d->depth == 0 || all_taps) && (d->w % 32) == 0 && d->cout >= 32) {
#define LAUNCHM(F, KD) hipLaunchKernelGGL((first_wgrad_mfma_kernel<F, KD>), dim3((int)b), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                          (const bf16_t*)a->dy, a->dwp, d->n, d->h, d->w, \
                                          (const bf16_t*)a->dy_y, a->dy_coef, a->dy_scale, a->dy_shift, part_mode, d->depth)
    if (k73) { if (d->cout == 32) LAUNCHM(32, 7); else LAUNCHM(64, 7); }
    else if (all_taps) { if (d->cout == 32) LAUNCHM(32, 3); else LAUNCHM(64, 3); }
    else { if (d->cout == 32) LAUNCHM(32, 1); else LAUNCHM(64, 1); }
#undef LAUNCHM
    int rcm = oct_check_launch("first_wgrad_mfma");
    return rcm ? rcm : 1;
  }
  if (all_taps) return 0;   // only the matrix-pipe kernel takes all depth taps at once (oct_conv_wgrad_all_depth_taps_ok)
#define LAUNCH(F) hipLaunchKernelGGL(first_wgrad_kernel<F>, dim3((int)b), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                     (const bf16_t*)a->dy, a->dwp, d->n, d->h, d->w, \
                                     (const bf16_t*)a->dy_y, a->dy_coef, a->dy_scale, a->dy_shift, part_mode)
#define LAUNCHZ(F) hipLaunchKernelGGL((first_wgrad_kernel<F, true>), dim3((int)b), dim3(256), 0, s, (const bf16_t*)a->x0, \
                                      (const bf16_t*)a->dy, a->dwp, d->n, d->h, d->w, \
                                      (const bf16_t*)a->dy_y, a->dy_coef, a->dy_scale, a->dy_shift, part_mode, d->depth, d->in_img_shift)
  if (d->depth > 0) { if (d->cout == 16) LAUNCHZ(16); else if (d->cout == 32) LAUNCHZ(32); else LAUNCHZ(64); }
  else if (d->cout == 16) LAUNCH(16); else if (d->cout == 32) LAUNCH(32); else LAUNCH(64);
#undef LAUNCH
#undef LAUNCHZ
  int rc = oct_check_launch("first_wgrad");
  return rc ? rc : 1;
}
