// Implicit-GEMM convolution on MFMA 32x32 tiles (gfx950).
//
//   D[row = output channel][col = pixel] += sum_k  W[row][k] * X[k][pixel]
//
// The activations are the MFMA "B" operand: a (TH+2)x(32+2) halo tile of KC input channels is
// staged ONCE in LDS (BN-apply + ReLU fused into the staging pass, concat and space-to-depth
// resolved there) and then read nine times, once per tap, with one ds_read_b128 per fragment --
// this is the im2col, never materialised.  The weights are the "A" operand: they are pre-packed
// in fragment order, so a wave fetches one fragment with a single fully coalesced 1 KiB load.
// Because the pixel index lives on the lane and four consecutive output channels live in four
// consecutive accumulator registers, the epilogue stores NHWC directly from registers
// (8 B/lane bf16, 16 B/lane fp32) and reduces the BatchNorm statistics with a transposing
// butterfly across the 32 pixel lanes.
#include "common.h"

struct IgemmParams {
  const void* x0; const void* x1;
  const float* sc0; const float* sh0; const float* sc1; const float* sh1;
  const void* wp; const float* bias;
  void* y0; void* y1; float* stats;
  int n, h, w;
  int c0, c1, ktot, nk16;
  int cout, nb32;
  int xf0, xf1, in_mode, out_mode, split;
  int tiles_x, tiles_y;
  // 3-D (volumes stored as n = N*D images): depth = D enables the depth taps.  Plain input: K = 3 * csrc, tap kdi reads
  // image img + kdi - 1 of the same volume (zero outside); S2D input (ConvTranspose3d dgrad): K = 8 * c0, k = (kdi, dy, dx, c)
  // gathers from image 2*img + kdi.  D2S output: image index = img * oimg_mul + oimg_add (ConvTranspose3d: 2*img + kdi).
  int depth, csrc, oimg_mul, oimg_add;
};

// ---- staging: 8 consecutive K-channels of one input pixel, transformed, as floats ------------
template <typename T>
__device__ __forceinline__ float fetch1(const IgemmParams& p, int img, int iy, int ix, int k) {
  if (k >= p.ktot) return 0.f;
  if (p.in_mode == OCT_IN_S2D) {
    int dydx = k / p.c0;
    const int c = k - dydx * p.c0;
    if (p.depth > 0) { img = 2 * img + (dydx >> 2); dydx &= 3; }
    const size_t pix = ((size_t)img * (2 * p.h) + (2 * iy + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * ix + (dydx & 1));
    float v = to_f32(reinterpret_cast<const T*>(p.x0)[pix * p.c0 + c]);
    if (p.xf0) v = fmaxf(fmaf(v, p.sc0[c], p.sh0[c]), xf_floor(p.xf0));
    return v;
  }
  if (p.depth > 0) {
    const int kdi = k / p.csrc;
    k -= kdi * p.csrc;
    const int dz = img % p.depth + kdi - 1;
    if (dz < 0 || dz >= p.depth) return 0.f;
    img += kdi - 1;
  }
  const size_t pix = ((size_t)img * p.h + iy) * (size_t)p.w + ix;
  if (k < p.c0) {
    float v = to_f32(reinterpret_cast<const T*>(p.x0)[pix * p.c0 + k]);
    if (p.xf0) v = fmaxf(fmaf(v, p.sc0[k], p.sh0[k]), xf_floor(p.xf0));
    return v;
  }
  const int c = k - p.c0;
  float v = to_f32(reinterpret_cast<const T*>(p.x1)[pix * p.c1 + c]);
  if (p.xf1) v = fmaxf(fmaf(v, p.sc1[c], p.sh1[c]), xf_floor(p.xf1));
  return v;
}

template <typename T>
__device__ __forceinline__ void fetch8(const IgemmParams& p, int img, int iy, int ix, int k, float (&v)[8]) {
  // fast paths: the 8-group lies inside one source and is 8-aligned there
  const T* src = nullptr; const float* sc = nullptr; const float* sh = nullptr; int c = 0; int cs = 0; bool xf = false;
  float lo = 0.f;   // clamp of the transform (xf_floor)
  size_t pix = 0;
  if (p.in_mode == OCT_IN_S2D) {
    if ((p.c0 & 7) == 0 && k + 8 <= p.ktot) {
      int dydx = k / p.c0; c = k - dydx * p.c0; cs = p.c0;
      int im = img;
      if (p.depth > 0) { im = 2 * img + (dydx >> 2); dydx &= 3; }
      pix = ((size_t)im * (2 * p.h) + (2 * iy + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * ix + (dydx & 1));
      src = reinterpret_cast<const T*>(p.x0); sc = p.sc0; sh = p.sh0; xf = p.xf0 != 0; lo = xf_floor(p.xf0);
    }
  } else if (p.depth > 0 && ((p.csrc & 7) != 0 || k + 8 > p.ktot)) {
    // depth taps on a channel count that is not a multiple of 8 (the 1-channel input volume): element by element
  } else {
    int kk = k, im = img;
    if (p.depth > 0) {
      const int kdi = k / p.csrc;
      kk = k - kdi * p.csrc;
      const int dz = img % p.depth + kdi - 1;
      if (dz < 0 || dz >= p.depth) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        return;
      }
      im = img + kdi - 1;
    }
    const int k = kk;   // channel inside the (virtually concatenated) source of this depth tap
    pix = ((size_t)im * p.h + iy) * (size_t)p.w + ix;
    if ((p.c0 & 7) == 0 && k + 8 <= p.c0) {
      src = reinterpret_cast<const T*>(p.x0); c = k; cs = p.c0; sc = p.sc0; sh = p.sh0; xf = p.xf0 != 0; lo = xf_floor(p.xf0);
    } else if ((p.c0 & 7) == 0 && (p.c1 & 7) == 0 && k >= p.c0 && k + 8 <= p.c0 + p.c1) {
      src = reinterpret_cast<const T*>(p.x1); c = k - p.c0; cs = p.c1; sc = p.sc1; sh = p.sh1; xf = p.xf1 != 0; lo = xf_floor(p.xf1);
    }
  }
  if (src) {
    load_vec<T, 8>(src + pix * cs + c, v);
    if (xf) {
      float s[8], b[8];
      load_vec<float, 4>(sc + c, reinterpret_cast<float(&)[4]>(s[0]));
      load_vec<float, 4>(sc + c + 4, reinterpret_cast<float(&)[4]>(s[4]));
      load_vec<float, 4>(sh + c, reinterpret_cast<float(&)[4]>(b[0]));
      load_vec<float, 4>(sh + c + 4, reinterpret_cast<float(&)[4]>(b[4]));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], s[j], b[j]), lo);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fetch1<T>(p, img, iy, ix, k + j);
  }
}

// ---- epilogue: four consecutive GEMM rows (output channels) of one pixel ------------------------
template <typename T>
__device__ __forceinline__ void store4(const IgemmParams& p, int img, int oy, int ox, int cb, const float (&v)[4]) {
  if (p.out_mode == OCT_OUT_D2S) {
    if (p.oimg_mul) img = img * p.oimg_mul + p.oimg_add;
    const int cr = p.cout >> 2;  // real output channels
    if ((cr & 3) == 0 && cb + 3 < p.cout) {
      const int dydx = cb / cr, co = cb - dydx * cr;
      const size_t pix = ((size_t)img * (2 * p.h) + (2 * oy + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * ox + (dydx & 1));
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = v[j] + (p.bias ? p.bias[co + j] : 0.f);
      store_vec<T, 4>(reinterpret_cast<T*>(p.y0) + pix * cr + co, o);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nn = cb + j;
        if (nn < p.cout) {
          const int dydx = nn / cr, co = nn - dydx * cr;
          const size_t pix = ((size_t)img * (2 * p.h) + (2 * oy + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * ox + (dydx & 1));
          reinterpret_cast<T*>(p.y0)[pix * cr + co] = from_f32<T>(v[j] + (p.bias ? p.bias[co] : 0.f));
        }
      }
    }
    return;
  }
  const size_t pix = ((size_t)img * p.h + oy) * (size_t)p.w + ox;
  const int c_a = p.split > 0 ? p.split : p.cout;  // channels of destination 0
  const int c_b = p.cout - c_a;
  if (cb + 3 < c_a && (c_a & 3) == 0) {
    store_vec<T, 4>(reinterpret_cast<T*>(p.y0) + pix * c_a + cb, v);
  } else if (cb >= c_a && cb + 3 < p.cout && (c_a & 3) == 0 && (c_b & 3) == 0) {
    store_vec<T, 4>(reinterpret_cast<T*>(p.y1) + pix * c_b + (cb - c_a), v);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cb + j;
      if (c < c_a) reinterpret_cast<T*>(p.y0)[pix * c_a + c] = from_f32<T>(v[j]);
      else if (c < p.cout) reinterpret_cast<T*>(p.y1)[pix * c_b + (c - c_a)] = from_f32<T>(v[j]);
    }
  }
}

// KH x KW kernel, stride 1, "same" padding ((KH-1)/2, (KW-1)/2): 3x3 and 1x1 for the U-Nets, 7x3 for ReLayNet's
// BasicBlock (ReLayNet_2017.py:155-160).  tap = ky*KW + kx in the packed weights.
template <typename T, int KH, int KW, int WM, int WN, int MF, int NF, int KC>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmParams p) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int TAPS = KH * KW;
  constexpr int TH = WM * MF, TW = 32;
  constexpr int HALO_H = (KH - 1) / 2, HALO_W = (KW - 1) / 2;
  constexpr int LH = TH + KH - 1, LW = TW + KW - 1;
  constexpr int PIXB = KC * (int)sizeof(T) + 16;  // LDS bytes per pixel (+16: conflict-free b128 reads)
  constexpr int NT = WN * NF * 32;
  constexpr int GROUPS = KC / 8;
  typedef Mma<T> M;
  typedef typename M::Frag Frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char ig_smem[];   // halo tile, then [WM][2][NT] statistics
  unsigned char* const tile = ig_smem;
  float (*const wg_stats)[2][NT] = reinterpret_cast<float (*)[2][NT]>(ig_smem + LH * LW * PIXB);
  static_assert((LH * LW * PIXB) % 16 == 0, "statistics scratch stays aligned");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int bx = blockIdx.x;
  const int txi = bx % p.tiles_x; bx /= p.tiles_x;
  const int tyi = bx % p.tiles_y; const int img = bx / p.tiles_y;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int nb0 = blockIdx.y * (NT / 32) + wn * NF;  // first 32-row block of this wave

  f32x16 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < NF; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const T* wp = reinterpret_cast<const T*>(p.wp);
  const int nchunks = (p.nk16 * 16 + KC - 1) / KC;
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    // ---- stage the halo tile of channels [ch*KC, ch*KC+KC) ----
    for (int idx = tid; idx < LH * LW * GROUPS; idx += 256) {
      const int pix = idx / GROUPS, grp = idx - pix * GROUPS;
      const int ly = pix / LW, lx = pix - ly * LW;
      const int iy = y0 + ly - HALO_H, ix = x0 + lx - HALO_W;
      float v[8];
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {
        fetch8<T>(p, img, iy, ix, ch * KC + grp * 8, v);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      store_vec<T, 8>(reinterpret_cast<T*>(tile + pix * PIXB) + grp * 8, v);
    }
    __syncthreads();
    // ---- MFMA over taps x k16 steps ----
    constexpr int TAP_UNROLL = TAPS <= 9 ? TAPS : 1;   // 21 taps x 2 steps x 8 fragment pairs: keep the big kernels rolled
#pragma unroll TAP_UNROLL
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ty = tap / KW, tx = tap % KW;
#pragma unroll
      for (int k16 = 0; k16 < KC / 16; ++k16) {
        const int kk = ch * (KC / 16) + k16;
        if (kk < p.nk16) {
          Frag wf[NF];
#pragma unroll
          for (int q = 0; q < NF; ++q) {
            const int nb = nb0 + q;
            if (nb < p.nb32)
              wf[q] = M::load(wp + ((size_t)(nb * TAPS + tap) * p.nk16 + kk) * 512 + lane * 8);
            else
              wf[q] = M::zero();
          }
          Frag xf[MF];
#pragma unroll
          for (int m = 0; m < MF; ++m) {
            const int row = wm * MF + m;
            xf[m] = M::load(tile + ((row + ty) * LW + (r + tx)) * PIXB + (k16 * 16 + 8 * hh) * (int)sizeof(T));
          }
#pragma unroll
          for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int q = 0; q < NF; ++q) M::mma(acc[m][q], wf[q], xf[m]);
        }
      }
    }
  }

  // ---- epilogue: NHWC stores straight from the accumulators ----
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int oy = y0 + wm * MF + m, ox = x0 + r;
    const bool valid = (oy < p.h) && (ox < p.w);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int nb = nb0 + q;
      if (valid && nb < p.nb32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cb = nb * 32 + 8 * g + 4 * hh;
          if (cb < p.cout) {
            const float v[4] = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
            store4<T>(p, img, oy, ox, cb, v);
          }
        }
      }
    }
  }

  // ---- BatchNorm partial statistics of the fp32 outputs ----
  if (p.stats) {
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      float s1[16], s2[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int oy = y0 + wm * MF + m, ox = x0 + r;
        const float msk = ((oy < p.h) && (ox < p.w)) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float a = acc[m][q][i] * msk;
          s1[i] += a;
          s2[i] = fmaf(a, a, s2[i]);
        }
      }
      const float t1 = reduce32_scatter16(s1, lane);
      const float t2 = reduce32_scatter16(s2, lane);
      if ((lane & 1) == 0) {
        const int reg = scatter16_reg_of_lane(lane);
        const int cl = (wn * NF + q) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        wg_stats[wm][0][cl] = t1;
        wg_stats[wm][1][cl] = t2;
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NT; i += 256) {
      const int st = i / NT, cl = i - st * NT;
      const int c = blockIdx.y * NT + cl;
      if (c < p.cout) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += wg_stats[w][st][cl];
        p.stats[((size_t)blockIdx.x * 2 + st) * p.cout + c] = s;
      }
    }
  }
}

// ---- weight packing -----------------------------------------------------------------------------
// wp[((nb*TAPS + tap)*nk16 + kk)*512 + lane*8 + j] = A[row = nb*32 + (lane&31)][tap][k = kk*16 + 8*(lane>>5) + j]
// mode OCT_PACK_CONV_FPROP / _DGRAD with taps = kh*kw of any kernel size (torch layout (Cout,Cin,kh,kw)): the dgrad
// filter is the point reflection of the kernel, i.e. the reversed flat tap index
template <typename T>
__global__ void pack_weights_kernel(int mode, const float* __restrict__ w, T* __restrict__ wp, int cout, int cin,
                                    int rows, int taps, int kch, int nk16, size_t total, int kdi = 0) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    const int lane = (i >> 3) & 63;
    size_t rest = i >> 9;
    const int kk = rest % nk16; rest /= nk16;
    const int tap = rest % taps;
    const int nb = rest / taps;
    const int row = nb * 32 + (lane & 31);
    const int k = kk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (row < rows && k < kch) {
      if (mode == OCT_PACK_CONV_FPROP) {            // row = co, k = ci
        v = w[((size_t)row * cin + k) * taps + tap];
      } else if (mode == OCT_PACK_CONV_DGRAD) {     // row = ci, k = co, flipped tap
        v = w[((size_t)k * cin + row) * taps + (taps - 1 - tap)];
      } else if (mode == OCT_PACK_DECONV_FPROP) {   // row = dydx*cout + co, k = ci
        const int dydx = row / cout, co = row - dydx * cout;
        v = w[((size_t)k * cout + co) * 4 + dydx];
      } else if (mode == OCT_PACK_DECONV_DGRAD) {   // row = ci, k = dydx*cout + co
        const int dydx = k / cout, co = k - dydx * cout;
        v = w[((size_t)row * cout + co) * 4 + dydx];
      } else if (mode == OCT_PACK_1X1_DGRAD) {      // row = ci, k = co
        v = w[(size_t)k * cin + row];
      } else if (mode == OCT_PACK_CONV3D_FPROP) {   // (Cout,Cin,3,3,3): row = co, k = (kd, ci), tap = (kh, kw)
        const int kd = k / cin, ci = k - kd * cin;
        v = w[(((size_t)row * cin + ci) * 3 + kd) * 9 + tap];
      } else if (mode == OCT_PACK_CONV3D_DGRAD) {   // row = ci, k = (kd, co), kernel point-reflected in all three axes
        const int kd = k / cout, co = k - kd * cout;
        v = w[(((size_t)co * cin + row) * 3 + (2 - kd)) * 9 + (8 - tap)];
      } else if (mode == OCT_PACK_DECONV3D_FPROP) { // (Cin,Cout,2,2,2), depth slice kdi: row = dydx*cout + co, k = ci
        const int dydx = row / cout, co = row - dydx * cout;
        v = w[(((size_t)k * cout + co) * 2 + kdi) * 4 + dydx];
      } else if (mode == OCT_PACK_DECONV3D_DGRAD) { // row = ci, k = (kd, dydx, co)
        const int q = k / cout, co = k - q * cout;
        v = w[(((size_t)row * cout + co) * 2 + (q >> 2)) * 4 + (q & 3)];
      } else {                                      // 1X1_FPROP: row = co, k = ci
        v = w[(size_t)row * cin + k];
      }
    }
    wp[i] = from_f32<T>(v);
  }
}

static void pack_dims(int mode, int cout, int cin, int* rows, int* taps, int* kch) {
  switch (mode) {
    case OCT_PACK_CONV_FPROP: *rows = cout; *taps = 9; *kch = cin; break;
    case OCT_PACK_CONV_DGRAD: *rows = cin; *taps = 9; *kch = cout; break;
    case OCT_PACK_DECONV_FPROP: *rows = 4 * cout; *taps = 1; *kch = cin; break;
    case OCT_PACK_DECONV_DGRAD: *rows = cin; *taps = 1; *kch = 4 * cout; break;
    case OCT_PACK_1X1_DGRAD: *rows = cin; *taps = 1; *kch = cout; break;
    case OCT_PACK_CONV3D_FPROP: *rows = cout; *taps = 9; *kch = 3 * cin; break;
    case OCT_PACK_CONV3D_DGRAD: *rows = cin; *taps = 9; *kch = 3 * cout; break;
    case OCT_PACK_DECONV3D_FPROP: *rows = 4 * cout; *taps = 1; *kch = cin; break;
    case OCT_PACK_DECONV3D_DGRAD: *rows = cin; *taps = 1; *kch = 8 * cout; break;
    default: *rows = cout; *taps = 1; *kch = cin; break;
  }
}

extern "C" size_t oct_packed_weight_elems(int rows, int taps, int kch) {
  return (size_t)ceil_div(rows, 32) * taps * ceil_div(kch, 16) * 512;
}

static int pack_weights_impl(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int taps_kk, void* stream, int kdi = 0);
extern "C" int oct_pack_weights(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, void* stream) {
  OCT_CHECK(mode >= 0 && mode <= 5, "oct_pack_weights: bad mode %d", mode);
  return pack_weights_impl(mode, dtype, w, wpacked, cout, cin, 0, stream);
}
extern "C" int oct_pack_weights3d(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int kdi, void* stream) {
  OCT_CHECK(mode >= OCT_PACK_CONV3D_FPROP && mode <= OCT_PACK_DECONV3D_DGRAD, "oct_pack_weights3d: bad mode %d", mode);
  OCT_CHECK(kdi == 0 || (mode == OCT_PACK_DECONV3D_FPROP && kdi == 1), "oct_pack_weights3d: kdi selects the depth slice of DECONV3D_FPROP (0 or 1)");
  return pack_weights_impl(mode, dtype, w, wpacked, cout, cin, 0, stream, kdi);
}
extern "C" int oct_pack_weights_kk(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int kh, int kw,
                                   void* stream) {
  OCT_CHECK(mode == OCT_PACK_CONV_FPROP || mode == OCT_PACK_CONV_DGRAD, "oct_pack_weights_kk: mode must be CONV_FPROP or CONV_DGRAD");
  OCT_CHECK(kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && kh * kw <= 49, "oct_pack_weights_kk: odd kernel sizes up to 7x7 (got %dx%d)", kh, kw);
  return pack_weights_impl(mode, dtype, w, wpacked, cout, cin, kh * kw, stream);
}
static int pack_weights_impl(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int taps_kk, void* stream, int kdi) {
  OCT_CHECK(mode >= 0 && mode <= OCT_PACK_DECONV3D_DGRAD, "oct_pack_weights: bad mode %d", mode);
  OCT_CHECK(cout > 0 && cin > 0 && w && wpacked, "oct_pack_weights: bad arguments");
  int rows, taps, kch;
  pack_dims(mode, cout, cin, &rows, &taps, &kch);
  if (taps_kk > 0) taps = taps_kk;
  const int nk16 = ceil_div(kch, 16);
  const size_t total = oct_packed_weight_elems(rows, taps, kch);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == OCT_DT_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), mode, w,
                       (bf16_t*)wpacked, cout, cin, rows, taps, kch, nk16, total, kdi);
  else if (dtype == OCT_DT_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), mode, w,
                       (float*)wpacked, cout, cin, rows, taps, kch, nk16, total, kdi);
  else
    OCT_CHECK(false, "oct_pack_weights: bad dtype %d", dtype);
  return oct_check_launch("pack_weights");
}

// All re-packings of a training step in ONE launch: the ~43 five-microsecond pack launches that follow
// every optimizer update were 0.2 ms of the step.  The job table travels in the kernel arguments.
struct PackJobs { OctPackJob j[OCT_PACK_BATCH_MAX]; };
template <typename T>
__global__ void pack_weights_batch_kernel(const PackJobs jobs) {
  const OctPackJob jb = jobs.j[blockIdx.y];
  int rows, taps, kch;
  switch (jb.mode) {
    case OCT_PACK_CONV_FPROP: rows = jb.cout; taps = 9; kch = jb.cin; break;
    case OCT_PACK_CONV_DGRAD: rows = jb.cin; taps = 9; kch = jb.cout; break;
    case OCT_PACK_DECONV_FPROP: rows = 4 * jb.cout; taps = 1; kch = jb.cin; break;
    case OCT_PACK_DECONV_DGRAD: rows = jb.cin; taps = 1; kch = 4 * jb.cout; break;
    case OCT_PACK_1X1_DGRAD: rows = jb.cin; taps = 1; kch = jb.cout; break;
    default: rows = jb.cout; taps = 1; kch = jb.cin; break;
  }
  const int nk16 = (kch + 15) / 16;
  const size_t total = (size_t)((rows + 31) / 32) * taps * nk16 * 512;
  const float* __restrict__ w = jb.w;
  T* __restrict__ wp = reinterpret_cast<T*>(jb.wpacked);
  const int cout = jb.cout, cin = jb.cin, mode = jb.mode;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    const int lane = (i >> 3) & 63;
    size_t rest = i >> 9;
    const int kk = rest % nk16; rest /= nk16;
    const int tap = rest % taps;
    const int nb = rest / taps;
    const int row = nb * 32 + (lane & 31);
    const int k = kk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (row < rows && k < kch) {
      if (mode == OCT_PACK_CONV_FPROP) v = w[((size_t)row * cin + k) * 9 + tap];
      else if (mode == OCT_PACK_CONV_DGRAD) v = w[((size_t)k * cin + row) * 9 + (8 - tap)];
      else if (mode == OCT_PACK_DECONV_FPROP) { const int dydx = row / cout, co = row - dydx * cout; v = w[((size_t)k * cout + co) * 4 + dydx]; }
      else if (mode == OCT_PACK_DECONV_DGRAD) { const int dydx = k / cout, co = k - dydx * cout; v = w[((size_t)row * cout + co) * 4 + dydx]; }
      else if (mode == OCT_PACK_1X1_DGRAD) v = w[(size_t)k * cin + row];
      else v = w[(size_t)row * cin + k];
    }
    wp[i] = from_f32<T>(v);
  }
}

extern "C" int oct_pack_weights_batch(int dtype, int count, const OctPackJob* jobs, void* stream) {
  OCT_CHECK(count >= 0 && (count == 0 || jobs), "oct_pack_weights_batch: bad arguments");
  OCT_CHECK(dtype == OCT_DT_BF16 || dtype == OCT_DT_F32, "oct_pack_weights_batch: bad dtype %d", dtype);
  for (int base = 0; base < count; base += OCT_PACK_BATCH_MAX) {
    PackJobs pj;
    const int n = count - base < OCT_PACK_BATCH_MAX ? count - base : OCT_PACK_BATCH_MAX;
    for (int i = 0; i < n; ++i) {
      pj.j[i] = jobs[base + i];
      OCT_CHECK(pj.j[i].mode >= 0 && pj.j[i].mode <= 5 && pj.j[i].cout > 0 && pj.j[i].cin > 0 && pj.j[i].w && pj.j[i].wpacked,
                "oct_pack_weights_batch: bad job %d", base + i);
    }
    if (dtype == OCT_DT_BF16)
      hipLaunchKernelGGL(pack_weights_batch_kernel<bf16_t>, dim3(96, n), dim3(256), 0, as_stream(stream), pj);
    else
      hipLaunchKernelGGL(pack_weights_batch_kernel<float>, dim3(96, n), dim3(256), 0, as_stream(stream), pj);
  }
  return oct_check_launch("pack_weights_batch");
}

// ---- host dispatch ------------------------------------------------------------------------------
struct TileCfg { int th; int nt; };
static TileCfg pick_cfg(int cout) {
  if (cout <= 32) return {8, 32};
  if (cout <= 64) return {8, 64};
  return {8, 128};
}

extern "C" int oct_conv_stat_blocks(const OctConvDesc* d) {
  if (!d) return 0;
  {
    const int f1 = oct_first_stat_rows(d);
    if (f1 >= 0) return f1;
    const int v2 = oct_conv_v2_stat_rows(d);
    if (v2 >= 0) return v2;
  }
  const TileCfg c = pick_cfg(d->cout);
  return ceil_div(d->w, 32) * ceil_div(d->h, c.th) * d->n;
}

template <typename T, int KH, int KW, int WM, int WN, int MF, int NF>
static void launch_igemm_cfg(const IgemmParams& p, dim3 grid, hipStream_t s) {
  constexpr int TH = WM * MF, NT = WN * NF * 32, PIXB = 32 * (int)sizeof(T) + 16;
  constexpr int lds = (TH + KH - 1) * (32 + KW - 1) * PIXB + WM * 2 * NT * (int)sizeof(float);
  if (lds > 64 * 1024) {   // 7x3 in fp32: 14 x 34 pixels x 144 B
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, KH, KW, WM, WN, MF, NF, 32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
  }
  hipLaunchKernelGGL((igemm_kernel<T, KH, KW, WM, WN, MF, NF, 32>), grid, dim3(256), lds, s, p);
}
template <typename T, int KH, int KW>
static int launch_igemm(const OctConvDesc* d, const IgemmParams& p, hipStream_t s) {
  const TileCfg c = pick_cfg(d->cout);
  dim3 grid(p.tiles_x * p.tiles_y * p.n, ceil_div(d->cout, c.nt));
  if (c.nt == 32) launch_igemm_cfg<T, KH, KW, 4, 1, 2, 1>(p, grid, s);
  else if (c.nt == 64) launch_igemm_cfg<T, KH, KW, 2, 2, 4, 1>(p, grid, s);
  else launch_igemm_cfg<T, KH, KW, 2, 2, 4, 2>(p, grid, s);
  return oct_check_launch("igemm");
}

// kernel size of a descriptor: kh = kw = 0 means "from taps" (9 -> 3x3, 1 -> 1x1), as before the fields existed
bool oct_conv_kernel_size(int taps, int kh_in, int kw_in, int* kh, int* kw) {
  if (kh_in == 0 && kw_in == 0) {
    if (taps == 9) { *kh = 3; *kw = 3; return true; }
    if (taps == 1) { *kh = 1; *kw = 1; return true; }
    return false;
  }
  *kh = kh_in; *kw = kw_in;
  return taps == kh_in * kw_in && ((kh_in == 3 && kw_in == 3) || (kh_in == 1 && kw_in == 1) || (kh_in == 7 && kw_in == 3));
}

extern "C" int oct_conv_forward(const OctConvDesc* d, const OctConvArgs* a, void* stream) {
  OCT_CHECK(d && a, "oct_conv_forward: null descriptor");
  OCT_CHECK(d->dtype == OCT_DT_BF16 || d->dtype == OCT_DT_F32, "oct_conv_forward: bad dtype %d", d->dtype);
  int kh = 0, kw = 0;
  OCT_CHECK(oct_conv_kernel_size(d->taps, d->kh, d->kw, &kh, &kw),
            "oct_conv_forward: kernel must be 3x3 (taps 9), 1x1 (taps 1) or 7x3 (taps 21, kh=7, kw=3); got taps=%d kh=%d kw=%d",
            d->taps, d->kh, d->kw);
  OCT_CHECK(kh != 7 || (d->in_mode == OCT_IN_PLAIN && d->out_mode == OCT_OUT_PLAIN), "oct_conv_forward: 7x3 runs plain -> plain");
  OCT_CHECK(d->depth >= 0 && (d->depth == 0 || (d->n % d->depth) == 0), "oct_conv_forward: n=%d is not a whole number of depth-%d volumes", d->n, d->depth);
  OCT_CHECK(d->depth == 0 || kh != 7, "oct_conv_forward: depth taps go with the 3x3 (3x3x3) and 1x1 (2x2x2 transposed) kernels");
  OCT_CHECK(d->out_img_mul == 0 || d->out_mode == OCT_OUT_D2S, "oct_conv_forward: the output image map belongs to D2S");
  OCT_CHECK(d->n > 0 && d->h > 0 && d->w > 0 && d->c0 > 0 && d->c1 >= 0 && d->cout > 0,
            "oct_conv_forward: bad shape n=%d h=%d w=%d c0=%d c1=%d cout=%d", d->n, d->h, d->w, d->c0, d->c1, d->cout);
  OCT_CHECK(a->x0 && a->wpacked && a->y0, "oct_conv_forward: null tensor");
  OCT_CHECK(d->c1 == 0 || a->x1, "oct_conv_forward: c1 > 0 but x1 is null");
  OCT_CHECK(!(d->in_mode == OCT_IN_S2D && d->c1 != 0), "oct_conv_forward: S2D input takes one source");
  OCT_CHECK(!(d->out_mode == OCT_OUT_D2S && (d->cout & 3)), "oct_conv_forward: D2S needs cout %% 4 == 0");
  OCT_CHECK(d->split >= 0 && d->split < d->cout, "oct_conv_forward: bad split %d", d->split);
  OCT_CHECK(d->split == 0 || a->y1, "oct_conv_forward: split without y1");
  OCT_CHECK(d->xform0 >= 0 && d->xform0 <= OCT_XF_AFFINE && d->xform1 >= 0 && d->xform1 <= OCT_XF_AFFINE, "oct_conv_forward: bad xform");
  OCT_CHECK(!(d->xform0 && (!a->scale0 || !a->shift0)), "oct_conv_forward: xform0 without scale/shift");
  OCT_CHECK(!(d->xform1 && (!a->scale1 || !a->shift1)), "oct_conv_forward: xform1 without scale/shift");
  OCT_CHECK(!(d->want_stats && !a->stat_partials), "oct_conv_forward: want_stats without buffer");
  OCT_CHECK((size_t)d->n * d->h * d->w < (1u << 31), "oct_conv_forward: too many pixels");
  {
    int took = oct_first_fprop(d, a, stream);
    if (took == 0) took = oct_conv_forward_v2(d, a, stream);
    if (took != 0) return took < 0 ? took : OCT_OK;
  }
  IgemmParams p;
  p.x0 = a->x0; p.x1 = a->x1; p.sc0 = a->scale0; p.sh0 = a->shift0; p.sc1 = a->scale1; p.sh1 = a->shift1;
  p.wp = a->wpacked; p.bias = a->bias; p.y0 = a->y0; p.y1 = a->y1;
  p.stats = d->want_stats ? a->stat_partials : nullptr;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.c1 = d->c1;
  p.depth = d->depth; p.csrc = d->c0 + d->c1; p.oimg_mul = d->out_img_mul; p.oimg_add = d->out_img_add;
  p.ktot = d->in_mode == OCT_IN_S2D ? (d->depth > 0 ? 8 : 4) * d->c0 : (d->depth > 0 ? 3 : 1) * (d->c0 + d->c1);
  p.nk16 = ceil_div(p.ktot, 16);
  p.cout = d->cout; p.nb32 = ceil_div(d->cout, 32);
  p.xf0 = d->xform0; p.xf1 = d->xform1; p.in_mode = d->in_mode; p.out_mode = d->out_mode; p.split = d->split;
  const TileCfg c = pick_cfg(d->cout);
  p.tiles_x = ceil_div(d->w, 32); p.tiles_y = ceil_div(d->h, c.th);
  hipStream_t s = as_stream(stream);
  if (d->dtype == OCT_DT_BF16)
    return kh == 7 ? launch_igemm<bf16_t, 7, 3>(d, p, s) : kh == 3 ? launch_igemm<bf16_t, 3, 3>(d, p, s) : launch_igemm<bf16_t, 1, 1>(d, p, s);
  return kh == 7 ? launch_igemm<float, 7, 3>(d, p, s) : kh == 3 ? launch_igemm<float, 3, 3>(d, p, s) : launch_igemm<float, 1, 1>(d, p, s);
}
