// Weight gradient of the 3x3 convolution / 2x2 transposed convolution on MFMA (gfx950).
//
//   dW[tap][co][ci] = sum over pixels  dY[p][co] * A[p + tap][ci]
//
// The contraction runs over PIXELS, so both operands are needed "pixel-major per channel".
// A workgroup stages a TH x 32 tile of dY and the matching halo tile of the (BN+ReLU-transformed,
// virtually concatenated) input in LDS, [pixel][channel] like the forward kernel, and each wave
// gathers its fragments pixel-strided out of those tiles.  Every wave owns one 32(co) x 32(ci)
// block and keeps all TAPS accumulators (9 x 16 registers) resident while the workgroup walks
// its share of the image tiles (persistent loop): the cross-workgroup reduction happens once per
// workgroup, as fp32 atomics into dwp[tap][co][ci] whose lanes run along ci (128-B segments).
#include "common.h"

struct WgradParams {
  const void* x0; const void* x1;
  const float* sc0; const float* sh0; const float* sc1; const float* sh1;
  const void* dy; float* dwp; float* dbias;
  int n, h, w;
  int c0, c1, ktot;
  int cout;            // GEMM rows (4*Cout for the deconv)
  int xf0, xf1, dy_mode;
  int tiles_x, tiles_y, ntiles;
  int ty0, pad_h, pad_w;   // first kernel row this launch covers, "same" padding of the whole kernel
  int depth, img_shift;    // 3-D: the input tile comes from slice d + img_shift of the same volume (zero outside)
  int dy_mul, dy_add;      // S2D dY: gathered from image img*dy_mul + dy_add (0: identity)
  int part_mode; size_t slab_elems; float* dbias_part;   // partials mode, see wgrad2.hip: slab = (blockIdx.x, wave)
};

template <typename T>
__device__ __forceinline__ float wg_fetch_in(const WgradParams& p, size_t pix, int k) {
  if (k >= p.ktot) return 0.f;
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
__device__ __forceinline__ void wg_fetch_in8(const WgradParams& p, size_t pix, int k, float (&v)[8]) {
  const T* src = nullptr; const float* sc = nullptr; const float* sh = nullptr; int c = 0, cs = 0; bool xf = false;
  float lo = 0.f;   // clamp of the transform (xf_floor)
  if ((p.c0 & 7) == 0 && k + 8 <= p.c0) {
    src = reinterpret_cast<const T*>(p.x0); c = k; cs = p.c0; sc = p.sc0; sh = p.sh0; xf = p.xf0 != 0; lo = xf_floor(p.xf0);
  } else if ((p.c0 & 7) == 0 && (p.c1 & 7) == 0 && k >= p.c0 && k + 8 <= p.ktot) {
    src = reinterpret_cast<const T*>(p.x1); c = k - p.c0; cs = p.c1; sc = p.sc1; sh = p.sh1; xf = p.xf1 != 0; lo = xf_floor(p.xf1);
  }
  if (src) {
    load_vec<T, 8>(src + pix * cs + c, v);
    if (xf) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], sc[c + j], sh[c + j]), lo);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = wg_fetch_in<T>(p, pix, k + j);
  }
}

// dY element: GEMM row `row` of GEMM pixel (img, y, x)
template <typename T>
__device__ __forceinline__ void wg_fetch_dy8(const WgradParams& p, int img, int y, int x, int row, float (&v)[8]) {
  const T* dy = reinterpret_cast<const T*>(p.dy);
  if (p.dy_mode == OCT_IN_S2D) {
    if (p.dy_mul) img = img * p.dy_mul + p.dy_add;
    const int cr = p.cout >> 2;
    if ((cr & 7) == 0 && row + 8 <= p.cout) {
      const int dydx = row / cr, co = row - dydx * cr;
      const size_t pix = ((size_t)img * (2 * p.h) + (2 * y + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * x + (dydx & 1));
      load_vec<T, 8>(dy + pix * cr + co, v);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = row + j;
        if (rr < p.cout) {
          const int dydx = rr / cr, co = rr - dydx * cr;
          const size_t pix = ((size_t)img * (2 * p.h) + (2 * y + (dydx >> 1))) * (size_t)(2 * p.w) + (2 * x + (dydx & 1));
          v[j] = to_f32(dy[pix * cr + co]);
        } else {
          v[j] = 0.f;
        }
      }
    }
    return;
  }
  const size_t pix = ((size_t)img * p.h + y) * (size_t)p.w + x;
  if ((p.cout & 7) == 0 && row + 8 <= p.cout) {
    load_vec<T, 8>(dy + pix * p.cout + row, v);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (row + j < p.cout) ? to_f32(dy[pix * p.cout + row + j]) : 0.f;
  }
}

// One launch covers TR consecutive kernel rows (all KW columns) starting at row p.ty0: 3x3 = one launch of
// <3,3>, 1x1 / deconv = <1,1>, ReLayNet's 7x3 = rows {0-2}, {3-5}, {6} -- the register budget (16 accumulators
// per tap) caps a launch at 9 taps.  Input rows are staged with the matching vertical offset.
template <typename T, int TR, int KW>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p) {
  constexpr int TAPS = TR * KW;
  constexpr int TH = 8, TW = 32;
  constexpr int LH = TH + TR - 1, LW = TW + KW - 1;
  constexpr int PIXE = 32 + 8 / (int)sizeof(T) * 2;  // elements per LDS pixel: 32 channels + pad (keeps 16-B rows)
  typedef Mma<T> M;
  typedef typename M::Frag Frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* in_tile = reinterpret_cast<T*>(smem);                 // [LH*LW][PIXE]
  T* dy_tile = in_tile + LH * LW * PIXE;                   // [TH*TW][PIXE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;  // bias gradient: this lane's output channel (co0 + r), its share of the pixels

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int bx = tile;
    const int txi = bx % p.tiles_x; bx /= p.tiles_x;
    const int tyi = bx % p.tiles_y; const int img = bx / p.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    __syncthreads();
    for (int idx = tid; idx < LH * LW * 4; idx += 256) {
      const int pix = idx >> 2, grp = idx & 3;
      const int ly = pix / LW, lx = pix - ly * LW;
      const int iy = y0 + ly + p.ty0 - p.pad_h, ix = x0 + lx - p.pad_w;
      float v[8];
      const int dz = p.depth > 0 ? img % p.depth + p.img_shift : 0;
      if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w && dz >= 0 && (p.depth == 0 || dz < p.depth)) {
        wg_fetch_in8<T>(p, ((size_t)(img + p.img_shift) * p.h + iy) * (size_t)p.w + ix, ci0 + grp * 8, v);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      store_vec<T, 8>(in_tile + pix * PIXE + grp * 8, v);
    }
    for (int idx = tid; idx < TH * TW * 4; idx += 256) {
      const int pix = idx >> 2, grp = idx & 3;
      const int ly = pix / TW, lx = pix - ly * TW;
      const int oy = y0 + ly, ox = x0 + lx;
      float v[8];
      if (oy < p.h && ox < p.w) {
        wg_fetch_dy8<T>(p, img, oy, ox, co0 + grp * 8, v);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      store_vec<T, 8>(dy_tile + pix * PIXE + grp * 8, v);
    }
    __syncthreads();
    // each wave: tile rows {2*wave, 2*wave+1}, two 16-pixel k-steps per row
#pragma unroll 1
    for (int step = 0; step < 4; ++step) {
      const int row = wave * 2 + (step >> 1);
      const int xs = (step & 1) * 16 + 8 * hh;  // first of this lane's 8 pixels
      Frag a = M::zero();
#pragma unroll
      for (int j = 0; j < 8; ++j) M::set(a, j, dy_tile[(row * TW + xs + j) * PIXE + r]);
      if (p.dbias && blockIdx.z == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += to_f32(dy_tile[(row * TW + xs + j) * PIXE + r]);
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ty = t / KW, tx = t % KW;
        Frag b = M::zero();
#pragma unroll
        for (int j = 0; j < 8; ++j) M::set(b, j, in_tile[((row + ty) * LW + xs + j + tx) * PIXE + r]);
        M::mma(acc[t], a, b);
      }
    }
  }
  const int slab = blockIdx.x * 4 + wave;
  if (p.dbias && blockIdx.z == 0) {
    bsum += __shfl_xor(bsum, 32);
    const int cr = (p.dy_mode == OCT_IN_S2D) ? (p.cout >> 2) : p.cout;
    if (hh == 0 && co0 + r < p.cout) {
      if (p.part_mode) p.dbias_part[(size_t)slab * p.cout + co0 + r] = bsum;
      else atomicAdd(&p.dbias[(co0 + r) % cr], bsum);
    }
  }
  // D[row = co][col = ci]: reg i -> co = co0 + (i&3) + 8*(i>>2) + 4*hh ; lane -> ci = ci0 + r
  const int ci = ci0 + r;
  if (ci < p.ktot) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (co < p.cout) {
          float* const q = &p.dwp[(p.part_mode ? (size_t)slab * p.slab_elems : 0) + ((size_t)(p.ty0 * KW + t) * p.cout + co) * p.ktot + ci];
          if (p.part_mode) *q = acc[t][i]; else atomicAdd(q, acc[t][i]);
        }
      }
  }
}

extern "C" int oct_conv_wgrad(const OctWgradDesc* d, const OctWgradArgs* a, void* stream) {
  OCT_CHECK(d && a, "oct_conv_wgrad: null descriptor");
  OCT_CHECK(d->dtype == OCT_DT_BF16 || d->dtype == OCT_DT_F32, "oct_conv_wgrad: bad dtype %d", d->dtype);
  int kh = 0, kw = 0;
  OCT_CHECK(oct_conv_kernel_size(d->taps, d->kh, d->kw, &kh, &kw),
            "oct_conv_wgrad: kernel must be 3x3 (taps 9), 1x1 (taps 1) or 7x3 (taps 21, kh=7, kw=3); got taps=%d kh=%d kw=%d",
            d->taps, d->kh, d->kw);
  OCT_CHECK(kh != 7 || (d->dy_mode == OCT_IN_PLAIN && !a->dy_coef), "oct_conv_wgrad: 7x3 takes a plain dY");
  OCT_CHECK(d->depth >= 0 && (d->depth == 0 || ((d->n % d->depth) == 0 && kh != 7)), "oct_conv_wgrad: bad depth %d for n=%d", d->depth, d->n);
  OCT_CHECK(((d->in_img_shift >= -1 && d->in_img_shift <= 1) || d->in_img_shift == OCT_IMG_SHIFT_ALL) && (d->in_img_shift == 0 || d->depth > 0),
            "oct_conv_wgrad: in_img_shift needs depth > 0");
  OCT_CHECK(d->in_img_shift != OCT_IMG_SHIFT_ALL || oct_conv_wgrad_all_depth_taps_ok(d),
            "oct_conv_wgrad: OCT_IMG_SHIFT_ALL is not available for this descriptor (oct_conv_wgrad_all_depth_taps_ok)");
  OCT_CHECK(d->dy_img_mul == 0 || d->dy_mode == OCT_IN_S2D, "oct_conv_wgrad: the dY image map belongs to S2D");
  OCT_CHECK(d->n > 0 && d->h > 0 && d->w > 0 && d->c0 > 0 && d->c1 >= 0 && d->cout > 0, "oct_conv_wgrad: bad shape");
  OCT_CHECK(a->x0 && a->dy && a->dwp, "oct_conv_wgrad: null tensor");
  OCT_CHECK(d->c1 == 0 || a->x1, "oct_conv_wgrad: c1 > 0 but x1 is null");
  OCT_CHECK(!(d->dy_mode == OCT_IN_S2D && (d->cout & 3)), "oct_conv_wgrad: S2D dy needs cout %% 4 == 0");
  OCT_CHECK(d->xform0 >= 0 && d->xform0 <= OCT_XF_AFFINE && d->xform1 >= 0 && d->xform1 <= OCT_XF_AFFINE, "oct_conv_wgrad: bad xform");
  OCT_CHECK(!(d->xform0 && (!a->scale0 || !a->shift0)), "oct_conv_wgrad: xform0 without scale/shift");
  OCT_CHECK(!(d->xform1 && (!a->scale1 || !a->shift1)), "oct_conv_wgrad: xform1 without scale/shift");
  OCT_CHECK(!d->partials || !a->dbias || a->dbias_partials, "oct_conv_wgrad: partials mode with a bias gradient needs dbias_partials");
  if (kh != 7) {
    int took = oct_first_wgrad(d, a, stream);
    if (took == 0 && a->dy_coef) OCT_CHECK(false, "oct_conv_wgrad: the fused BN-backward apply is only implemented for the 1->F first layer in bf16");
    if (took == 0) took = oct_conv_wgrad_v2(d, a, stream);
    if (took != 0) return took < 0 ? took : OCT_OK;
  } else {
    int took = oct_first_wgrad(d, a, stream);            // Conv2d(1 -> F, 7x3): the matrix-pipe first-layer kernel
    if (took == 0) took = oct_conv_wgrad_v2(d, a, stream);   // 64-channel-block shapes: three row-shifted launches of the 3x3 kernel
    if (took != 0) return took < 0 ? took : OCT_OK;
  }
  WgradParams p;
  p.x0 = a->x0; p.x1 = a->x1; p.sc0 = a->scale0; p.sh0 = a->shift0; p.sc1 = a->scale1; p.sh1 = a->shift1;
  p.dy = a->dy; p.dwp = a->dwp; p.dbias = a->dbias;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.c1 = d->c1; p.ktot = d->c0 + d->c1; p.cout = d->cout;
  p.xf0 = d->xform0; p.xf1 = d->xform1; p.dy_mode = d->dy_mode;
  p.depth = d->depth; p.img_shift = d->in_img_shift; p.dy_mul = d->dy_img_mul; p.dy_add = d->dy_img_add;
  p.tiles_x = ceil_div(d->w, 32); p.tiles_y = ceil_div(d->h, 8); p.ntiles = p.tiles_x * p.tiles_y * d->n;
  const int nco = ceil_div(d->cout, 32), nci = ceil_div(p.ktot, 32);
  // persistent workgroups: ~4 per CU over all channel-block pairs
  int per_pair = 1024 / (nco * nci);
  if (per_pair < 1) per_pair = 1;
  if (per_pair > p.ntiles) per_pair = p.ntiles;
  dim3 grid(per_pair, nco, nci);
  p.part_mode = d->partials ? 1 : 0; p.slab_elems = (size_t)d->taps * d->cout * p.ktot; p.dbias_part = a->dbias_partials;
  const int esz = d->dtype == OCT_DT_BF16 ? 2 : 4;
  const int pixe = 32 + 8 / esz * 2;
  p.pad_h = (kh - 1) / 2; p.pad_w = (kw - 1) / 2;
  hipStream_t s = as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float, 1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr_set = true;
  }
  const bool bf = d->dtype == OCT_DT_BF16;
  for (int ty0 = 0; ty0 < kh; ty0 += 3) {
    const int tr = kh - ty0 >= 3 ? 3 : 1;           // kh in {1, 3, 7}: groups of three rows, then single rows
    p.ty0 = ty0;
    if (ty0 > 0) p.dbias = nullptr;                  // the bias gradient belongs to one launch only
    const size_t lds = (size_t)((8 + tr - 1) * (32 + kw - 1) + 8 * 32) * pixe * esz;
    if (tr == 3 && kw == 3) {
      if (bf) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 3, 3>), grid, dim3(256), lds, s, p);
      else hipLaunchKernelGGL((wgrad_kernel<float, 3, 3>), grid, dim3(256), lds, s, p);
    } else if (tr == 1 && kw == 3) {
      if (bf) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 1, 3>), grid, dim3(256), lds, s, p);
      else hipLaunchKernelGGL((wgrad_kernel<float, 1, 3>), grid, dim3(256), lds, s, p);
    } else {
      if (bf) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 1, 1>), grid, dim3(256), lds, s, p);
      else hipLaunchKernelGGL((wgrad_kernel<float, 1, 1>), grid, dim3(256), lds, s, p);
    }
  }
  return oct_check_launch("wgrad");
}

// Number of partial slabs a launch with this descriptor writes in partials mode (OctWgradDesc.partials = 1): the caller
// sizes dwp as [slabs][taps][cout][ktot] (and dbias_partials as [slabs][cout]) and hands `slabs` to the unpack pass.
extern "C" int oct_conv_wgrad_partials(const OctWgradDesc* d) {
  if (!d) return 0;
  int kh = 0, kw = 0;
  if (!oct_conv_kernel_size(d->taps, d->kh, d->kw, &kh, &kw)) return 0;
  int q = 0;
  if (kh != 7) {
    if (oct_first_wgrad(d, nullptr, nullptr, &q) == 1) return q;
    if (oct_conv_wgrad_v2(d, nullptr, nullptr, &q) == 1) return q;
  } else {
    if (oct_first_wgrad(d, nullptr, nullptr, &q) == 1) return q;
    if (oct_conv_wgrad_v2(d, nullptr, nullptr, &q) == 1) return q;
  }
  const int ktot = d->c0 + d->c1;
  const int nco = ceil_div(d->cout, 32), nci = ceil_div(ktot, 32);
  const int ntiles = ceil_div(d->w, 32) * ceil_div(d->h, 8) * d->n;
  int per_pair = 1024 / (nco * nci);
  if (per_pair < 1) per_pair = 1;
  if (per_pair > ntiles) per_pair = ntiles;
  return per_pair * 4;
}

// dwp[slab][tap][rows][kch] -> torch-layout gradient; the slabs (1 for the atomics mode) are summed in index order.
// Convolutions: with R = cout*cin the layouts are dwp[tap][r] and grad[r][tap] -- a [T][R] -> [R][T] transpose.  A wave
// takes 64 consecutive r: T coalesced 256-B row reads (per slab), a wave-private LDS tile written at stride T (odd for
// 3x3 / 7x3: conflict-free), T coalesced 256-B writes.  The first version gathered with four 64-bit divisions per element
// and one cache line per lane (0.2 TB/s on AttU_Net's 35 M parameters: 1.5 ms per step).
#define UNPACK_TMAX 21
__device__ __forceinline__ void unpack_conv_rows(const float* __restrict__ dwp, float* __restrict__ grad, unsigned R, int taps,
                                                 int nparts, size_t slab, int accumulate, unsigned wave0, unsigned nwaves,
                                                 float* tile /* [64 * taps], wave-private */) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned nchunk = (R + 63) / 64;
  for (unsigned ch = wave0; ch < nchunk; ch += nwaves) {
    const unsigned r = ch * 64 + lane;
    if (r < R) {
      for (int t = 0; t < taps; ++t) {
        const float* src = dwp + (size_t)t * R + r;
        float v = src[0];
        for (int g = 1; g < nparts; ++g) v += src[(size_t)g * slab];
        tile[lane * taps + t] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned n = min(64u, R - ch * 64) * (unsigned)taps;
    float* dst = grad + (size_t)ch * 64 * taps;
    for (unsigned k = lane; k < n; k += 64) dst[k] = accumulate ? dst[k] + tile[k] : tile[k];
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void __launch_bounds__(256) unpack_wgrad_kernel(int mode, const float* __restrict__ dwp, float* __restrict__ grad, int cout,
                                                           int cin, int accumulate, size_t total, int taps, int nparts, size_t slab) {
  if (mode == OCT_PACK_CONV_FPROP) {  // grad[co][ci][tap]
    __shared__ float tiles[4][64 * UNPACK_TMAX];
    unpack_conv_rows(dwp, grad, (unsigned)(cout * cin), taps, nparts, slab, accumulate, blockIdx.x * 4 + (threadIdx.x >> 6),
                     gridDim.x * 4, tiles[threadIdx.x >> 6]);
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t o;
    if (mode == OCT_PACK_DECONV_FPROP) {  // grad[ci][co][dydx] ; dwp[0][dydx*cout+co][ci]
      const int dydx = i & 3; const size_t r = i >> 2; const int co = r % cout; const int ci = r / cout;
      o = ((size_t)dydx * cout + co) * cin + ci;
    } else {  // 1x1: grad[co][ci] = dwp[0][co][ci]
      o = i;
    }
    float v = dwp[o];
    for (int g = 1; g < nparts; ++g) v += dwp[(size_t)g * slab + o];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

// bias gradient of partials mode: dbias[c] (+)= sum over slabs (in order) and over the `fold` GEMM rows that share channel c
__global__ void reduce_bias_partials_kernel(const float* __restrict__ part, int nparts, int rows, int cr, float* __restrict__ dbias,
                                            int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cr) return;
  float v = 0.f;
  for (int g = 0; g < nparts; ++g)
    for (int q = c; q < rows; q += cr) v += part[(size_t)g * rows + q];
  dbias[c] = accumulate ? dbias[c] + v : v;
}
extern "C" int oct_reduce_bias_partials(const float* part, int nparts, int rows, int channels, float* dbias, int accumulate,
                                        void* stream) {
  OCT_CHECK(part && dbias && nparts > 0 && rows > 0 && channels > 0 && rows % channels == 0, "oct_reduce_bias_partials: bad arguments");
  hipLaunchKernelGGL(reduce_bias_partials_kernel, dim3(ceil_div(channels, 256)), dim3(256), 0, as_stream(stream), part, nparts,
                     rows, channels, dbias, accumulate);
  return oct_check_launch("reduce_bias_partials");
}

// every unpacking of a backward pass in one launch (22 five-microsecond launches otherwise)
struct UnpackJobs { OctUnpackJob j[OCT_PACK_BATCH_MAX]; };
__global__ void __launch_bounds__(256) unpack_wgrad_batch_kernel(const UnpackJobs jobs) {
  const OctUnpackJob jb = jobs.j[blockIdx.y];
  const int mode = jb.mode, cout = jb.cout, cin = jb.cin;
  const float* __restrict__ dwp = jb.dwp;
  float* __restrict__ grad = jb.grad;
  const size_t total = (size_t)cout * cin * (mode == OCT_PACK_CONV_FPROP ? 9 : mode == OCT_PACK_DECONV_FPROP ? 4 : 1);
  const int nparts = jb.nparts > 1 ? jb.nparts : 1;
  if (mode == OCT_PACK_CONV_FPROP) {
    __shared__ float tiles[4][64 * 9];
    unpack_conv_rows(dwp, grad, (unsigned)(cout * cin), 9, nparts, total, jb.accumulate, blockIdx.x * 4 + (threadIdx.x >> 6),
                     gridDim.x * 4, tiles[threadIdx.x >> 6]);
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t o;
    if (mode == OCT_PACK_DECONV_FPROP) {
      const int dydx = i & 3; const size_t r = i >> 2; const int co = r % cout; const int ci = r / cout;
      o = ((size_t)dydx * cout + co) * cin + ci;
    } else {
      o = i;
    }
    float v = dwp[o];
    for (int g = 1; g < nparts; ++g) v += dwp[(size_t)g * total + o];   // slab size == total for these modes
    grad[i] = jb.accumulate ? grad[i] + v : v;
  }
}

extern "C" int oct_unpack_wgrad_batch(int count, const OctUnpackJob* jobs, void* stream) {
  OCT_CHECK(count >= 0 && (count == 0 || jobs), "oct_unpack_wgrad_batch: bad arguments");
  for (int base = 0; base < count; base += OCT_PACK_BATCH_MAX) {
    UnpackJobs uj;
    const int n = count - base < OCT_PACK_BATCH_MAX ? count - base : OCT_PACK_BATCH_MAX;
    for (int i = 0; i < n; ++i) {
      uj.j[i] = jobs[base + i];
      const int m = uj.j[i].mode;
      OCT_CHECK((m == OCT_PACK_CONV_FPROP || m == OCT_PACK_DECONV_FPROP || m == OCT_PACK_1X1_FPROP) && uj.j[i].dwp &&
                uj.j[i].grad && uj.j[i].cout > 0 && uj.j[i].cin > 0, "oct_unpack_wgrad_batch: bad job %d", base + i);
    }
    // grid.x follows the largest job (a wave per 64 (cout, cin) pairs, four waves per workgroup); partial-slab jobs read
    // nparts x as much: spread them over the whole chip
    size_t big = 0;
    for (int i = 0; i < n; ++i) { const size_t r = (size_t)uj.j[i].cout * uj.j[i].cin; if (r > big) big = r; }
    int gx = (int)((big + 1023) / 1024);
    if (gx < 64) gx = 64;
    if (gx > 1024) gx = 1024;
    for (int i = 0; i < n; ++i) if (uj.j[i].nparts > 1) gx = 1024;
    hipLaunchKernelGGL(unpack_wgrad_batch_kernel, dim3(gx, n), dim3(256), 0, as_stream(stream), uj);
  }
  return oct_check_launch("unpack_wgrad_batch");
}

extern "C" int oct_unpack_wgrad(int mode, const float* dwp, float* grad, int cout, int cin, int accumulate, void* stream) {
  OCT_CHECK(mode == OCT_PACK_CONV_FPROP || mode == OCT_PACK_DECONV_FPROP || mode == OCT_PACK_1X1_FPROP,
            "oct_unpack_wgrad: bad mode %d", mode);
  OCT_CHECK(dwp && grad && cout > 0 && cin > 0, "oct_unpack_wgrad: bad arguments");
  const size_t total = (size_t)cout * cin * (mode == OCT_PACK_CONV_FPROP ? 9 : mode == OCT_PACK_DECONV_FPROP ? 4 : 1);
  const size_t work = mode == OCT_PACK_CONV_FPROP ? (size_t)cout * cin : total;   // conv: a wave per 64 (cout, cin) pairs
  const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), mode, dwp, grad, cout, cin,
                     accumulate, total, 9, 1, (size_t)0);
  return oct_check_launch("unpack_wgrad");
}

// 3-D unpacking (see oct_hip.h)
__global__ void unpack_wgrad3d_kernel(int mode, const float* __restrict__ dwp, float* __restrict__ grad, int cout, int cin,
                                      int kdi, int accumulate, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v; size_t o;
    if (mode == OCT_PACK_CONV3D_FPROP) {   // i over grad[co][ci][kd][tap]
      const int tap = i % 9; size_t r = i / 9; const int kd = r % 3; r /= 3; const int ci = r % cin; const int co = r / cin;
      v = dwp[(((size_t)kd * 9 + tap) * cout + co) * cin + ci];
      o = i;
    } else {                               // i over (ci, co, dydx) of depth slice kdi: grad[ci][co][kdi][dydx]
      const int dydx = i & 3; const size_t r = i >> 2; const int co = r % cout; const int ci = r / cout;
      v = dwp[((size_t)dydx * cout + co) * cin + ci];
      o = (((size_t)ci * cout + co) * 2 + kdi) * 4 + dydx;
    }
    grad[o] = accumulate ? grad[o] + v : v;
  }
}
extern "C" int oct_unpack_wgrad3d(int mode, const float* dwp, float* grad, int cout, int cin, int kdi, int accumulate,
                                  void* stream) {
  OCT_CHECK(mode == OCT_PACK_CONV3D_FPROP || mode == OCT_PACK_DECONV3D_FPROP, "oct_unpack_wgrad3d: bad mode %d", mode);
  OCT_CHECK(dwp && grad && cout > 0 && cin > 0 && (kdi == 0 || (kdi == 1 && mode == OCT_PACK_DECONV3D_FPROP)), "oct_unpack_wgrad3d: bad arguments");
  const size_t total = (size_t)cout * cin * (mode == OCT_PACK_CONV3D_FPROP ? 27 : 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpack_wgrad3d_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), mode, dwp, grad, cout, cin, kdi,
                     accumulate, total);
  return oct_check_launch("unpack_wgrad3d");
}

// dwp[kh*kw][cout][cin] -> grad (Cout, Cin, kh, kw) for any kernel size (ReLayNet's 7x3)
extern "C" int oct_unpack_wgrad_kk(const float* dwp, float* grad, int cout, int cin, int kh, int kw, int accumulate,
                                   void* stream) {
  OCT_CHECK(dwp && grad && cout > 0 && cin > 0 && kh > 0 && kw > 0, "oct_unpack_wgrad_kk: bad arguments");
  OCT_CHECK(kh * kw <= UNPACK_TMAX, "oct_unpack_wgrad_kk: at most %d taps", UNPACK_TMAX);
  const size_t total = (size_t)cout * cin * kh * kw;
  const size_t work = (size_t)cout * cin;
  const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), OCT_PACK_CONV_FPROP, dwp, grad,
                     cout, cin, accumulate, total, kh * kw, 1, (size_t)0);
  return oct_check_launch("unpack_wgrad_kk");
}
