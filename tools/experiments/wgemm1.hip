// EXPERIMENT (round 3), NOT part of the library: an eight-wave 256 x 256-block weight-gradient GEMM for the transposed
// convolutions.  Bit-exact against the oracle, but slower than wgrad2 (0.372 / 0.246 / 0.168 ms against 0.289 / 0.191 / 0.170 on
// upconv2-4): wgrad2 one-tap launches run at the HBM rate of their re-reads (1.07 GB at 5.6 TB/s), this kernel halves the bytes but
// sustains only ~2.5 TB/s with one tile of loads in flight per workgroup.  Kept for the record.
// Weight gradient of the transposed convolutions (and of plain 1x1 convolutions) as a GEMM over the pixels with a
// 256 x 256 (or 256 x 128) output block per workgroup and ALL EIGHT waves staging and multiplying:
//
//   dW[row][ci] = sum over pixels  dY[pixel][row] * relu(bn(X))[pixel][ci]        row = (dy, dx, co) for ConvTranspose2d k2 s2
//
// wgrad2's one-tap form (four producer waves, 128 x 128 block, 4-row tiles) staged 32 KB per 32 MFMAs of a wave: 4 k cycles of
// staging per 1 k cycles of matrix work, and upconv4 (Cin 512, N 1024) re-read X eight times and dY four times
// (profiles/r02_cfg2_launch_table.txt rows 32, 38, 44: 2-5 x their floors).  Here a 64-pixel tile (two image rows x 32) of 256 rows
// and 256 (128) input channels is staged by all 512 threads (eight 16-byte pieces each: loaded a tile ahead into registers, BN + ReLU
// applied on the way into LDS), wave (cw, iw) keeps 2 x 4 (2 x 2) accumulators of 32 x 32 and multiplies 32 (16) MFMAs per tile from
// transposed LDS reads (ds_read_b64_tr_b16, [pixel][64 B] blocks as in wgrad2.hip); twice the matrix work per staged byte.
// fp32 atomics into dwp[row][ktot] at the end, the bias gradient as one more MFMA per k-step against a ones fragment: same
// results and layout as wgrad2 (the exact-arithmetic tests do not tell them apart).  Partials (deterministic) mode, ragged sizes
// and the volumetric modes stay on wgrad2.
#include "common.h"
#include <stdlib.h>

struct Wgemm1Params {
  const bf16_t* x; const float* sc; const float* sh; const bf16_t* dy; float* dwp; float* dbias;
  int n, h, w, c0, cout, xf, s2d, tiles_x, tiles_y, ntiles;
};

typedef unsigned int wg1_u32x4 __attribute__((ext_vector_type(4)));
typedef short wg1_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 wg1_tr_frag(const unsigned char* base_lo) {
  typedef __attribute__((address_space(3))) wg1_s16x4 lds_s16x4;
  const wg1_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
  const wg1_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo + 4 * 64));  // pixels +4
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ unsigned wg1_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}

constexpr int WG1_TP = 64;                    // pixels per tile: two image rows x 32
constexpr int WG1_BLKB = WG1_TP * 64;         // one 32-channel block of a tile: [pixel][64 B]

// IB: 32-channel input blocks per workgroup (8 or 4); eight row blocks always
template <int IB, bool XF>
__global__ void __launch_bounds__(512) wgemm1_kernel(const Wgemm1Params p) {
  typedef Mma<bf16_t> M;
  constexpr int CB = 8, WCB = 2, WIB = IB / 2;
  constexpr int STAGEB = (IB + CB) * WG1_BLKB;
  constexpr int KX = (IB * 4) / 8;            // input passes per thread: 512 threads cover two blocks (2 x 64 px x 4 pieces) per pass
  constexpr int KD = (CB * 4) / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const sxf = reinterpret_cast<float*>(smem + 2 * STAGEB);   // [2][32 * IB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave >> 1, iw = wave & 1;
  const int co_sb = blockIdx.y * (32 * CB), ci_sb = blockIdx.z * (32 * IB);
  if ((int)blockIdx.x >= p.ntiles) return;
  const int nstage = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  if (XF) {
    for (int i = tid; i < 32 * IB; i += 512) { sxf[i] = p.sc[ci_sb + i]; sxf[32 * IB + i] = p.sh[ci_sb + i]; }
  }

  // ---- staging map: pass k covers blocks 2k and 2k + 1 (waves 0-3 / 4-7); inside a block 256 threads = 64 pixels x 4 pieces ----
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8);      // 0 / 1: which of the pass's two blocks (wave-uniform)
  const int t8 = tid & 255, pix = t8 >> 2, g = t8 & 3;
  const int ly = pix >> 5, lx = pix & 31;
  const int cs = p.s2d ? (p.cout >> 2) : p.cout;                   // channels of the dY tensor
  const unsigned xoff = (unsigned)(ly * p.w + lx) * (unsigned)(2 * p.c0) + (unsigned)g * 16u;
  const unsigned doff = (unsigned)(p.s2d ? (2 * ly) * (2 * p.w) + 2 * lx : ly * p.w + lx) * (unsigned)(2 * cs) + (unsigned)g * 16u;
  const unsigned lds_lane = (unsigned)(pix * 64 + g * 16);

  // tile iterator (counter chain; the workgroup walks tiles blockIdx.x, + gridDim.x, ...)
  const int sx = (int)gridDim.x % p.tiles_x, sy = ((int)gridDim.x / p.tiles_x) % p.tiles_y, simg = (int)gridDim.x / (p.tiles_x * p.tiles_y);
  int i_left = nstage - 1;
  int i_txi, i_tyi, i_img;
  {
    int t = blockIdx.x;
    i_txi = t % p.tiles_x; t /= p.tiles_x;
    i_tyi = t % p.tiles_y; i_img = t / p.tiles_y;
  }
  auto advance = [&]() {
    if (i_left > 0) {
      --i_left;
      i_txi += sx; if (i_txi >= p.tiles_x) { i_txi -= p.tiles_x; ++i_tyi; }
      i_tyi += sy; if (i_tyi >= p.tiles_y) { i_tyi -= p.tiles_y; ++i_img; }
      i_img += simg;
    }
  };
  wg1_u32x4 RX[KX], RD[KD];
  // the (dy, dx) quarter and first co of this thread's 32-row block in pass k: fixed for the whole launch (no division per tile)
  int d_dydx[KD], d_co[KD];
#pragma unroll
  for (int k = 0; k < KD; ++k) {
    const int row = co_sb + (2 * k + half) * 32;
    d_dydx[k] = p.s2d ? row / cs : 0;
    d_co[k] = row - d_dydx[k] * cs;
  }
  // the tile's two operands are fetched (and committed) half a stage apart: the workgroup always has one of them in flight and
  // its requests leave in two bursts per stage instead of one (the kernel moves 64 KB per 2 k cycles of MFMAs: memory-bound)
  auto issue_x = [&]() {
    const size_t origin = ((size_t)i_img * p.h + i_tyi * 2) * p.w + i_txi * 32;
    const unsigned char* const xb = reinterpret_cast<const unsigned char*>(p.x + origin * p.c0 + ci_sb);
#pragma unroll
    for (int k = 0; k < KX; ++k) RX[k] = *reinterpret_cast<const wg1_u32x4*>(xb + (size_t)((2 * k + half) * 64) + (size_t)xoff);
  };
  auto issue_d = [&]() {
    const size_t origin = ((size_t)i_img * p.h + i_tyi * 2) * p.w + i_txi * 32;
#pragma unroll
    for (int k = 0; k < KD; ++k) {
      const unsigned char* db;   // 32-row block: one (dy, dx) quarter, 32 consecutive co
      if (p.s2d) {
        const int dydx = d_dydx[k], co = d_co[k];
        const size_t o2 = ((size_t)i_img * (2 * p.h) + 2 * (i_tyi * 2) + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * (i_txi * 32) + (dydx & 1);
        db = reinterpret_cast<const unsigned char*>(p.dy + o2 * cs + co);
      } else {
        db = reinterpret_cast<const unsigned char*>(p.dy + origin * cs + d_co[k]);
      }
      RD[k] = *reinterpret_cast<const wg1_u32x4*>(db + (size_t)doff);
    }
  };
  auto commit_x = [&](unsigned char* buf) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      wg1_u32x4 v = RX[k];
      const int blk = 2 * k + half;
      if (XF) {
        const float* sc = sxf + blk * 32 + g * 8;
        const float* sh = sxf + 32 * IB + blk * 32 + g * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
        const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const float b[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = fmaxf(fmaf(__uint_as_float(v[e] << 16), s[2 * e], b[2 * e]), 0.f);
          const float hi = fmaxf(fmaf(__uint_as_float(v[e] & 0xffff0000u), s[2 * e + 1], b[2 * e + 1]), 0.f);
          v[e] = wg1_pack(lo, hi);
        }
      }
      *reinterpret_cast<wg1_u32x4*>(buf + blk * WG1_BLKB + lds_lane) = v;
    }
  };
  auto commit_d = [&](unsigned char* buf) {
#pragma unroll
    for (int k = 0; k < KD; ++k)
      *reinterpret_cast<wg1_u32x4*>(buf + (IB + 2 * k + half) * WG1_BLKB + lds_lane) = RD[k];
  };

  f32x16 acc[WCB * WIB];
#pragma unroll
  for (int t = 0; t < WCB * WIB; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // bias gradient = sum over pixels of dY: one more MFMA per k-step against an all-ones operand.  A wave reads BOTH row blocks of
  // its pair anyway, so wave (cw, iw) sums row block iw of the pair: one extra accumulator per wave instead of two
  const bool do_bias = (p.dbias != nullptr) && (blockIdx.z == 0);
  f32x16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  // transposed-read lane address inside a [pixel][64 B] block: pixel 8*(g4>>1) + (li>>2), channel 16*(g4&1) + 4*(li&3)
  const int g4 = lane >> 4, li = lane & 15;
  const int lane_off = (8 * (g4 >> 1) + (li >> 2)) * 64 + (16 * (g4 & 1) + 4 * (li & 3)) * 2;

  __syncthreads();   // coefficient table
  issue_x(); issue_d(); advance();
  commit_x(smem); commit_d(smem);
  issue_x(); issue_d(); advance();
  __syncthreads();

  int cur = 0;
  for (int s = 0; s < nstage; ++s) {
    const unsigned char* const in_t = smem + cur * STAGEB + (iw * WIB) * WG1_BLKB + lane_off;
    const unsigned char* const dy_t = smem + cur * STAGEB + (IB + cw * WCB) * WG1_BLKB + lane_off;
    // four k-steps of 16 pixels.  256-channel blocks (IB = 8) have no registers for a second fragment set (8 + 1 accumulators,
    // eight staged pieces): the reads of a step are issued right before its MFMAs and the SIMD partner covers their latency
    constexpr bool DBUF = IB < 8;
    bf16x8 af[DBUF ? 2 : 1][WCB], bf[DBUF ? 2 : 1][WIB];
    if constexpr (DBUF) {
#pragma unroll
      for (int j = 0; j < WCB; ++j) af[0][j] = wg1_tr_frag(dy_t + j * WG1_BLKB);
#pragma unroll
      for (int i = 0; i < WIB; ++i) bf[0][i] = wg1_tr_frag(in_t + i * WG1_BLKB);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cs_ = DBUF ? (k & 1) : 0;
      if constexpr (DBUF) {
        if (k + 1 < 4) {
#pragma unroll
          for (int j = 0; j < WCB; ++j) af[(k + 1) & 1][j] = wg1_tr_frag(dy_t + j * WG1_BLKB + (k + 1) * 16 * 64);
#pragma unroll
          for (int i = 0; i < WIB; ++i) bf[(k + 1) & 1][i] = wg1_tr_frag(in_t + i * WG1_BLKB + (k + 1) * 16 * 64);
        }
      } else {
#pragma unroll
        for (int j = 0; j < WCB; ++j) af[0][j] = wg1_tr_frag(dy_t + j * WG1_BLKB + k * 16 * 64);
#pragma unroll
        for (int i = 0; i < WIB; ++i) bf[0][i] = wg1_tr_frag(in_t + i * WG1_BLKB + k * 16 * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < WCB; ++j) {
#pragma unroll
        for (int i = 0; i < WIB; ++i) M::mma(acc[j * WIB + i], af[cs_][j], bf[cs_][i]);
      }
      if (do_bias) { if (iw) M::mma(accb, af[cs_][1], ones); else M::mma(accb, af[cs_][0], ones); }
      if (k == 1) {   // half way: the next tile's input operand goes into the other buffer, the one after next is requested
        commit_x(smem + (cur ^ 1) * STAGEB);
        issue_x();
      }
    }
    // the tile after next is in the registers (issued a stage ago): into the other buffer, then fetch the one after that
    commit_d(smem + (cur ^ 1) * STAGEB);
    issue_d(); advance();
    __syncthreads();
    cur ^= 1;
  }

  // D[row = co][col = ci]
  const int r = lane & 31, hh = lane >> 5;
  const int cr = p.s2d ? (p.cout >> 2) : p.cout;
#pragma unroll
  for (int j = 0; j < WCB; ++j) {
#pragma unroll
    for (int ii = 0; ii < WIB; ++ii) {
      const int ci = ci_sb + (iw * WIB + ii) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co_sb + (cw * WCB + j) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        atomicAdd(&p.dwp[(size_t)co * p.c0 + ci], acc[j * WIB + ii][i]);
      }
    }
  }
  if (do_bias && r == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co_sb + (cw * WCB + iw) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      atomicAdd(&p.dbias[co % cr], accb[i]);
    }
  }
}

static bool wg1_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_WGEMM1"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}

template <int IB>
static void wg1_launch(const Wgemm1Params& p, int gy, int gz, hipStream_t s) {
  constexpr int lds = 2 * (IB + 8) * WG1_BLKB + 2 * 32 * IB * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgemm1_kernel<IB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgemm1_kernel<IB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  int gx = 256 / (gy * gz);
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  if (p.xf) hipLaunchKernelGGL((wgemm1_kernel<IB, true>), dim3(gx, gy, gz), dim3(512), lds, s, p);
  else hipLaunchKernelGGL((wgemm1_kernel<IB, false>), dim3(gx, gy, gz), dim3(512), lds, s, p);
}

// returns 1 when taken, 0 when the shape is not eligible (the caller falls through to wgrad2)
int oct_conv_wgrad_g1(const OctWgradDesc* d, const OctWgradArgs* a, void* stream) {
  if (!wg1_enabled() || d->taps != 1 || d->dtype != OCT_DT_BF16 || d->partials) return 0;
  if (d->c1 != 0 || d->depth != 0 || d->dy_img_mul != 0) return 0;
  if ((d->w % 32) != 0 || (d->h % 2) != 0) return 0;
  if ((d->cout % 256) != 0 || (d->c0 % 128) != 0) return 0;
  if (d->dy_mode == OCT_IN_S2D && ((d->cout >> 2) % 32) != 0) return 0;
  if (d->dy_mode != OCT_IN_S2D && d->dy_mode != OCT_IN_PLAIN) return 0;
  if (d->xform0 != OCT_XF_NONE && d->xform0 != OCT_XF_AFFINE_RELU) return 0;
  if (a->dy_coef != nullptr) return 0;   // fused BN-backward apply: first layer only
  Wgemm1Params p;
  p.x = (const bf16_t*)a->x0; p.sc = a->scale0; p.sh = a->shift0; p.dy = (const bf16_t*)a->dy; p.dwp = a->dwp; p.dbias = a->dbias;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.cout = d->cout; p.xf = d->xform0 != OCT_XF_NONE; p.s2d = d->dy_mode == OCT_IN_S2D;
  p.tiles_x = d->w / 32; p.tiles_y = d->h / 2; p.ntiles = p.tiles_x * p.tiles_y * d->n;
  hipStream_t s = as_stream(stream);
  const int gy = d->cout / 256;
  if (d->c0 % 256 == 0) wg1_launch<8>(p, gy, d->c0 / 256, s);
  else wg1_launch<4>(p, gy, d->c0 / 128, s);
  int rc = oct_check_launch("wgemm1");
  return rc ? rc : 1;
}
