// Transposed convolutions (ConvTranspose2d k2 s2, YNet_2022.py:526-540) as what they are: plain GEMMs over the pixels.
//
//   forward        Y[pixel][(dy,dx,co)] = relu(bn(X))[pixel][ci] . W[ci][(dy,dx,co)] + b[co]     (depth-to-space store)
//   data gradient  dX[pixel][ci]        = dY[2y+dy, 2x+dx][co]  . W^T                            (space-to-depth gather)
//
// Round 2 ran them on igemm2's 3x3 machinery with one tap: four producer waves staged a 256-pixel x 32-channel chunk (16 KB) for
// 16 MFMAs per MFMA wave -- 2,250 cycles of staging per 512 cycles of matrix work, and the 256 x K tile was re-staged for every
// 128-channel block of N (8 x for upconv4): 0.17 ms for a 0.03-ms problem (profiles/r02_cfg2_launch_table.txt rows 10, 13, 39, 45).
// Here ALL EIGHT waves stage and multiply:
//   * workgroup tile 256 pixels x 256 channels of N, K in chunks of 64; wave (pr, nq) owns 4 tile rows x 64 channels
//     (4 x 2 accumulators = 128 registers), 32 MFMAs per chunk -- twice the matrix work per staged byte and half the passes over X
//     (the first version's 32-channel chunks spent as many cycles on a chunk's ~250 instructions of staging / addressing as on
//     its 16 MFMAs: 3.3 k cycles per chunk);
//   * every thread stages four 16-byte pieces per chunk: loaded two chunks ahead into registers, BN + ReLU applied on the way into
//     LDS one chunk ahead; three LDS chunk buffers (144-byte pixel pitch: conflict-free ds_read_b128), one barrier per chunk;
//   * the two waves of a SIMD run the chunk's two halves in opposite order (waves 0-3: multiply, then commit; waves 4-7: commit,
//     then multiply), so one wave's vector / LDS-write work sits beside the other's MFMAs (MI355X_MICROARCH.md, two waves per SIMD, 9);
//   * weights straight from L2 into registers in packed fragment order (one coalesced KB per fragment): the chunk multiplies
//     k16-major, and the two fragments of a k16 step are re-fetched for the NEXT chunk as soon as its four rows are done.
// Same packed weights (OCT_PACK_DECONV_FPROP / _DGRAD), same addressing modes and results as igemm2's one-tap instantiations, which
// stay for N % 256 != 0, ragged tiles, BatchNorm sums and the volumetric modes.
#include "common.h"
#include <stdlib.h>

struct Gemm1Params {
  const bf16_t* x; const float* sc; const float* sh; const bf16_t* wp; bf16_t* y; const float* bias;
  int n, h, w, c0, cout, ktot, nch, nk16, nblk, tiles_x, tiles_y, nitems, per_wg;
};

typedef unsigned int g1_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int g1_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned g1_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}

constexpr int G1_PIXB = 144;                // 64 bf16 + 16 B pad: conflict-free ds_read_b128 for the 32x32x16 lane map (9 r mod 16)
constexpr int G1_BUFB = 256 * G1_PIXB;      // one 256-pixel x 64-channel chunk
constexpr int G1_NBUF = 3;

// XF: BN + ReLU on load (forward); S2D: space-to-depth gather of the 2H x 2W input (data gradient); D2S: depth-to-space store
template <bool XF, bool S2D, bool D2S>
__global__ void __launch_bounds__(512) gemm1_kernel(const Gemm1Params p) {
  typedef Mma<bf16_t> M;
  typedef M::Frag Frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const buf0 = smem;
  float* const sxf = reinterpret_cast<float*>(smem + G1_NBUF * G1_BUFB);            // [2][c0] scale | shift (XF)
  unsigned char* const oscr = smem + G1_NBUF * G1_BUFB + 2 * 1024 * 4;                 // 8 waves x 32 px x 80 B
  float* const sbias = reinterpret_cast<float*>(oscr + 8 * 32 * 80);                   // [cout/4] (D2S)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = wave & 1, nq = (wave >> 1) & 3;       // waves w and w + 4 share a SIMD: they differ in nq (bit 1 of nq), not in role
  const bool late = wave >= 4;                         // second-dispatched half: commit first, multiply second
  const int r = lane & 31, hh = lane >> 5;

  const int it0 = blockIdx.x * p.per_wg, it1 = min(it0 + p.per_wg, p.nitems);
  if (it0 >= it1) return;
  const int nstage = (it1 - it0) * p.nch;

  if (XF) {
    for (int i = tid; i < p.c0; i += 512) { sxf[i] = p.sc[i]; sxf[p.c0 + i] = p.sh[i]; }
  }
  if (D2S)   // (zeros without a bias: the epilogue adds unconditionally)
    for (int i = tid; i < (p.cout >> 2); i += 512) sbias[i] = p.bias ? p.bias[i] : 0.f;

  // ---- staging: this thread's two 16-byte pieces of a chunk ----
  const unsigned cs2 = 2u * (unsigned)p.c0;
  unsigned goff[4];      // byte offset of the piece's pixel from the tile origin (+ its 16-B group)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int id = tid + 512 * k;
    const int pix = id >> 3, grp = id & 7;
    const int ly = pix >> 5, lx = pix & 31;
    const int rel = S2D ? (2 * ly) * (2 * p.w) + 2 * lx : ly * p.w + lx;
    goff[k] = (unsigned)rel * cs2 + (unsigned)grp * 16u;
  }
  // iterator of the stage being ISSUED (two ahead of the one being multiplied); clamps at the last stage.  Everything is a
  // counter chain: decoding (tile, block, chunk) with integer divisions cost the first version ~2 k cycles per stage.
#ifndef G1_ROTATE
#define G1_ROTATE 0   /* measured (r3, same box): no difference on any launch -- L2 hot-spotting of the lockstep weight reads is not what the chunk waits for */
#endif
  // ROTATE: workgroup b walks the K chunks of every item starting at chunk rot(b) (wrapping): without it all 256 workgroups
  // ask the L2 for the SAME weight fragments at the same moment (they run in lockstep through identical items), and every
  // byte has to be delivered 32 times per XCD through the few channels that hold it.  fp32 accumulation order changes with
  // the workgroup (not from run to run); exact-arithmetic tests are order-independent by construction.
  const int rot = G1_ROTATE ? (int)((blockIdx.x >> 3) % (unsigned)p.nch) : 0;
  const int rot_dydx = S2D ? (rot * 64) / p.c0 : 0, rot_cc = S2D ? rot * 64 - rot_dydx * p.c0 : 0;
  struct It { int ch, pch, nb, txi, tyi, img, dydx, cc; };   // ch: chunks done in this item; pch: the physical chunk
  auto it_init = [&](int item) {
    It t;
    t.ch = 0; t.pch = rot; t.dydx = rot_dydx; t.cc = rot_cc;
    t.nb = item % p.nblk;
    int tile = item / p.nblk;
    t.txi = tile % p.tiles_x; tile /= p.tiles_x;
    t.tyi = tile % p.tiles_y; t.img = tile / p.tiles_y;
    return t;
  };
  auto it_next = [&](It& t) {
    t.cc += 64;
    if (S2D && t.cc == p.c0) { t.cc = 0; ++t.dydx; }
    if (++t.pch == p.nch) { t.pch = 0; t.cc = 0; t.dydx = 0; }
    if (++t.ch == p.nch) {
      t.ch = 0; t.pch = rot; t.cc = rot_cc; t.dydx = rot_dydx;
      if (++t.nb == p.nblk) {
        t.nb = 0;
        if (++t.txi == p.tiles_x) { t.txi = 0; if (++t.tyi == p.tiles_y) { t.tyi = 0; ++t.img; } }
      }
    }
  };
  int i_left = nstage - 1;
  It it = it_init(it0);
  auto stage_base = [&]() -> const unsigned char* {
    if (S2D) {   // k = (dy*2+dx)*C + c of the 2H x 2W tensor
      const size_t o2 = ((size_t)it.img * (2 * p.h) + 2 * it.tyi * 8 + (it.dydx >> 1)) * (size_t)(2 * p.w) + 2 * it.txi * 32 + (it.dydx & 1);
      return reinterpret_cast<const unsigned char*>(p.x + o2 * p.c0 + it.cc);
    }
    const size_t origin = ((size_t)it.img * p.h + it.tyi * 8) * p.w + it.txi * 32;
    return reinterpret_cast<const unsigned char*>(p.x + origin * p.c0 + it.pch * 64);
  };
  It wt = it_init(it0);  // the stage whose weights are being fetched: one ahead of the one being multiplied
  int w_left = nstage - 1;
  auto wfrag_base = [&](const It& t) -> const unsigned char* {   // fragment (N-fragment 0 of this wave, k16 = 4*ch) of the item's block
    const int nb = t.nb * 8 + nq * 2;
    return reinterpret_cast<const unsigned char*>(p.wp + ((size_t)nb * p.nk16 + t.pch * 4) * 512);   // uniform; the lane adds lane * 16
  };
  const unsigned wl = (unsigned)lane * 16u;
  g1_u32x4 R[2][4];
  Frag W[4][2];          // [k16][q] of the chunk being multiplied; slot k16 is refilled for the next chunk right after its last use
  const size_t qs = (size_t)p.nk16 * 1024;
  auto issue_x = [&](g1_u32x4 (&Rr)[4]) {
    const unsigned char* const b = stage_base();
#pragma unroll
    for (int k = 0; k < 4; ++k) Rr[k] = *reinterpret_cast<const g1_u32x4*>(b + (size_t)goff[k]);
  };
  auto issue_w1 = [&](const unsigned char* b, int k16) {
#pragma unroll
    for (int q = 0; q < 2; ++q) W[k16][q] = M::load(b + q * qs + k16 * 1024 + (size_t)wl);
  };
  auto advance = [&]() {
    if (i_left > 0) { --i_left; it_next(it); }
  };
  int c_ch = rot;        // physical chunk of the stage being COMMITTED (for the BN coefficients); every item has nch stages
  auto commit = [&](unsigned char* buf, const g1_u32x4 (&Rr)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g1_u32x4 v = Rr[k];
      if (XF) {
        const int cg = c_ch * 64 + (tid & 7) * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sxf + cg), s1 = *reinterpret_cast<const f32x4*>(sxf + cg + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sxf + p.c0 + cg), b1 = *reinterpret_cast<const f32x4*>(sxf + p.c0 + cg + 4);
        const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = fmaxf(fmaf(__uint_as_float(v[j] << 16), s[2 * j], bb[2 * j]), 0.f);
          const float hi = fmaxf(fmaf(__uint_as_float(v[j] & 0xffff0000u), s[2 * j + 1], bb[2 * j + 1]), 0.f);
          v[j] = g1_pack(lo, hi);
        }
      }
      *reinterpret_cast<g1_u32x4*>(buf + ((tid + 512 * k) >> 3) * G1_PIXB + (tid & 7) * 16) = v;
    }
    if (++c_ch == p.nch) c_ch = 0;
  };

  __syncthreads();   // coefficient tables
  // prologue: stages 0 and 1 in flight, stage 0 committed
  issue_x(R[0]); advance();
  issue_x(R[1]); advance();
  {
    const unsigned char* const b = wfrag_base(wt);
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) issue_w1(b, k16);
    if (w_left > 0) { --w_left; it_next(wt); }     // -> stage 1
  }
  commit(buf0, R[0]);
  issue_x(R[0]);     // stage 2
  __syncthreads();

  f32x16 acc[4][2];
  It mt = it_init(it0);              // stage being multiplied
  int m_ch = 0;
  int cur = 0;                       // its LDS buffer
  const unsigned char* const lbase = buf0 + ((4 * pr) * 32 + r) * G1_PIXB + hh * 16;

  // sixteen activation fragments (k16, row m) per chunk, each feeding two MFMAs; a ring of three keeps the LDS read of fragment
  // f + 2 in flight while fragment f multiplies (the SIMD partner covers the rest of the latency).  k16-major: the weight slot of
  // a k16 step is free after its fourth row and is refilled with the next chunk's fragments there.
  auto multiply = [&]() {
    const unsigned char* lb = lbase + cur * G1_BUFB;
    const unsigned char* const wb = wfrag_base(wt);
    Frag xb[3];
    auto xoff = [](int f) constexpr { return (f & 3) * 32 * G1_PIXB + (f >> 2) * 32; };   // f = 4*k16 + m
    xb[0] = M::load(lb + xoff(0));
    xb[1] = M::load(lb + xoff(1));
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      if (f + 2 < 16) xb[(f + 2) % 3] = M::load(lb + xoff(f + 2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) M::mma(acc[f & 3][q], W[f >> 2][q], xb[f % 3]);
      if ((f & 3) == 3) issue_w1(wb, f >> 2);
    }
    if (w_left > 0) { --w_left; it_next(wt); }
  };
  auto epilogue = [&]() {   // the finished item m_item: 8 fragments, bf16, through a wave-private LDS transpose
    const int nbi = mt.nb, txi = mt.txi, tyi = mt.tyi, img = mt.img;
    unsigned char* const sc = oscr + wave * (32 * 80);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cb0 = nbi * 256 + nq * 64 + q * 32;
      int cd, co, dydx = 0;
      if (D2S) { cd = p.cout >> 2; dydx = cb0 / cd; co = cb0 - dydx * cd; } else { cd = p.cout; co = cb0; }
      constexpr bool has_bias = D2S;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int oy = tyi * 8 + 4 * pr + m;
        const size_t pix0 = D2S ? ((size_t)img * (2 * p.h) + 2 * oy + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * (txi * 32) + (dydx & 1)
                                : ((size_t)img * p.h + oy) * p.w + txi * 32;
        unsigned char* const fb = reinterpret_cast<unsigned char*>(p.y + pix0 * cd + co);
        const unsigned pstep = (D2S ? 4u : 2u) * (unsigned)cd;
        // lane (r, hh) holds channels 8g + 4hh .. +3 of pixel r in registers 4g .. 4g+3
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float a0 = acc[m][q][4 * g], a1 = acc[m][q][4 * g + 1], a2 = acc[m][q][4 * g + 2], a3 = acc[m][q][4 * g + 3];
          if (has_bias) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + co + 8 * g + 4 * hh);
            a0 += b4[0]; a1 += b4[1]; a2 += b4[2]; a3 += b4[3];
          }
          const g1_u32x2 v = {g1_pack(a0, a1), g1_pack(a2, a3)};
          *reinterpret_cast<g1_u32x2*>(sc + r * 80 + (8 * g + 4 * hh) * 2) = v;
        }
        g1_u32x4 tv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int chunk = lane + 64 * k;
          tv[k] = *reinterpret_cast<const g1_u32x4*>(sc + (chunk >> 2) * 80 + (chunk & 3) * 16);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int chunk = lane + 64 * k;
          *reinterpret_cast<g1_u32x4*>(fb + (__umul24((unsigned)(chunk >> 2), pstep) + (unsigned)(chunk & 3) * 16u)) = tv[k];
        }
      }
    }
  };

  // two stages per trip: the register rings' slots are compile-time; the stage count is padded to even (clamped iterators
  // make the extra stage a harmless repeat whose result is never stored)
  const int nstage2 = (nstage + 1) & ~1;
  for (int g0 = 0; g0 < nstage2; g0 += 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = g0 + j;
      const bool live = g < nstage;
      if (live && m_ch == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
      }
      const int nb = cur + 1 == G1_NBUF ? 0 : cur + 1;
      // R[j] holds stage g + 2 (issued one trip ago), R[j ^ 1] stage g + 1: commit that one while this stage multiplies
      if (late) commit(buf0 + nb * G1_BUFB, R[j ^ 1]);
      multiply();              // (a padded last stage multiplies once more into accumulators nobody stores)
      if (!late) commit(buf0 + nb * G1_BUFB, R[j ^ 1]);
      // refill the slot just committed: activations of stage g + 3
      advance();
      issue_x(R[j ^ 1]);
      if (live && m_ch == p.nch - 1) epilogue();
      __syncthreads();
      cur = nb;
      if (live) { it_next(mt); if (++m_ch == p.nch) m_ch = 0; }
    }
  }
}

static bool g1_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_GEMM1"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}

// returns 1 when taken, 0 when the shape is not eligible (the caller falls through to igemm2's one-tap kernels)
int oct_conv_forward_g1(const OctConvDesc* d, const OctConvArgs* a, void* stream) {
  if (!g1_enabled() || d->taps != 1 || d->dtype != OCT_DT_BF16) return 0;
  const bool fwd = d->in_mode == OCT_IN_PLAIN && d->out_mode == OCT_OUT_D2S;
  const bool bwd = d->in_mode == OCT_IN_S2D && d->out_mode == OCT_OUT_PLAIN;
  if (!fwd && !bwd) return 0;
  if (d->c1 != 0 || d->want_stats || d->split != 0 || d->depth != 0 || d->out_img_mul != 0) return 0;
  if ((d->w % 32) != 0 || (d->h % 8) != 0 || (d->c0 % 64) != 0 || (d->cout % 256) != 0) return 0;   // 64-channel chunks
  if (fwd && ((d->cout >> 2) % 32) != 0) return 0;
  if (fwd && d->xform0 != OCT_XF_AFFINE_RELU && d->xform0 != OCT_XF_NONE) return 0;
  if (bwd && d->xform0 != OCT_XF_NONE) return 0;
  const int ktot = bwd ? 4 * d->c0 : d->c0;
  if (d->c0 > 1024 || d->cout > 4096) return 0;
  if (ktot < 256) return 0;   // K = 128 (upconv2 forward at 128 x 256): HBM-bound, 3 % slower here than on igemm2 (same box, r3)
  Gemm1Params p;
  p.x = (const bf16_t*)a->x0; p.sc = a->scale0; p.sh = a->shift0; p.wp = (const bf16_t*)a->wpacked;
  p.y = (bf16_t*)a->y0; p.bias = a->bias;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.cout = d->cout; p.ktot = ktot; p.nch = ktot / 64; p.nk16 = ktot / 16;
  p.nblk = d->cout / 256; p.tiles_x = d->w / 32; p.tiles_y = d->h / 8;
  p.nitems = p.tiles_x * p.tiles_y * d->n * p.nblk;
  int grid = p.nitems < 256 ? p.nitems : 256;
  p.per_wg = (p.nitems + grid - 1) / grid;
  grid = (p.nitems + p.per_wg - 1) / p.per_wg;
  const int lds = G1_NBUF * G1_BUFB + 2 * 1024 * 4 + 8 * 32 * 80 + 1024 * 4;
  hipStream_t s = as_stream(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  if (fwd) {
    if (d->xform0 != OCT_XF_NONE) hipLaunchKernelGGL((gemm1_kernel<true, false, true>), dim3(grid), dim3(512), lds, s, p);
    else hipLaunchKernelGGL((gemm1_kernel<false, false, true>), dim3(grid), dim3(512), lds, s, p);
  } else {
    hipLaunchKernelGGL((gemm1_kernel<false, true, false>), dim3(grid), dim3(512), lds, s, p);
  }
  int rc = oct_check_launch("gemm1");
  return rc ? rc : 1;
}
