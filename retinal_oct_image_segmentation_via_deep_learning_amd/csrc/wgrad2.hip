// Weight gradient, pipelined bf16 path for regular shapes (W % 32 == 0, H % 8 == 0, all channel
// counts multiples of 32).  Same math as wgrad.hip:
//
//   dW[tap][co][ci] = sum over pixels  dY[p][co] * A[p + tap][ci]
//
// Round-1 profile: the generic kernel gathered every MFMA fragment with 8 scalar LDS reads and was
// LDS-issue bound (26.8 of 77 ms per step).  Here
//  * the contraction index (pixel) is brought onto the MFMA k axis with ds_read_b64_tr_b16: the LDS
//    tiles keep the NHWC order they are staged in ([pixel][32 channels], 64-B rows, so four pixel
//    rows of a transposed read cover all 64 banks exactly once) and the hardware transposes
//    4 pixels x 16 channels per 16-lane group -- two reads per fragment instead of eight;
//  * two producer waves stage dY (plain copy) and the input halo tile (BN+ReLU fused, virtual
//    concat) two stages ahead of four MFMA waves, double-buffered LDS, one barrier per stage;
//  * each MFMA wave keeps all 9 tap accumulators of one 32(co) x 32(ci) block (144 registers)
//    across the whole persistent tile loop and issues its fp32 atomics once at the end.
#include "common.h"
#include <stdlib.h>

struct Wgrad2Params {
  const bf16_t* x0; const bf16_t* x1;
  const float* sc0; const float* sh0; const float* sc1; const float* sh1;
  const bf16_t* dy; float* dwp; float* dbias;
  int n, h, w, c0, c1, ktot, cout, xf0, xf1, dy_mode;
  int tiles_x, tiles_y, ntiles, per_wg;
  int interleave;   // 1: workgroup x walks tiles x, x + gridDim.x, ... (see igemm2.hip), 0: a contiguous tile range
  int depth, img_shift;   // 3-D: the input tile comes from slice d + img_shift of the same volume (all zero outside)
  int dy_mul, dy_add;     // S2D dY gathered from image img*dy_mul + dy_add (0: identity)
  // partials mode (OctWgradDesc.partials): every (workgroup column, row strip) writes its accumulators with plain stores
  // into its own slab dwp[slab][tap][cout][ktot] (and dbias_part[slab][cout]); oct_unpack_wgrad sums the slabs in order.
  // No atomics: the result does not depend on scheduling, and nothing has to be zeroed.
  int part_mode; size_t slab_elems; float* dbias_part;
  // RSH (7x3 kernels, three launches of three tap rows): the staged input tile starts ty0 rows below the 7x3 halo corner
  // (y0 - pad_y, x0 - 1); only the first `nstore` of the nine accumulators are real taps (the last launch has one tap row)
  int ty0, pad_y, nstore;
  unsigned long long* trace;   // diagnostic builds only (-DOCT_TRACE): s_memtime stamps of workgroup (0,0,0)
};

#ifdef OCT_TRACE
#define W2TRACE(slot, idx)                                                                                     \
  do {                                                                                                         \
    if (p.trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 && (idx) < 256) \
      p.trace[(slot) * 256 + (idx)] = __builtin_amdgcn_s_memtime();                                            \
  } while (0)
static unsigned long long* g_trace_w2 = nullptr;
extern "C" void oct_debug_set_trace_w2(void* buf) { g_trace_w2 = (unsigned long long*)buf; }
#else
#define W2TRACE(slot, idx) do {} while (0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned w2_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}

// one MFMA operand fragment (8 contraction pixels per lane) through two transposed LDS reads
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base_lo) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo + 4 * 64));  // pixels +4
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// the same with the two reads addressed separately (swizzled tiles: the second read's half can differ per lane)
__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* first, const unsigned char* second) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(first));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(second));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// D3: depth shift of the input tile / image map of an S2D dY (volumetric network) -- compile-time, see igemm2.hip
// RSH: the nine taps are rows ty0 .. ty0 + 2 of a TALLER kernel (ReLayNet's 7x3, ReLayNet_2017.py:155-160, padding (3, 1)):
// dW[ty0 + ty][tx] = sum dY[y][x] * A[y - pad_y + ty0 + ty][x - 1 + tx].  Same tile, same loop; the input tile is fetched
// ty0 - pad_y + 1 rows further down and its rows are checked against the image by range instead of by halo flags (a shifted
// tile reaches padding on more tile rows than the first and the last).
template <int TAPS, int CB, int IB, int TH, bool RAGGED, bool D3 = false, bool RSH = false>
__global__ void __launch_bounds__(512) wgrad2_kernel(const Wgrad2Params p) {
  constexpr int TW = 32;
  // TAPS = 9: 3x3; TAPS = 3: one row of three (the seventh row of ReLayNet's 7x3, always row-shifted: RSH); TAPS = 1: 1x1
  constexpr int HALO = (TAPS != 1) ? 1 : 0;        // columns
  constexpr int HALO_Y = (TAPS == 9) ? 1 : 0;      // rows
  static_assert(TAPS != 3 || RSH, "the 1x3 kernel exists as a row group of a taller one");
  constexpr int LH = TH + 2 * HALO_Y, LW = TW + 2 * HALO;
  constexpr int NPI = LH * LW, NPD = TH * TW;          // pixels of the input / dY tile
  constexpr int INB = NPI * 64, DYB = NPD * 64;        // bytes per 32-channel block
  constexpr int STAGEB = IB * INB + CB * DYB;
#ifndef W2_DYDMA
#define W2_DYDMA 0   /* measured (r3, same box): +3.5 % on the weight-gradient launches, see DESIGN.md */
#endif
  // DMA: dY needs no transform on the way in and its LDS image ([pixel][64 B], dense) is exactly the order in which the
  // producer lanes fetch it, so it goes global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction): no
  // registers, no ds_write, no commit work -- the producers' commit + issue (5.76 k cycles per stage against 5.35 k of MFMA
  // phase, profiles/r02_wgrad2_timeline.txt) was what the stage waited for.  Ragged tiles (dY rows beyond the image must
  // read as zero) keep the register path.
  constexpr bool DMA = W2_DYDMA && !RAGGED;
#ifndef W2_M16
#define W2_M16 1
#endif
#ifndef W2_SWZ
#define W2_SWZ 1
#endif
  // W16 (3x3 kernels): v_mfma_f32_16x16x32_bf16 -- a fragment is 32 pixels (a whole tile row) x 16 channels, D = 16 co x 16 ci
  // in quarter S = 2*(co half) + (ci half) of the tap's accumulator: register 4S + e = (co 16a + 4*(lane>>4) + e, ci 16b + (lane&15)).
  // Same FLOPs, LDS reads and loop as the 32x32x16 form (pixel halves become channel halves); the chip holds a higher clock
  // on this shape (igemm2.hip, M16).
  constexpr bool W16 = W2_M16 && TAPS == 9;
  // SWZ: in the W16 form the two 16-lane groups that a ds_read_b64_tr_b16 services together read pixels p and p + 8 of the SAME
  // 32-byte channel half -- 512 B apart, i.e. the same banks: a 2-way conflict on every transposed read (4 LDS cycles instead
  // of 2; tools/lds_swizzle_check.py).  Pixels whose tile column has bit 3 set therefore store their two 32-B halves swapped
  // (chunk g at g ^ 2), and a reader lane takes half h ^ bit3(column).
  constexpr bool SWZ = W2_SWZ && W16;
  // MULTI (1 x 1 kernels only: one accumulator per pair, 16 registers): 8 or 16 pairs per workgroup, the four waves as a
  // 2 x 2 grid over (co, ci), each owning WCB x WIB pairs of the same staged tile.  A 1 x 1 weight gradient is a plain GEMM
  // over the pixels with nothing but staging between memory and the matrix pipe: with 64 x 64 blocks the deep transposed
  // convolutions staged their input 16x and dY 8x (upconv4: 2.1 GB for a 0.2 GB problem, 0.29 ms) -- 128 x 128 halves both.
  constexpr int PAIRS = CB * IB;
  constexpr bool MULTI = PAIRS > 4;
  constexpr int WCB = MULTI ? CB / 2 : 1, WIB = MULTI ? IB / 2 : 1;
  constexpr int PS = MULTI ? 1 : 4 / PAIRS, ROWS = TH / PS;
  static_assert(PAIRS == 1 || PAIRS == 2 || PAIRS == 4 || (TAPS == 1 && CB == 4 && (IB == 2 || IB == 4)),
                "1, 2 or 4 channel-block pairs per workgroup (1 x 1: also 4 x 2 and 4 x 4)");
  typedef Mma<bf16_t> M;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co_sb = blockIdx.y * (32 * CB), ci_sb = blockIdx.z * (32 * IB);
  int t0, tstep, nstage;
  if (p.interleave) {
    if ((int)blockIdx.x >= p.ntiles) return;
    t0 = blockIdx.x; tstep = gridDim.x;
    nstage = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  } else {
    t0 = blockIdx.x * p.per_wg; tstep = 1;
    const int t1 = min(t0 + p.per_wg, p.ntiles);
    if (t0 >= t1) return;
    nstage = t1 - t0;
  }
  constexpr int DRING = ((CB * IB >= 2 && TH == 8) || TH == 16 || MULTI) ? 2 : 4;   // stages of loads in flight per producer thread (register budget)
  const int nstage_pad = (nstage + DRING - 1) / DRING * DRING;

  // BN scale/shift of this workgroup's input channels in LDS (kept off the in-order vmcnt queue)
  float* const sxf = reinterpret_cast<float*>(smem + 2 * STAGEB);   // [2][32*IB]
  for (int i = tid; i < 32 * IB; i += 512) {
    const int cg = ci_sb + i;
    const bool second = cg >= p.c0;
    const bool xf = second ? (p.xf1 != 0) : (p.xf0 != 0);
    sxf[i] = xf ? (second ? p.sc1[cg - p.c0] : p.sc0[cg]) : 1.f;
    sxf[32 * IB + i] = xf ? (second ? p.sh1[cg - p.c0] : p.sh0[cg]) : 0.f;
  }
  __syncthreads();

  if (wave >= 4) {
    // ============================== producer waves ==============================
#ifdef W2_PROD_PRIO
    __builtin_amdgcn_s_setprio(W2_PROD_PRIO);
#endif
    const int ptid = tid - 256, g = ptid & 3, pb = ptid >> 2;
    constexpr int SIB = (NPI * 4 + 255) / 256;   // input slots per 32-channel block (256 producer threads)
    constexpr int SDB = (NPD * 4 + 255) / 256;   // dY slots per 32-row block
    constexpr int D = DRING;
    struct Stage { u32x4 ri[IB][SIB]; u32x4 rd[DMA ? 1 : CB][DMA ? 1 : SDB]; unsigned vm[IB]; unsigned vd; int img, tyi, txi; };
    Stage R[D];
    const int pw = __builtin_amdgcn_readfirstlane(wave) - 4;   // producer wave 0..3, scalar: the LDS-DMA destination is wave-uniform
    // per-slot constants (shared by all blocks): pixel offset from the tile origin + border code
    int reli[SIB], reld[SDB];
    int lyv[RSH ? SIB : 1];   // RSH: local row of the slot
    unsigned code[SIB], dcode[SDB];   // bits 8-15 / 16-23: local row / column (ragged last tiles), low bits: halo flags
#pragma unroll
    for (int j = 0; j < SIB; ++j) {
      const int pix = pb + 64 * j;
      const int ly = pix / LW, lx = pix - ly * LW;
      reli[j] = (RSH ? ly + p.ty0 : ly) * p.w + lx;   // relative to the tile's halo corner: never negative (a scalar base + unsigned lane offset per load)
      if constexpr (RSH) lyv[j] = ly;
      // bottom / right flags against the LAST tile row / column of the image (see igemm2.hip): the halo
      // row / column for whole tiles, everything beyond H, W for a ragged size
      const int ylast = p.h - (p.tiles_y - 1) * TH + HALO_Y, xlast = p.w - (p.tiles_x - 1) * TW + HALO;
      unsigned c = pix >= NPI ? 16u : 0u;
      if (HALO) c |= ((!RSH && HALO_Y && ly == 0) ? 1u : 0u) | (lx == 0 ? 4u : 0u);
      c |= ((!RSH && ly >= ylast) ? 2u : 0u) | (lx >= xlast ? 8u : 0u);
      code[j] = c | (SWZ ? (unsigned)(((lx >> 3) & 1) << 1) << 8 : 0u);   // bits 8-9: chunk swizzle of this pixel
    }
    unsigned dswz[SDB];
#pragma unroll
    for (int j = 0; j < SDB; ++j) {
      const int pix = pb + 64 * j;  // NPD is a multiple of 64: every dY slot is live
      const int ly = pix / TW, lx = pix - ly * TW;
      dswz[j] = SWZ ? (unsigned)(((lx >> 3) & 1) << 1) : 0u;
      reld[j] = (p.dy_mode == OCT_IN_S2D) ? (2 * ly) * (2 * p.w) + 2 * lx : ly * p.w + lx;
      dcode[j] = RAGGED ? ((ly >= p.h - (p.tiles_y - 1) * TH ? 2u : 0u) | (lx >= p.w - (p.tiles_x - 1) * TW ? 8u : 0u)) : 0u;
    }
    auto issue = [&](int s, Stage& S) {
      int t = t0 + s * tstep;
      const int txi = t % p.tiles_x; t /= p.tiles_x;
      const int tyi = t % p.tiles_y; const int img = t / p.tiles_y;
      const unsigned edge = 16u | (tyi == 0 ? 1u : 0u) | (tyi == p.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) |
                            (txi == p.tiles_x - 1 ? 8u : 0u);
      const size_t origin = ((size_t)img * p.h + tyi * TH) * p.w + txi * TW;
      S.img = img; S.tyi = tyi; S.txi = txi;
      // RSH: local rows [ylo, ylo + ynum) of the shifted tile lie inside the image (row ly is image row y0 - pad_y + ty0 + ly)
      int ylo = 0; unsigned ynum = 0;
      if constexpr (RSH) {
        const int ytop = tyi * TH - p.pad_y + p.ty0;
        ylo = max(0, -ytop);
        ynum = (unsigned)max(0, min(LH, p.h - ytop) - ylo);
      }
      // 3-D: the input tile of this depth tap lies one slice up / down; outside the volume it is padding (the loads
      // then re-read the tile's own slice and every slot is marked dead)
      bool zok = true;
      size_t in_origin = origin;
      if (D3 && p.depth > 0) {
        const int dz = img % p.depth + p.img_shift;
        zok = dz >= 0 && dz < p.depth;
        if (zok) in_origin = ((size_t)(img + p.img_shift) * p.h + tyi * TH) * p.w + txi * TW;
      }
#pragma unroll
      for (int blk = 0; blk < IB; ++blk) {
        const int cg = ci_sb + blk * 32;
        const bool second = cg >= p.c0;
        const int cs = second ? p.c1 : p.c0;
        // wave-uniform 64-bit base = the tile's halo corner (it may lie outside the tensor: such slots are dead and read
        // the tile origin instead) + an unsigned 32-bit lane offset: `global_load_dwordx4 v, v_off, s[base]` -- the per-lane
        // 64-bit address arithmetic this replaces was 4-5 vector instructions per load, 20 loads per stage, on the waves
        // the stage waits for (producers 5.76 k cycles against 5.35 k of MFMA phase, profiles/r02_wgrad2_timeline.txt)
        const bf16_t* base = second ? p.x1 + in_origin * p.c1 + (cg - p.c0) : p.x0 + in_origin * p.c0 + cg;
        const unsigned corner = RSH ? (unsigned)(p.pad_y * p.w + HALO) : (unsigned)(HALO * (p.w + 1));   // pixels from the halo corner to the tile origin
        const unsigned cs2 = 2u * (unsigned)cs, safe = corner * cs2 + (unsigned)g * 16u;
        const unsigned char* const hb = reinterpret_cast<const unsigned char*>(base) - (size_t)corner * cs2;
        // unconditional loads (see igemm2.hip): invalid slots re-read the tile origin, zeroed at commit
        unsigned vm = 0;
#pragma unroll
        for (int j = 0; j < SIB; ++j) {
          bool ok = (!D3 || zok) && (code[j] & edge) == 0;
          if constexpr (RSH) ok = ok && (unsigned)(lyv[j] - ylo) < ynum;   // image rows only
          S.ri[blk][j] = *reinterpret_cast<const u32x4*>(hb + (ok ? __umul24((unsigned)reli[j], cs2) + (unsigned)g * 16u : safe));
          vm |= ok ? (1u << j) : 0u;
        }
        S.vm[blk] = vm;
      }
      if constexpr (!DMA) {
#pragma unroll
      for (int blk = 0; blk < CB; ++blk) {
        const int row = co_sb + blk * 32;  // GEMM row = output channel (or (dydx, co) for the deconv)
        const bf16_t* base;
        int cs;
        if (p.dy_mode == OCT_IN_S2D) {
          cs = p.cout >> 2;
          const int dydx = row / cs, co = row - dydx * cs;
          const int img2 = (D3 && p.dy_mul) ? img * p.dy_mul + p.dy_add : img;
          const size_t o2 = ((size_t)img2 * (2 * p.h) + 2 * tyi * TH + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * txi * TW + (dydx & 1);
          base = p.dy + o2 * cs + co;
        } else {
          cs = p.cout;
          base = p.dy + origin * cs + row;
        }
        const unsigned cs2 = 2u * (unsigned)cs;
        const unsigned char* const db = reinterpret_cast<const unsigned char*>(base);   // uniform base + unsigned lane offset
        unsigned vd = 0;
#pragma unroll
        for (int j = 0; j < SDB; ++j) {
          const bool ok = !RAGGED || (dcode[j] & edge) == 0;   // dY pixels of a ragged last tile beyond the image
          S.rd[blk][j] = *reinterpret_cast<const u32x4*>(db + ((ok ? __umul24((unsigned)reld[j], cs2) : 0u) + (unsigned)g * 16u));
          vd |= ok ? (1u << j) : 0u;
        }
        if (RAGGED) S.vd = vd;
      }
      }
    };
    // dY tile of the stage whose coordinates `S` holds -> LDS, no registers in between.  Lane i of producer wave pw fetches
    // the 16 B that belong at byte (256 j + 64 pw + i) * 16 of the block: one contiguous KB per instruction.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gl_void;
    auto dma_dy = [&](unsigned char* buf, const Stage& S) {
      const int img = S.img, tyi = S.tyi, txi = S.txi;
#pragma unroll
      for (int blk = 0; blk < CB; ++blk) {
        const int row = co_sb + blk * 32;
        const bf16_t* base;
        int cs;
        if (p.dy_mode == OCT_IN_S2D) {
          cs = p.cout >> 2;
          const int dydx = row / cs, co = row - dydx * cs;
          const int img2 = (D3 && p.dy_mul) ? img * p.dy_mul + p.dy_add : img;
          const size_t o2 = ((size_t)img2 * (2 * p.h) + 2 * tyi * TH + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * txi * TW + (dydx & 1);
          base = p.dy + o2 * cs + co;
        } else {
          cs = p.cout;
          base = p.dy + (((size_t)img * p.h + tyi * TH) * p.w + txi * TW) * cs + row;
        }
        unsigned char* const dst = buf + IB * INB + blk * DYB + pw * 1024;
#pragma unroll
        for (int j = 0; j < SDB; ++j)   // a DMA writes LDS in lane order: the swizzle goes on the source chunk
          __builtin_amdgcn_global_load_lds((gl_void*)(base + __mul24(reld[j], cs) + (int)(((unsigned)g ^ dswz[j]) * 8u)), (lds_void*)(dst + j * 4096), 16, 0, 0);
      }
    };
    // s_waitcnt vmcnt(N), everything else at its maximum (gfx9 encoding: vmcnt = simm16[15:14 | 3:0])
    constexpr int NXL = IB * SIB;   // input loads of one stage: issued AFTER the stage's DMA, they stay in flight across the wait
    constexpr int WAIT_DMA = (NXL & 15) | ((NXL >> 4) << 14) | (7 << 4) | (0 << 8);   // ... and lgkmcnt(0): the commit's ds_writes
    // Stage barrier of the producers.  With a DMA in the stage it is the counted wait + a raw s_barrier: __syncthreads()
    // would add vmcnt(0) (the pending LDS write of a DMA sits on the VM counter) and drain the prefetched stages too.
    auto stage_barrier = [&]() {
      if constexpr (DMA) { __builtin_amdgcn_s_waitcnt(WAIT_DMA); asm volatile("s_barrier" ::: "memory"); }
      else __syncthreads();
    };
    auto commit = [&](unsigned char* buf, const Stage& S) {
#pragma unroll
      for (int blk = 0; blk < IB; ++blk) {
        const int cg = ci_sb + blk * 32;
        const bool second = cg >= p.c0;
        const bool xf = second ? (p.xf1 != 0) : (p.xf0 != 0);
        const float flo = xf_floor(second ? p.xf1 : p.xf0);   // wave-uniform: 0 (BN + ReLU) or -inf (plain affine)
        const unsigned flo_pk = xf_floor_pk(second ? p.xf1 : p.xf0);
        float s[8], b[8];
        if (xf) {
          const float* sc = sxf + blk * 32 + g * 8;
          const float* sh = sxf + 32 * IB + blk * 32 + g * 8;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[j] = s0[j]; s[4 + j] = s1[j]; b[j] = b0[j]; b[4 + j] = b1[j]; }
        }
#pragma unroll
        for (int j = 0; j < SIB; ++j) {
          if ((code[j] & 16u) == 0) {
            u32x4 v = S.ri[blk][j];
            if (xf) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (OCT_PK_RELU) {
                  v[e] = pk_clamp_bf16(w2_pack(fmaf(__uint_as_float(v[e] << 16), s[2 * e], b[2 * e]),
                                               fmaf(__uint_as_float(v[e] & 0xffff0000u), s[2 * e + 1], b[2 * e + 1])), flo_pk);
                } else {
                const float lo = fmaxf(fmaf(__uint_as_float(v[e] << 16), s[2 * e], b[2 * e]), flo);
                const float hi = fmaxf(fmaf(__uint_as_float(v[e] & 0xffff0000u), s[2 * e + 1], b[2 * e + 1]), flo);
                v[e] = w2_pack(lo, hi);
                }
              }
            }
            const bool live = ((S.vm[blk] >> j) & 1u) != 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = live ? v[e] : 0u;
            *reinterpret_cast<u32x4*>(buf + blk * INB + (pb + 64 * j) * 64 + ((unsigned)g ^ ((code[j] >> 8) & 3u)) * 16) = v;
          }
        }
      }
      if constexpr (!DMA)
#pragma unroll
      for (int blk = 0; blk < CB; ++blk)
#pragma unroll
        for (int j = 0; j < SDB; ++j) {
          u32x4 v = S.rd[blk][j];
          if (RAGGED) {
            const bool live = ((S.vd >> j) & 1u) != 0;   // dY rows / columns of a ragged tile beyond the image: zero
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = live ? v[e] : 0u;
          }
          *reinterpret_cast<u32x4*>(buf + IB * INB + blk * DYB + (pb + 64 * j) * 64 + ((unsigned)g ^ dswz[j]) * 16) = v;
        }
    };
    const int last = nstage - 1;
#pragma unroll
    for (int j = 0; j < D; ++j) issue(min(j, last), R[j]);
    commit(smem, R[0]);
    if constexpr (DMA) dma_dy(smem, R[0]);
    issue(min(D, last), R[0]);
    // The DMA sits between this stage's commit (whose registers were waited for with a COUNTED vmcnt: no DMA was pending
    // then) and the next loads; the explicit wait retires it and leaves those loads in flight.  It must be the builtin:
    // hipcc has to see the DMA retired, or every later wait of the ring becomes vmcnt(0).
    stage_barrier();
    // branch-free steady state over the padded stage count (see igemm2.hip)
    for (int s0 = 0; s0 < nstage_pad; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int nx = s0 + j + 1;
        if (wave == 4) W2TRACE(4, nx - 1);
        commit(smem + (nx & 1) * STAGEB, R[(j + 1) % D]);
        if constexpr (DMA) dma_dy(smem + (nx & 1) * STAGEB, R[(j + 1) % D]);   // stage nx's tile: buffer nx & 1 was last read in stage nx - 2
        if (wave == 4) W2TRACE(5, nx - 1);
        issue(min(nx + D, last), R[(j + 1) % D]);
        if (wave == 4) W2TRACE(6, nx - 1);
        stage_barrier();
        if (wave == 4) W2TRACE(7, nx - 1);
      }
    }
    return;
  }

  // ================================ MFMA waves ================================
#ifndef W2_MFMA_PRIO
#define W2_MFMA_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(W2_MFMA_PRIO);  // win issue arbitration against the co-resident producer wave
  const int pair = MULTI ? 0 : wave % PAIRS, psx = MULTI ? 0 : wave / PAIRS;
  const int cb = MULTI ? (wave >> 1) * WCB : pair / IB, ib = MULTI ? (wave & 1) * WIB : pair % IB;   // MULTI: first pair of the wave's sub-block
  const int g4 = lane >> 4, li = lane & 15;
  // transposed-read lane address inside a [pixel][64 B] block: pixel 8*(g4>>1) + (li>>2), channel 16*(g4&1) + 4*(li&3)
  // (W16: pixel 8*g4 + (li>>2), channel 4*(li&3) of the 16-channel half)
  const int lane_off = W16 ? (8 * g4 + (li >> 2)) * 64 + (4 * (li & 3)) * 2
                           : (8 * (g4 >> 1) + (li >> 2)) * 64 + (16 * (g4 & 1) + 4 * (li & 3)) * 2;
  // SWZ: lane offsets with the channel half folded in.  lo1[h]: pixel 8*g4 + (li>>2) (tile column bit 3 = g4 & 1: the
  // tap column tx <= 2 and li>>2 <= 3 cannot carry into it), half h ^ (g4 & 1).  lo2[tx-1][h]: the SECOND read of an input
  // fragment (pixel + 4) at tap column tx = 1, 2, where tx + (li>>2) + 4 can reach 8 and flip the bit for some lanes.
  int lo1[2], lo2[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    lo1[h] = lane_off + ((h ^ (g4 & 1)) * 32);
#pragma unroll
    for (int tx = 1; tx <= 2; ++tx) {
      const int carry = ((tx + (li >> 2) + 4) >> 3) & 1;
      lo2[tx - 1][h] = lane_off + 4 * 64 + ((h ^ (g4 & 1) ^ carry) * 32);
    }
  }

  constexpr int NACC = MULTI ? WCB * WIB : TAPS;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // bias gradient = sum over pixels of dY: one more MFMA per k-step against an all-ones operand
  const bool do_bias = (p.dbias != nullptr) && (ib == 0) && (blockIdx.z == 0);
  f32x16 accb, accb2[MULTI ? WCB : 1];   // MULTI: one bias accumulator per co block of the wave
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
#pragma unroll
  for (int j = 0; j < (MULTI ? WCB : 1); ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accb2[j][i] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  __syncthreads();
  int cur = 0;
  for (int s = 0; s < nstage_pad; ++s) {
    if (s >= nstage) { __syncthreads(); continue; }   // padded stages keep the barrier count in step
    if (wave == 0) W2TRACE(0, s);
    const unsigned char* in_t = smem + cur * STAGEB + ib * INB + lane_off;
    const unsigned char* dy_t = smem + cur * STAGEB + IB * INB + cb * DYB + lane_off;
    // flattened (row, half, tap) sequence with the transposed reads running two MFMAs ahead
    constexpr int NK = ROWS * 2;                 // k-steps (16 pixels) of this wave per stage
    auto a_off = [&](int k) { return ((psx * ROWS + (k >> 1)) * TW + (k & 1) * 16) * 64; };
    auto b_off = [&](int k, int t) {
      const int ty = (TAPS == 9) ? t / 3 : 0, tx = (TAPS != 1) ? t % 3 : 0;
      return ((psx * ROWS + (k >> 1) + ty) * LW + (k & 1) * 16 + tx) * 64;
    };
    if constexpr (TAPS == 9) {
      // Walk the INPUT rows of the wave's strip: the fragment of (input row i, half, tx) is read once and
      // multiplied with the dY fragments of output rows i, i-1, i-2 (taps ty = 0, 1, 2), which sit in a
      // four-slot register window.  0.44 transposed reads per MFMA instead of 1.11: with one read per MFMA
      // the four waves asked the LDS for 142 B/clk, more than the 128 B/clk it delivers.
      constexpr int NI = ROWS + 2, NB = NI * 6;
      // j = (input row, half, tx): half = 16-pixel half of the row, or (W16) 16-channel half of the 32-pixel fragment
      auto bo = [&](int j) {
        return W16 ? ((psx * ROWS + j / 6) * LW + j % 3) * 64 + ((j / 3) & 1) * 32
                   : ((psx * ROWS + j / 6) * LW + ((j / 3) & 1) * 16 + j % 3) * 64;
      };
      auto ao = [&](int k) { return W16 ? ((psx * ROWS + (k >> 1)) * TW) * 64 + (k & 1) * 32 : a_off(k); };
      // swizzled tiles: per-stage lane bases (the row / tap column of a fragment stays a compile-time DS offset)
      const unsigned char* const in_b = smem + cur * STAGEB + ib * INB + (psx * ROWS) * LW * 64;
      const unsigned char* const dy_b = smem + cur * STAGEB + IB * INB + cb * DYB + (psx * ROWS) * TW * 64;
      const unsigned char* in1[2] = {in_b + lo1[0], in_b + lo1[1]};
      const unsigned char* in2[2][2] = {{in_b + lo2[0][0], in_b + lo2[0][1]}, {in_b + lo2[1][0], in_b + lo2[1][1]}};
      const unsigned char* dy1[2] = {dy_b + lo1[0], dy_b + lo1[1]};
      auto in_frag = [&](int j) -> bf16x8 {
        if constexpr (SWZ) {
          const int h = (j / 3) & 1, tx = j % 3, rc = ((j / 6) * LW + tx) * 64;
          return tx == 0 ? tr_frag2(in1[h] + rc, in1[h] + rc + 4 * 64) : tr_frag2(in1[h] + rc, in2[tx - 1][h] + rc);
        } else return tr_frag(in_t + bo(j));
      };
      auto dy_frag = [&](int k) -> bf16x8 {
        if constexpr (SWZ) { const int rc = ((k >> 1) * TW) * 64; return tr_frag2(dy1[k & 1] + rc, dy1[k & 1] + rc + 4 * 64); }
        else return tr_frag(dy_t + ao(k));
      };
      // input fragments run LA iterations (up to 3 MFMAs = 96 matrix cycles each) ahead of their use.  LA = 2 left the
      // reads ~190 cycles of lead, about one loaded-LDS round trip; W2_LA (default 4) doubles it for 8 VGPRs.
#ifndef W2_LA
#define W2_LA 2   /* A/B on the box (LA = 2, 4, 6): no difference on any launch -- the kernel is not LDS-latency bound */
#endif
      constexpr int LA = W2_LA;
      bf16x8 aw[4][2];
      bf16x8 bq[LA + 1];
      aw[0][0] = dy_frag(0);
      aw[0][1] = dy_frag(1);
#pragma unroll
      for (int j = 0; j < LA; ++j) bq[j] = in_frag(j);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int hx = 0; hx < 6; ++hx) {
          const int j = i * 6 + hx, h = hx / 3, tx = hx % 3;
          if (j + LA < NB) bq[(j + LA) % (LA + 1)] = in_frag(j + LA);
          if (hx == 0 && i + 1 < ROWS) {   // dY of the next output row, a whole input row ahead of its first use
            aw[(i + 1) & 3][0] = dy_frag(2 * (i + 1));
            aw[(i + 1) & 3][1] = dy_frag(2 * (i + 1) + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const int k = i - ty;   // output row that sees input row i through tap row ty
            if (k < 0 || k >= ROWS) continue;
            if constexpr (W16) {   // h = ci half of the input fragment; both co halves of the dY row multiply it
              if (do_bias && ty == 0 && tx == 0 && h == 0) { M::template mma16<0>(accb, aw[k & 3][0], ones); M::template mma16<2>(accb, aw[k & 3][1], ones); }
              if (h == 0) { M::template mma16<0>(acc[ty * 3 + tx], aw[k & 3][0], bq[j % (LA + 1)]); M::template mma16<2>(acc[ty * 3 + tx], aw[k & 3][1], bq[j % (LA + 1)]); }
              else { M::template mma16<1>(acc[ty * 3 + tx], aw[k & 3][0], bq[j % (LA + 1)]); M::template mma16<3>(acc[ty * 3 + tx], aw[k & 3][1], bq[j % (LA + 1)]); }
            } else {
            if (do_bias && ty == 0 && tx == 0) M::mma(accb, aw[k & 3][h], ones);
            M::mma(acc[ty * 3 + tx], aw[k & 3][h], bq[j % (LA + 1)]);
            }
          }
        }
    } else if constexpr (MULTI) {
      // per 16-pixel k-step: WCB dY fragments x WIB input fragments -> WCB * WIB MFMAs, the fragments of step k + 1 read
      // while step k multiplies
      bf16x8 af[2][WCB], bf[2][WIB];
#pragma unroll
      for (int j = 0; j < WCB; ++j) af[0][j] = tr_frag(dy_t + j * DYB + a_off(0));
#pragma unroll
      for (int i = 0; i < WIB; ++i) bf[0][i] = tr_frag(in_t + i * INB + b_off(0, 0));
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (k + 1 < NK) {
#pragma unroll
          for (int j = 0; j < WCB; ++j) af[(k + 1) & 1][j] = tr_frag(dy_t + j * DYB + a_off(k + 1));
#pragma unroll
          for (int i = 0; i < WIB; ++i) bf[(k + 1) & 1][i] = tr_frag(in_t + i * INB + b_off(k + 1, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < WCB; ++j) {
          if (do_bias) M::mma(accb2[j], af[k & 1][j], ones);
#pragma unroll
          for (int i = 0; i < WIB; ++i) M::mma(acc[j * WIB + i], af[k & 1][j], bf[k & 1][i]);
        }
      }
    } else {
      constexpr int NSEQ = NK * TAPS;
      bf16x8 bq[3];
      bf16x8 aq[2];
      aq[0] = tr_frag(dy_t + a_off(0));
      bq[0] = tr_frag(in_t + b_off(0, 0));
      if (NSEQ > 1) bq[1] = tr_frag(in_t + b_off(1 / TAPS, 1 % TAPS));
#pragma unroll
      for (int i = 0; i < NSEQ; ++i) {
        const int k = i / TAPS, t = i % TAPS;
        if (i + 2 < NSEQ) bq[(i + 2) % 3] = tr_frag(in_t + b_off((i + 2) / TAPS, (i + 2) % TAPS));
        if (t == 0 && k + 1 < NK) aq[(k + 1) & 1] = tr_frag(dy_t + a_off(k + 1));
        __builtin_amdgcn_sched_barrier(0);
        if (do_bias && t == 0) M::mma(accb, aq[k & 1], ones);
        M::mma(acc[t], aq[k & 1], bq[i % 3]);
      }
    }
    if (wave == 0) W2TRACE(1, s);
    __syncthreads();
    if (wave == 0) W2TRACE(2, s);
    cur ^= 1;
  }
  if (wave == 0) W2TRACE(3, 0);

  // D[row = co][col = ci]
  const int r = lane & 31, hh = lane >> 5;
  const int slab = blockIdx.x * PS + psx;
  float* const out = p.dwp + (p.part_mode ? (size_t)slab * p.slab_elems : 0);
  const int cr = (p.dy_mode == OCT_IN_S2D) ? (p.cout >> 2) : p.cout;
  if constexpr (MULTI) {
#pragma unroll
    for (int j = 0; j < WCB; ++j) {
#pragma unroll
      for (int ii = 0; ii < WIB; ++ii) {
        const int ci = ci_sb + (ib + ii) * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co_sb + (cb + j) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          float* const q = &out[(size_t)co * p.ktot + ci];
          if (p.part_mode) *q = acc[j * WIB + ii][i]; else atomicAdd(q, acc[j * WIB + ii][i]);
        }
      }
      if (do_bias && r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co_sb + (cb + j) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (p.part_mode) p.dbias_part[(size_t)slab * p.cout + co] = accb2[j][i];
          else atomicAdd(&p.dbias[co % cr], accb2[j][i]);
        }
      }
    }
  } else {
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    if (RSH && t >= p.nstore) continue;   // the tap rows beyond the kernel (last launch of a 7x3: one row of three)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // W16: register 4S + e, S = 2a + b -> (co 16a + 4*(lane>>4) + e, ci 16b + (lane&15))
      const int co = co_sb + cb * 32 + (W16 ? 16 * (i >> 3) + 4 * g4 + (i & 3) : (i & 3) + 8 * (i >> 2) + 4 * hh);
      const int ci = ci_sb + ib * 32 + (W16 ? 16 * ((i >> 2) & 1) + li : r);
      float* const q = &out[((size_t)t * p.cout + co) * p.ktot + ci];
      if (p.part_mode) *q = acc[t][i]; else atomicAdd(q, acc[t][i]);
    }
  }
  if (do_bias && (W16 ? li == 0 : r == 0)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (W16 && ((i >> 2) & 1)) continue;   // the bias sums sit in the b = 0 quarters
      const int co = co_sb + cb * 32 + (W16 ? 16 * (i >> 3) + 4 * g4 + (i & 3) : (i & 3) + 8 * (i >> 2) + 4 * hh);
      if (p.part_mode) p.dbias_part[(size_t)slab * p.cout + co] = accb[i];
      else atomicAdd(&p.dbias[co % cr], accb[i]);
    }
  }
  }
#ifdef OCT_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave == 0) W2TRACE(3, 1);
#endif
}

static bool w2_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_DISABLE_V2"); on = (e && e[0] == '1') ? 0 : 1; }
  return on == 1;
}

template <int TAPS, int CB, int IB, int TH, bool RAGGED, bool D3 = false, bool RSH = false>
static void launch_w2r(Wgrad2Params& p, int nco, int nci, hipStream_t s) {
  constexpr int halo = TAPS != 1 ? 1 : 0, halo_y = TAPS == 9 ? 1 : 0;
  constexpr int stage = IB * (TH + 2 * halo_y) * (32 + 2 * halo) * 64 + CB * TH * 32 * 64;
  constexpr int lds = 2 * stage + 2 * 32 * IB * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2_kernel<TAPS, CB, IB, TH, RAGGED, D3, RSH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  p.tiles_x = (p.w + 31) / 32; p.tiles_y = (p.h + TH - 1) / TH; p.ntiles = p.tiles_x * p.tiles_y * p.n;
  const int gy = nco / CB, gz = nci / IB;
  int gx = (2 * stage * 2 <= 160 * 1024 ? 512 : 256) / (gy * gz);  // 2 workgroups per CU when LDS allows
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  p.per_wg = (p.ntiles + gx - 1) / gx;
  p.interleave = p.ntiles >= 2 * gx ? 1 : 0;
  if (!p.interleave) gx = (p.ntiles + p.per_wg - 1) / p.per_wg;
  if (p.part_mode < 0) { p.part_mode = gx * (CB * IB > 4 ? 1 : 4 / (CB * IB)); return; }   // query: slabs this launch would write
  hipLaunchKernelGGL((wgrad2_kernel<TAPS, CB, IB, TH, RAGGED, D3, RSH>), dim3(gx, gy, gz), dim3(512), lds, s, p);
}
template <int TAPS, int CB, int IB, int TH>
static void launch_w2(Wgrad2Params& p, int nco, int nci, hipStream_t s) {
  // whole tiles take the instantiation without the dY validity mask (no register cost on the bench shapes)
  const bool whole = (p.w % 32) == 0 && (p.h % TH) == 0;
  if (p.depth > 0 || p.dy_mul) launch_w2r<TAPS, CB, IB, TH, false, true>(p, nco, nci, s);   // volumetric: whole tiles (checked by the caller)
  else if (whole) launch_w2r<TAPS, CB, IB, TH, false>(p, nco, nci, s);
  else launch_w2r<TAPS, CB, IB, TH, true>(p, nco, nci, s);
}

// returns 1 when taken, 0 when the shape is not eligible, <0 on error; query != nullptr: no launch, *query = number
// of partial slabs the launch would write in partials mode
int oct_conv_wgrad_v2(const OctWgradDesc* d, const OctWgradArgs* a, void* stream, int* query) {
  if (!w2_enabled()) return 0;
  if (d->kh == 7) {
    // 7x3 (ReLayNet): 64 x 64 channel blocks, atomics mode only; everything else stays on the generic kernel
    if (d->taps != 21 || d->kw != 3 || d->depth > 0 || d->dy_img_mul != 0 || d->dy_mode != OCT_IN_PLAIN || d->partials ||
        d->dtype != OCT_DT_BF16 || (d->c0 % 32) != 0 || (d->c1 % 32) != 0 || (d->cout % 64) != 0 || ((d->c0 + d->c1) % 64) != 0)
      return 0;
  }
  if ((d->depth > 0 || d->dy_img_mul != 0) && ((d->w % 32) != 0 || (d->h % 16) != 0)) return 0;   // volumetric: whole tiles
  const int ktot = d->c0 + d->c1;
  // plain 3x3 / 1x1: any H, W (ragged last tiles are predicated); the deconv mode needs whole tiles
  const bool whole = (d->w % 32) == 0 && (d->h % 8) == 0;
  const bool ok = d->dtype == OCT_DT_BF16 && (whole || d->dy_mode == OCT_IN_PLAIN) && (d->c0 % 32) == 0 &&
                  (d->c1 % 32) == 0 && (d->cout % 32) == 0 &&
                  (d->dy_mode == OCT_IN_PLAIN || ((d->cout >> 2) % 32) == 0);
  if (!ok) return 0;
  Wgrad2Params p;
  static const OctWgradArgs no_args = {};
  if (query) a = &no_args;
  p.x0 = (const bf16_t*)a->x0; p.x1 = (const bf16_t*)a->x1;
  p.sc0 = a->scale0; p.sh0 = a->shift0; p.sc1 = a->scale1; p.sh1 = a->shift1;
  p.dy = (const bf16_t*)a->dy; p.dwp = a->dwp; p.dbias = a->dbias;
#ifdef OCT_TRACE
  p.trace = g_trace_w2;
#else
  p.trace = nullptr;
#endif
  p.part_mode = query ? -1 : (d->partials ? 1 : 0);
  p.slab_elems = (size_t)d->taps * d->cout * ktot; p.dbias_part = a->dbias_partials;
  if (p.part_mode == 1 && a->dbias && !a->dbias_partials) { oct_set_error("oct_conv_wgrad: partials mode with a bias gradient needs dbias_partials"); return OCT_E_INVALID; }
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.c1 = d->c1; p.ktot = ktot; p.cout = d->cout;
  p.xf0 = d->xform0; p.xf1 = d->xform1; p.dy_mode = d->dy_mode;
  p.depth = d->depth; p.img_shift = d->in_img_shift; p.dy_mul = d->dy_img_mul; p.dy_add = d->dy_img_add;
  const int nco = d->cout / 32, nci = ktot / 32;
  const bool big = (nco % 2 == 0) && (nci % 2 == 0);
  hipStream_t s = as_stream(stream);
  p.ty0 = 0; p.pad_y = 1; p.nstore = 9;
  if (d->kh == 7) {
    // three launches: tap rows 0-2 and 3-5 on the nine-tap kernel, row 6 on its one-row (three-tap) form (the first version ran
    // row 6 on the nine-tap kernel too and dropped two rows of accumulators: 27 taps of MFMA work for 21); the bias gradient
    // rides on the first
    const bool whole = (p.w % 32) == 0 && (p.h % 8) == 0;
    float* const dwp0 = p.dwp;
    for (int ty0 = 0; ty0 < 7; ty0 += 3) {
      p.ty0 = ty0; p.pad_y = 3; p.nstore = ty0 == 6 ? 3 : 9;
      p.dwp = dwp0 + (size_t)ty0 * 3 * d->cout * ktot;
      if (ty0 > 0) p.dbias = nullptr;
      if (ty0 == 6) {
        if (whole) launch_w2r<3, 2, 2, 8, false, false, true>(p, nco, nci, s);
        else launch_w2r<3, 2, 2, 8, true, false, true>(p, nco, nci, s);
      } else if (whole) launch_w2r<9, 2, 2, 8, false, false, true>(p, nco, nci, s);
      else launch_w2r<9, 2, 2, 8, true, false, true>(p, nco, nci, s);
      if (query) break;
    }
  } else if (d->taps == 9) {
    // 64 x 64 channel blocks: 8-row tiles (halo overhead 10/8 instead of 6/4 on the staged input, half the
    // barriers; the two 76-KB stage buffers fill the LDS and the producer ring drops to 2 stages): -5 %
    // 32 x 32 channel blocks (the 32-channel full-resolution layers): 16-row tiles, halo overhead 18/16 -- -6 %
    // 32 x 64 / 64 x 32 channel blocks (an odd block count on one side: Cout = 32 with Cin = 64, Cin = 32 with Cout = 64 --
    // dec1 conv1, enc2 conv1): two pairs per workgroup share the staged tile of the 32-channel side, which 1 x 1 blocks
    // read twice (the 64 -> 32 layer at 512 x 1024: 4.3 GB per launch instead of 3.2)
    static int pairs2 = -1;
    if (pairs2 < 0) { const char* e = getenv("OCT_W2_PAIRS2"); pairs2 = (e && e[0] == '0') ? 0 : 1; }
    if (big) launch_w2<9, 2, 2, 8>(p, nco, nci, s);
    else if (pairs2 && nci % 2 == 0 && (d->h % 8) == 0 && d->dy_img_mul == 0) launch_w2<9, 1, 2, 8>(p, nco, nci, s);
    else if (pairs2 && nco % 2 == 0 && (d->h % 8) == 0 && d->dy_img_mul == 0) launch_w2<9, 2, 1, 8>(p, nco, nci, s);
    else launch_w2<9, 1, 1, 16>(p, nco, nci, s);
  } else {
    static int multi = -1;
    if (multi < 0) { const char* e = getenv("OCT_W2_MULTI"); multi = (e && e[0] == '0') ? 0 : 1; }
    const bool m_ok = multi && nco % 4 == 0 && (d->h % 4) == 0 && d->depth == 0 && d->dy_img_mul == 0;
    if (m_ok && nci % 4 == 0) launch_w2<1, 4, 4, 4>(p, nco, nci, s);        // 128 x 128 channel blocks, 4-row tiles
    else if (m_ok && nci % 2 == 0) launch_w2<1, 4, 2, 4>(p, nco, nci, s);   // 128 x 64 (upconv1: Cin = 64)
    else if (big) launch_w2<1, 2, 2, 8>(p, nco, nci, s);
    else launch_w2<1, 1, 1, 8>(p, nco, nci, s);   // 8 rows: -14 % vs 4
  }
  if (query) { *query = p.part_mode; return 1; }
  int rc = oct_check_launch("wgrad2");
  return rc ? rc : 1;
}
