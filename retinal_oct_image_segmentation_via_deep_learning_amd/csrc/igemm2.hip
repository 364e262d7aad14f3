// Pipelined, persistent implicit-GEMM 3x3 convolution for bf16 with every channel count a multiple
// of 32 -- the production path of conv3x3 fprop and dgrad.  Any H, W: the last tile row / column
// of a size that is not a multiple of the 8(16) x 32 tile is predicated (loads zeroed, stores and
// BatchNorm sums masked).  Other channel counts and the fp32 parity mode stay on igemm.hip.
//
// Differences to the generic kernel, all driven by the rocprof numbers of round 1:
//  * persistent workgroups walk a contiguous range of (tile, channel-block) items, so the halo
//    rows shared by vertically adjacent tiles are re-read from the same XCD's L2;
//  * the LDS halo tile is double buffered and the global loads of stage s+1 are issued BEFORE the
//    MFMA phase of stage s (raw bf16 parked in registers, BN+ReLU applied when they are written to
//    LDS after the phase): one barrier per stage, HBM latency hidden under the MFMAs;
//  * WRES: for Cout <= 64 with Cin = 32 the whole filter (18 fragments = 72 VGPRs) stays resident
//    in registers for the life of the workgroup -- no weight traffic at all in the full-resolution
//    layers, which are HBM bound; otherwise weight fragments stream from L2 one step ahead;
//  * epilogue: bf16 pack + v_permlane32_swap so every lane stores 16 contiguous bytes;
//  * BatchNorm partial sums stay in registers across all tiles of the workgroup (WRES) and are
//    reduced across lanes once.
#include "common.h"
#include <stdlib.h>

struct Igemm2Params {
  const bf16_t* x0; const bf16_t* x1;
  const float* sc0; const float* sh0; const float* sc1; const float* sh1;
  const bf16_t* wp;
  bf16_t* y0; bf16_t* y1; float* stats; const float* bias;
  int n, h, w, c0, c1, cout, split, xf0, xf1, in_mode, out_mode;
  int tiles_x, tiles_y, nblk, nitems, per_wg, nch, nk16;
  int interleave;   // 1: workgroup b walks tiles b, b + grid, b + 2*grid, ... (all channel blocks of a tile), 0: a contiguous item range
  // 3-D (see OctConvDesc.depth): nchc = 32-channel chunks per depth tap; chunk ch = kd*nchc + c reads slice d + kd - 1.
  // S2D with depth: k = (kd, dy, dx, c) gathers from slice 2d + kd.  D2S: output image = img*oimg_mul + oimg_add.
  int depth, nchc, oimg_mul, oimg_add;
  unsigned long long* trace;  // diagnostic builds only (-DOCT_TRACE): s_memtime stamps of workgroup 0
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef OCT_TRACE
#define TRACE(slot, idx)                                                                              \
  do {                                                                                                \
    if (p.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (idx) < 256)                          \
      p.trace[(slot) * 256 + (idx)] = __builtin_amdgcn_s_memtime();                                    \
  } while (0)
static unsigned long long* g_trace = nullptr;
extern "C" void oct_debug_set_trace(void* buf) { g_trace = (unsigned long long*)buf; }
#else
#define TRACE(slot, idx) do {} while (0)
#endif

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// RAGGED: H or W is not a multiple of the tile; whole-tile shapes (the benchmark) run the instantiation
// without any of the predication below (measured: 2-3 % when it was unconditional).
// D3: the depth taps / image maps of the volumetric network (OctConvDesc.depth, out_img_*).  A template switch, not a
// runtime one: the same code guarded by `p.depth > 0` inside the producers' issue path cost the 2-D benchmark 12 % of
// its igemm2 time (measured: 13.7 -> 15.4 ms per step), these kernels being bound by exactly that path.
#ifndef IG2_M16
#define IG2_M16 1
#endif
#ifndef IG2_M16_NF1
#define IG2_M16_NF1 0
#endif
#ifndef IG2_PIXB16
#define IG2_PIXB16 96
#endif
// which instantiations multiply with v_mfma_f32_16x16x32_bf16 (see M16 in the kernel)
template <int TAPS, int NF, bool WRES> constexpr bool ig2_m16() { return IG2_M16 && TAPS != 1 && !WRES && (NF == 2 || IG2_M16_NF1); }
// Pixel pitch of the LDS halo tile (bytes; 32 bf16 = 64 B of payload).  Register staging pads the pixel so that a wave's
// ds_read_b128 of one fragment touches every bank once -- and which pad does that depends on the lane -> pixel map of the
// MFMA shape (bank model of MI355X_MICROARCH.md, LDS: ds_read_b128 is serviced in four 16-lane groups):
//   32x32x16 (lane = pixel l & 31, k half l >> 5):        80 B: 4 LDS cycles per read   (96 B: 8)
//   16x16x32 (lane = pixel l & 15, k quarter l >> 4):     96 B: 4 LDS cycles per read   (80 B: 8 -- the pitch round 2 ran the
//                                                          16x16x32 kernels on: every activation read paid a 2-way conflict)
// DMA tiles are dense (64 B) with the swizzle on the source address.  tools/lds_swizzle_check.py enumerates all of these.
template <int TAPS, int NF, bool WRES, bool DMA> constexpr int ig2_pixb() { return DMA ? 64 : (ig2_m16<TAPS, NF, WRES>() ? IG2_PIXB16 : 80); }

// 16 zero bytes in device memory: the source of every LDS-DMA lane whose pixel is padding (a DMA cannot write a constant)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

// DMA (data gradients: the input dY needs no transform on load): the halo tile goes global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave instruction, no registers, no ds_write).  A DMA writes LDS linearly in lane
// order, so the tile is dense ([pixel][64 B], no pad) and the bank-conflict swizzle moves to the SOURCE address: the 16-B
// chunk stored at position c of pixel (row, col) is channel chunk c ^ key(col), key = (col >> KSH) & 3 with KSH = 2 for
// the 32x32x16 read pattern and 1 for the 16x16x32 one (each conflict-free for all three tap columns; exhaustive check in
// tools/lds_swizzle_check.py).  Three (3x3) or six (1x1) LDS buffers: the DMAs of stage s + NBUF - 1 are issued while stage s
// multiplies, the producers wait with a COUNTED vmcnt for stage s + 1 and the stage barrier is a raw s_barrier.
// WLDS (Cout = 32 with Cin = 64: dec1 conv1 forward, the data gradient of enc2 conv1): the whole filter (36 KB) is copied into
// LDS once per workgroup and the MFMA waves read their weight fragments from there.  Streamed from L2, every one of the four
// waves fetched all 36 fragments per 16 x 32 tile -- 144 KB of weight requests next to 78 KB of activations on the
// full-resolution layer, through the same per-CU load path: it ran at 3.9 TB/s where its 32 -> 32 siblings (resident
// weights in registers) reach 5.4.  Registers cannot hold 36 fragments (144 VGPRs) beside the accumulators.
template <int TAPS, int WM, int WN, int MF, int NF, bool WRES, bool STATS, bool RAGGED = false, bool D3 = false, bool DMA = false,
          bool WLDS = false>
__global__ void __launch_bounds__(512) igemm2_kernel(const Igemm2Params p) {
  static_assert(WM * WN == 4, "four MFMA waves");
  static_assert(!DMA || (!WRES && !STATS && !RAGGED && !D3), "DMA staging: streamed weights, whole tiles, 2-D, no BatchNorm sums");
  static_assert(!WLDS || (TAPS == 9 && !WRES && !D3 && !DMA && WN * NF == 1), "LDS-resident weights: 3x3, one 32-channel block");
  constexpr int TH = WM * MF, TW = 32;
  // TAPS = 9: 3x3; TAPS = 21: 7x3 (ReLayNet_2017.py:155-160, padding (3, 1)): tap = ky * 3 + kx either way; TAPS = 1: 1x1
  constexpr int HALO = (TAPS != 1) ? 1 : 0;              // columns
  constexpr int HALO_Y = (TAPS == 21) ? 3 : HALO;        // rows
  constexpr int LH = TH + 2 * HALO_Y, LW = TW + 2 * HALO;
  constexpr int NPIX = LH * LW;
  constexpr int NSLOT = (NPIX + 63) / 64;       // producers (4 waves): 64 pixels x 4 channel groups per pass
  constexpr int PIXB = ig2_pixb<TAPS, NF, WRES, DMA>();
  constexpr int BUFB = DMA ? NSLOT * 4096 : NPIX * PIXB;
  constexpr int NBUF = DMA ? (TAPS != 1 ? 3 : 6) : 2;
  constexpr int NT = WN * NF * 32;
  constexpr int KSTEPS = TAPS * 2;              // k16 steps per 32-channel chunk
  // M16: the streamed-weight 3x3 kernels multiply with v_mfma_f32_16x16x32_bf16 -- 18 half-steps (tap, 16-pixel half) of
  // MF * 2NF MFMAs per stage instead of 18 k16 steps of MF * NF.  Same FLOPs per cycle, same LDS and weight bytes, same
  // packed-weight layout (a lane gathers its 16 B from the 32x32x16 fragment order); an accumulator acc[m][q] holds four
  // 16x16 blocks, register 4*S + e of quarter S = 2*half + cc = channel 16*cc + 4*(lane>>4) + e at pixel 16*half + (lane&15).
  // Why: these kernels are power-bound and the chip holds a higher clock on this shape -- a timing-only
  // build that issued the same FLOPs as 16x16x32 measured -9 % on fprop / dgrad of the Cout >= 128 layers.
  constexpr bool M16 = ig2_m16<TAPS, NF, WRES>();   // NF == 1: 5-12 % slower with the deferred epilogue riding on the half-steps,
                                                                                    // +0.06 ms per step with the un-deferred one (-DIG2_M16_NF1=1): stays on 32x32x16
  typedef Mma<bf16_t> M;
  typedef M::Frag Frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const buf0 = smem;
  float* const wg_stats = reinterpret_cast<float*>(smem + NBUF * BUFB);  // [2 parity][WM][2][NT]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: everything derived from it stays scalar
  // the 3x3 kernels only run plain -> plain (plan_v2); the deconv modes belong to the 1x1 instantiations
  const bool s2d = TAPS == 1 && p.in_mode == OCT_IN_S2D;
  const bool d2s = TAPS == 1 && p.out_mode == OCT_OUT_D2S;
  // Work of this workgroup: `nitems_wg` items (tile, channel block) starting at (tile t_first, block nbi_first),
  // the tile index advancing by `tstep` whenever the channel blocks of a tile are done.  Large layers interleave
  // the workgroups over the tiles (tstep = grid): at any time the chip works on ~256 consecutive tiles, i.e. on
  // whole image rows -- DRAM pages are swept linearly and the halo rows shared by vertically adjacent tiles are
  // fetched by workgroups running at the same time (Infinity Cache hits instead of a second HBM read).
  int t_first, nbi_first, nitems_wg, tstep;
  if (p.interleave) {
    if ((int)blockIdx.x >= p.nitems / p.nblk) return;
    t_first = blockIdx.x; nbi_first = 0; tstep = gridDim.x;
    nitems_wg = ((p.nitems / p.nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * p.nblk;
  } else {
    const int it0 = blockIdx.x * p.per_wg;
    const int it1 = min(it0 + p.per_wg, p.nitems);
    if (it0 >= it1) return;
    t_first = it0 / p.nblk; nbi_first = it0 - t_first * p.nblk; tstep = 1; nitems_wg = it1 - it0;
  }
  const int sx = tstep % p.tiles_x, sy = (tstep / p.tiles_x) % p.tiles_y, simg = tstep / (p.tiles_x * p.tiles_y);
  const int nstage = nitems_wg * p.nch;
  const int nstage_pad = (nstage + 3) / 4 * 4;   // a multiple of the producer ring depth (D = 2; 4 keeps the unrolled bodies' parity)

  // BN scale/shift of every input channel live in LDS: reading them with ds_read keeps them off the
  // vmcnt queue (a global load issued at commit time would be YOUNGER than the prefetched stages and
  // waiting for it would drain the whole ring -- vmcnt retires in order)
  // WLDS trims the tables to what a 3x3 kernel with <= 64 input channels needs, to make room for the filter
  constexpr int SXF_N = WLDS ? 64 : 1024, OSCR_SLOTS = WLDS ? 1 : 2, SBIAS_N = WLDS ? 0 : 1024, WLDS_BYTES = WLDS ? 2 * 18 * 1024 : 0;
  float* const sxf = reinterpret_cast<float*>(smem + NBUF * BUFB) + (2 * WM * 2 * NT + 4);
  unsigned char* const oscr = smem + NBUF * BUFB + (2 * WM * 2 * NT + 4) * 4 + 2 * SXF_N * 4;  // OSCR_SLOTS x 4 waves x 32 px x 80 B
  static_assert((NBUF * BUFB + (2 * WM * 2 * NT + 4) * 4) % 16 == 0, "scratch must stay 16-B aligned");
  unsigned char* const wlds = oscr + OSCR_SLOTS * 4 * 32 * 80;               // [tap][k16][1 KB fragment] (WLDS)
  float* const sbias = reinterpret_cast<float*>(wlds + WLDS_BYTES);          // [cout/4] deconv bias (D2S only)
  // streamed-weight kernels: BatchNorm partial sums of ALL items of this workgroup, [2][cout]; one row
  // per workgroup reaches memory instead of one per tile (bn_finalize then reads <= 512 rows, not 16 k)
  float* const wgacc = sbias + SBIAS_N;
  if (STATS && !WRES) {
    for (int i = tid; i < 2 * p.cout; i += 512) wgacc[i] = 0.f;
  }
  {
    const int kx = s2d ? p.c0 : p.c0 + p.c1;
    for (int i = tid; i < kx; i += 512) {
      const bool first = i < p.c0;
      const bool xf = first ? (p.xf0 != 0) : (p.xf1 != 0);
      sxf[i] = xf ? (first ? p.sc0[i] : p.sc1[i - p.c0]) : 1.f;
      sxf[kx + i] = xf ? (first ? p.sh0[i] : p.sh1[i - p.c0]) : 0.f;
    }
    if (d2s && p.bias)
      for (int i = tid; i < (p.cout >> 2); i += 512) sbias[i] = p.bias[i];
    if constexpr (WLDS) {   // the filter of this 32-channel block, packed fragment order as it lies in memory
      const u32x4* src = reinterpret_cast<const u32x4*>(p.wp);
      u32x4* dst = reinterpret_cast<u32x4*>(wlds);
      for (int i = tid; i < TAPS * p.nk16 * 64; i += 512) dst[i] = src[i];
    }
  }
  __syncthreads();

  if constexpr (DMA) {
   if (wave >= 4) {
    // ====================== producer waves (4), LDS-DMA staging: no registers, no commit ======================
    const int ptid = tid - 256, grp = ptid & 3, pbase = ptid >> 2;
    constexpr int KSH = M16 ? 1 : 2;
    const unsigned cs2 = 2u * (unsigned)p.c0;           // one source, no transform (checked by the host)
    unsigned goff[NSLOT], code[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int pix = pbase + 64 * i;
      const int ly = pix / LW, lx = pix - ly * LW;
      const int relp = s2d ? (2 * ly) * (2 * p.w) + 2 * lx : ly * p.w + lx;   // relative to the halo corner
      goff[i] = (unsigned)relp * cs2 + (unsigned)((grp ^ ((lx >> KSH) & 3)) * 16);
      unsigned c = pix >= NPIX ? 16u : 0u;
      if (HALO) c |= (ly < HALO_Y ? 1u : 0u) | (lx == 0 ? 4u : 0u) | (ly >= LH - HALO_Y ? 2u : 0u) | (lx == LW - 1 ? 8u : 0u);
      code[i] = c;
    }
    const int last = nstage - 1;
    int i_sidx = 0, i_ch = 0, i_nbi, i_txi, i_tyi, i_img;
    {
      int t = t_first;
      i_nbi = nbi_first;
      i_txi = t % p.tiles_x; t /= p.tiles_x;
      i_tyi = t % p.tiles_y; i_img = t / p.tiles_y;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gl_void;
    const unsigned char* const zsrc = reinterpret_cast<const unsigned char*>(g_zero16);
    const int pw = wave - 4;
    auto dma = [&](unsigned char* buf) {   // the next stage (counters as in the register path below) -> buf
      const int ch = i_ch, txi = i_txi, tyi = i_tyi, img = i_img;
      if (i_sidx < last) {
        ++i_sidx;
        if (++i_ch == p.nch) {
          i_ch = 0;
          if (++i_nbi == p.nblk) {
            i_nbi = 0;
            i_txi += sx; if (i_txi >= p.tiles_x) { i_txi -= p.tiles_x; ++i_tyi; }
            i_tyi += sy; if (i_tyi >= p.tiles_y) { i_tyi -= p.tiles_y; ++i_img; }
            i_img += simg;
          }
        }
      }
      const unsigned edge = 16u | (tyi == 0 ? 1u : 0u) | (tyi == p.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) |
                            (txi == p.tiles_x - 1 ? 8u : 0u);
      const bf16_t* base;
      if (s2d) {  // k = (dy*2+dx)*C + c of the 2H x 2W tensor
        const int dydx = (ch * 32) / p.c0;
        const int cc = ch * 32 - dydx * p.c0;
        const size_t o2 = ((size_t)img * (2 * p.h) + 2 * tyi * TH + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * txi * TW + (dydx & 1);
        base = p.x0 + o2 * p.c0 + cc;
      } else {
        base = p.x0 + (((size_t)img * p.h + tyi * TH) * p.w + txi * TW) * p.c0 + ch * 32;
      }
      // the tile's halo corner (it may lie outside the tensor: such lanes read the zero block instead)
      const unsigned char* const hb = reinterpret_cast<const unsigned char*>(base) - (size_t)(HALO_Y * p.w + HALO) * cs2;
      unsigned char* const dst = buf + pw * 1024;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        const bool ok = (code[i] & edge) == 0;
        const unsigned char* src = ok ? hb + goff[i] : zsrc;
#ifndef IG2_DMA_NT
#define IG2_DMA_NT 0
#endif
        __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(dst + i * 4096), 16, 0, IG2_DMA_NT ? 2 : 0);
#ifndef IG2_DMA_SLEEP
#define IG2_DMA_SLEEP 6
#endif
        // the producers have nothing else to do: pace the tile's requests over the stage instead of sending them as one burst
        // in front of the MFMA waves' weight loads (same reasoning as SPREAD below); s_sleep counts 64-cycle units
        // (the one-fragment kernels' stages are 2.3 k cycles: a third of the pause; same box, r3: -5...-7 % on the Cout >= 128
        //  launches with 6, +3 % on the 64-channel ones with 6, neutral with 2)
        if (TAPS != 1 && IG2_DMA_SLEEP > 0 && i + 1 < NSLOT) __builtin_amdgcn_s_sleep(NF == 2 ? IG2_DMA_SLEEP : IG2_DMA_SLEEP / 3);
      }
    };
    // all but the youngest NBUF - 2 stages have landed (vmcnt = simm16[15:14 | 3:0]); then the raw barrier
    constexpr int NW = NSLOT * (NBUF - 2);
    static_assert(NW < 64, "vmcnt is a 6-bit counter");
    auto stage_barrier = [&]() {
      __builtin_amdgcn_s_waitcnt((NW & 15) | ((NW >> 4) << 14) | (7 << 4) | (15 << 8));
      asm volatile("s_barrier" ::: "memory");
    };
    int wb = 0;   // buffer of the next stage to fetch
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) { dma(buf0 + wb * BUFB); wb = wb + 1 == NBUF ? 0 : wb + 1; }
    stage_barrier();   // stage 0 is in LDS
    for (int s0 = 0; s0 < nstage_pad; ++s0) {   // while stage s0 multiplies: fetch stage s0 + NBUF - 1 into the buffer stage s0 - 1 left
      dma(buf0 + wb * BUFB); wb = wb + 1 == NBUF ? 0 : wb + 1;
      stage_barrier();
    }
    return;
   }
  } else
  if (wave >= 4) {
    // =============================== producer waves (4) ===============================
    // global -> registers (issued two stages ahead) -> BN+ReLU -> LDS halo tile of the next stage
    // (no raised priority for the producers: the MFMA waves interleave their epilogue with the matrix steps)
    const int ptid = tid - 256, grp = ptid & 3, pbase = ptid >> 2;
    constexpr int D = 2;            // stages of global loads in flight per producer thread (2 stages = 6 us of lead; 4 measured 1-5 % slower on the full-resolution layers)
    u32x4 R[D][NSLOT];
    unsigned vmask[D];
    // per-slot constants: pixel offset relative to the tile origin and a border code
    // (bit0 top halo row, bit1 bottom, bit2 left, bit3 right, bit4 slot beyond the tile)
    int relp[NSLOT];
    unsigned code[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int pix = pbase + 64 * i;
      const int ly = pix / LW, lx = pix - ly * LW;
      relp[i] = s2d ? (2 * ly) * (2 * p.w) + 2 * lx : ly * p.w + lx;   // relative to the halo corner: never negative
      // bottom / right flags are set against the LAST tile row / column of the image: for whole tiles that is
      // the halo row / column (zero padding), for a ragged size also every local row / column beyond H, W.
      // They only take effect on tiles of that last row / column (edge bits 1 and 3 below).
      const int ylast = p.h - (p.tiles_y - 1) * TH + HALO_Y, xlast = p.w - (p.tiles_x - 1) * TW + HALO;
      unsigned c = pix >= NPIX ? 16u : 0u;
      if (HALO) c |= (ly < HALO_Y ? 1u : 0u) | (lx == 0 ? 4u : 0u);
      c |= (ly >= ylast ? 2u : 0u) | (lx >= xlast ? 8u : 0u);
      code[i] = c;
    }
    // Stages are issued and committed strictly in order (0, 1, 2, ... clamped at the last one), so the
    // (chunk, tile x, tile y, image) of the next stage are counters: the five integer divisions they replace
    // were ~150 scalar instructions per stage on the producer's critical path.
    const int last = nstage - 1;
    int i_sidx = 0, i_ch = 0, i_nbi, i_txi, i_tyi, i_img, c_sidx = 0, c_ch = 0;
    {
      int t = t_first;
      i_nbi = nbi_first;
      i_txi = t % p.tiles_x; t /= p.tiles_x;
      i_tyi = t % p.tiles_y; i_img = t / p.tiles_y;
    }
    struct IssueSt { const unsigned char* hb; unsigned edge, cs2, safe; bool zok; };
    auto issue_begin = [&]() -> IssueSt {
      const int ch = i_ch, txi = i_txi, tyi = i_tyi, img = i_img;
      if (i_sidx < last) {
        ++i_sidx;
        if (++i_ch == p.nch) {
          i_ch = 0;
          if (++i_nbi == p.nblk) {
            i_nbi = 0;
            i_txi += sx; if (i_txi >= p.tiles_x) { i_txi -= p.tiles_x; ++i_tyi; }
            i_tyi += sy; if (i_tyi >= p.tiles_y) { i_tyi -= p.tiles_y; ++i_img; }
            i_img += simg;
          }
        }
      }
      const unsigned edge = 16u | (tyi == 0 ? 1u : 0u) | (tyi == p.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) |
                            (txi == p.tiles_x - 1 ? 8u : 0u);
      // depth taps (3-D): chunk ch = kd * nchc + chc reads slice d + kd - 1 of the same volume; a slice outside the
      // volume is all padding -- its loads re-read the tile's own slice (valid memory) and every slot is dead
      int chc = ch, srcimg = img;
      bool zok = true;
      if (D3 && !s2d) {
        const int kdi = ch / p.nchc;
        chc = ch - kdi * p.nchc;
        const int dz = img % p.depth + kdi - 1;
        zok = dz >= 0 && dz < p.depth;
        srcimg = zok ? img + kdi - 1 : img;
      }
      const size_t origin = ((size_t)srcimg * p.h + tyi * TH) * p.w + txi * TW;
      const bool second = !s2d && chc * 32 >= p.c0;  // uniform: a 32-channel chunk lies in one source
      const int cs = second ? p.c1 : p.c0;
      const bf16_t* base;
      if (s2d) {  // k = (dy*2+dx)*C + c of the 2H x 2W tensor (3-D: k = (kd,dy,dx,c), slice 2d + kd)
        int dydx = (ch * 32) / p.c0;
        const int cc = ch * 32 - dydx * p.c0;
        int img2 = img;
        if (D3) { img2 = 2 * img + (dydx >> 2); dydx &= 3; }
        const size_t o2 = ((size_t)img2 * (2 * p.h) + 2 * tyi * TH + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * txi * TW + (dydx & 1);
        base = p.x0 + o2 * p.c0 + cc + grp * 8;
      } else {
        base = (second ? p.x1 + origin * p.c1 + (chc * 32 - p.c0) : p.x0 + origin * p.c0 + chc * 32) + grp * 8;
      }
      // Loads are UNCONDITIONAL (out-of-image / dead slots re-read the tile origin and are zeroed at
      // commit): a load under a divergent branch makes hipcc fall back to s_waitcnt vmcnt(0) at every
      // use, which drains the whole multi-stage prefetch ring.
      // scalar 64-bit base (the tile's halo corner; it may lie outside the tensor, dead slots read the tile
      // origin instead) + unsigned 32-bit lane offset: one global_load with an SGPR base per slot
      const unsigned cs2 = 2u * (unsigned)cs, safe = (unsigned)(HALO_Y * p.w + HALO) * cs2;
      const unsigned char* const hb = reinterpret_cast<const unsigned char*>(base) - safe;
      return IssueSt{hb, edge, cs2, safe, zok};
    };
    auto issue_slot = [&](const IssueSt& st, int i, u32x4& r, unsigned& vm) {
      const bool ok = (!D3 || st.zok) && (code[i] & st.edge) == 0;
#ifndef IG2_PNT
#define IG2_PNT 0
#endif
      if (IG2_PNT) r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(st.hb + (ok ? __umul24((unsigned)relp[i], st.cs2) : st.safe)));
      else r = *reinterpret_cast<const u32x4*>(st.hb + (ok ? __umul24((unsigned)relp[i], st.cs2) : st.safe));
      vm |= ok ? (1u << i) : 0u;
    };
    auto issue = [&](u32x4 (&Rr)[NSLOT], unsigned& vm) {
      const IssueSt st = issue_begin();
      vm = 0;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) issue_slot(st, i, Rr[i], vm);
    };
    struct CommitSt { float s[8], b[8]; float flo; unsigned flo_pk; bool xf; };
    auto commit_begin = [&](bool always) -> CommitSt {
      CommitSt st;
      const int ch = c_ch;
      if (c_sidx < last) { ++c_sidx; if (++c_ch == p.nch) c_ch = 0; }
      const int chc = (D3 && !s2d) ? ch % p.nchc : ch;   // channel chunk inside its depth tap
      int cg = chc * 32 + grp * 8;
      if (s2d) cg -= ((ch * 32) / p.c0) * p.c0;
      // wave-uniform on purpose (a 32-channel chunk lies in one source): a per-lane select between the
      // two kernel arguments would become a VECTOR load + s_waitcnt vmcnt(0) in the middle of the ring
      const bool first = s2d || chc * 32 < p.c0;
      st.xf = first ? (p.xf0 != 0) : (p.xf1 != 0);
      st.flo = xf_floor(first ? p.xf0 : p.xf1);   // wave-uniform: 0 (BN + ReLU) or -inf (plain affine)
      st.flo_pk = xf_floor_pk(first ? p.xf0 : p.xf1);
      if (st.xf || always) {
        const int kx = s2d ? p.c0 : p.c0 + p.c1;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sxf + cg), s1 = *reinterpret_cast<const f32x4*>(sxf + cg + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sxf + kx + cg), b1 = *reinterpret_cast<const f32x4*>(sxf + kx + cg + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { st.s[j] = s0[j]; st.s[4 + j] = s1[j]; st.b[j] = b0[j]; st.b[4 + j] = b1[j]; }
      }
      return st;
    };
    // always: the caller knows every source is transformed (branch-free form, used by the interleaved commit / issue loop)
    auto commit_slot = [&](const CommitSt& st, unsigned char* buf, int i, const u32x4& r, unsigned vm, bool always) {
      const int pix = pbase + 64 * i;
      if (pix < NPIX) {
        u32x4 v = r;
#ifndef ABL_NO_XFORM
        if (always || st.xf)
#else
        if (st.xf && p.n == 12345)
#endif
        {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (OCT_PK_RELU) {
              v[j] = pk_clamp_bf16(pack_bf16x2(fmaf(bf16lo(v[j]), st.s[2 * j], st.b[2 * j]), fmaf(bf16hi(v[j]), st.s[2 * j + 1], st.b[2 * j + 1])), st.flo_pk);
            } else {
            const float lo = fmaxf(fmaf(bf16lo(v[j]), st.s[2 * j], st.b[2 * j]), st.flo);
            const float hi = fmaxf(fmaf(bf16hi(v[j]), st.s[2 * j + 1], st.b[2 * j + 1]), st.flo);
            v[j] = pack_bf16x2(lo, hi);
            }
          }
        }
        // out-of-image pixels are exactly zero (padding applies to the activated tensor)
        const bool live = (vm & (1u << i)) != 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = live ? v[j] : 0u;
        *reinterpret_cast<u32x4*>(buf + pix * PIXB + grp * 16) = v;
      }
    };
    auto commit = [&](unsigned char* buf, const u32x4 (&Rr)[NSLOT], unsigned vm) {
      const CommitSt st = commit_begin(false);
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) commit_slot(st, buf, i, Rr[i], vm, false);
    };
#pragma unroll
    for (int j = 0; j < D; ++j) issue(R[j], vmask[j]);   // stages past the end re-read the last one
    commit(buf0, R[0], vmask[0]);
    issue(R[0], vmask[0]);
    __syncthreads();
    // Stage k lives in ring slot k % D; while the MFMA waves work on stage cs, stage cs+1 is written to
    // the other LDS buffer and its slot is refilled with the loads of stage cs+1+D.  The body is
    // branch-free (the stage count is padded to a multiple of D, both roles run the padded count,
    // indices clamp to the last stage): with branches around the loads hipcc protects the ring
    // registers with s_waitcnt vmcnt(0) and the prefetch collapses.
#ifndef IG2_PSPREAD
#define IG2_PSPREAD 1
#endif
    // SPREAD (3x3 kernels whose sources are all transformed on load): slot i of the next stage is committed and IMMEDIATELY
    // re-issued for the stage after next, so the stage's global loads leave one by one over the ~3.7 k cycles of commit work
    // instead of as one burst behind it.  Why: the MFMA waves stream their weights from L2 through the same per-CU
    // load path, and while a burst of HBM-missing tile loads sat in it the weight loads' latency exceeded the ring's lead
    // (profiles/r03_ig2_traces.txt: stages that coincide with a new tile's loads ran 30-60 % longer).
    // (not the resident-weight kernels: their MFMA waves load nothing, and an HBM-bound kernel wants its requests out early)
    const bool spread = IG2_PSPREAD && TAPS != 1 && !D3 && !WRES && p.xf0 != 0 && (p.c1 == 0 || p.xf1 != 0);
    if (spread) {
      for (int s0 = 0; s0 < nstage_pad; s0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int nx = s0 + j + 1;
          if (wave == 4) TRACE(4, nx - 1);
          unsigned char* const buf = buf0 + (nx & 1) * BUFB;
          const CommitSt cst = commit_begin(true);
          const IssueSt ist = issue_begin();
          const unsigned vold = vmask[(j + 1) % D];
          unsigned vnew = 0;
#pragma unroll
          for (int i = 0; i < NSLOT; ++i) {
            commit_slot(cst, buf, i, R[(j + 1) % D][i], vold, true);
            __builtin_amdgcn_sched_barrier(0);
            issue_slot(ist, i, R[(j + 1) % D][i], vnew);
            __builtin_amdgcn_sched_barrier(0);
          }
          vmask[(j + 1) % D] = vnew;
          if (wave == 4) TRACE(6, nx - 1);
          __syncthreads();
          if (wave == 4) TRACE(7, nx - 1);
        }
      }
    } else
    for (int s0 = 0; s0 < nstage_pad; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int nx = s0 + j + 1;
        if (wave == 4) TRACE(4, nx - 1);
        commit(buf0 + (nx & 1) * BUFB, R[(j + 1) % D], vmask[(j + 1) % D]);
        if (wave == 4) TRACE(5, nx - 1);
        issue(R[(j + 1) % D], vmask[(j + 1) % D]);
        if (wave == 4) TRACE(6, nx - 1);
        __syncthreads();
        if (wave == 4) TRACE(7, nx - 1);
      }
    }
    if (STATS) __syncthreads();  // matches the barrier of the final statistics reduction
    return;
  }

  // ================================= MFMA waves (4) =================================
  // each SIMD hosts one MFMA wave and one producer wave: the MFMA wave must win issue arbitration
  // against the partner's VALU-dense staging code (static priority, MI355X_MICROARCH.md item 4)
#ifndef IG2_MFMA_PRIO
#define IG2_MFMA_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(IG2_MFMA_PRIO);
  const int r = lane & 31, hh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  Frag wres[WRES ? KSTEPS : 1][NF];
  if (WRES) {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int nb = wn * NF + q;
        const int tap = s >> 1, k16 = s & 1;
        wres[s][q] = M::load(p.wp + ((size_t)(nb * TAPS + tap) * p.nk16 + k16) * 512 + lane * 8);
      }
    // Wait for the filter once and hide its origin from hipcc's waitcnt bookkeeping: otherwise every
    // loop iteration re-waits "vmcnt(17..0)" for these registers, which in steady state means waiting
    // for the previous tile's output STORES (vmcnt retires in order) before each MFMA.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int q = 0; q < NF; ++q) asm volatile("" : "+v"(wres[s][q]));
  }

#ifndef IG2_PF4
#define IG2_PF4 9
#endif
#ifndef IG2_PF8
#define IG2_PF8 3
#endif
  constexpr int PF = (KSTEPS == 2) ? 2 : ((MF * NF >= 8) ? ((KSTEPS % IG2_PF8 == 0) ? IG2_PF8 : 3) : ((MF * NF >= 4) ? (TAPS == 21 ? 7 : IG2_PF4) : 9));   // PF4: 6 until round 3 (42 steps: 7)
  static_assert(KSTEPS % PF == 0, "ring slots must line up across stages");
  Frag wring[(WRES || M16) ? 1 : PF][NF];
  // M16: weights of two taps.  Tap t sits in slot t & 1 and tap t + 1 is fetched while it multiplies; a stage has nine taps,
  // so the next stage's tap 0 lands in slot 1 and is moved to slot 0 when that stage starts (16 v_mov per stage; a third
  // slot instead cost 16 more registers and spilled the 128-channel kernels)
#ifndef IG2_WRING3
#define IG2_WRING3 1
#endif
  // WR3: three weight slots, tap t in slot t % 3, tap t + 2 fetched while tap t multiplies (two taps = ~1000 matrix cycles
  // of lead; nine taps per stage, so the slots line up across stages and nothing has to be moved).  With two slots the
  // lead was ONE tap (~500 cycles), less than an L2 hit takes while the producers' loads of a new tile miss to HBM: the
  // stages that coincide with those loads ran 30-60 % longer (in-kernel timeline, profiles/r03_ig2_traces.txt: MFMA phase
  // 5.5 k cycles on even stages, 7.3-9.3 k on odd ones).
#ifndef IG2_WR3_RAGGED
#define IG2_WR3_RAGGED 0
#endif
  constexpr bool WR3 = IG2_WRING3 && M16 && (!RAGGED || IG2_WR3_RAGGED);   // (the ragged instantiations spill 6-10 dwords with the third slot)
#ifndef IG2_WRING4
#define IG2_WRING4 0
#endif
  // WR4: FOUR weight slots (tap t + 3 fetched while tap t multiplies: three taps = ~1.5 k matrix cycles of lead).  Nine taps do not
  // line up with four slots: the next stage's taps 0-2 land in slots 1-3 and are moved down by one when that stage starts
  // (48 v_mov per stage).  The 16 registers come from the activation fragments: a ring of six instead of two sets of four.
  constexpr bool WR4 = IG2_WRING4 && WR3 && MF == 4;
  constexpr int NWS = WR4 ? 4 : (WR3 ? 3 : 2);
  Frag a16[M16 ? NWS : 1][M16 ? 2 * NF : 1];
  f32x16 acc[MF][NF];
  float s1[STATS ? NF : 1][16], s2[STATS ? NF : 1][16];  // BN partial sums (lane = pixel column)
  if (STATS) {
#pragma unroll
    for (int q = 0; q < NF; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) { s1[q][i] = 0.f; s2[q][i] = 0.f; }
  }

#ifdef OCT_TRACE
  if (p.trace && blockIdx.x == 0 && wave == 0) {
    // calibration: 64 back-to-back MFMAs on register operands, shader clock vs 100 MHz real time
    f32x16 cacc[2];
    for (int i = 0; i < 16; ++i) { cacc[0][i] = 0.f; cacc[1][i] = 0.f; }
    Frag ca = M::load(p.wp + lane * 8), cb = M::load(p.wp + 512 + lane * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) M::mma(cacc[i & 1], ca, cb);
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float sink = 0.f;
    for (int i = 0; i < 16; ++i) sink += cacc[0][i] + cacc[1][i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { p.trace[2040] = t1 - t0; p.trace[2041] = r1 - r0; p.trace[2042] = (unsigned long long)(sink != 12345.f); p.trace[2043] = r0; }
  }
#endif
  // ---- deferred epilogue -------------------------------------------------------------------------
  // At the end of an item the accumulators are only packed to bf16 (32 registers per 4 fragments);
  // the LDS transpose + global stores of those fragments are interleaved with the MFMA steps of the
  // NEXT stage (a fragment every ESTRIDE steps), where they ride in the issue slots the matrix pipe
  // leaves free.  Done back to back they cost 3.5-5.8k cycles per item with the pipe idle.
  constexpr bool DEFER = !WRES && NF == 1 && !M16;   // register budget: 8 VGPRs per deferred fragment (M16: the MFMAs leave half the issue slots of the 32x32x16 form)
  constexpr int NFR = MF * NF;
  constexpr int ESTRIDE = (KSTEPS - 2) / NFR > 0 ? (KSTEPS - 2) / NFR : 1;
  constexpr int EHANDLED = ((KSTEPS - 1 + ESTRIDE - 1) / ESTRIDE) < NFR ? ((KSTEPS - 1 + ESTRIDE - 1) / ESTRIDE) : NFR;
  unsigned packed[DEFER ? MF : 1][DEFER ? NF : 1][8];
  bool pend = false;
  // Output addressing of the item being stored, all wave-uniform (SGPRs): the 64-bit base of the wave's first
  // fragment row per channel fragment q, the byte step between fragment rows and between adjacent pixels.
  // Set once per item by set_item(); a fragment then costs one 64-bit add instead of a full NHWC index.
  int e_tyi = 0, e_txi = 0;
  unsigned char* e_fb[NF];
  unsigned e_rowb[NF], e_pstep[NF];
#pragma unroll
  for (int q = 0; q < NF; ++q) { e_fb[q] = nullptr; e_rowb[q] = 0; e_pstep[q] = 0; }
  auto set_item = [&](int img, int tyi, int txi, int nbi) {
    e_tyi = tyi; e_txi = txi;
    const int oy0 = tyi * TH + wm * MF;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int cb0 = (nbi * (NT / 32) + wn * NF + q) * 32;
      bf16_t* dst; int cd, co, dydx = 0;
      if (d2s) { cd = p.cout >> 2; dydx = cb0 / cd; co = cb0 - dydx * cd; dst = p.y0; }
      else if (p.split > 0 && cb0 >= p.split) { dst = p.y1; cd = p.cout - p.split; co = cb0 - p.split; }
      else { dst = p.y0; cd = p.split > 0 ? p.split : p.cout; co = cb0; }
      const int oimg = (D3 && d2s) ? img * p.oimg_mul + p.oimg_add : img;
      const size_t pix0 = d2s ? ((size_t)oimg * (2 * p.h) + 2 * oy0 + (dydx >> 1)) * (size_t)(2 * p.w) + 2 * (txi * TW) + (dydx & 1)
                              : ((size_t)img * p.h + oy0) * p.w + txi * TW;
      e_fb[q] = reinterpret_cast<unsigned char*>(dst + pix0 * cd + co);
      e_rowb[q] = (d2s ? 8u : 2u) * (unsigned)p.w * (unsigned)cd;   // one output row down (two for the deconv scatter)
      e_pstep[q] = (d2s ? 4u : 2u) * (unsigned)cd;                  // bytes between horizontally adjacent output pixels
    }
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  // A fragment leaves in three moves: (1) bf16 pairs into the wave-private LDS scratch, pixel-major, so that
  // (2) consecutive lanes read back consecutive 16-B chunks (whole 64-B channel rows per pixel) and (3) store them.
  auto frag_to_lds = [&](const unsigned (&pk)[8], int slot = 0) {
    unsigned char* sc = oscr + (slot * 4 + wave) * (32 * 80);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 v = {pk[2 * g], pk[2 * g + 1]};
      if (M16)   // quarter g = 2*half + cc: pixel 16*half + (lane&15), channels 16*cc + 4*(lane>>4) ..+3
        *reinterpret_cast<u32x2*>(sc + (16 * (g >> 1) + (lane & 15)) * 80 + (16 * (g & 1) + 4 * (lane >> 4)) * 2) = v;
      else
        *reinterpret_cast<u32x2*>(sc + r * 80 + (8 * g + 4 * hh) * 2) = v;   // pixel r, channels 8g+4hh..+3
    }
  };
  auto frag_from_lds = [&](u32x4 (&tv)[2], int slot = 0) {
    const unsigned char* sc = oscr + (slot * 4 + wave) * (32 * 80);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int chunk = lane + 64 * k;   // 128 chunks of 16 B: pixel = chunk / 4, part = chunk % 4
      tv[k] = *reinterpret_cast<const u32x4*>(sc + (chunk >> 2) * 80 + (chunk & 3) * 16);
    }
  };
  auto frag_store = [&](int m, int q, const u32x4 (&tv)[2]) {
    if (RAGGED && e_tyi * TH + wm * MF + m >= p.h) return;   // ragged last tile row (wave-uniform)
    const int wlim = (RAGGED && (e_txi + 1) * TW > p.w) ? p.w - e_txi * TW : TW;
    unsigned char* const fb = e_fb[q] + (size_t)m * e_rowb[q];
    const unsigned pstep = e_pstep[q];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int chunk = lane + 64 * k;
      const int px = chunk >> 2, part = chunk & 3;
      if (RAGGED && px >= wlim) continue;   // ragged last tile column
#ifndef ABL_NO_STORE
      *reinterpret_cast<u32x4*>(fb + (__umul24((unsigned)px, pstep) + (unsigned)part * 16u)) = tv[k];
#else   // diagnostic build: same instruction stream, the store never executes
      if (tv[k][0] == 0x12345678u && tv[k][1] == 0x9abcdef0u) *reinterpret_cast<u32x4*>(fb + (__umul24((unsigned)px, pstep) + (unsigned)part * 16u)) = tv[k];
#endif
    }
  };
#ifndef IG2_PERM_EPI
#define IG2_PERM_EPI 0   /* measured (r3, same box, two runs): +1...+6 % on every 16x16x32 launch -- 32-B store segments instead of 64-B ones */
#endif
  // 16x16x32 accumulators leave WITHOUT the LDS transpose: lane (r16, kg) holds, per channel half cc, channels 16cc + 4kg ..+3
  // of pixel r16 (quarter half 0) and of pixel 16 + r16 (half 1) -- v_permlane16_swap trades the half-1 piece of the even kg
  // rows for the half-0 piece of the odd ones, after which a lane owns 8 consecutive channels (16 B) of ONE pixel:
  // pixel 16*(kg&1) + r16, channels 16cc + 8*(kg>>1).  Two 16-B stores per fragment, every pixel's 32 B written by two
  // neighbouring lanes; no ds_write / ds_read / lgkmcnt round trips (the un-deferred eight-fragment epilogue through LDS was
  // 4.7-5.0 k cycles per item with the matrix pipe idle: profiles/r03_ig2_traces.txt).
  constexpr bool PERM_EPI = IG2_PERM_EPI && M16 && !DEFER;
  auto frag_store_perm = [&](int m, int q, const unsigned (&pk)[8]) {
    if (RAGGED && e_tyi * TH + wm * MF + m >= p.h) return;   // ragged last tile row (wave-uniform)
    const int wlim = (RAGGED && (e_txi + 1) * TW > p.w) ? p.w - e_txi * TW : TW;
    const int kg = lane >> 4, r16 = lane & 15;
    const int px = 16 * (kg & 1) + r16;
    unsigned char* const fb = e_fb[q] + (size_t)m * e_rowb[q];
    const unsigned loff = __umul24((unsigned)px, e_pstep[q]) + (unsigned)(kg >> 1) * 16u;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      // quarters S = cc (pixel half 0) and 2 + cc (pixel half 1): two dwords each
      unsigned a0 = pk[2 * cc], a1 = pk[2 * cc + 1], b0 = pk[2 * (2 + cc)], b1 = pk[2 * (2 + cc) + 1];
      auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
      if (RAGGED && px >= wlim) continue;   // ragged last tile column
      *reinterpret_cast<u32x4*>(fb + (loff + (unsigned)cc * 32u)) = v;
    }
  };
  auto store_frag = [&](int m, int q) {   // fragment (m, q) of the item recorded by set_item, back to back
    u32x4 tv[2];
    frag_to_lds(packed[DEFER ? m : 0][DEFER ? q : 0]);
    frag_from_lds(tv);
    frag_store(m, q, tv);
  };

#ifndef IG2_EARLYB
#define IG2_EARLYB 0   /* measured (r3, same box, two runs each): 11.36 / 11.36 ms without, 11.42 / 11.34 ms with, over the 3x3 launches */
#endif
  // EARLYB (16x16x32 kernels, off): the activation fragments of a stage's FIRST half-step are read right behind the barrier that
  // publishes the stage, before the ~240 instructions of per-stage set-up (statistics flush, accumulator clear, weight
  // addresses) that precede them in program order.  The idea: the first MFMA of a stage waits an LDS round trip behind that
  // set-up.  The in-kernel timeline (profiles/r03_ig2_timeline_late.txt) shows the stage IS bound by the MFMA waves (they wait
  // ~100 cycles at the barrier, the producers ~2.9 k), but moving the reads changed no launch: the set-up is long enough to
  // cover the round trip either way.
  constexpr bool EARLYB = IG2_EARLYB && M16 && !(IG2_WRING4 && MF == 4 && !RAGGED);
  Frag b16e[EARLYB ? MF : 1];
  auto early_b = [&](int cur_) {
    if constexpr (EARLYB) {
      const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        if constexpr (DMA) b16e[m] = M::load(smem + (unsigned)(cur_ * BUFB + ((wm * MF + m) * LW + r16) * 64 + ((kg ^ ((r16 >> 1) & 3)) * 16)));
        else b16e[m] = M::load(buf0 + cur_ * BUFB + ((wm * MF + m) * LW + r16) * PIXB + kg * 16);
      }
    }
  };
  __syncthreads();  // stage 0 is in LDS
  early_b(0);
  int cur = 0, pending_tile = -1, pending_nbi = 0, parity = 0;
  int item = 0, ch = 0;                                        // item = index within this workgroup's items
  int tile_c = t_first, nbi_c = nbi_first;                     // (tile, channel block) of `item`, kept as counters:
                                                             // two integer divisions per stage cost ~200 cycles
  int txi_c = tile_c % p.tiles_x, tyi_c = (tile_c / p.tiles_x) % p.tiles_y, img_c = tile_c / (p.tiles_x * p.tiles_y);
  for (int sidx = 0; sidx < nstage_pad; ++sidx) {
    if (sidx >= nstage) { __syncthreads(); continue; }   // padded stages: keep the barrier count in step
    // flush the statistics of the previous item (written to wg_stats before the last barrier)
    if (STATS && !WRES && pending_tile >= 0) {
      const float* ws = wg_stats + (parity ^ 1) * (WM * 2 * NT);
      for (int i = tid; i < 2 * NT; i += 256) {
        const int st = i / NT, cl = i - st * NT;
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) s += ws[(w_ * 2 + st) * NT + cl];
        wgacc[st * p.cout + pending_nbi * NT + cl] += s;   // thread i owns (st, cl) for every item: no race
      }
      pending_tile = -1;
    }

    const int tile = tile_c, nbi = nbi_c;
    if constexpr (WRES) {
      // ---- resident weights (Cin = 32, one channel fragment per wave): every stage is a whole item ----
      // The MFMAs run fragment-major (all 18 steps of output row m, then row m+1: a single accumulation chain
      // issues at the full matrix rate) and the epilogue of row m -- bf16 pack, LDS transpose, BatchNorm sums,
      // stores: ~35 instructions with two LDS round trips -- is spread over the MFMA steps of row m+1, each
      // wait ~4 steps behind its issue.  Only the last row's epilogue is exposed.  Done back to back after the
      // phase, the four epilogues took 3.0 k cycles per stage next to 2.7 k cycles of MFMAs (timeline, round 1).
      static_assert(!WRES || NF == 1, "resident-weight kernels hold one channel fragment per wave");
      if (wave == 0) TRACE(0, sidx);
      set_item(img_c, tyi_c, txi_c, nbi);
      const bool rag = RAGGED && STATS && ((tyi_c + 1) * TH > p.h || (txi_c + 1) * TW > p.w);
      const bool col_in = !RAGGED || (txi_c * TW + r) < p.w;   // lane = pixel column
      const unsigned char* lb = buf0 + cur * BUFB + ((wm * MF) * LW + r) * PIXB + 8 * hh * 2;
      auto xoff = [](int s, int m) constexpr {
        const int tap = s >> 1, k16 = s & 1;
        return ((m + tap / 3) * LW + tap % 3) * PIXB + k16 * 32;
      };
      unsigned pk[8];
      u32x4 tv[2];
      auto epi = [&](int mp, int s) __attribute__((always_inline)) {   // piece s (1..10) of the epilogue of output row mp
        if (s == 1) {
          if (RAGGED && STATS) {   // ragged last tile: pixels outside the image must not reach the BatchNorm sums
            const bool in = !rag || (col_in && (tyi_c * TH + wm * MF + mp) < p.h);   // branch-free: a branch here
#pragma unroll                                                                        // keeps hipcc from unrolling
            for (int i = 0; i < 16; ++i) acc[mp][0][i] = in ? acc[mp][0][i] : 0.f;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            pk[2 * g] = pack_bf16x2(acc[mp][0][4 * g], acc[mp][0][4 * g + 1]);
            pk[2 * g + 1] = pack_bf16x2(acc[mp][0][4 * g + 2], acc[mp][0][4 * g + 3]);
          }
          frag_to_lds(pk);
        }
        if (STATS && s >= 2 && s < 6) {
#pragma unroll
          for (int i = 4 * (s - 2); i < 4 * (s - 2) + 4; ++i) {
            s1[0][i] += acc[mp][0][i];
            s2[0][i] = fmaf(acc[mp][0][i], acc[mp][0][i], s2[0][i]);
            asm volatile("" : "+v"(s1[0][i]), "+v"(s2[0][i]));   // pin the sums to this step: LLVM otherwise sinks
          }                                                        // all of them behind the stage's barrier
        }
        if (s == 6) frag_from_lds(tv);
        if (s == 10) frag_store(mp, 0, tv);
      };
      constexpr int LD = 2, KT = MF * KSTEPS;   // one ring of LD+1 activation fragments across all rows
      Frag xr[LD + 1];
#pragma unroll
      for (int k = 0; k < LD; ++k) xr[k] = M::load(lb + xoff(k % KSTEPS, k / KSTEPS));
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {   // (nested: hipcc gives up on one 72-trip loop in the ragged instantiations)
        const int k = m * KSTEPS + s;
        if (k + LD < KT) xr[(k + LD) % (LD + 1)] = M::load(lb + xoff((k + LD) % KSTEPS, (k + LD) / KSTEPS));
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of step k+LD ahead of the MFMA of step k
        if (s == 0) M::mma0(acc[m][0], wres[0][0], xr[k % (LD + 1)]);
        else M::mma(acc[m][0], wres[s][0], xr[k % (LD + 1)]);
        if (m > 0) epi(m - 1, s);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (wave == 0) TRACE(1, sidx);
#pragma unroll
      for (int s = 1; s <= 10; ++s) epi(MF - 1, s);
    } else {
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int q = 0; q < NF; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
    }

    if (wave == 0) TRACE(0, sidx);
    // ---- MFMA phase over the staged 32-channel chunk ----
    // Streamed weights ride a ring of PF fragments-steps that stays PF steps (>= ~700 cycles of MFMA
    // work) ahead of the consumer, across stage boundaries, so an L2 round trip never stalls a step.
    {
      const unsigned char* tb = buf0 + cur * BUFB;
      const bf16_t* wbase = nullptr; const bf16_t* wbase_n = nullptr;
      const bf16_t* wuni = nullptr; const bf16_t* wuni_n = nullptr;   // the same without the lane term (wave-uniform)
      const size_t qstride = (size_t)TAPS * p.nk16 * 512, tstride = (size_t)p.nk16 * 512;
      if (!WRES) {
        const int nb0 = nbi * (NT / 32) + wn * NF;
        const bf16_t* const wsrc = WLDS ? reinterpret_cast<const bf16_t*>(wlds) : p.wp;   // WLDS: nb0 = 0 (one 32-channel block)
        wbase = wsrc + ((size_t)nb0 * TAPS * p.nk16 + ch * 2) * 512 + lane * 8;
        if constexpr (M16 && !STATS) wuni = p.wp + ((size_t)nb0 * TAPS * p.nk16 + ch * 2) * 512;
        int n_ch = ch + 1, n_item = item;
        if (n_ch == p.nch) { n_ch = 0; n_item = item + 1; }
        if (n_item >= nitems_wg) { n_item = item; n_ch = ch; }  // last stage: harmless re-read of valid memory
        const int n_nbi = n_item == item ? nbi : (nbi + 1 == p.nblk ? 0 : nbi + 1);
        wbase_n = wsrc + ((size_t)(n_nbi * (NT / 32) + wn * NF) * TAPS * p.nk16 + n_ch * 2) * 512 + lane * 8;
        if constexpr (M16 && !STATS) wuni_n = p.wp + ((size_t)(n_nbi * (NT / 32) + wn * NF) * TAPS * p.nk16 + n_ch * 2) * 512;
        if (sidx == 0 && !M16) {
#pragma unroll
          for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int q = 0; q < NF; ++q) wring[j][q] = M::load(wbase + q * qstride + (j >> 1) * tstride + (j & 1) * 512);
        }
      }
      // activation fragments ping-pong between two register sets: the ds_reads of step s+1 are issued
      // before the MFMAs of step s, one lane base address + compile-time offsets for the whole stage
      const unsigned char* lb = tb + ((wm * MF) * LW + r) * PIXB + 8 * hh * 2;
      auto xoff = [](int s, int m) constexpr {
        const int tap = s >> 1, k16 = s & 1;
        const int ty = (TAPS != 1) ? tap / 3 : 0, tx = (TAPS != 1) ? tap % 3 : 0;
        return ((m + ty) * LW + tx) * PIXB + k16 * 32;
      };
      // DMA tiles (dense, swizzled): the chunk position depends on the tile column r + tx, so each (tx, k16) has its own
      // lane base; the row (m + ty) stays a compile-time offset
      constexpr int NTX = (TAPS != 1) ? 3 : 1;
      unsigned lbs[DMA ? NTX : 1][2];   // byte offsets from smem (32-bit: six 64-bit pointers cost the dgrad kernels their last registers)
      if constexpr (DMA && !M16) {
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx)
#pragma unroll
          for (int k16 = 0; k16 < 2; ++k16) {
            const int col = r + tx;
            lbs[tx][k16] = (unsigned)(cur * BUFB + ((wm * MF) * LW + col) * 64 + (((2 * k16 + hh) ^ ((col >> 2) & 3)) * 16));
          }
      }
      auto xptr = [&](int s, int m) -> const unsigned char* {
        if constexpr (DMA) {
          const int tap = s >> 1, k16 = s & 1;
          const int ty = (TAPS != 1) ? tap / 3 : 0, tx = (TAPS != 1) ? tap % 3 : 0;
          return smem + (lbs[tx][k16] + (unsigned)((m + ty) * LW * 64));
        } else return lb + xoff(s, m);
      };
      if constexpr (M16) {
        const int r16 = lane & 15, kg = lane >> 4;
        const unsigned char* lb16 = tb + ((wm * MF) * LW + r16) * PIXB + kg * 16;
        auto boff = [](int u, int m) constexpr { const int t = u >> 1; return ((m + t / 3) * LW + t % 3 + 16 * (u & 1)) * PIXB; };
        unsigned lbs16[DMA ? 3 : 1];   // DMA tiles: lane byte offset per tap column; key = (col >> 1) & 3 is the same for both pixel
        if constexpr (DMA) {           // halves (col + 16), so the half is a compile-time +16 pixels
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const int col = r16 + tx;
            lbs16[tx] = (unsigned)(cur * BUFB + ((wm * MF) * LW + col) * 64 + ((kg ^ ((col >> 1) & 3)) * 16));
          }
        }
        auto bptr = [&](int u, int m) -> const unsigned char* {
          if constexpr (DMA) { const int t = u >> 1; return smem + (lbs16[t % 3] + (unsigned)(((m + t / 3) * LW + 16 * (u & 1)) * 64)); }
          else return lb16 + boff(u, m);
        };
        // this lane's 16 B of a weight fragment pair: rows 16*cc + r16 of k16 fragment kg>>1, k half kg&1 (32x32x16 order).
        // Addressed as a wave-uniform 64-bit base (stage, channel fragment, tap: SGPRs) + ONE 32-bit lane offset, so that every
        // weight load is `global_load_dwordx4 v, v_off, s[base]`: per-fragment 64-bit VGPR pointers cost 16 registers here and
        // pushed the three-slot ring into scratch.
        const unsigned wlane = (unsigned)(((kg >> 1) * 512 + ((kg & 1) * 32 + r16) * 8) * 2);
        const unsigned char* const wu = reinterpret_cast<const unsigned char*>(wuni);
        const unsigned char* const wun = reinterpret_cast<const unsigned char*>(wuni_n);
        // (register allocation decides which form fits: with BatchNorm sums in the kernel the per-lane pointer form stays
        //  under 240 registers and the scalar-base form spills three dwords; without them it is the other way round)
        constexpr bool WSB = !STATS;
        const int lpart = (kg >> 1) * 512 + ((kg & 1) * 32 + r16) * 8 - lane * 8;   // wbase / wbase_n carry lane*8
        auto aoff = [&](int c) { return (size_t)(c >> 1) * qstride + (c & 1) * 128 + lpart; };
        auto wfrag = [&](bool next, int c, int tt) {
          if constexpr (WSB) {
            const size_t off = ((size_t)(c >> 1) * qstride + (size_t)tt * tstride) * 2 + (c & 1) * 256;
            return M::load((next ? wun : wu) + off + (size_t)wlane);
          } else {
            const bf16_t* wb = next ? wbase_n : wbase;
            return M::load(wb + aoff(c) + tt * tstride);
          }
        };
        if (sidx == 0) {
#pragma unroll
          for (int c = 0; c < 2 * NF; ++c) a16[0][c] = wfrag(false, c, 0);
          if constexpr (WR3) {
#pragma unroll
            for (int c = 0; c < 2 * NF; ++c) a16[1][c] = wfrag(false, c, 1);
          }
          if constexpr (WR4) {
#pragma unroll
            for (int c = 0; c < 2 * NF; ++c) a16[2][c] = wfrag(false, c, 2);
          }
        } else if constexpr (WR4) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2 * NF; ++c) a16[j][c] = a16[j + 1][c];   // taps 0-2 of this stage, fetched under the previous one's taps 6-8
        } else if constexpr (!WR3) {
#pragma unroll
          for (int c = 0; c < 2 * NF; ++c) a16[0][c] = a16[1][c];   // fetched under the previous stage's last tap
        }
        if constexpr (WR4) {
          // activation fragments f = s * MF + m (72 per stage) through a ring of six: the read of fragment f + 5 is issued before
          // the MFMAs of fragment f (320 matrix cycles of lead)
          constexpr int RB = 6, NFRAG = KSTEPS * MF;
          static_assert(NFRAG % RB == 0, "the fragment ring must line up across stages");
          Frag bq[RB];
#pragma unroll
          for (int f = 0; f < RB - 1; ++f) bq[f] = M::load(bptr(f / MF, f % MF));
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) {
            const int t = s >> 1;
#ifndef ABL_NO_WLOAD
            if ((s & 1) == 0) {   // weights of tap t + 3 (the next stage's taps 0-2 under this one's taps 6-8)
              const int t2 = t + 3;
              const int tt = t2 < TAPS ? t2 : t2 - TAPS;
#pragma unroll
              for (int c = 0; c < 2 * NF; ++c) a16[t2 % 4][c] = wfrag(t2 >= TAPS, c, tt);
            }
#endif
#pragma unroll
            for (int m = 0; m < MF; ++m) {
              const int f = s * MF + m;
              if (f + RB - 1 < NFRAG) bq[(f + RB - 1) % RB] = M::load(bptr((f + RB - 1) / MF, (f + RB - 1) % MF));
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int c = 0; c < 2 * NF; ++c) {
                const Frag& wf = a16[t % 4][c];
                if (s & 1) { if (c & 1) M::template mma16<3>(acc[m][c >> 1], wf, bq[f % RB]); else M::template mma16<2>(acc[m][c >> 1], wf, bq[f % RB]); }
                else { if (c & 1) M::template mma16<1>(acc[m][c >> 1], wf, bq[f % RB]); else M::template mma16<0>(acc[m][c >> 1], wf, bq[f % RB]); }
              }
            }
          }
        } else {
        Frag b16[2][MF];
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          if constexpr (EARLYB) b16[0][m] = b16e[m];   // read right behind the barrier
          else b16[0][m] = M::load(bptr(0, m));
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {   // half-step s = 2*tap + pixel half
          const int t = s >> 1;
          if (s + 1 < KSTEPS) {
#pragma unroll
            for (int m = 0; m < MF; ++m) b16[(s + 1) & 1][m] = M::load(bptr(s + 1, m));
          }
#ifndef ABL_NO_WLOAD
          if ((s & 1) == 0) {   // weights of tap t + 1 (WR3: t + 2); the next stage's first tap(s) under this one's last
            const int t2 = t + (WR3 ? 2 : 1);
            const int tt = t2 < TAPS ? t2 : t2 - TAPS;
#pragma unroll
            for (int c = 0; c < 2 * NF; ++c) a16[WR3 ? t2 % 3 : (t2 & 1)][c] = wfrag(t2 >= TAPS, c, tt);
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int c = 0; c < 2 * NF; ++c) {
              const Frag& wf = a16[WR3 ? t % 3 : (t & 1)][c];
              if (s & 1) { if (c & 1) M::template mma16<3>(acc[m][c >> 1], wf, b16[s & 1][m]); else M::template mma16<2>(acc[m][c >> 1], wf, b16[s & 1][m]); }
              else { if (c & 1) M::template mma16<1>(acc[m][c >> 1], wf, b16[s & 1][m]); else M::template mma16<0>(acc[m][c >> 1], wf, b16[s & 1][m]); }
            }
          if (DEFER && s >= 1 && (s - 1) % ESTRIDE == 0 && (s - 1) / ESTRIDE < NFR) {
            if (pend) store_frag(((s - 1) / ESTRIDE) / NF, ((s - 1) / ESTRIDE) % NF);
          }
        }
        if (DEFER && pend) {
#pragma unroll
          for (int idx = EHANDLED; idx < NFR; ++idx) store_frag(idx / NF, idx % NF);
          pend = false;
        }
        }   // !WR4
      } else {
      // ring of LD+1 fragment sets: the reads of step s+LD are in flight while step s multiplies
#ifndef IG2_LD8
#define IG2_LD8 1
#endif
      constexpr int LD = (MF * NF >= 8) ? IG2_LD8 : ((MF * NF >= 4) ? 2 : 3);
      Frag xr[LD + 1][MF];
#pragma unroll
      for (int j = 0; j < LD; ++j)
#pragma unroll
        for (int m = 0; m < MF; ++m) xr[j][m] = M::load(xptr(j, m));
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if (s + LD < KSTEPS) {
#pragma unroll
          for (int m = 0; m < MF; ++m) xr[(s + LD) % (LD + 1)][m] = M::load(xptr(s + LD, m));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of step s+LD ahead of the MFMAs of step s
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int q = 0; q < NF; ++q)
            M::mma(acc[m][q], WRES ? wres[s][q] : wring[s % PF][q], xr[s % (LD + 1)][m]);
#ifndef ABL_NO_WLOAD
        if (!WRES) {
          const int s2 = s + PF;  // refill the slot just consumed
          const bf16_t* wb = s2 < KSTEPS ? wbase : wbase_n;
          const int s3 = s2 < KSTEPS ? s2 : s2 - KSTEPS;
#pragma unroll
          for (int q = 0; q < NF; ++q)
            wring[s % PF][q] = M::load(wb + q * qstride + (s3 >> 1) * tstride + (s3 & 1) * 512);
        }
#endif
        if (DEFER && s >= 1 && (s - 1) % ESTRIDE == 0 && (s - 1) / ESTRIDE < NFR) {
          if (pend) store_frag(((s - 1) / ESTRIDE) / NF, ((s - 1) / ESTRIDE) % NF);
        }
      }
      if (DEFER && pend) {
#pragma unroll
        for (int idx = EHANDLED; idx < NFR; ++idx) store_frag(idx / NF, idx % NF);
        pend = false;
      }
      }   // !M16
    }

    if (wave == 0) TRACE(1, sidx);
    // ---- epilogue of an item: NHWC stores, BN sums ----
    if (ch == p.nch - 1) {
      const int txi = txi_c, tyi = tyi_c, img = img_c;
      if (RAGGED && STATS && ((tyi + 1) * TH > p.h || (txi + 1) * TW > p.w)) {
        // ragged last tile: pixels outside the image must not reach the BatchNorm sums (their stores are
        // skipped anyway).  Zeroed in place, inside this wave-uniform branch: no register cost on whole tiles.
        const bool col_in = (txi * TW + r) < p.w;   // lane = pixel column
        const bool col_in0 = (txi * TW + (lane & 15)) < p.w, col_in1 = (txi * TW + 16 + (lane & 15)) < p.w;   // M16: per pixel half
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const bool rin = (tyi * TH + wm * MF + m) < p.h;
#pragma unroll
          for (int q = 0; q < NF; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const bool in = rin && (M16 ? (i < 8 ? col_in0 : col_in1) : col_in);
              acc[m][q][i] = in ? acc[m][q][i] : 0.f;
            }
        }
      }
      if (!DEFER) set_item(img, tyi, txi, nbi);
      // Not deferred (two channel fragments per wave): fragment f's bf16 pairs go to LDS scratch slot f & 1 while fragment
      // f - 1 is read back from the other slot and stored, and the BatchNorm sums of f fill the LDS round trip -- back to
      // back (write, wait, read, wait, store per fragment) the eight fragments took 6.3 k cycles per item (timeline, round 2).
      auto stats_to_ws = [&](int q) {   // the BatchNorm sums of channel fragment q of this item: registers -> wg_stats[parity]
        float* ws = wg_stats + parity * (WM * 2 * NT);
        {
          if constexpr (M16) {
            // registers 4*(2*half + cc) + e: both pixel halves carry the same channels 16*cc + 4*(lane>>4) + e
            float v1[8], v2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v1[k] = s1[q][k] + s1[q][8 + k]; v2[k] = s2[q][k] + s2[q][8 + k]; }
            const float t1 = reduce16_scatter8(v1, lane);
            const float t2 = reduce16_scatter8(v2, lane);
            if ((lane & 1) == 0) {
              const int reg = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
              const int cl = (wn * NF + q) * 32 + 16 * (reg >> 2) + 4 * (lane >> 4) + (reg & 3);
              ws[(wm * 2 + 0) * NT + cl] = t1;
              ws[(wm * 2 + 1) * NT + cl] = t2;
            }
          } else {
          const float t1 = reduce32_scatter16(s1[q], lane);
          const float t2 = reduce32_scatter16(s2[q], lane);
          if ((lane & 1) == 0) {
            const int reg = scatter16_reg_of_lane(lane);
            const int cl = (wn * NF + q) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            ws[(wm * 2 + 0) * NT + cl] = t1;
            ws[(wm * 2 + 1) * NT + cl] = t2;
          }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) { s1[q][i] = 0.f; s2[q][i] = 0.f; }
        }
      };
#ifndef IG2_MRG
#define IG2_MRG 1
#endif
      // MRG (two channel fragments per wave, 16x16x32): the wave's 64 channels of a pixel are 128 contiguous bytes -- one cache
      // line -- but stored fragment by fragment every store instruction wrote sixteen HALF lines (16 pixels x 64 B).  What a
      // store costs the CU is per line touched, not per byte (a timing build without the stores ran the 3x3 launches 15 % faster,
      // 17-27 % on the data gradients; halving the MFMA waves' store COUNT by handing fragments to other waves changed nothing):
      // both fragments of an output row go through one [32 px][128 B] scratch tile and leave as four instructions of eight FULL
      // lines each.  One scratch slot (LDS executes a wave's accesses in order: row m + 1 is written after row m's reads were
      // issued); the stores of row m are issued behind the packing and sums of row m + 1.
      constexpr bool MRG = IG2_MRG && M16 && !DEFER && NF == 2 && TAPS != 1 && !PERM_EPI && !RAGGED;   // (the ragged instantiations spill 6 dwords with it)
      // (plan_v2 keeps a concat split that falls between the two fragments of a wave -- split % 64 != 0 -- off these tilings)
      constexpr bool mrg_done = MRG;
      if constexpr (MRG) {
        {
          constexpr int P2 = 144;        // scratch pitch: 128 B of channels + 16 B pad
          unsigned char* const sc2 = oscr + wave * (32 * P2);
          static_assert(32 * P2 <= OSCR_SLOTS * 32 * 80, "the merged tile fits the wave's scratch");
          u32x4 tv4[4];
          int pm2 = -1;
          auto store4 = [&](int mm) {
            if (RAGGED && e_tyi * TH + wm * MF + mm >= p.h) return;
            const int wlim = (RAGGED && (e_txi + 1) * TW > p.w) ? p.w - e_txi * TW : TW;
            unsigned char* const fb = e_fb[0] + (size_t)mm * e_rowb[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int chunk = lane + 64 * k, px = chunk >> 3, part = chunk & 7;
              if (RAGGED && px >= wlim) continue;
              *reinterpret_cast<u32x4*>(fb + (__umul24((unsigned)px, e_pstep[0]) + (unsigned)part * 16u)) = tv4[k];
            }
          };
          if (STATS) {   // sums first, one channel fragment at a time, and out of the registers before the stores need them
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < NF; ++q) {
#pragma unroll
              for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                  const f32x2 av = {acc[m][q][i], acc[m][q][i + 1]};
                  f32x2 sv = {s1[q][i], s1[q][i + 1]}, tv = {s2[q][i], s2[q][i + 1]};
                  sv += av;
                  tv = __builtin_elementwise_fma(av, av, tv);
                  s1[q][i] = sv[0]; s1[q][i + 1] = sv[1]; s2[q][i] = tv[0]; s2[q][i + 1] = tv[1];
                }
              if (!WRES) stats_to_ws(q);
            }
          }
#pragma unroll
          for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int q = 0; q < NF; ++q) {
              unsigned pk[8];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                pk[2 * g] = pack_bf16x2(acc[m][q][4 * g], acc[m][q][4 * g + 1]);
                pk[2 * g + 1] = pack_bf16x2(acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]);
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {   // quarter g = 2*half + cc: pixel 16*half + (lane&15), channels 32q + 16cc + 4*(lane>>4) ..+3
                const u32x2 v = {pk[2 * g], pk[2 * g + 1]};
                *reinterpret_cast<u32x2*>(sc2 + (16 * (g >> 1) + (lane & 15)) * P2 + q * 64 + (16 * (g & 1) + 4 * (lane >> 4)) * 2) = v;
              }
            }
            if (pm2 >= 0) store4(pm2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int chunk = lane + 64 * k;   // 256 chunks of 16 B: pixel = chunk / 8, part = chunk % 8
              tv4[k] = *reinterpret_cast<const u32x4*>(sc2 + (chunk >> 3) * P2 + (chunk & 7) * 16);
            }
            pm2 = m;
          }
          store4(pm2);
        }
      }
      u32x4 tvp[2];
      int pm_ = -1, pq_ = 0, fidx = 0;
      if constexpr (!mrg_done)
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        // deconv bias from LDS (staged at kernel start): a global load here would put a vmcnt(0) --
        // i.e. a drain of the weight ring and of all earlier output stores -- into every epilogue
        const bool has_bias = d2s && p.bias != nullptr;   // compile-time false in the 3x3 kernels
        float bv[16];
        if (has_bias) {
          const int cb0 = (nbi * (NT / 32) + wn * NF + q) * 32;
          const int co = cb0 % (p.cout >> 2);   // channel inside the (dy, dx) quarter
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + co + 8 * g + 4 * hh);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[4 * g + j] = b4[j];
          }
        }
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          if (has_bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              packed[DEFER ? m : 0][DEFER ? q : 0][2 * g] = pack_bf16x2(acc[m][q][4 * g] + bv[4 * g], acc[m][q][4 * g + 1] + bv[4 * g + 1]);
              packed[DEFER ? m : 0][DEFER ? q : 0][2 * g + 1] = pack_bf16x2(acc[m][q][4 * g + 2] + bv[4 * g + 2], acc[m][q][4 * g + 3] + bv[4 * g + 3]);
            }
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              packed[DEFER ? m : 0][DEFER ? q : 0][2 * g] = pack_bf16x2(acc[m][q][4 * g], acc[m][q][4 * g + 1]);
              packed[DEFER ? m : 0][DEFER ? q : 0][2 * g + 1] = pack_bf16x2(acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]);
            }
          }
          if (PERM_EPI) frag_store_perm(m, q, packed[0][0]);
          else if (!DEFER) {
            frag_to_lds(packed[0][0], fidx & 1);
            if (pm_ >= 0) { frag_from_lds(tvp, (fidx - 1) & 1); frag_store(pm_, pq_, tvp); }
            pm_ = m; pq_ = q; ++fidx;
          }
          if (STATS) {
#ifndef IG2_PKSTATS
#define IG2_PKSTATS 1
#endif
            if constexpr (IG2_PKSTATS && !DEFER) {
              // un-deferred epilogue (no MFMA of this wave in flight): the 32 sum updates of a fragment as 16 packed-f32
              // instructions -- same IEEE arithmetic, half the vector issue slots on a SIMD that also hosts a producer wave
              typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
              for (int i = 0; i < 16; i += 2) {
                const f32x2 av = {acc[m][q][i], acc[m][q][i + 1]};
                f32x2 sv = {s1[q][i], s1[q][i + 1]}, tv = {s2[q][i], s2[q][i + 1]};
                sv += av;
                tv = __builtin_elementwise_fma(av, av, tv);
                s1[q][i] = sv[0]; s1[q][i + 1] = sv[1]; s2[q][i] = tv[0]; s2[q][i + 1] = tv[1];
              }
            } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              s1[q][i] += acc[m][q][i];
              s2[q][i] = fmaf(acc[m][q][i], acc[m][q][i], s2[q][i]);
            }
            }
          }
        }
      }
      if (!DEFER && !PERM_EPI && !mrg_done) { frag_from_lds(tvp, (fidx - 1) & 1); frag_store(pm_, pq_, tvp); }
      if (DEFER) { set_item(img, tyi, txi, nbi); pend = true; }
      if (STATS && !WRES) {
        if constexpr (!mrg_done) {
#pragma unroll
          for (int q = 0; q < NF; ++q) stats_to_ws(q);
        }
        pending_tile = tile; pending_nbi = nbi; parity ^= 1;
      }
    }

    }
    if (wave == 0) TRACE(2, sidx);
    __syncthreads();
    if (wave == 0) TRACE(3, sidx);
#ifdef OCT_TRACE
    if (p.trace && blockIdx.x == 0 && wave == 0 && lane == 0 && sidx == nstage - 1) p.trace[2044] = __builtin_amdgcn_s_memrealtime();
#endif
    cur = DMA ? (cur + 1 == NBUF ? 0 : cur + 1) : (cur ^ 1);
    early_b(cur);
    if (++ch == p.nch) {
      ch = 0; ++item;
      if (++nbi_c == p.nblk) {
        nbi_c = 0; tile_c += tstep;
        txi_c += sx; if (txi_c >= p.tiles_x) { txi_c -= p.tiles_x; ++tyi_c; }
        tyi_c += sy; if (tyi_c >= p.tiles_y) { tyi_c -= p.tiles_y; ++img_c; }
        img_c += simg;
      }
    }
  }

  if (DEFER && pend) {  // the last item's fragments
#pragma unroll
    for (int idx = 0; idx < NFR; ++idx) store_frag(idx / NF, idx % NF);
  }

  // ---- final statistics ----
  if (STATS) {
    if (WRES) {
      float* ws = wg_stats;
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const float t1 = reduce32_scatter16(s1[q], lane);
        const float t2 = reduce32_scatter16(s2[q], lane);
        if ((lane & 1) == 0) {
          const int reg = scatter16_reg_of_lane(lane);
          const int cl = (wn * NF + q) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
          ws[(wm * 2 + 0) * NT + cl] = t1;
          ws[(wm * 2 + 1) * NT + cl] = t2;
        }
      }
      __syncthreads();  // the producer waves join this barrier before they exit
      for (int i = tid; i < 2 * NT; i += 256) {
        const int st = i / NT, cl = i - st * NT;
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) s += ws[(w_ * 2 + st) * NT + cl];
        if (cl < p.cout) p.stats[((size_t)blockIdx.x * 2 + st) * p.cout + cl] = s;
      }
    } else {
      if (pending_tile >= 0) {
        const float* ws = wg_stats + (parity ^ 1) * (WM * 2 * NT);
        for (int i = tid; i < 2 * NT; i += 256) {
          const int st = i / NT, cl = i - st * NT;
          float s = 0.f;
#pragma unroll
          for (int w_ = 0; w_ < WM; ++w_) s += ws[(w_ * 2 + st) * NT + cl];
          wgacc[st * p.cout + pending_nbi * NT + cl] += s;
        }
      }
      // the four MFMA waves of this workgroup wrote disjoint (st, cl) slots with the same thread each time;
      // make them visible to each other, then one row [2][cout] per workgroup
      __syncthreads();   // the producer waves join this barrier before they exit
      for (int i = tid; i < 2 * p.cout; i += 256) p.stats[(size_t)blockIdx.x * 2 * p.cout + i] = wgacc[i];
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------
struct V2Plan { bool ok; int nt; int th; bool wres; int grid; int per_wg; int nitems; int nblk; int stat_rows; int interleave; };

static bool v2_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_DISABLE_V2"); on = (e && e[0] == '1') ? 0 : 1; }
  return on == 1;
}

static V2Plan plan_v2(const OctConvDesc* d) {
  V2Plan pl = {};
  if (!v2_enabled()) return pl;
  if (d->kh == 7) {
    // 7x3 (ReLayNet): the 64- and 128-channel tilings with three halo rows above and below an 8-row tile.  A last tile row of
    // one or two image rows would put padding into the bottom halo of the tile row ABOVE it, which the border codes do not
    // express (they flag rows against the last tile row only): such heights stay on the generic kernel.
    const int rem = d->h % 8;
    if (d->taps != 21 || d->kw != 3 || d->depth > 0 || d->out_img_mul != 0 || rem == 1 || rem == 2 || (d->cout % 64) != 0) return pl;
  }
  if ((d->depth > 0 || d->out_img_mul != 0) && ((d->w % 32) != 0 || (d->h % 16) != 0)) return pl;   // volumetric: whole tiles
  const int cin = d->in_mode == OCT_IN_S2D ? (d->depth > 0 ? 8 : 4) * d->c0 : d->c0 + d->c1;   // channels per depth tap
  // plain 3x3: any H, W (ragged last tiles are predicated); the deconv modes need whole tiles
  const bool whole = (d->w % 32) == 0 && (d->h % 8) == 0;
  // plain -> plain (3x3, and 1x1: the attention gates' W_g / W_x, classifier heads, their data gradients): any H, W;
  // the deconv modes (1x1 with depth-to-space / space-to-depth addressing) need whole tiles
  const bool plain = d->in_mode == OCT_IN_PLAIN && d->out_mode == OCT_OUT_PLAIN;
  pl.ok = d->dtype == OCT_DT_BF16 && (whole || plain) && (d->c0 % 32) == 0 &&
          (d->c1 % 32) == 0 && (d->cout % 32) == 0 && (d->split % 32) == 0;
  pl.ok = pl.ok && (d->c0 + d->c1) <= 1024 && d->cout <= 4096;   // LDS tables: 2 x 1024 BN coefficients, 1024 bias values
  if (d->taps != 1) pl.ok = pl.ok && plain;
  else pl.ok = pl.ok && (plain ||
               (!d->want_stats && d->split == 0 &&
                ((d->in_mode == OCT_IN_PLAIN && d->out_mode == OCT_OUT_D2S && ((d->cout >> 2) % 32) == 0) ||
                 (d->in_mode == OCT_IN_S2D && d->out_mode == OCT_OUT_PLAIN && d->c1 == 0))));
  if (!pl.ok) return pl;
  pl.nt = d->cout == 32 ? 32 : (d->cout % 128 == 0 ? 128 : (d->cout % 64 == 0 ? 64 : 0));
  if (pl.nt == 128 && d->split > 0 && (d->split % 64) != 0) pl.nt = 64;   // a wave's two channel fragments leave as one 128-B line (MRG): same destination
  if (pl.nt == 0) { pl.ok = false; return pl; }
  pl.wres = (d->taps == 9) && (cin == 32) && (d->cout == pl.nt) && pl.nt <= 64 && d->depth == 0;   // 3-D: three chunks per item
  pl.nblk = d->cout / pl.nt;
  // Cout = 32 (full-resolution, HBM-bound layers): 16-row tiles halve the halo overhead per output pixel
  // 16-row tiles: Cout = 32 always; Cout = 64 with streamed weights in fprop (each wave then owns 4 rows x 64
  // channels: half the LDS reads per MFMA, halo overhead 18/16) -- measured -2 % on fprop, neutral to worse on dgrad
#ifndef IG2_TH16_DGRAD
#define IG2_TH16_DGRAD 1   /* 64-channel data gradients on the 16-row two-fragment tiling too (full-line stores, MRG): 0.358 -> 0.309, 0.334 -> 0.297, 0.155 -> 0.144 ms on the three launches of cfg2 (same box) */
#endif
  // (the two-fragment tilings store a wave's 64 channels as one line: a concat split inside them -- split % 64 != 0 -- keeps the one-fragment tiling)
  pl.th = (d->taps == 9 && (pl.nt == 32 || (pl.nt == 64 && (d->want_stats || IG2_TH16_DGRAD) && !pl.wres && (d->split % 64) == 0)) && (d->h % 16) == 0) ? 16 : 8;
  const int ntiles = ((d->w + 31) / 32) * ((d->h + pl.th - 1) / pl.th) * d->n;
  pl.nitems = ntiles * pl.nblk;
  int target = 256;  // one persistent workgroup per CU (only one fits the LDS); 512 measured 2 % slower on dgrad, 1024 4 %
  if (target > pl.nitems) target = pl.nitems;
  pl.per_wg = (pl.nitems + target - 1) / target;
  pl.grid = (pl.nitems + pl.per_wg - 1) / pl.per_wg;
  pl.interleave = 0;
  if (ntiles >= 2 * target) {   // interleaved tile walk: one workgroup per CU, tiles b, b + 256, ...
    pl.grid = target;
    pl.interleave = 1;
  }
  pl.stat_rows = pl.grid;   // one row [2][cout] per (persistent) workgroup
  return pl;
}

int oct_conv_v2_stat_rows(const OctConvDesc* d) {
  const int roll = oct_conv_roll3d_stat_rows(d);   // first-level volumetric layers: the depth-rolling kernel (roll3d.hip)
  if (roll >= 0) return roll;
  const V2Plan pl = plan_v2(d);
  return pl.ok ? pl.stat_rows : -1;
}

template <int WM, int WN, int MF, int NF, bool WRES, int TAPS = 9>
static void launch_v2(const Igemm2Params& p, int grid, hipStream_t s) {
  constexpr int TH = WM * MF;
  constexpr int LH = TH + (TAPS == 21 ? 6 : 2);   // halo rows: 3 + 3 for the 7x3 kernel
  const int lds = 2 * LH * 34 * ig2_pixb<TAPS, NF, WRES, false>() + (2 * WM * 2 * (WN * NF * 32) + 4 + 2 * 1024) * (int)sizeof(float) + 2 * 4 * 32 * 80 + 1024 * (int)sizeof(float) +
                  (p.stats ? 2 * p.cout * (int)sizeof(float) : 0);
  {
    // > 64 KB of dynamic LDS: opt in once per instantiation (all eight variants of this shape share the size class)
    static bool attr = false;
    if (!attr) {
      const int cap = 160 * 1024;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      if constexpr (!WRES && TAPS == 9) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
      }
      attr = true;
    }
  }
  const bool ragged = (p.w % 32) != 0 || (p.h % TH) != 0;
  if (p.depth > 0) {   // volumetric: whole tiles only (plan_v2), streamed weights
    if constexpr (!WRES && TAPS == 9) {
      if (p.stats) hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, false, true, false, true>), dim3(grid), dim3(512), lds, s, p);
      else hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, false, false, false, true>), dim3(grid), dim3(512), lds, s, p);
    }
  } else if (ragged) {
    if (p.stats) hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, true, true>), dim3(grid), dim3(512), lds, s, p);
    else hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, false, true>), dim3(grid), dim3(512), lds, s, p);
  } else if (p.stats) hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, true>), dim3(grid), dim3(512), lds, s, p);
  else hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, WRES, false>), dim3(grid), dim3(512), lds, s, p);
}
template <int WM, int WN, int MF, int NF>
static void launch_v2_1x1(const Igemm2Params& p, int grid, hipStream_t s) {
  constexpr int TH = WM * MF;
  const int lds = 2 * TH * 32 * 80 + (2 * WM * 2 * (WN * NF * 32) + 4 + 2 * 1024) * (int)sizeof(float) + 2 * 4 * 32 * 80 + 1024 * (int)sizeof(float) +
                  (p.stats ? 2 * p.cout * (int)sizeof(float) : 0);
  const bool ragged = (p.w % 32) != 0 || (p.h % TH) != 0;
  if (p.in_mode == OCT_IN_PLAIN && p.out_mode == OCT_OUT_PLAIN && (p.stats || ragged)) {   // plain 1x1 convolution with BN sums / ragged tiles
    if (ragged) {
      if (p.stats) hipLaunchKernelGGL((igemm2_kernel<1, WM, WN, MF, NF, false, true, true>), dim3(grid), dim3(512), lds, s, p);
      else hipLaunchKernelGGL((igemm2_kernel<1, WM, WN, MF, NF, false, false, true>), dim3(grid), dim3(512), lds, s, p);
    } else hipLaunchKernelGGL((igemm2_kernel<1, WM, WN, MF, NF, false, true>), dim3(grid), dim3(512), lds, s, p);
  } else if (p.depth > 0 || p.oimg_mul)
    hipLaunchKernelGGL((igemm2_kernel<1, WM, WN, MF, NF, false, false, false, true>), dim3(grid), dim3(512), lds, s, p);
  else
    hipLaunchKernelGGL((igemm2_kernel<1, WM, WN, MF, NF, false, false>), dim3(grid), dim3(512), lds, s, p);
}

// LDS-resident weights (see WLDS in the kernel): Cout = 32, Cin <= 64, streamed-weight 3x3 kernels on whole tiles
static bool wlds_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_IG2_WLDS"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
template <int WM, int MF>
static void launch_v2_wlds(const Igemm2Params& p, int grid, hipStream_t s) {
  constexpr int TH = WM * MF;
  const int lds = 2 * (TH + 2) * 34 * 80 + (2 * WM * 2 * 32 + 4) * (int)sizeof(float) + 2 * 64 * (int)sizeof(float) + 4 * 32 * 80 + 2 * 18 * 1024 +
                  (p.stats ? 2 * p.cout * (int)sizeof(float) : 0);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<9, WM, 1, MF, 1, false, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<9, WM, 1, MF, 1, false, false, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (p.stats) hipLaunchKernelGGL((igemm2_kernel<9, WM, 1, MF, 1, false, true, false, false, false, true>), dim3(grid), dim3(512), lds, s, p);
  else hipLaunchKernelGGL((igemm2_kernel<9, WM, 1, MF, 1, false, false, false, false, false, true>), dim3(grid), dim3(512), lds, s, p);
}

// LDS-DMA staging (see the kernel): data gradients -- one source, no transform on load, whole 8-row tiles, streamed weights
static bool dma_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("OCT_IG2_DMA"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
template <int TAPS, int WM, int WN, int MF, int NF>
static void launch_v2_dma(const Igemm2Params& p, int grid, hipStream_t s) {
  constexpr int TH = WM * MF, HALO = TAPS != 1 ? 1 : 0, HALO_Y = TAPS == 21 ? 3 : HALO;
  constexpr int NSLOT = ((TH + 2 * HALO_Y) * (32 + 2 * HALO) + 63) / 64, NBUF = TAPS != 1 ? 3 : 6;
  constexpr int lds = NBUF * NSLOT * 4096 + (2 * WM * 2 * (WN * NF * 32) + 4 + 2 * 1024) * (int)sizeof(float) + 2 * 4 * 32 * 80 + 1024 * (int)sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<TAPS, WM, WN, MF, NF, false, false, false, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((igemm2_kernel<TAPS, WM, WN, MF, NF, false, false, false, false, true>), dim3(grid), dim3(512), lds, s, p);
}

// returns 1 when the launch was taken by this path, 0 when the shape is not eligible, <0 on error
int oct_conv_forward_v2(const OctConvDesc* d, const OctConvArgs* a, void* stream) {
  if (d->depth > 0) {   // 3x3x3 with 32 input channels per depth tap: the depth-rolling walk (roll3d.hip)
    const int rc = oct_conv_forward_roll3d(d, a, stream);
    if (rc != 0) return rc;
  }
  const V2Plan pl = plan_v2(d);
  if (!pl.ok) return 0;
  if (d->taps == 1) {   // transposed convolutions with N % 256 == 0: the eight-wave GEMM kernel (gemm1.hip)
    const int rc = oct_conv_forward_g1(d, a, stream);
    if (rc != 0) return rc;
  }
  Igemm2Params p;
  p.x0 = (const bf16_t*)a->x0; p.x1 = (const bf16_t*)a->x1;
  p.sc0 = a->scale0; p.sh0 = a->shift0; p.sc1 = a->scale1; p.sh1 = a->shift1;
  p.wp = (const bf16_t*)a->wpacked; p.y0 = (bf16_t*)a->y0; p.y1 = (bf16_t*)a->y1;
  p.stats = d->want_stats ? a->stat_partials : nullptr;
  p.bias = a->bias; p.in_mode = d->in_mode; p.out_mode = d->out_mode;
#ifdef OCT_TRACE
  p.trace = g_trace;
#else
  p.trace = nullptr;
#endif
  p.n = d->n; p.h = d->h; p.w = d->w; p.c0 = d->c0; p.c1 = d->c1; p.cout = d->cout; p.split = d->split;
  p.xf0 = d->xform0; p.xf1 = d->xform1;
  p.tiles_x = (d->w + 31) / 32; p.tiles_y = (d->h + pl.th - 1) / pl.th; p.nblk = pl.nblk; p.nitems = pl.nitems; p.per_wg = pl.per_wg;
  p.interleave = pl.interleave;
  const int ktot = d->in_mode == OCT_IN_S2D ? (d->depth > 0 ? 8 : 4) * d->c0 : (d->depth > 0 ? 3 : 1) * (d->c0 + d->c1);
  p.nch = ktot / 32; p.nk16 = ktot / 16;
  p.depth = d->depth; p.nchc = (d->c0 + d->c1) / 32; p.oimg_mul = d->out_img_mul; p.oimg_add = d->out_img_add;
  if (p.depth > 0 && p.oimg_mul == 0) p.oimg_mul = 1;   // the D3 instantiation always applies the image map
  hipStream_t s = as_stream(stream);
  const bool dma = dma_enabled() && !d->xform0 && !d->xform1 && d->c1 == 0 && !d->want_stats && !pl.wres && pl.th == 8 &&
                   d->depth == 0 && d->out_img_mul == 0 && (d->w % 32) == 0 && (d->h % 8) == 0 && pl.nt >= 64 &&
                   ((d->taps == 9) || (d->in_mode == OCT_IN_S2D && d->out_mode == OCT_OUT_PLAIN));   // (7x3: the LDS-DMA variant spills 8 dwords and measured 1 % slower on ReLayNet's data gradients)
  if (dma) {
    if (d->taps == 9) { if (pl.nt == 64) launch_v2_dma<9, 2, 2, 4, 1>(p, pl.grid, s); else launch_v2_dma<9, 2, 2, 4, 2>(p, pl.grid, s); }
    else { if (pl.nt == 64) launch_v2_dma<1, 2, 2, 4, 1>(p, pl.grid, s); else launch_v2_dma<1, 2, 2, 4, 2>(p, pl.grid, s); }
  } else if (d->taps == 21) {
    if (pl.nt == 64) launch_v2<2, 2, 4, 1, false, 21>(p, pl.grid, s); else launch_v2<2, 2, 4, 2, false, 21>(p, pl.grid, s);
  } else if (d->taps == 1) {
    if (pl.nt == 32) launch_v2_1x1<4, 1, 2, 1>(p, pl.grid, s);
    else if (pl.nt == 64) launch_v2_1x1<2, 2, 4, 1>(p, pl.grid, s);
    else launch_v2_1x1<2, 2, 4, 2>(p, pl.grid, s);
  } else if (pl.nt == 32 && pl.th == 16) {
    const bool wlds = wlds_enabled() && !pl.wres && d->cout == 32 && (d->c0 + d->c1) <= 64 && d->depth == 0 && d->out_img_mul == 0 &&
                      (d->w % 32) == 0 && (d->h % 16) == 0;
    if (pl.wres) launch_v2<4, 1, 4, 1, true>(p, pl.grid, s);
    else if (wlds) launch_v2_wlds<4, 4>(p, pl.grid, s);
    else launch_v2<4, 1, 4, 1, false>(p, pl.grid, s);
  } else if (pl.nt == 32) {
    if (pl.wres) launch_v2<4, 1, 2, 1, true>(p, pl.grid, s); else launch_v2<4, 1, 2, 1, false>(p, pl.grid, s);
  } else if (pl.nt == 64 && !pl.wres && pl.th == 16) {
    launch_v2<4, 1, 4, 2, false>(p, pl.grid, s);
  } else if (pl.nt == 64) {
    if (pl.wres) launch_v2<2, 2, 4, 1, true>(p, pl.grid, s); else launch_v2<2, 2, 4, 1, false>(p, pl.grid, s);
  } else {
    launch_v2<2, 2, 4, 2, false>(p, pl.grid, s);
  }
  int rc = oct_check_launch("igemm2");
  return rc ? rc : 1;
}
