// Depth-rolling 3x3x3 convolution (volumetric U-Net, first level: Cin = 32 per depth tap) -- fprop and dgrad.
//
// igemm2's volumetric mode stages one (tile, depth tap) per stage: every input slice tile is fetched, transformed and written
// to LDS THREE times (as slice d - 1, d and d + 1 of three different items): 241 FLOP per staged byte at Cin = Cout = 32,
// and the first-level launches of cfg5 sat at 5-6 ms for a 2 ms matrix floor with the four producer waves as the limiter
// (profiles/r02_cfg5_v3_kernel_table.txt, r03_cfg5_kernel_table.txt).  Here a workgroup walks a COLUMN -- one 8 x 32 tile
// position through all slices of a volume, d = 0 .. depth - 1 in order -- with a ring of FOUR slice tiles in LDS: item d
// multiplies slices d - 1, d, d + 1 (27 taps x 2 k16 steps = 54 MFMA steps per output fragment, one barrier) while the
// producers stage slice d + 2 into the fourth buffer.  One tile staged per item instead of three: 723 FLOP per staged byte.
//   stage j of a column (j = 0 .. depth + 1): producers deliver slice z = j - 1 (all zero for z = -1 and z = depth: the depth
//   padding) into ring slot g & 3 (g = running stage index of the workgroup); the MFMA waves compute item d = j - 2 from the
//   slots of stages g - 2, g - 1, g.  Two bubbles per column (j = 0, 1) in depth + 2 stages: 3 % at depth 64.
// Everything else is igemm2's one-fragment (NF = 1, v_mfma_f32_32x32x16_bf16) design: four producer waves (global -> registers
// two stages ahead -> on-load transform -> LDS, 80-byte pixel pitch), four MFMA waves, weights streamed from L2 through a
// nine-step register ring (the sequence repeats every item, so the ring simply cycles), deferred epilogue (the previous item's
// fragments are transposed through LDS and stored between the MFMA steps of the next one), BatchNorm sums in registers.
// No reference counterpart (the volumetric network is BASELINE configs[4]); bit-exact against torch's float64 Conv3d on
// exactly representable operands (tests/test_gpu_exact.py).
#include "common.h"
#include <stdlib.h>

struct RollParams {
  const bf16_t* x; const float* sc; const float* sh; const bf16_t* wp;
  bf16_t* y0; bf16_t* y1; float* stats;
  int nvol, depth, h, w, cout, split, xf;
  int tiles_x, tiles_y, ncols;
};

typedef unsigned int r3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int r3_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned r3_pack(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned, v);
}

template <int WM, int WN, int MF, bool STATS>
__global__ void __launch_bounds__(512) roll3d_kernel(const RollParams p) {
  static_assert(WM * WN == 4 && WM * MF == 8, "four MFMA waves, 8-row tiles");
  constexpr int TH = 8, TW = 32, LH = TH + 2, LW = TW + 2, NPIX = LH * LW;
  constexpr int NSLOT = (NPIX + 63) / 64, PIXB = 80, BUFB = NPIX * PIXB, NRING = 4;
  constexpr int NT = WN * 32;
  constexpr int KSTEPS = 54;   // (depth tap, tap, k16): s = kd * 18 + tap * 2 + k16
  typedef Mma<bf16_t> M;
  typedef M::Frag Frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const buf0 = smem;
  float* const sxf = reinterpret_cast<float*>(smem + NRING * BUFB);            // [2][32] scale, shift
  unsigned char* const oscr = smem + NRING * BUFB + 64 * 4;                       // 4 waves x 32 px x 80 B
  float* const wg_stats = reinterpret_cast<float*>(oscr + 4 * 32 * 80);          // [WM][2][NT]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncols_wg = (p.ncols - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // columns b, b + grid, ...
  if (ncols_wg <= 0) return;
  const int spc = p.depth + 2;                       // stages per column
  const int nstage = ncols_wg * spc;
  const int nstage_pad = (nstage + 1) / 2 * 2;

  if (tid < 32) {
    sxf[tid] = p.xf ? p.sc[tid] : 1.f;
    sxf[32 + tid] = p.xf ? p.sh[tid] : 0.f;
  }
  __syncthreads();

  if (wave >= 4) {
    // =============================== producer waves ===============================
    const int ptid = tid - 256, grp = ptid & 3, pbase = ptid >> 2;
    constexpr int D = 2;
    r3_u32x4 R[D][NSLOT];
    unsigned vmask[D];
    int relp[NSLOT];
    unsigned code[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int pix = pbase + 64 * i;
      const int ly = pix / LW, lx = pix - ly * LW;
      relp[i] = ly * p.w + lx;   // relative to the halo corner
      code[i] = (pix >= NPIX ? 16u : 0u) | (ly == 0 ? 1u : 0u) | (ly == LH - 1 ? 2u : 0u) | (lx == 0 ? 4u : 0u) | (lx == LW - 1 ? 8u : 0u);
    }
    float s[8], b[8];
    {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sxf + grp * 8), s1 = *reinterpret_cast<const f32x4*>(sxf + grp * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sxf + 32 + grp * 8), b1 = *reinterpret_cast<const f32x4*>(sxf + 32 + grp * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] = s0[j]; s[4 + j] = s1[j]; b[j] = b0[j]; b[4 + j] = b1[j]; }
    }
    const float flo = xf_floor(p.xf);
    const unsigned cs2 = 64u;   // 32 channels x 2 B
    const unsigned safe = (unsigned)(p.w + 1) * cs2;
    // stage counters of the issue side: column (txi, tyi, vol) and j
    const int last = nstage - 1;
    int i_sidx = 0, i_j = 0, i_col = blockIdx.x;
    int i_txi, i_tyi, i_vol;
    auto decode = [&](int col, int& txi, int& tyi, int& vol) {
      txi = col % p.tiles_x; const int t = col / p.tiles_x;
      tyi = t % p.tiles_y; vol = t / p.tiles_y;
    };
    decode(i_col, i_txi, i_tyi, i_vol);
    auto issue = [&](r3_u32x4 (&Rr)[NSLOT], unsigned& vm) {
      const int j = i_j, txi = i_txi, tyi = i_tyi, vol = i_vol;
      if (i_sidx < last) {
        ++i_sidx;
        if (++i_j == spc) { i_j = 0; i_col += gridDim.x; decode(i_col, i_txi, i_tyi, i_vol); }
      }
      const int z = j - 1;
      const bool zok = z >= 0 && z < p.depth;
      const int img = vol * p.depth + (zok ? z : 0);   // a slice outside the volume is padding: its loads re-read slice 0, every slot dead
      const unsigned edge = 16u | (tyi == 0 ? 1u : 0u) | (tyi == p.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) |
                            (txi == p.tiles_x - 1 ? 8u : 0u);
      const size_t origin = ((size_t)img * p.h + tyi * TH) * p.w + txi * TW;
      const unsigned char* const hb = reinterpret_cast<const unsigned char*>(p.x + origin * 32 + grp * 8) - safe;
      vm = 0;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        const bool ok = zok && (code[i] & edge) == 0;
        Rr[i] = *reinterpret_cast<const r3_u32x4*>(hb + (ok ? __umul24((unsigned)relp[i], cs2) : safe));
        vm |= ok ? (1u << i) : 0u;
      }
    };
    auto commit = [&](unsigned char* buf, const r3_u32x4 (&Rr)[NSLOT], unsigned vm) {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        const int pix = pbase + 64 * i;
        if (pix < NPIX) {
          r3_u32x4 v = Rr[i];
          if (p.xf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = fmaxf(fmaf(__uint_as_float(v[j] << 16), s[2 * j], b[2 * j]), flo);
              const float hi = fmaxf(fmaf(__uint_as_float(v[j] & 0xffff0000u), s[2 * j + 1], b[2 * j + 1]), flo);
              v[j] = r3_pack(lo, hi);
            }
          }
          const bool live = (vm & (1u << i)) != 0;   // padding is exactly zero (it applies to the activated tensor)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = live ? v[j] : 0u;
          *reinterpret_cast<r3_u32x4*>(buf + pix * PIXB + grp * 16) = v;
        }
      }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) issue(R[j], vmask[j]);
    commit(buf0, R[0], vmask[0]);
    issue(R[0], vmask[0]);
    __syncthreads();
    // while the MFMA waves work on stage cs, stage cs + 1 goes to ring slot (cs + 1) & 3 and its register slot is refilled
    // with the loads of stage cs + 1 + D (branch-free over the padded stage count, indices clamp to the last stage)
    for (int s0 = 0; s0 < nstage_pad; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int nx = s0 + j + 1;
        commit(buf0 + (nx & 3) * BUFB, R[(j + 1) % D], vmask[(j + 1) % D]);
        issue(R[(j + 1) % D], vmask[(j + 1) % D]);
        __syncthreads();
      }
    }
    if (STATS) __syncthreads();
    return;
  }

  // ================================= MFMA waves =================================
  __builtin_amdgcn_s_setprio(3);
  const int r = lane & 31, hh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  constexpr int PF = 9;
  static_assert(KSTEPS % PF == 0, "the weight ring lines up across items");
  Frag wring[PF];
  f32x16 acc[MF];
  float s1[16], s2[16];   // BatchNorm sums (dead code without STATS)
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  }
  // packed filter: [32-channel block][tap][nk16 = 6][512], k = (kd, ci): the fragment of (kd, tap, k16) is k16 index 2 kd + k16
  // addressed as a wave-uniform 64-bit base (SGPRs) + compile-time byte offset + one 32-bit lane offset: `global_load_dwordx4 v,
  // v_off, s[base]`.  (With a per-lane pointer hipcc hoists the 54 fragment addresses of the unrolled item out of the stage
  // loop -- 108 VGPRs of loop-invariant pointers, 80-316 bytes of scratch; the base is therefore re-laundered every stage.)
  // (an explicit GLOBAL pointer: after the laundering hipcc no longer knows where a generic pointer points, emits flat_load and
  //  then waits with vmcnt(0) lgkmcnt(0) at every use -- six drains of the ring and of all earlier stores per item)
  typedef const __attribute__((address_space(1))) unsigned char* gbytes;
  typedef const __attribute__((address_space(1))) Frag* gfrag;
  gbytes wu = (gbytes)reinterpret_cast<const unsigned char*>(p.wp + ((size_t)wn * 9 * 6) * 512);
  const unsigned wlane = (unsigned)lane * 16u;
  auto wload = [&](int off) -> Frag { return *(gfrag)(wu + off + wlane); };
  auto woff = [](int s) constexpr { const int kd = s / 18, t = (s % 18) >> 1, k16 = s & 1; return (t * 6 + kd * 2 + k16) * 1024; };
#pragma unroll
  for (int j = 0; j < PF; ++j) wring[j] = wload(woff(j));

  // deferred epilogue (igemm2.hip): the item's fragments are only packed to bf16; the LDS transpose and the stores ride
  // between the MFMA steps of the next item
  constexpr int NFR = MF;
  constexpr int ESTRIDE = (KSTEPS - 2) / NFR;
  // The deferred stores are UNCONDITIONAL: a wave-uniform `if (pending)` around them made hipcc lose count of the outstanding
  // memory operations and wait with s_waitcnt vmcnt(0) -- a drain of the weight ring and of every earlier store -- six times
  // per item (4.5 ms per launch for 1.9 ms of matrix work).  So there is always a "previous item": before the first one it is
  // the first item itself with zeros (same wave, same lanes, same addresses: the real values follow one stage later, in
  // program order), and a column's last item simply waits through the two bubble stages of the next column.
  unsigned packed[MF][8];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int q = 0; q < 8; ++q) packed[m][q] = 0u;
  unsigned char* e_fb = nullptr;
  unsigned e_rowb = 0, e_pstep = 0;
  auto set_item = [&](int img, int tyi, int txi) {
    const int cb0 = wn * 32;
    bf16_t* dst; int cd, co;
    if (p.split > 0 && cb0 >= p.split) { dst = p.y1; cd = p.cout - p.split; co = cb0 - p.split; }
    else { dst = p.y0; cd = p.split > 0 ? p.split : p.cout; co = cb0; }
    const size_t pix0 = ((size_t)img * p.h + tyi * TH + wm * MF) * p.w + txi * TW;
    e_fb = reinterpret_cast<unsigned char*>(dst + pix0 * cd + co);
    e_rowb = 2u * (unsigned)p.w * (unsigned)cd;
    e_pstep = 2u * (unsigned)cd;
  };
  auto store_frag = [&](int m) {
    unsigned char* sc = oscr + wave * (32 * 80);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const r3_u32x2 v = {packed[m][2 * g], packed[m][2 * g + 1]};
      *reinterpret_cast<r3_u32x2*>(sc + r * 80 + (8 * g + 4 * hh) * 2) = v;   // pixel r, channels 8g + 4hh ..+3
    }
    r3_u32x4 tv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int chunk = lane + 64 * k;   // 128 chunks of 16 B: pixel = chunk / 4, part = chunk % 4
      tv[k] = *reinterpret_cast<const r3_u32x4*>(sc + (chunk >> 2) * 80 + (chunk & 3) * 16);
    }
    unsigned char* const fb = e_fb + (size_t)m * e_rowb;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int chunk = lane + 64 * k;
      *reinterpret_cast<r3_u32x4*>(fb + (__umul24((unsigned)(chunk >> 2), e_pstep) + (unsigned)(chunk & 3) * 16u)) = tv[k];
    }
  };
  auto flush = [&]() {
#pragma unroll
    for (int idx = 0; idx < NFR; ++idx) store_frag(idx);
  };

  __syncthreads();   // stage 0 is in LDS
  int j = 0, col = blockIdx.x;
  int txi = col % p.tiles_x, tyi = (col / p.tiles_x) % p.tiles_y, vol = col / (p.tiles_x * p.tiles_y);
  set_item(vol * p.depth, tyi, txi);
  for (int g = 0; g < nstage_pad; ++g) {
    if (g >= nstage) { __syncthreads(); continue; }
    if (j >= 2) {   // (j = 0, 1: a column starts, nothing to multiply yet)
      asm volatile("" : "+s"(wu));   // keeps the fragment addresses of this stage out of the loop preheader (see above)
      const int lane_off = ((wm * MF) * LW + r) * PIXB + 16 * hh;
      const unsigned char* lbk[3];
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) lbk[kd] = buf0 + ((g - 2 + kd) & 3) * BUFB + lane_off;
      auto xptr = [&](int s, int m) -> const unsigned char* {
        const int kd = s / 18, t = (s % 18) >> 1, k16 = s & 1;
        return lbk[kd] + ((m + t / 3) * LW + t % 3) * PIXB + k16 * 32;
      };
      constexpr int LD = (MF >= 4) ? 2 : 3;
      Frag xr[LD + 1][MF];
#pragma unroll
      for (int q = 0; q < LD; ++q)
#pragma unroll
        for (int m = 0; m < MF; ++m) xr[q][m] = M::load(xptr(q, m));
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if (s + LD < KSTEPS) {
#pragma unroll
          for (int m = 0; m < MF; ++m) xr[(s + LD) % (LD + 1)][m] = M::load(xptr(s + LD, m));
        }
        __builtin_amdgcn_sched_barrier(0);   // the reads of step s + LD stay ahead of the MFMAs of step s
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          if (s == 0) M::mma0(acc[m], wring[s % PF], xr[s % (LD + 1)][m]);
          else M::mma(acc[m], wring[s % PF], xr[s % (LD + 1)][m]);
        }
        wring[s % PF] = wload(woff((s + PF) % KSTEPS));   // the sequence repeats every item: the ring cycles
        if (s >= 1 && (s - 1) % ESTRIDE == 0 && (s - 1) / ESTRIDE < NFR) store_frag((s - 1) / ESTRIDE);   // previous item
      }
      // ---- this item: pack, BatchNorm sums, hand the stores to the next stage ----
#pragma unroll
      for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          packed[m][2 * q] = r3_pack(acc[m][4 * q], acc[m][4 * q + 1]);
          packed[m][2 * q + 1] = r3_pack(acc[m][4 * q + 2], acc[m][4 * q + 3]);
        }
        if (STATS) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { s1[i] += acc[m][i]; s2[i] = fmaf(acc[m][i], acc[m][i], s2[i]); }
        }
      }
      set_item(vol * p.depth + (j - 2), tyi, txi);
    }
    __syncthreads();
    if (++j == spc) {
      j = 0; col += gridDim.x;
      txi = col % p.tiles_x; tyi = (col / p.tiles_x) % p.tiles_y; vol = col / (p.tiles_x * p.tiles_y);
    }
  }
  flush();

  if constexpr (STATS) {
    const float t1 = reduce32_scatter16(s1, lane);
    const float t2 = reduce32_scatter16(s2, lane);
    if ((lane & 1) == 0) {
      const int reg = scatter16_reg_of_lane(lane);
      const int cl = wn * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
      wg_stats[(wm * 2 + 0) * NT + cl] = t1;
      wg_stats[(wm * 2 + 1) * NT + cl] = t2;
    }
    __syncthreads();   // the producer waves join this barrier before they exit
    for (int i = tid; i < 2 * NT; i += 256) {
      const int st = i / NT, cl = i - st * NT;
      float v = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) v += wg_stats[(w_ * 2 + st) * NT + cl];
      p.stats[((size_t)blockIdx.x * 2 + st) * p.cout + cl] = v;   // one row [2][cout] per workgroup
    }
  }
}

static bool roll_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("OCT_ROLL3D");
    const char* v2 = getenv("OCT_DISABLE_V2");
    on = ((e && e[0] == '0') || (v2 && v2[0] == '1')) ? 0 : 1;
  }
  return on == 1;
}

// Which launches: measured on cfg5 (profiles/r03_cfg5_kernel_table.txt, same box): 32 -> 32 forward with BatchNorm sums 4.9 -> 3.96 ms,
// the 32 -> 64 data gradient 7.6 -> 6.6 ms, but the 32 -> 32 data gradient (no sums) 3.25 -> 3.97 ms -- igemm2's 16-row tile gives
// each weight fragment four MFMAs, the 8-row tile the ring forces (four 16-row slice tiles do not fit the LDS) only two, and at
// Cout = 32 all four MFMA waves stream the SAME 54 KB of filter per item: 62 B/clk per CU of L2 requests beside 128 B/clk of LDS
// reads, both at the CU's limits.  Those stay on igemm2 (OCT_ROLL3D=2 forces every eligible launch here).
static bool roll_ok(const OctConvDesc* d) {
  static int all = -1;
  if (all < 0) { const char* e = getenv("OCT_ROLL3D"); all = (e && e[0] == '2') ? 1 : 0; }
  if (!all && d->cout == 32 && !d->want_stats) return false;
  return roll_enabled() && d->dtype == OCT_DT_BF16 && d->depth > 0 && d->taps == 9 && d->kh != 7 && d->in_mode == OCT_IN_PLAIN &&
         d->out_mode == OCT_OUT_PLAIN && d->c0 == 32 && d->c1 == 0 && (d->cout == 32 || d->cout == 64) &&
         (d->split == 0 || (d->cout == 64 && d->split == 32)) && (d->w % 32) == 0 && (d->h % 8) == 0 && (d->n % d->depth) == 0 &&
         d->out_img_mul == 0 && d->depth >= 2 && (size_t)d->n * d->h * d->w < (1ull << 31);
}
static int roll_grid(const OctConvDesc* d) {
  const int ncols = (d->w / 32) * (d->h / 8) * (d->n / d->depth);
  return ncols < 256 ? ncols : 256;   // one persistent workgroup per CU (the ring fills the LDS)
}

// BatchNorm partial rows the rolling kernel writes for this descriptor, or -1 when it does not take it
int oct_conv_roll3d_stat_rows(const OctConvDesc* d) { return roll_ok(d) ? roll_grid(d) : -1; }

template <int WM, int WN, int MF>
static void launch_roll(const RollParams& p, int grid, bool stats, hipStream_t s) {
  constexpr int lds = 4 * (10 * 34 * 80) + 64 * 4 + 4 * 32 * 80 + WM * 2 * (WN * 32) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roll3d_kernel<WM, WN, MF, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&roll3d_kernel<WM, WN, MF, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  if (stats) hipLaunchKernelGGL((roll3d_kernel<WM, WN, MF, true>), dim3(grid), dim3(512), lds, s, p);
  else hipLaunchKernelGGL((roll3d_kernel<WM, WN, MF, false>), dim3(grid), dim3(512), lds, s, p);
}

// returns 1 when the launch was taken, 0 when the shape is not eligible, <0 on error
int oct_conv_forward_roll3d(const OctConvDesc* d, const OctConvArgs* a, void* stream) {
  if (!roll_ok(d)) return 0;
  if (d->xform0 && (!a->scale0 || !a->shift0)) return 0;
  RollParams p;
  p.x = (const bf16_t*)a->x0; p.sc = a->scale0; p.sh = a->shift0; p.wp = (const bf16_t*)a->wpacked;
  p.y0 = (bf16_t*)a->y0; p.y1 = (bf16_t*)a->y1; p.stats = d->want_stats ? a->stat_partials : nullptr;
  p.nvol = d->n / d->depth; p.depth = d->depth; p.h = d->h; p.w = d->w; p.cout = d->cout; p.split = d->split; p.xf = d->xform0;
  p.tiles_x = d->w / 32; p.tiles_y = d->h / 8; p.ncols = p.tiles_x * p.tiles_y * p.nvol;
  const int grid = roll_grid(d);
  hipStream_t s = as_stream(stream);
  if (d->cout == 32) launch_roll<4, 1, 2>(p, grid, p.stats != nullptr, s);
  else launch_roll<2, 2, 4>(p, grid, p.stats != nullptr, s);
  const int rc = oct_check_launch("roll3d");
  return rc ? rc : 1;
}
