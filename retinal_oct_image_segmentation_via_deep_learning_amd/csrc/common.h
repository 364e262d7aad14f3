// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/oct_hip.h"

typedef __bf16 bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// thread-local error string (host side)
void oct_set_error(const char* fmt, ...);
#define OCT_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      oct_set_error(__VA_ARGS__);       \
      return OCT_E_INVALID;             \
    }                                   \
  } while (0)
int oct_check_launch(const char* what);
// igemm2.hip: pipelined bf16 path for regular shapes (returns 1 taken / 0 not eligible / <0 error)
int oct_conv_forward_v2(const OctConvDesc* d, const OctConvArgs* a, void* stream);
int oct_conv_v2_stat_rows(const OctConvDesc* d);
// gemm1.hip: transposed-convolution forward / data gradient as an eight-wave GEMM (1 taken / 0 not eligible / <0 error)
int oct_conv_forward_g1(const OctConvDesc* d, const OctConvArgs* a, void* stream);
int oct_conv_forward_roll3d(const OctConvDesc* d, const OctConvArgs* a, void* stream);   // roll3d.hip: 1 taken, 0 not eligible, < 0 error
int oct_conv_roll3d_stat_rows(const OctConvDesc* d);                                        // BatchNorm partial rows it writes, or -1
// igemm.hip: (kh, kw) of a descriptor (0, 0 -> from taps); false for unsupported sizes
bool oct_conv_kernel_size(int taps, int kh_in, int kw_in, int* kh, int* kw);
int oct_conv_wgrad_v2(const OctWgradDesc* d, const OctWgradArgs* a, void* stream, int* query = nullptr);
// first.hip: direct kernels for Conv2d(1 -> F)
int oct_first_stat_rows(const OctConvDesc* d);
int oct_first_fprop(const OctConvDesc* d, const OctConvArgs* a, void* stream);
int oct_first_wgrad(const OctWgradDesc* d, const OctWgradArgs* a, void* stream, int* query = nullptr);

// lower clamp of the on-load transform a = max(x*scale + shift, floor): 0 for BN + ReLU (OCT_XF_AFFINE_RELU), -inf for the
// plain per-channel affine (OCT_XF_AFFINE: a deferred bias add) -- the same v_max either way
__device__ __forceinline__ float xf_floor(int xf) { return xf == 2 ? -__builtin_inff() : 0.f; }
// The same clamp applied AFTER the pair has been rounded to bf16: one v_pk_max_i16 per pair instead of two v_max_f32 (the
// producers' vector instructions come straight out of the co-resident MFMA wave's cycles: MI355X_MICROARCH.md, "Two waves per
// SIMD").  A bf16 as a signed 16-bit integer is negative exactly when its sign bit is set, so max(x, 0) as integers is the ReLU
// of the rounded value -- and rounding is monotonic with round(0) = 0, so relu(round(x)) == round(relu(x)) bit for bit; the
// plain-affine clamp is the most negative integer (identity).  (NaN: a positive NaN stays NaN where fmaxf returned 0.)
// Measured (r3, same box, two runs each): igemm2 3x3 launches 11.30 / 11.39 ms without, 11.31 / 11.33 ms with; wgrad2 5.88 / 5.86 ms
// without, 6.01 / 6.04 ms WITH (+2.7 %): like the packed-f32 forms (MI355X_MICROARCH.md constants table) the packed 16-bit op costs
// more beside MFMAs than the two scalar ops it replaces.  Off; bit-exact either way (tests/test_gpu_exact.py ran on both).
#ifndef OCT_PK_RELU
#define OCT_PK_RELU 0
#endif
__device__ __forceinline__ unsigned xf_floor_pk(int xf) { return xf == 2 ? 0x80008000u : 0u; }
__device__ __forceinline__ unsigned pk_clamp_bf16(unsigned v, unsigned floor_pk) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), __builtin_bit_cast(s16x2, floor_pk));
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// V contiguous elements of T with natural vector alignment
template <typename T, int V> struct alignas((sizeof(T) * V) > 16 ? 16 : (sizeof(T) * V)) VecT { T v[V]; };

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* p, float (&out)[V]) {
  VecT<T, V> t = *reinterpret_cast<const VecT<T, V>*>(p);
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = to_f32(t.v[i]);
}
template <typename T, int V>
__device__ __forceinline__ void store_vec(T* p, const float (&in)[V]) {
  VecT<T, V> t;
#pragma unroll
  for (int i = 0; i < V; ++i) t.v[i] = from_f32<T>(in[i]);
  *reinterpret_cast<VecT<T, V>*>(p) = t;
}

// streaming variants: the data is not re-read by this kernel (nontemporal = "nt" cache policy)
template <typename T, int V>
__device__ __forceinline__ void load_vec_nt(const T* p, float (&out)[V]) {
  typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
  constexpr int NQ = (sizeof(T) * V) / 16;
  static_assert(NQ >= 1 && (sizeof(T) * V) % 16 == 0, "16-B multiples only");
  VecT<T, V> t;
  nt_u4* q = reinterpret_cast<nt_u4*>(&t);
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p) + i);
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = to_f32(t.v[i]);
}
template <typename T, int V>
__device__ __forceinline__ void store_vec_nt(T* p, const float (&in)[V]) {
  typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
  constexpr int NQ = (sizeof(T) * V) / 16;
  static_assert(NQ >= 1 && (sizeof(T) * V) % 16 == 0, "16-B multiples only");
  VecT<T, V> t;
#pragma unroll
  for (int i = 0; i < V; ++i) t.v[i] = from_f32<T>(in[i]);
  const nt_u4* q = reinterpret_cast<const nt_u4*>(&t);
#pragma unroll
  for (int i = 0; i < NQ; ++i) __builtin_nontemporal_store(q[i], reinterpret_cast<nt_u4*>(p) + i);
}

// MFMA wrappers: one "k16 step" = 16 contraction elements, 8 per lane (lane>>5 picks the half).
// D[row][col] += sum_k A[row][k] * B[k][col]; lane l holds A[row=l&31][8*(l>>5)+j] and
// B[8*(l>>5)+j][col=l&31], j=0..7; D: col = l&31, row = (i&3) + 8*(i>>2) + 4*(l>>5), i = reg.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  // first step of an accumulation: C is the inline constant 0, the accumulator needs no zero fill
  static __device__ __forceinline__ void mma0(f32x16& acc, const Frag& a, const Frag& b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
  }
  // v_mfma_f32_16x16x32_bf16 on quarter S of a 16-register accumulator: D[row][col] += sum_k A[row][k] B[k][col] with
  // lane l holding A[row = l&15][8*(l>>4)+j] and B[8*(l>>4)+j][col = l&15], j = 0..7, and D[row = 4*(l>>4)+e][col = l&15]
  // in register 4*S + e.  Same FLOPs per cycle as the 32x32x16 form; the chip holds a higher clock on it (MI355X_MICROARCH.md,
  // DVFS give-back item 7).
  template <int S>
  static __device__ __forceinline__ void mma16(f32x16& acc, const Frag& a, const Frag& b) {
    f32x4 v = {acc[4 * S], acc[4 * S + 1], acc[4 * S + 2], acc[4 * S + 3]};
    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, v, 0, 0, 0);
    acc[4 * S] = v[0]; acc[4 * S + 1] = v[1]; acc[4 * S + 2] = v[2]; acc[4 * S + 3] = v[3];
  }
  static __device__ __forceinline__ Frag load(const void* p) { return *reinterpret_cast<const Frag*>(p); }
  static __device__ __forceinline__ Frag zero() {
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)0.0f;
    return f;
  }
  static __device__ __forceinline__ void set(Frag& f, int j, bf16_t v) { f[j] = v; }
};
struct F32Frag { f32x4 lo, hi; };
template <> struct Mma<float> {
  typedef F32Frag Frag;
  // exact fp32: eight 32x32x2 MFMAs; MFMA #j contracts k = {j, 8+j} of the 16-step
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
  }
  static __device__ __forceinline__ Frag load(const void* p) {
    Frag f;
    f.lo = reinterpret_cast<const f32x4*>(p)[0];
    f.hi = reinterpret_cast<const f32x4*>(p)[1];
    return f;
  }
  static __device__ __forceinline__ Frag zero() {
    Frag f;
    f.lo = f32x4{0, 0, 0, 0};
    f.hi = f32x4{0, 0, 0, 0};
    return f;
  }
  static __device__ __forceinline__ void set(Frag& f, int j, float v) {
    if (j < 4) f.lo[j] = v; else f.hi[j - 4] = v;
  }
};

// sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Reduce 16 per-lane values over the 32 lanes that share (lane>>5): transposing butterfly
// (reduce-scatter).  The returned value is the total of register index
//   reg = 8*bit4 + 4*bit3 + 2*bit2 + bit1   (bits of the lane id); lanes l and l^1 hold the same.
__device__ __forceinline__ float reduce32_scatter16(const float (&v)[16], int lane) {
  float t8[8], t4[4], t2[2];
  bool b = (lane & 16) != 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float keep = b ? v[k + 8] : v[k];
    float send = b ? v[k] : v[k + 8];
    t8[k] = keep + __shfl_xor(send, 16);
  }
  b = (lane & 8) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float keep = b ? t8[k + 4] : t8[k];
    float send = b ? t8[k] : t8[k + 4];
    t4[k] = keep + __shfl_xor(send, 8);
  }
  b = (lane & 4) != 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float keep = b ? t4[k + 2] : t4[k];
    float send = b ? t4[k] : t4[k + 2];
    t2[k] = keep + __shfl_xor(send, 4);
  }
  b = (lane & 2) != 0;
  float keep = b ? t2[1] : t2[0];
  float send = b ? t2[0] : t2[1];
  float s = keep + __shfl_xor(send, 2);
  s += __shfl_xor(s, 1);
  return s;
}
// The same over the 16 lanes that share (lane>>4), 8 values: the returned value is the total of register index
//   reg = 4*bit3 + 2*bit2 + bit1 (bits of the lane id); lanes l and l^1 hold the same.
__device__ __forceinline__ float reduce16_scatter8(const float (&v)[8], int lane) {
  float t4[4], t2[2];
  bool b = (lane & 8) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float keep = b ? v[k + 4] : v[k];
    float send = b ? v[k] : v[k + 4];
    t4[k] = keep + __shfl_xor(send, 8);
  }
  b = (lane & 4) != 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float keep = b ? t4[k + 2] : t4[k];
    float send = b ? t4[k] : t4[k + 2];
    t2[k] = keep + __shfl_xor(send, 4);
  }
  b = (lane & 2) != 0;
  float keep = b ? t2[1] : t2[0];
  float send = b ? t2[0] : t2[1];
  float s = keep + __shfl_xor(send, 2);
  s += __shfl_xor(s, 1);
  return s;
}
__device__ __forceinline__ int scatter16_reg_of_lane(int lane) {
  return ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
