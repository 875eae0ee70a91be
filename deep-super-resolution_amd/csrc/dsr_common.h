// Shared device helpers for the gfx950 (MI355X / CDNA4) super-resolution kernels.
// Activations are NHWC with the channel count padded to a multiple of 8 ("Cp"), 16-bit
// storage (bf16 for training, f16 for inference), fp32 accumulation everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define DSR_DTYPE_BF16 0
#define DSR_DTYPE_F16 1

// activation codes shared by conv epilogues and the pointwise kernels
#define DSR_ACT_NONE 0
#define DSR_ACT_LEAKY 1   // slope passed by value
#define DSR_ACT_PRELU 2   // slope read from a 1-element device tensor
#define DSR_ACT_RELU 3
#define DSR_ACT_TANH 4
#define DSR_ACT_SIGMOID 5
#define DSR_ACT_ELU 6     // alpha = 1 (nn.ELU() of models/DIP/utils.py:69)

#define DSR_PAD_ZERO 0
#define DSR_PAD_REFLECT 1
#define DSR_PAD_REPLICATE 2

// ---- 16-bit <-> fp32 (bit exact round-to-nearest-even; NaN stays NaN via the compiler cast)
template <int DT>
__device__ __forceinline__ float h2f(unsigned short h) {
  if constexpr (DT == DSR_DTYPE_BF16) {
    return __uint_as_float(((unsigned)h) << 16);
  } else {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
  }
}
template <int DT>
__device__ __forceinline__ unsigned short f2h(float f) {
  if constexpr (DT == DSR_DTYPE_BF16) {
    __bf16 v = (__bf16)f;
    unsigned short h;
    __builtin_memcpy(&h, &v, 2);
    return h;
  } else {
    _Float16 v = (_Float16)f;
    unsigned short h;
    __builtin_memcpy(&h, &v, 2);
    return h;
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned U4;   // one 16-byte vector = 8 x 16-bit channels

template <int DT>
__device__ __forceinline__ void unpack8(const U4& v, float* f) {
  f[0] = h2f<DT>((unsigned short)(v.x & 0xffff));
  f[1] = h2f<DT>((unsigned short)(v.x >> 16));
  f[2] = h2f<DT>((unsigned short)(v.y & 0xffff));
  f[3] = h2f<DT>((unsigned short)(v.y >> 16));
  f[4] = h2f<DT>((unsigned short)(v.z & 0xffff));
  f[5] = h2f<DT>((unsigned short)(v.z >> 16));
  f[6] = h2f<DT>((unsigned short)(v.w & 0xffff));
  f[7] = h2f<DT>((unsigned short)(v.w >> 16));
}
template <int DT>
__device__ __forceinline__ U4 pack8(const float* f) {
  U4 v;
  v.x = (unsigned)f2h<DT>(f[0]) | ((unsigned)f2h<DT>(f[1]) << 16);
  v.y = (unsigned)f2h<DT>(f[2]) | ((unsigned)f2h<DT>(f[3]) << 16);
  v.z = (unsigned)f2h<DT>(f[4]) | ((unsigned)f2h<DT>(f[5]) << 16);
  v.w = (unsigned)f2h<DT>(f[6]) | ((unsigned)f2h<DT>(f[7]) << 16);
  return v;
}

__device__ __forceinline__ float act_apply(int act, float v, float slope) {
  // branch-light on purpose: this is inlined 64x in the conv epilogue.  tanh/sigmoid share one exp:
  // sigmoid(t) = 1/(1+e^-t), tanh(v) = 2*sigmoid(2v) - 1 (abs error ~1e-7, saturates correctly).
  if (act >= DSR_ACT_TANH) {   // wave-uniform
    if (act == DSR_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
    const float t = act == DSR_ACT_TANH ? 2.f * v : v;
    const float sg = __fdividef(1.f, 1.f + __expf(-t));
    return act == DSR_ACT_TANH ? 2.f * sg - 1.f : sg;
  }
  const float neg = act == DSR_ACT_RELU ? 0.f : (act == DSR_ACT_NONE ? v : v * slope);
  return v >= 0.f ? v : neg;
}
// derivative expressed through the activation OUTPUT o (valid for slope > 0)
__device__ __forceinline__ float act_grad_from_out(int act, float o, float slope) {
  if (act == DSR_ACT_TANH) return 1.f - o * o;
  if (act == DSR_ACT_SIGMOID) return o * (1.f - o);
  if (act == DSR_ACT_NONE) return 1.f;
  if (act == DSR_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == DSR_ACT_ELU) return o > 0.f ? 1.f : o + 1.f;   // d/dv (e^v - 1) = o + 1
  return o >= 0.f ? 1.f : slope;
}

// g * act'(o) on 8 packed 16-bit values, o = the activation OUTPUT (ReLU, or Leaky/PReLU with slope > 0): what act_bwd_kernel
// computes per element (fp32 product, one rounding), so a kernel that applies it to its own rounded output is bit-identical
// to that kernel followed by the separate pass.
template <int DT>
__device__ __forceinline__ U4 act_mask8(const U4& gv, const U4& ov, int act, float slope) {
  float g[8], o[8];
  unpack8<DT>(gv, g);
  unpack8<DT>(ov, o);
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = g[k] * act_grad_from_out(act, o[k], slope);
  return pack8<DT>(g);
}

__device__ __forceinline__ int pad_index(int i, int n, int mode, bool& inb) {
  // maps a possibly out-of-range coordinate (branch-free); inb=false means "contributes zero"
  const bool in = (unsigned)i < (unsigned)n;
  const int refl = i < 0 ? -i : 2 * (n - 1) - i;
  const int repl = i < 0 ? 0 : n - 1;
  const int alt = mode == DSR_PAD_REFLECT ? refl : repl;
  const bool alt_ok = (mode != DSR_PAD_ZERO) && ((unsigned)alt < (unsigned)n);
  inb = inb && (in || alt_ok);
  return in ? i : (alt_ok ? alt : 0);
}

// predicated 16-byte load without a branch: masked lanes read element 0 of the tensor (always mapped)
__device__ __forceinline__ U4 load16_or_zero(const unsigned short* base, size_t off, bool ok) {
  U4 v = *reinterpret_cast<const U4*>(base + (ok ? off : 0));
  const U4 z = {0u, 0u, 0u, 0u};
  return ok ? v : z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// MFMA 16x16x32, bf16 or f16 operands (8 elements / lane), fp32 accumulate
template <int DT>
__device__ __forceinline__ f32x4 mfma16(const U4& a, const U4& b, f32x4 c) {
  if constexpr (DT == DSR_DTYPE_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
}

// MFMA 32x32x16: lane l supplies A[m = l%32][k = 8(l/32)..+7] and B[k = 8(l/32)..+7][n = l%32];
// D[m = 8(i/4) + 4(l/32) + i%4][n = l%32] in register i of 16.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int DT>
__device__ __forceinline__ f32x16 mfma32(const U4& a, const U4& b, f32x16 c) {
  if constexpr (DT == DSR_DTYPE_BF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
}

// Adam (torch.optim.Adam defaults, no weight decay): ONE statement of the arithmetic, used by every kernel that applies it
// (pointwise.hip adam_kernel / adam_multi_kernel, linear.hip linear_wgrad_adam_kernel), with floating-point contraction
// off so that each of them rounds identically -- the fused weight-gradient + Adam launch is tested bit-for-bit against
// the two separate launches.
struct AdamCoef {
  float step_size, rbc2, b1, b2, eps, grad_scale;
};
__device__ __forceinline__ AdamCoef adam_coef(const int* __restrict__ step, float lr, float b1, float b2, float eps,
                                              float grad_scale) {
  const int t = *step;
  AdamCoef c;
  c.step_size = lr / (1.f - powf(b1, (float)t));
  c.rbc2 = 1.f / sqrtf(1.f - powf(b2, (float)t));
  c.b1 = b1;
  c.b2 = b2;
  c.eps = eps;
  c.grad_scale = grad_scale;
  return c;
}
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamCoef& c) {
#pragma clang fp contract(off)
  const float gk = g * c.grad_scale;          // 1/S un-does a static loss scale (fp16 storage); 1 otherwise
  m = c.b1 * m + (1.f - c.b1) * gk;
  v = c.b2 * v + (1.f - c.b2) * gk * gk;
  p -= c.step_size * (m / (sqrtf(v) * c.rbc2 + c.eps));
}

// LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes x 16 B, lane-linear from the LDS address in M0) issued by INLINE ASM.
// hipcc's wait-count pass tracks the builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds) as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of every LDS access it cannot prove disjoint from it -- 8-byte LDS stores and every transposing
// read (ds_read_b64_tr_b16): a prefetch issued before a tile's MFMA phase is then waited for at that phase's FIRST operand
// read, and the two never overlap.  Issued by asm the DMA is invisible to the pass; the kernel waits itself (s_waitcnt
// vmcnt(N) + barrier) where the tile is needed.  Compiler-generated vmcnt waits stay correct: vmcnt retires in order, so
// operations the pass does not know about only ever make one of its waits stricter.
typedef __attribute__((ext_vector_type(4))) unsigned dsr_u32x4;
struct BufSrd {
  dsr_u32x4 w;   // raw buffer resource: base[47:0], stride 0, num_records = bytes, flags as __builtin_amdgcn_make_buffer_rsrc(.., 0x00020000)
};
__device__ __forceinline__ BufSrd make_srd(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  BufSrd r;
  r.w = dsr_u32x4{(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}
// `lds`: wave-uniform pointer into shared memory (this wave's 1 KB slot row); `voff`: per-lane byte offset into the buffer
// (an out-of-range offset writes zeros)
__device__ __forceinline__ void lds_dma16(const BufSrd& srd, const void* lds, unsigned voff) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) const void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l), "v"(voff), "s"(srd.w) : "memory");
}

// Sum over the 16 lanes of a DPP row (the 16 pixels of an MFMA fragment); every lane ends up with it.  Four DPP adds -- lane ^ 1
// and lane ^ 2 inside a quad, then the mirrored quad pair and the mirrored half row (every lane of a quad holds the quad's sum
// by then) -- i.e. the summation tree of __shfl_xor(1, 2, 4, 8) without its four ds_bpermute round trips through the LDS.
template <int CTRL>
__device__ __forceinline__ float dsr_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dsr_dpp_add<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v = dsr_dpp_add<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v = dsr_dpp_add<0x141>(v);      // row_half_mirror
  v = dsr_dpp_add<0x140>(v);      // row_mirror
  return v;
}

// XCD-aware, bijective block-id remap (consecutive logical tiles share an XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
