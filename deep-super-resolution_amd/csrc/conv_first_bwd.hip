// Backward of a first layer whose input needs no gradient: Conv2d(<=3 -> 64, 3x3, stride 1, zero pad 1) + LeakyReLU /
// ReLU / nothing (discriminator.py:22,25-27 on the HR-size images).  The reference computes, in three passes over
// 64-channel tensors,  g = dout * act'(y);  db = sum_p g;  dW = sum_p g[p] x[p+tap]  (and no dx: the input is an image).
// Here that is ONE pass: per tile the kernel
//   * DMAs the dout and y tiles (128 pixels x 64 channels each) and the 4 x 66 pixel image halo into LDS,
//   * rewrites the dout tile in place as g = dout * act'(y),
//   * builds the im2col image B[p][col] with col = 3*tap + ci (27 columns), col 27 = 1.0 -- so that the GEMM
//         D[co][col] = sum_p g[p][co] * B[p][col]         (pixels are the contraction index)
//     yields dW in columns 0..26 and db in column 27,
//   * issues 8 MFMAs per wave (both operands by transposing LDS reads, as in the other weight-gradient kernels).
// It reads dout and y once (2 x 128 B per pixel) and nothing else of that size; the old path (act_bwd + colsum +
// wgrad) moved 5 such tensors.  Deterministic: one partial [64][32] per block, folded by first_bwd_finalize_kernel.
//
// RC (dsr_conv_first_bwd_recompute): y is not read at all.  The only thing the kernel needs from it is the SIGN of the
// pre-activation, and the im2col image it builds anyway IS the forward layer's A operand: one more MFMA K-step per
// (16 pixels x 16 channels) against the layer's own 27 x 64 weights recomputes v = conv(x) + b in fp32, and g = dout * act'(v).
// The 1.07 GB activation of D's first layer (config 3) is then read by this pass no more: 2.15 -> 1.1 GB.
#include <stdlib.h>
#include <string.h>

#include "../../include/dsr_hip.h"
#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int FB_TR = 2, FB_TW = 64, FB_PX = FB_TR * FB_TW;          // 128 pixels per tile
constexpr int FB_HC = FB_TW + 2, FB_HR = FB_TR + 2;                  // 66 x 4 halo
constexpr int FB_T = FB_PX * 128;                                    // one 64-channel tile: 16 KB
constexpr int FB_XRAW = 5 * 1024;                                    // 264 halo pixels x 16 B, rounded to whole DMA pieces
constexpr int FB_B = FB_PX * 64;                                     // im2col image [128][32] bf16
constexpr int FB_STAGE = 2 * FB_T + FB_XRAW;
}   // namespace

__device__ __forceinline__ s16x4 fb_tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// NS = 2: two stages, the next tile's DMA under this tile's work, two blocks per CU (74 KB).  NS = 1: one stage, FOUR blocks
// per CU (37 KB): a tile's phases are short dependent LDS round trips between barriers, so what hides them is more resident
// waves, and the other three blocks cover a block's DMA wait as well as a second stage would.
template <int DT, int NS, bool RC = false>
__global__ __launch_bounds__(256, RC ? 5 : (NS == 1 ? 4 : 2)) void conv_first_bwd_kernel(const FirstBwdArgs a) {
  // stage s: [dout tile | y tile | image halo]; the im2col image is built over the y tile once g has been formed
  static_assert(FB_B <= FB_T, "im2col image must fit in the y tile");
  // (RC: no y tile is fetched -- the im2col image lives where it would be, in 8 of its 16 KB: 29 KB per block, FIVE blocks per CU)
  __shared__ __attribute__((aligned(16))) unsigned char smem[RC ? FB_T + FB_B + FB_XRAW : NS * FB_STAGE];
  constexpr int XOFF = RC ? FB_T + FB_B : 2 * FB_T;                  // the image halo's place in a stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dout), 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.y), 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int per_img = a.tiles_y * a.tiles_x;
  const int chunk = (tid & 7) ^ ((tid >> 3) & 7);                    // source-side swizzle of the lane-linear 128-byte rows

  auto dma = [&](int t, int buf) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int ty = rem / a.tiles_x;
    const int oy0 = ty * FB_TR, ox0 = (rem - ty * a.tiles_x) * FB_TW;
    unsigned char* st = smem + buf * FB_STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                    // 128 pixels x 8 chunks = 16 pieces per tensor, 4 per wave
      const int p = 32 * u + 8 * wave + (lane >> 3);
      const int oy = oy0 + (p >> 6), ox = ox0 + (p & 63);
      const bool ok = oy < a.H && ox < a.W;
      const unsigned off = ok ? (unsigned)((((n * a.H + oy) * a.W + ox) * 64 + chunk * 8) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_ptr)(st + (32 * u + 8 * wave) * 128), 16, off, 0, 0, 0);
      if constexpr (!RC)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, (lds_ptr)(st + FB_T + (32 * u + 8 * wave) * 128), 16, off, 0, 0, 0);
    }
    {                                                                // halo: 264 pixels of 16 B = 5 pieces (waves 0..3 + wave 0 again)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int piece = wave + 4 * u;                              // wave-uniform
        if (piece < 5) {
          const int q = 64 * piece + lane;
          const int hr = q / FB_HC, hc = q - hr * FB_HC;
          const int iy = oy0 - 1 + hr, ix = ox0 - 1 + hc;
          const bool ok = q < FB_HR * FB_HC && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(st + XOFF + piece * 1024), 16,
                                                   ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * 8) * 2) : OOB, 0, 0, 0);
        }
      }
    }
  };

  // RC: this wave's 16 output channels of the layer's weights as ONE MFMA A fragment (rows co = 16 wave + l16, k = column
  // 3 tap + ci of the im2col image; the 16-bit rounding the forward's packed weights went through), and its fp32 bias
  [[maybe_unused]] U4 fw0 = U4{0u, 0u, 0u, 0u};
  [[maybe_unused]] f32x4 bias0 = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (RC) {
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = 8 * g + i;
      const int tap = col / 3, ci = col - 3 * tap;
      e[i] = (col < 27 && ci < a.Cin) ? f2h<DT>(a.w0[((16 * wave + l16) * a.Cin + ci) * 9 + tap]) : (unsigned short)0;
    }
    fw0.x = e[0] | ((unsigned)e[1] << 16);
    fw0.y = e[2] | ((unsigned)e[3] << 16);
    fw0.z = e[4] | ((unsigned)e[5] << 16);
    fw0.w = e[6] | ((unsigned)e[7] << 16);
    if (a.b0) bias0 = f32x4{a.b0[16 * wave + 4 * g], a.b0[16 * wave + 4 * g + 1], a.b0[16 * wave + 4 * g + 2], a.b0[16 * wave + 4 * g + 3]};
  }
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = blockIdx.x;
  if (NS == 2 && t < a.ntiles) dma(t, 0);
  int buf = 0;
  for (; t < a.ntiles; t += gridDim.x, buf ^= (NS - 1)) {
    if constexpr (NS == 1) {
      __builtin_amdgcn_s_barrier();                                  // every wave is done with the previous tile
      asm volatile("" ::: "memory");
      dma(t, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // this tile landed; the previous step's reads are done
    asm volatile("" ::: "memory");
    if (NS == 2 && t + (int)gridDim.x < a.ntiles) dma(t + gridDim.x, buf ^ 1);
    unsigned char* st = smem + buf * FB_STAGE;
    // ---- g = dout * act'(y), in place (4 chunks per thread; slot (p, pos) holds channel chunk pos ^ (p & 7) in both tiles)
    if constexpr (!RC) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int off = (tid + 256 * u) * 16;
      const U4 dv = *reinterpret_cast<const U4*>(st + off);
      const U4 yv = *reinterpret_cast<const U4*>(st + FB_T + off);
      float d[8], yy[8];
      unpack8<DT>(dv, d);
      unpack8<DT>(yv, yy);
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] *= act_grad_from_out(a.act, yy[i], a.slope);
      *reinterpret_cast<U4*>(st + off) = pack8<DT>(d);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every thread is done reading the y tile
    asm volatile("" ::: "memory");
    }
    unsigned char* sBim = st + FB_T;
    // ---- im2col image: row p (64 B), chunk c = 8 columns col = 3*tap + ci; col 27 = 1, 28..31 = 0
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;                                 // 128 pixels x 4 chunks
      const int p = idx >> 2, c = idx & 3;
      const int row = p >> 6, xx = p & 63;
      const unsigned char* xr = st + XOFF;
      unsigned short e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int col = 8 * c + i;
        const int tap = col / 3, ci = col - 3 * tap;
        const int kh = tap / 3, kw = tap - 3 * kh;
        unsigned short v = 0;
        if (col < 27) v = *reinterpret_cast<const unsigned short*>(xr + ((row + kh) * FB_HC + xx + kw) * 16 + ci * 2);
        if (col == 27) v = f2h<DT>(1.f);
        e[i] = v;
      }
      U4 v;
      v.x = e[0] | ((unsigned)e[1] << 16);
      v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16);
      v.w = e[6] | ((unsigned)e[7] << 16);
      *reinterpret_cast<U4*>(sBim + p * 64 + ((c ^ ((p >> 1) & 3)) << 4)) = v;
    }
    if constexpr (RC) {
      if (a.act != DSR_ACT_NONE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the im2col image is complete
        asm volatile("" ::: "memory");
        // ---- v[co][p] = W0[co][:] . im2col[p][:] + b (weights as the A operand: a lane ends up with 4 consecutive channels
        // 16 wave + 4 g .. + 3 of pixel 16 pg + l16 = 8 contiguous bytes of that pixel's dout row), g = dout * act'(v) in place
#pragma unroll
        for (int pg = 0; pg < 8; ++pg) {
          const int p = 16 * pg + l16;
          const U4 fa = *reinterpret_cast<const U4*>(sBim + p * 64 + ((g ^ ((p >> 1) & 3)) << 4));
          const f32x4 v = mfma16<DT>(fw0, fa, bias0);
          unsigned char* dp = st + p * 128 + (((2 * wave + (g >> 1)) ^ (p & 7)) << 4) + (g & 1) * 8;
          const uint2 dv = *reinterpret_cast<const uint2*>(dp);
          float d0 = h2f<DT>((unsigned short)(dv.x & 0xffff)) * act_grad_from_out(a.act, v[0], a.slope);
          float d1 = h2f<DT>((unsigned short)(dv.x >> 16)) * act_grad_from_out(a.act, v[1], a.slope);
          float d2 = h2f<DT>((unsigned short)(dv.y & 0xffff)) * act_grad_from_out(a.act, v[2], a.slope);
          float d3 = h2f<DT>((unsigned short)(dv.y >> 16)) * act_grad_from_out(a.act, v[3], a.slope);
          uint2 o;
          o.x = (unsigned)f2h<DT>(d0) | ((unsigned)f2h<DT>(d1) << 16);
          o.y = (unsigned)f2h<DT>(d2) | ((unsigned)f2h<DT>(d3) << 16);
          *reinterpret_cast<uint2*>(dp) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // g and the im2col image are complete
    asm volatile("" ::: "memory");
    // ---- D[co = 16*wave + ..][col] += sum over the tile's 128 pixels (4 K-steps of 32)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int p1 = 32 * ks + 4 * g + q4, p2 = p1 + 16;
      U4 fa, fb[2];
      {
        const int ch = 16 * wave + cc;                               // A: g[p][co], co = 16*wave + (l16 after the transpose)
        const s16x4 lo = fb_tr_read(st + p1 * 128 + (((ch >> 3) ^ (p1 & 7)) << 4) + (ch & 7) * 2);
        const s16x4 hi = fb_tr_read(st + p2 * 128 + (((ch >> 3) ^ (p2 & 7)) << 4) + (ch & 7) * 2);
        fa = __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int col = 16 * nf + cc;                                // B: im2col[p][col]
        const s16x4 lo = fb_tr_read(sBim + p1 * 64 + (((col >> 3) ^ ((p1 >> 1) & 3)) << 4) + (col & 7) * 2);
        const s16x4 hi = fb_tr_read(sBim + p2 * 64 + (((col >> 3) ^ ((p2 >> 1) & 3)) << 4) + (col & 7) * 2);
        fb[nf] = __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      acc[0] = mfma16<DT>(fa, fb[0], acc[0]);
      acc[1] = mfma16<DT>(fa, fb[1], acc[1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // partial[block][co][32]: lane (g, l16) holds D[co = 16*wave + 4g + j][col = 16nf + l16]
  float* P = a.partial + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int j = 0; j < 4; ++j) P[(16 * wave + 4 * g + j) * 32 + 16 * nf + l16] = acc[nf][j];
}

// dw[co][ci][tap] (OIHW, fp32) and db[co] from the per-block partials (fixed order: deterministic).  One block per output
// channel: thread (r, col) adds the rows r, r + 8, ... of its column in double, the 8 row classes are folded in order.
__global__ __launch_bounds__(256) void first_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int Cin,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double part[8][32];
  const int co = blockIdx.x, r = threadIdx.x >> 5, col = threadIdx.x & 31;
  const float* src = partial + co * 32 + col;
  double s = 0.0;
  int b = r;
  for (; b + 56 < blocks; b += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + 8 * u) * 2048];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; b < blocks; b += 8) s += (double)src[(size_t)b * 2048];
  part[r][col] = s;
  __syncthreads();
  if (r != 0) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += part[q][col];
  if (col < 27) {
    const int tap = col / 3, ci = col - 3 * tap;
    if (ci < Cin) dw[((size_t)co * Cin + ci) * 9 + tap] = (float)s;
  } else if (col == 27 && db != nullptr) {
    db[co] = (float)s;
  }
}

static int first_bwd_stages() {          // tuning switch DSR_FIRST_BWD_STAGES: 1 (default) | 2
  const char* e = getenv("DSR_FIRST_BWD_STAGES");
  return (e && e[0] == '2') ? 2 : 1;
}
static int first_bwd_blocks(long long ntiles, bool rc = true) {      // four (five: recompute form) resident blocks per CU
  const long long cap = rc ? 1280 : 1024;
  return (int)(ntiles < cap ? ntiles : cap);
}

extern "C" size_t dsr_conv_first_bwd_workspace(const dsr_conv_desc* d) {
  if (!d) return 0;
  const long long ntiles = (long long)d->N * ((d->H + FB_TR - 1) / FB_TR) * ((d->W + FB_TW - 1) / FB_TW);
  return (size_t)first_bwd_blocks(ntiles) * 64 * 32 * sizeof(float);
}

extern "C" int dsr_conv_first_bwd_supported(const dsr_conv_desc* d, int act) {
  return d && d->Cin <= 3 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == DSR_PAD_ZERO && (act == DSR_ACT_NONE || act == DSR_ACT_LEAKY || act == DSR_ACT_RELU) &&
         (size_t)d->N * d->H * d->W * 128 < (1ull << 31);
}

static int first_bwd_impl(const dsr_conv_desc* d, const void* x, const void* dout, const void* y, const float* w0, const float* b0,
                          int act, float slope, float* dw, float* db, void* workspace, size_t ws_bytes, dsr_stream_t s) {
  const bool rc = w0 != nullptr;
  if (!dsr_conv_first_bwd_supported(d, act)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_first_bwd: unsupported layer");
  if (!x || !dout || (!y && !rc) || !dw) return dsr_fail(DSR_E_ARG, "conv_first_bwd: null pointer");
  if (act == DSR_ACT_LEAKY && !rc && !(slope > 0.f))   // the derivative is read off the stored output: needs slope > 0
    return dsr_fail(DSR_E_ARG, "conv_first_bwd: LeakyReLU slope %g must be > 0", (double)slope);
  const size_t need = dsr_conv_first_bwd_workspace(d);
  if (!workspace || ws_bytes < need) return dsr_fail(DSR_E_WORKSPACE, "conv_first_bwd: workspace %zu < %zu", ws_bytes, need);
  FirstBwdArgs a;
  a.x = x;
  a.dout = dout;
  a.y = y;
  a.partial = (float*)workspace;
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.act = act;
  a.slope = slope;
  a.w0 = w0;
  a.b0 = b0;
  a.Cin = d->Cin;
  a.tiles_y = (d->H + FB_TR - 1) / FB_TR;
  a.tiles_x = (d->W + FB_TW - 1) / FB_TW;
  a.ntiles = d->N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * 16);
  a.y_bytes = (unsigned)((size_t)d->N * d->H * d->W * 128);
  const int blocks = first_bwd_blocks(a.ntiles, rc);
  const bool two = first_bwd_stages() == 2;
  if (rc) {      // (one stage, four blocks per CU: 29 KB of LDS each)
    if (d->dtype == DSR_BF16)
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_BF16, 1, true>), dim3(blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_F16, 1, true>), dim3(blocks), dim3(256), 0, s, a);
  } else if (d->dtype == DSR_BF16) {
    if (two)
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_BF16, 2>), dim3(blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_BF16, 1>), dim3(blocks), dim3(256), 0, s, a);
  } else {
    if (two)
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_F16, 2>), dim3(blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((conv_first_bwd_kernel<DSR_DTYPE_F16, 1>), dim3(blocks), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(first_bwd_finalize_kernel, dim3(64), dim3(256), 0, s, (const float*)workspace, blocks, d->Cin, dw, db);
  return dsr_launch_status("dsr_conv_first_bwd");
}

extern "C" int dsr_conv_first_bwd(const dsr_conv_desc* d, const void* x, const void* dout, const void* y, int act,
                                  float slope, float* dw, float* db, void* workspace, size_t ws_bytes, dsr_stream_t s) {
  return first_bwd_impl(d, x, dout, y, nullptr, nullptr, act, slope, dw, db, workspace, ws_bytes, s);
}

// The same pass without the activation output: the sign of the pre-activation is recomputed from the image and the layer's own
// fp32 weights `w` (OIHW [64][Cin][3][3]) and bias (nullable) -- see the RC note at the top of this file.
extern "C" int dsr_conv_first_bwd_recompute(const dsr_conv_desc* d, const void* x, const void* dout, const float* w,
                                            const float* bias, int act, float slope, float* dw, float* db, void* workspace,
                                            size_t ws_bytes, dsr_stream_t s) {
  if (!w) return dsr_fail(DSR_E_ARG, "conv_first_bwd_recompute: null weight pointer");
  return first_bwd_impl(d, x, dout, nullptr, w, bias, act, slope, dw, db, workspace, ws_bytes, s);
}

// ---- the stride-2 layer's input gradient and this layer's backward as ONE launch (conv_dgrad_s2_kernel<FB>, conv_dgrad_s2.hip):
// d0 = the image layer (discriminator.py:25), d1 = the 3x3 stride-2 64 -> 64 layer on top of it (:29).  dy = the gradient of d1's
// conv output, w1_dgrad = d1's packed input-gradient weights; the gradient of the 64-channel activation in between is formed per
// tile in LDS, masked by this layer's activation derivative (pre-activation recomputed from `img`, `w0`, `b0`) and contracted
// with the im2col image there -- it never exists in HBM.  Only dw0 / db0 come out: for a step that needs no image gradient.
static bool dgrad_first_bwd_ok(const dsr_conv_desc* d0, const dsr_conv_desc* d1, int act0) {
  const char* e = getenv("DSR_DGRAD_FIRST_BWD");     // tuning switch, read per call: 0 = never
  if (e && e[0] == '0') return false;
  if (!d0 || !d1 || !dsr_conv_first_bwd_supported(d0, act0)) return false;
  if (d1->dtype != d0->dtype || d1->N != d0->N || d1->H != d0->H || d1->W != d0->W || d1->Cin != 64 || d1->Cout != 64) return false;
  if (d1->KH != 3 || d1->KW != 3 || d1->stride != 2 || d1->pad != 1 || d1->pad_mode != DSR_PAD_ZERO) return false;
  if ((d1->H & 1) || (d1->W & 1) || (d1->W / 2) % 256 != 0) return false;      // a tile = 256 consecutive dY pixels of ONE row
  return (size_t)d1->N * d1->H * d1->W * 128 < (1ull << 31);
}
extern "C" int dsr_conv_dgrad_first_bwd_supported(const dsr_conv_desc* d0, const dsr_conv_desc* d1, int act0) {
  return dgrad_first_bwd_ok(d0, d1, act0) ? 1 : 0;
}
extern "C" size_t dsr_conv_dgrad_first_bwd_workspace(const dsr_conv_desc* d1) {
  if (!d1) return 0;
  return (size_t)2 * dsr_dgrad_s2_blocks(d1->N, d1->H, d1->W, 64) * 64 * 32 * sizeof(float);
}
extern "C" int dsr_conv_dgrad_first_bwd(const dsr_conv_desc* d0, const dsr_conv_desc* d1, const void* dy, const void* w1_dgrad,
                                        const void* img, const float* w0, const float* b0, int act0, float slope0, float* dw0,
                                        float* db0, void* workspace, size_t ws_bytes, dsr_stream_t s) {
  if (!dgrad_first_bwd_ok(d0, d1, act0)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad_first_bwd: unsupported layer pair");
  if (!dy || !w1_dgrad || !img || !w0 || !dw0) return dsr_fail(DSR_E_ARG, "conv_dgrad_first_bwd: null pointer");
  const size_t need = dsr_conv_dgrad_first_bwd_workspace(d1);
  if (!workspace || ws_bytes < need) return dsr_fail(DSR_E_WORKSPACE, "conv_dgrad_first_bwd: workspace %zu < %zu", ws_bytes, need);
  DgradS2Args q;
  memset(&q, 0, sizeof(q));
  q.dy = dy;
  q.w = w1_dgrad;
  q.dx = nullptr;                                    // (never written)
  q.H = d1->H;
  q.W = d1->W;
  q.CinP = 64;
  q.CoutP = 64;
  q.img = img;
  q.w0 = w0;
  q.b0 = b0;
  q.fb_partial = (float*)workspace;
  q.img_bytes = (unsigned)((size_t)d0->N * d0->H * d0->W * 16);
  q.Cin0 = d0->Cin;
  q.act0 = act0;
  q.slope0 = slope0;
  dsr_launch_dgrad_s2(q, d1->N, d1->dtype, s);
  const int rows = 2 * dsr_dgrad_s2_blocks(d1->N, d1->H, d1->W, 64);
  hipLaunchKernelGGL(first_bwd_finalize_kernel, dim3(64), dim3(256), 0, s, (const float*)workspace, rows, d0->Cin, dw0, db0);
  return dsr_launch_status("dsr_conv_dgrad_first_bwd");
}
