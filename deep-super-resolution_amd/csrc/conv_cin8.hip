// First-layer convolution on gfx950: 3x3, stride 1, zero pad 1, Cin <= 8 (the RGB image padded to 8 channels) -> 64.
//   discriminator.py:22 (Conv2d(3,64,3,1,1) + LeakyReLU) at the HR resolution, and VGG19 features[0] (utils/GAN.py).
// The layer is bound by writing its 64-channel output (128 B / pixel against 16 B / pixel read), so the kernel is
// built around that store: one 8x32-pixel tile per block iteration, the 10x34 input halo staged once in LDS
// (5.4 KB), every tap read from it as a 16-byte fragment, and the result transposed through a wave-private LDS
// slab so that each pixel's 128-byte channel row leaves as one full line.
//
// GEMM view per wave: D[m = cout][n = pixel] = sum_k A[m][k] B[k][n] with k = (tap, cin): one MFMA 16x16x32 k-step
// covers 4 taps x 8 channels, so the 9 taps take 3 k-steps (the last three slots read a zero chunk).  A (weights,
// 12 fragments) stays in registers for the life of the persistent block.
#include "../../include/dsr_hip.h"
#include <stdlib.h>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int TH = 8, TW = 32, HW_ = TW + 2, HROWS = TH + 2;
constexpr int HALO = HROWS * HW_;           // 340 chunks of 16 B
constexpr int SH_BYTES = (HALO + 1) * 16;   // + one zero chunk for the unused tap slots (bias: 256 B behind it)
constexpr int SC_WAVE = 64 * 128;           // 64 pixels x 64 channels x 2 B
}   // namespace

template <int DT>
__global__ __launch_bounds__(256, 3) void conv_cin8_kernel(const Cin8Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SH_BYTES + 256 + 4 * SC_WAVE];
  unsigned char* sH = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  unsigned char* sC = smem + SH_BYTES + 256 + wave * SC_WAVE;
  const unsigned short* Wt = (const unsigned short*)a.w;
  unsigned short* Y = (unsigned short*)a.y;

  // weights: A[m = cout][k-chunk g] of k-step ks <-> tap 4ks+g, 8 input channels
  U4 wr[3][4];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = 4 * ks + g;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) wr[ks][mf] = load16_or_zero(Wt, ((size_t)tap * 64 + 16 * mf + l16) * 8, tap < 9);
  }
  const float slope = a.prelu ? a.prelu[0] : a.slope;
  const int act = a.act == DSR_ACT_PRELU ? DSR_ACT_LEAKY : a.act;
  // B fragment addresses (bytes into sH): pixel (row 2*wave + (f>>1), col 16*(f&1) + l16) shifted by this lane's tap;
  // the three unused slots of the last k-step read the zero chunk
  const int pbase = ((2 * wave) * HW_ + l16) * 16;
  int toff[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = 4 * ks + g;
    toff[ks] = tap < 9 ? ((tap / 3) * HW_ + tap % 3) * 16 : -1;
  }
  float* sBias = reinterpret_cast<float*>(smem + SH_BYTES);       // 64 floats behind the halo
  if (tid < 64) sBias[tid] = a.bias ? a.bias[tid] : 0.f;
  if (tid == 0) *reinterpret_cast<U4*>(sH + HALO * 16) = U4{0u, 0u, 0u, 0u};

  const int per_img = a.tiles_y * a.tiles_x;
  // buffer loads: an out-of-range offset returns zeros, so padding needs no select on the data (a select would
  // make the prefetch wait for its own loads -- and, vmcnt being in-order, for every store issued before them)
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int hy0 = tid / HW_, hx0 = tid - hy0 * HW_;
  const int hy1 = (tid + 256) / HW_, hx1 = (tid + 256) - hy1 * HW_;
  const bool has1 = tid + 256 < HALO;
  U4 h0, h1;
  auto gload = [&](int tile) {
    const int n = tile / per_img, rem = tile - n * per_img;
    const int ty = rem / a.tiles_x;
    const int y0 = ty * TH - 1, x0 = (rem - ty * a.tiles_x) * TW - 1;
    const int nb = n * a.H * a.W;
    {
      const int iy = y0 + hy0, ix = x0 + hx0;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      h0 = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(
                                      xrsrc, ok ? (unsigned)(nb + iy * a.W + ix) * 16u : OOB, 0, 0));
    }
    {
      const int iy = y0 + hy1, ix = x0 + hx1;
      const bool ok = has1 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      h1 = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(
                                      xrsrc, ok ? (unsigned)(nb + iy * a.W + ix) * 16u : OOB, 0, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) gload(tile);
  while (tile < a.ntiles) {
    *reinterpret_cast<U4*>(sH + tid * 16) = h0;
    if (has1) *reinterpret_cast<U4*>(sH + (tid + 256) * 16) = h1;
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) gload(next);

    const int tn = tile / per_img, trem = tile - tn * per_img;
    const int tty = trem / a.tiles_x;
    const int oy0 = tty * TH + 2 * wave, ox0 = (trem - tty * a.tiles_x) * TW, nrow = tn * a.H;
    // two passes of 32 pixels (one tile row of this wave each): 32 accumulator registers instead of 64
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      f32x4 acc[4][2];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + 16 * mf + 4 * g);
        acc[mf][0] = b4;
        acc[mf][1] = b4;
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        U4 fb[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int off = toff[ks] >= 0 ? pbase + (half * HW_ + 16 * c) * 16 + toff[ks] : HALO * 16;
          fb[c] = *reinterpret_cast<const U4*>(sH + off);
        }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          acc[mf][0] = mfma16<DT>(wr[ks][mf], fb[0], acc[mf][0]);
          acc[mf][1] = mfma16<DT>(wr[ks][mf], fb[1], acc[mf][1]);
        }
      }
      // lane holds couts 16mf+4g..+3 of pixel p = 32*half + 16c + l16 -> 8 bytes into the wave's slab
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int p = 32 * half + 16 * c + l16;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = act_apply(act, acc[mf][c][j], slope);
          uint2 h;
          h.x = (unsigned)f2h<DT>(v[0]) | ((unsigned)f2h<DT>(v[1]) << 16);
          h.y = (unsigned)f2h<DT>(v[2]) | ((unsigned)f2h<DT>(v[3]) << 16);
          const int c16 = 2 * mf + (g >> 1);
          *reinterpret_cast<uint2*>(sC + p * 128 + ((c16 ^ (p & 7)) << 4) + (g & 1) * 8) = h;
        }
      }
      // this tile row (32 pixels x 128 B = 4 KB contiguous in NHWC) leaves as full lines
      const int oy = oy0 + half;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        const int p = 32 * half + (idx >> 3), c16 = idx & 7;
        const U4 v = *reinterpret_cast<const U4*>(sC + p * 128 + ((c16 ^ (p & 7)) << 4));
        const int ox = ox0 + (idx >> 3);
        if (oy < a.H && ox < a.W) {
          U4* dst = reinterpret_cast<U4*>(Y + ((size_t)(nrow + oy) * a.W + ox) * 64 + c16 * 8);
          if (a.nt_store) __builtin_nontemporal_store(v, dst); else *dst = v;     // (uniform)
        }
      }
    }
    __syncthreads();   // every wave is done with the halo before the next tile overwrites it
    tile = next;
  }
}

void dsr_launch_conv_cin8(Cin8Args& a, int N, int dtype, hipStream_t st) {
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.H * a.W * 16);
  {
    const char* e = getenv("DSR_CIN8_NT");               // tuning switch (measured: 0.284 vs 0.281 ms on D's first layer: off)
    a.nt_store = e ? atoi(e) : 0;
  }
  const int blocks = a.ntiles < 768 ? a.ntiles : 768;   // persistent: three 4-wave blocks per CU
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_cin8_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_cin8_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, a);
}
