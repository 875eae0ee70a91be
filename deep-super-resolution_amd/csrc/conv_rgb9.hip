// The generator's head on gfx950: 9x9, stride 1, zero pad 4, RGB (Cin <= 3, stored padded to 8 channels) -> 64, + PReLU.
//   models/GAN/generator.py:48 (Conv2d(3, 64, 9, 1, 4) + PReLU) at the LR resolution, training and inference.
// On the gather kernel this layer ran 81 taps of 8 stored channels each (648 k for 243 real ones, a quarter of every MFMA
// k-step used: 0.141 ms = 115 TFLOP/s at config 3).  Like conv_cin8.hip the layer is bound by writing its output (128 B per pixel
// against 16 B read), so the kernel is built around that store; what differs is the k dimension:
//
// One kernel ROW per MFMA k-step.  The halo of a tile is staged in LDS with its three real channels PACKED (6 bytes per pixel),
// so the 9 taps x 3 channels of kernel row ky under output pixel x are the 27 CONSECUTIVE elements row[3x .. 3x + 26] of halo
// row y + ky: an MFMA 16x16x32 B fragment (lane = pixel l16, k chunk g) is the 16 bytes at 6 (x + l16) + 16 g, and 9 k-steps
// cover the whole 9x9 window (27 of 32 k used).  Those 16 bytes are 4-byte aligned for even pixels only, and a 2-byte-aligned
// ds_read_b128 -- legal on gfx950 -- is served element by element (measured: 68 us for the layer, the LDS reads all of it); so
// the halo is kept TWICE, the second copy one element further right: odd pixels read that one, every fragment is two
// ds_read2_b32 at full rate, and the second copy costs 3 more 2-byte LDS writes per halo pixel.
// Slots 27..31 lie over the next pixels of the row: finite data that must not count, so the chunk g = 3 is masked after the read
// (weights of those slots are zero as well; the mask keeps an Inf / NaN two pixels outside a window out of it).
// A (weights: 9 rows x 4 cout tiles = 36 fragments, 144 VGPRs) stays in registers for the life of the persistent block, compacted
// once per block from the forward weight image [81][64][8] through LDS.
#include "../../include/dsr_hip.h"
#include <stdlib.h>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int KS = 9;
constexpr int TH = 8, TW = 32, HROWS = TH + KS - 1, HCOLS = TW + KS - 1;   // 16 x 40 halo pixels
constexpr int HALO = HROWS * HCOLS;          // 640 = 2.5 per thread
constexpr int ROW_BYTES = 256;               // 40 pixels x 6 B = 240, + 16 zero bytes (what the last fragments of a row run into)
constexpr int SH_COPY = HROWS * ROW_BYTES;   // 4 KB
constexpr int SH_BYTES = 2 * SH_COPY;        // even-pixel copy | odd-pixel copy (shifted by one element)
constexpr int SC_WAVE = 64 * 128;            // 64 pixels x 64 channels x 2 B
static_assert(6 * (TW - 1) + 2 + 16 * 3 + 16 <= ROW_BYTES, "a fragment stays inside its halo row");
}   // namespace

template <int DT>
__global__ __launch_bounds__(256, 2) void conv_rgb9_kernel(const Cin8Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SH_BYTES + 256 + 4 * SC_WAVE];
  unsigned char* sH = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  unsigned char* sC = smem + SH_BYTES + 256 + wave * SC_WAVE;
  const unsigned short* Wt = (const unsigned short*)a.w;
  unsigned short* Y = (unsigned short*)a.y;

  // weights: A[m = cout 16 mf + l16][k = 8 g + j] of kernel row ky <-> tap (ky, k / 3), channel k % 3; zero for k >= 27.
  // Compacted once per block through LDS (36 KB over the regions used later): [ky][cout][32 k], then 36 aligned 16-byte reads.
  {
    unsigned short* sW = reinterpret_cast<unsigned short*>(smem);
    static_assert(KS * 64 * 32 * 2 <= SH_BYTES + 256 + 4 * SC_WAVE, "the compacted weights fit the block's LDS");
    // 81 x 64 chunks of 16 bytes (tap, cout: 8 stored channels, 3 real): 21 independent loads per thread, all in flight at once
    // (element-wise loads in a rolled loop were 18 dependent round trips: a third of the launch at config 3)
    constexpr int NCH = KS * KS * 64, PER = (NCH + 255) / 256;
    U4 ch[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      ch[i] = q < NCH ? *reinterpret_cast<const U4*>(Wt + (size_t)q * 8) : U4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      if (q < NCH) {
        const int tap = q >> 6, co = q & 63;
        const int ky = tap / KS, kx = tap - ky * KS;
        unsigned short* d = sW + (ky * 64 + co) * 32 + 3 * kx;
        d[0] = (unsigned short)ch[i].x;
        d[1] = (unsigned short)(ch[i].x >> 16);
        d[2] = (unsigned short)ch[i].y;
        if (kx == KS - 1) d[3] = d[4] = d[5] = d[6] = d[7] = 0;     // k = 27 .. 31 of the row
      }
    }
    __syncthreads();
  }
  U4 wr[KS][4];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
      wr[ky][mf] = *reinterpret_cast<const U4*>(smem + ((ky * 64 + 16 * mf + l16) * 32 + 8 * g) * 2);
  __syncthreads();                              // every lane has its fragments: the regions may be reused
  const float slope = a.prelu ? a.prelu[0] : a.slope;
  const int act = a.act == DSR_ACT_PRELU ? DSR_ACT_LEAKY : a.act;
  // chunk g = 3 holds k = 24 .. 31: elements 24, 25, 26 count
  const unsigned m1 = g == 3 ? 0x0000FFFFu : 0xFFFFFFFFu, m2 = g == 3 ? 0u : 0xFFFFFFFFu;
  // fragment of (tile row 2 wave, column l16), kernel row 0: 4-byte aligned in the copy of this lane's pixel parity
  const int pbase = (2 * wave) * ROW_BYTES + 6 * l16 + 16 * g + ((l16 & 1) ? SH_COPY + 2 : 0);
  float* sBias = reinterpret_cast<float*>(smem + SH_BYTES);          // 64 floats behind the halo
  if (tid < 64) sBias[tid] = a.bias ? a.bias[tid] : 0.f;
  // both copies zeroed once: the bytes behind a row's 40 pixels (what its last fragments run into) are never written again
  static_assert(SH_COPY == 256 * 16, "one 16-byte store per thread and copy");
  *reinterpret_cast<U4*>(sH + tid * 16) = U4{0u, 0u, 0u, 0u};
  *reinterpret_cast<U4*>(sH + SH_COPY + tid * 16) = U4{0u, 0u, 0u, 0u};
  __syncthreads();

  const int per_img = a.tiles_y * a.tiles_x;
  // buffer loads: an out-of-range offset returns zeros, so the zero padding needs no select on the data
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  int hy[3], hx[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hp = tid + 256 * i;
    hy[i] = hp / HCOLS;
    hx[i] = hp - hy[i] * HCOLS;
  }
  const bool has2 = tid + 512 < HALO;
  typedef __attribute__((ext_vector_type(2))) unsigned U2;
  U2 hv[3];                                   // channels 0..3 of a halo pixel (8 of its 16 stored bytes)
  auto gload = [&](int tile) {
    const int n = tile / per_img, rem = tile - n * per_img;
    const int ty = rem / a.tiles_x;
    const int y0 = ty * TH - KS / 2, x0 = (rem - ty * a.tiles_x) * TW - KS / 2;
    const int nb = n * a.H * a.W;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int iy = y0 + hy[i], ix = x0 + hx[i];
      const bool ok = (i < 2 || has2) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      hv[i] = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, ok ? (unsigned)(nb + iy * a.W + ix) * 16u : OOB, 0, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) gload(tile);
  while (tile < a.ntiles) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || has2) {
        unsigned short* d = reinterpret_cast<unsigned short*>(sH + hy[i] * ROW_BYTES + hx[i] * 6);
        d[0] = (unsigned short)hv[i].x;
        d[1] = (unsigned short)(hv[i].x >> 16);
        d[2] = (unsigned short)hv[i].y;
        d[SH_COPY / 2 + 1] = (unsigned short)hv[i].x;
        d[SH_COPY / 2 + 2] = (unsigned short)(hv[i].x >> 16);
        d[SH_COPY / 2 + 3] = (unsigned short)hv[i].y;
      }
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
      for (int ky = 0; ky < KS; ++ky) {
        U4 fb[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned* q = reinterpret_cast<const unsigned*>(sH + pbase + (half + ky) * ROW_BYTES + 96 * c);   // 4-byte aligned
          fb[c] = U4{q[0], q[1], q[2], q[3]};
          fb[c].y &= m1;
          fb[c].z &= m2;
          fb[c].w &= m2;
        }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          acc[mf][0] = mfma16<DT>(wr[ky][mf], fb[0], acc[mf][0]);
          acc[mf][1] = mfma16<DT>(wr[ky][mf], fb[1], acc[mf][1]);
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
        if (oy < a.H && ox < a.W) *reinterpret_cast<U4*>(Y + ((size_t)(nrow + oy) * a.W + ox) * 64 + c16 * 8) = v;
      }
    }
    __syncthreads();   // every wave is done with the halo before the next tile overwrites it
    tile = next;
  }
}

int dsr_conv_rgb9_supported(int KH, int KW, int stride, int pad, int pad_mode, int Cin, int Cout) {
  const char* e = getenv("DSR_CONV_RGB9");       // tuning switch, read per call (a test compares the two): 0 = the gather kernel
  return !(e && e[0] == '0') && KH == KS && KW == KS && stride == 1 && pad == KS / 2 && pad_mode == DSR_PAD_ZERO && Cin <= 3 && Cout == 64;
}

void dsr_launch_conv_rgb9(Cin8Args& a, int N, int dtype, hipStream_t st) {
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.H * a.W * 16);
  a.nt_store = 0;
  const int blocks = a.ntiles < 512 ? a.ntiles : 512;   // persistent: two 4-wave blocks per CU (144 weight registers each wave)
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_rgb9_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_rgb9_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------ weight gradient of the same layer
// dW[co][c][ky][kx] = sum_{n,y,x} dy[n][y][x][co] * img[n][y + ky - 4][x + kx - 4][c].  With the packed halo row the 27 products
// of kernel row ky under pixel x read row[3x + k'], k' = 3 kx + c: per (output row y, ky) ONE GEMM over the 32 pixels of a tile
// row,  D_ky[co][k'] += A[co][x] B[x][k'],  A = dy^T (transposing LDS reads of the staged dy tile), B[x][k'] = row_{y+ky}[3x + k']
// -- 8 MFMAs (4 cout tiles x 2 k' tiles) per (row, ky) against 324 on the tap-per-MFMA kernel (81 taps x 4, 3 of 16 columns
// used: 0.123 ms at config 3).  B through the same transposing read: a lane hands in the 8 bytes (pixel p, k' .. k' + 3) at
// 6 p + 2 k', which must be 8-byte aligned: FOUR copies of the halo, copy r one element further right than copy r - 1, pixel p
// reads copy p % 4.  The pixel order inside a k-step is the transposing read's (4g + j, 16 + 4g + j), the same for A and B.
// Waves split the kernel rows (0: ky 0, 4, 8; w: ky w, w + 4), every wave reads all four A fragments of a tile row.
// Each block leaves ONE partial slab [9][64][32] (k' = 27..31: unused columns); rgb9_wgrad_reduce_kernel sums the slabs.
namespace {
constexpr int WG_COPY = HROWS * ROW_BYTES;          // 4 KB per halo copy
constexpr int WG_X = 4 * WG_COPY;                   // 16 KB
constexpr int WG_Y = TH * TW * 128;                 // 32 KB: dy tile [pixel][64 cout], 16-byte chunks swizzled by pixel
constexpr int WG_SLAB = KS * 64 * 32;               // floats per partial slab
__device__ __forceinline__ s16x4 tr_read8(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
}   // namespace

template <int DT>
__global__ __launch_bounds__(256, 2) void conv_rgb9_wgrad_kernel(const Rgb9WgradArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WG_X + WG_Y];
  unsigned char* sX = smem;
  unsigned char* sY = smem + WG_X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, c4 = l16 & 3;
  const int nky = wave == 0 ? 3 : 2;                 // kernel rows wave, wave + 4 (, 8)

  f32x4 acc[3][4][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[t][i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // all four copies zeroed once: the bytes around a row's 40 pixels are never written again (they feed k' >= 27 only, but must
  // not be NaN patterns next to real columns of the same MFMA: 0 * NaN)
#pragma unroll
  for (int i = 0; i < WG_X / (256 * 16); ++i) *reinterpret_cast<U4*>(sX + (tid + 256 * i) * 16) = U4{0u, 0u, 0u, 0u};
  __syncthreads();

  const int per_img = a.tiles_y * a.tiles_x;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  int hy[3], hx[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hp = tid + 256 * i;
    hy[i] = hp / HCOLS;
    hx[i] = hp - hy[i] * HCOLS;
  }
  const bool has2 = tid + 512 < HALO;
  typedef __attribute__((ext_vector_type(2))) unsigned U2;
  U2 hv[3];
  U4 yv[TH];                                         // this thread's 16-byte chunk (tid & 7) of pixel (tid >> 3) of every tile row
  auto gload = [&](int tile) {
    const int n = tile / per_img, rem = tile - n * per_img;
    const int ty = rem / a.tiles_x;
    const int oy0 = ty * TH, ox0 = (rem - ty * a.tiles_x) * TW;
    const int nb = n * a.H * a.W;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int iy = oy0 - KS / 2 + hy[i], ix = ox0 - KS / 2 + hx[i];
      const bool ok = (i < 2 || has2) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      hv[i] = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, ok ? (unsigned)(nb + iy * a.W + ix) * 16u : OOB, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const int oy = oy0 + r, ox = ox0 + (tid >> 3);
      const bool ok = oy < a.H && ox < a.W;
      yv[r] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(
                                         yrsrc, ok ? (unsigned)((nb + oy * a.W + ox) * 128 + (tid & 7) * 16) : OOB, 0, 0));
    }
  };
  // B fragment address of this lane inside a halo row: pixel 4g + q4 (its copy: q4), k' = 4 c4 .. + 3 (+ 16 nt, + 6 * 16 for hi)
  const int boff = q4 * WG_COPY + 6 * (4 * g + q4) + 2 * q4 + 8 * c4;
  int tile = blockIdx.x;
  if (tile < a.ntiles) gload(tile);
  while (tile < a.ntiles) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || has2) {
        unsigned short* d = reinterpret_cast<unsigned short*>(sX + hy[i] * ROW_BYTES + hx[i] * 6);
        const unsigned short e0 = (unsigned short)hv[i].x, e1 = (unsigned short)(hv[i].x >> 16), e2 = (unsigned short)hv[i].y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r * (WG_COPY / 2) + r] = e0;
          d[r * (WG_COPY / 2) + r + 1] = e1;
          d[r * (WG_COPY / 2) + r + 2] = e2;
        }
      }
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const int p = r * TW + (tid >> 3);
      *reinterpret_cast<U4*>(sY + p * 128 + (((tid & 7) ^ (p & 7)) << 4)) = yv[r];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) gload(next);

#pragma unroll 1
    for (int r = 0; r < TH; ++r) {
      U4 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = i * 16 + 4 * c4;
        const int p1 = r * TW + 4 * g + q4, p2 = p1 + 16;
        const s16x4 lo = tr_read8(sY + p1 * 128 + (((ch >> 3) ^ (p1 & 7)) << 4) + (ch & 7) * 2);
        const s16x4 hi = tr_read8(sY + p2 * 128 + (((ch >> 3) ^ (p2 & 7)) << 4) + (ch & 7) * 2);
        fa[i] = __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t < nky) {                               // wave-uniform
          const int ky = wave + 4 * t;
          const unsigned char* row = sX + (r + ky) * ROW_BYTES + boff;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const s16x4 lo = tr_read8(row + 32 * nt);
            const s16x4 hi = tr_read8(row + 32 * nt + 96);
            const U4 fb = __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i][nt] = mfma16<DT>(fa[i], fb, acc[t][i][nt]);
          }
        }
      }
    }
    __syncthreads();   // every wave is done with the tile before the next one overwrites it
    tile = next;
  }
  float* P = a.partial + (size_t)blockIdx.x * WG_SLAB;
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < nky) {
      const int ky = wave + 4 * t;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[(ky * 64 + 16 * i + 4 * g + r) * 32 + 16 * nt + l16] = acc[t][i][nt][r];
    }
}

// dw[co][c][ky][kx] = sum over the slabs of P[ky][co][3 kx + c].  A block = 32 consecutive (ky, co, k') x 8 slab groups: group q
// sums slabs q, q + 8, ... (8 loads in flight), the 8 group sums are added in group order through LDS -- fixed order, double:
// deterministic.  (One thread per output walking all 256 slabs was a 64-deep load chain: 40 of the launch pair's 67 us.)
__global__ __launch_bounds__(256) void rgb9_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int slabs,
                                                                int Cin) {
  __shared__ double red[8][32];
  const int idx = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;     // (ky, co, k') ; slab group
  double s = 0.0;
  int z = q;
  for (; z + 56 < slabs; z += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(z + 8 * u) * WG_SLAB + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; z < slabs; z += 8) s += (double)partial[(size_t)z * WG_SLAB + idx];
  red[q][threadIdx.x & 31] = s;
  __syncthreads();
  if (q != 0) return;
#pragma unroll
  for (int u = 1; u < 8; ++u) s += red[u][threadIdx.x];
  const int kp = idx & 31, co = (idx >> 5) & 63, ky = idx >> 11;
  const int kx = kp / 3, c = kp - 3 * kx;
  if (kp >= 27 || c >= Cin) return;
  dw[((size_t)co * Cin + c) * (KS * KS) + ky * KS + kx] = (float)s;
}

// blocks (= partial slabs of WG_SLAB floats) the launch will use
int dsr_wgrad_rgb9_blocks(int N, int H, int W) {
  const long long nt = (long long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  static const int cap = [] { const char* e = getenv("DSR_RGB9_WGRAD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();   // tuning switch (measured at config 3: 256 blocks 47 us, 512 43, 1024 slower: the slabs)
  return (int)(nt < cap ? nt : cap);                  // persistent, two blocks per CU: one slab per block, 4 tiles each at config 3
}
size_t dsr_wgrad_rgb9_slab_floats() { return WG_SLAB; }

void dsr_launch_wgrad_rgb9(Rgb9WgradArgs& a, int N, int Cin, float* dw, int dtype, hipStream_t st) {
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.H * a.W * 16);
  a.dy_bytes = (unsigned)((size_t)N * a.H * a.W * 128);
  const int blocks = dsr_wgrad_rgb9_blocks(N, a.H, a.W);
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_rgb9_wgrad_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_rgb9_wgrad_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(rgb9_wgrad_reduce_kernel, dim3(WG_SLAB / 32), dim3(256), 0, st, (const float*)a.partial, dw, blocks, Cin);
}
