// Convolutions with very few channels on one side, large kernels (the generator's 9x9 head and tail,
// models/GAN/generator.py:47,62): the gather-GEMM kernel re-reads every input pixel once per tap from L2 (81x for
// 9x9) to feed a 16-wide N tile, so these layers were L2-bound at ~2 % of MFMA peak.  Here the input HALO tile is
// staged in LDS once per block and every tap is an address offset into it.
//
//  conv_smalln_kernel      forward, Cout <= 16, stride 1, zero padding: 4 rows x 32 columns of pixels per block,
//                          one row per wave; weights [tap][16][Cin] stream from L1/L2 straight into B fragments.
//  conv_wgrad_taps_kernel  weight gradient for KW-wide tap rows when Cout <= 16 or Cin <= 16: a block stages a
//                          (R+KHB-1) x (31+KW) halo of X and an R x 32 tile of dY, its 4 waves split the KHB*KW taps.
#include <string.h>

#include "dsr_common.h"
#include "dsr_kernels.h"

// Persistent, one block per CU.  LDS = weights (all taps, only the `NB` real rows: lanes of the 16-wide B fragment
// beyond them re-read row 0 and feed output columns that are never stored) + one 8-row x 32-column halo tile.
// The tap loop touches LDS only (A: 4 fragments, B: 1 fragment per 4 MFMAs); the NEXT tile's halo travels
// HBM -> registers underneath it, so no memory wait sits between two tiles' MFMA phases.
template <int DT, int KWT>
__global__ __launch_bounds__(512, 2) void conv_smalln_kernel(const SmallNArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  constexpr int TR = 8;                                  // output rows per tile: 8 waves, one row each (two waves per
                                                         // SIMD hide each other's LDS-read latency)
  const int HR = TR + a.KH - 1;
  constexpr int HC = (32 + KWT - 1 + 7) & ~7;            // multiple of 8: the read swizzle depends on (kw + lane) only
  const int ntaps = a.KH * KWT;
  const int WR = (a.cout + 3) & ~3;                      // weight rows kept per tap (the real output channels)
  unsigned char* sW = smem;                              // [tap][WR rows][128 B]
  unsigned char* sX = smem + ntaps * WR * 128;           // [HR][HC][64 ch], 16-byte chunks XOR-swizzled by pixel
  const int per_img = a.tiles_y * a.tiles_x;
  const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(a.w);
  const int c = tid & 7, pb = tid >> 3;                  // pb in 0..63

  for (int i = tid; i < ntaps * WR * 8; i += 512) {      // global [tap][NB][64] 16-bit -> LDS [tap][WR][64]
    const int ch = i & 7, row = (i >> 3) % WR, tap = (i >> 3) / WR;
    const int src = a.flip ? ntaps - 1 - tap : tap;      // dgrad: mirrored taps of the [tap][ci][co] image
    reinterpret_cast<U4*>(sW)[i] = reinterpret_cast<const U4*>(W)[(src * a.NB + row) * 8 + ch];
  }
  const int wrow = r16 < WR ? r16 : 0;
  const int w_off = wrow * 128 + g * 16;                 // + tap * WR*128 + kk*64

  int lds_off[KWT][2];
#pragma unroll
  for (int kw = 0; kw < KWT; ++kw)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) lds_off[kw][kk] = (kw + r16) * 128 + (((4 * kk + g) ^ ((kw + r16) & 7)) << 4);

  constexpr int NV = (16 * HC + 63) / 64;                // halo vectors per thread: HR(<=16) x HC pixels x 8 chunks / 512
  // (the slot -> (row, column) walk is recomputed where needed instead of being held in 3 x NV registers: the tap
  //  loop needs those registers to keep several LDS reads in flight ahead of the MFMAs)
  const int hr0 = pb / HC, hc0 = pb - hr0 * HC;
  U4 v[NV];
  // buffer loads (out-of-range offset -> zeros): a select on the loaded data would pin the wait for this prefetch
  // right behind its issue, in front of the tap loop it is meant to run under
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  auto fetch = [&](int t) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * TR - a.pad, ox0 = (rem % a.tiles_x) * 32 - a.pad;
    int hr = hr0, hc = hc0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int iy = oy0 + hr, ix = ox0 + hc;
      const bool ok = hr < HR && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      v[u] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(
                                        xrsrc, ok ? (unsigned)((((n * a.IH + iy) * a.IW + ix) * 64 + c * 8) * 2) : OOB, 0, 0));
      hc += 64;
      while (hc >= HC) {
        hc -= HC;
        ++hr;
      }
    }
  };
  auto stash = [&]() {
    int hr = hr0, hc = hc0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int q = hr * HC + hc;
      if (hr < HR) *reinterpret_cast<U4*>(sX + q * 128 + ((c ^ (q & 7)) << 4)) = v[u];
      hc += 64;
      while (hc >= HC) {
        hc -= HC;
        ++hr;
      }
    }
  };
  const int col = r16;
  const float bv = (a.bias && col < a.cout) ? a.bias[col] : 0.f;
  const float slope = a.prelu ? a.prelu[0] : a.slope;

  int t = xcd_remap(blockIdx.x, gridDim.x);          // blocks of one XCD walk neighbouring tiles (shared halo rows)
  const int tstep = gridDim.x;
  if (t < a.ntiles) fetch(t);
  for (; t < a.ntiles; t += tstep) {
    __syncthreads();                                 // previous tile's fragment reads are done (and sW is written)
    stash();
    __syncthreads();
    if (t + tstep < a.ntiles) fetch(t + tstep);

    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < a.KH; ++kh) {
      const unsigned char* rowp = sX + (wave + kh) * HC * 128;              // uniform
      const unsigned char* wp = sW + kh * KWT * WR * 128 + w_off;
#pragma unroll
      for (int kw = 0; kw < KWT; ++kw) {
        // all 10 fragments of this tap first, then its 8 MFMAs: the scheduler can run the next tap's reads under them
        U4 fb[2], fa[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          fb[kk] = *reinterpret_cast<const U4*>(wp + kw * WR * 128 + kk * 64);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            fa[kk][mt] = *reinterpret_cast<const U4*>(rowp + lds_off[kw][kk] + mt * 2048);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = mfma16<DT>(fa[kk][i], fb[kk], acc[i]);
      }
    }

    // epilogue: lane (g, col=r16) holds rows 4g..4g+3 = 4 consecutive x positions of channel `col`
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oyb = (rem / a.tiles_x) * TR + wave, ox0 = (rem % a.tiles_x) * 32;
    if (col < a.cout) {
#pragma unroll
      for (int rr = 0; rr < 1; ++rr) {
        const int oy = oyb + rr;
        if (oy >= a.OH) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int ox = ox0 + mt * 16 + 4 * g;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = act_apply(a.act, acc[mt][r] + bv, slope);
          if (a.out_f32) {
            float* dst = a.out_f32 + (((size_t)n * a.cout + col) * a.OH + oy) * a.OW + ox;
            if (ox + 3 < a.OW && (a.OW & 3) == 0) {
              *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
              for (int r = 0; r < 4; ++r)
                if (ox + r < a.OW) dst[r] = o[r];
            }
          } else {
            unsigned short* Y = reinterpret_cast<unsigned short*>(a.y);
            for (int r = 0; r < 4; ++r)
              if (ox + r < a.OW) Y[((size_t)(n * a.OH + oy) * a.OW + ox + r) * a.CoutP + col] = f2h<DT>(o[r]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ 9x9, 64 -> <=3 channels: Toeplitz column mapping
// conv_smalln_kernel feeds one tap per MFMA with 3 of its 16 output columns useful.  Here the kw offsets go into the
// GEMM's N dimension instead: for tap row kh and kw group G (kw = 0..4 | 5..8),
//     T_G[p][(d, co)] = sum_kh sum_ci in[y + kh][p][ci] * W[kh][kwbase_G + d][ci][co]         (p = input column)
// is one GEMM with N = 5 shifts x 3 channels = 15 columns and K = 9 x 64, and the output is the diagonal sum
//     out[y][x][co] = sum_G sum_d T_G[x + kwbase_G + d][(d, co)].
// 108 MFMAs per 32 output pixels instead of 324.  T is written to LDS (aliasing a consumed halo stage) with a 17-float row
// pitch and summed by the wave that produced it.
// Round 3 form (the one-row-per-wave, register-staged form before it ran at 0.14 of the MFMA peak, bound by the LDS port:
// 90 fragment reads per 108 MFMAs, and by its 2.5x halo):
//   * tile = 16 rows x 32 columns; the input channels are walked in two K-blocks of 32 (64 bytes per pixel: a wave's
//     fragment read covers 16 consecutive pixels = 1 KB contiguous, no swizzle, and a K-block is one mfma_f32_16x16x32), so
//     the 24 x 40 halo of one K-block is 60 KB and TWO stages fit beside the 36 KB of weights: the next K-block's (next
//     tile's) halo arrives by LDS-DMA under this one's MFMAs instead of through registers;
//   * a wave owns TWO output rows: the A fragments of halo row h serve row y0 (tap row h - y0) and row y0 + 1 (tap row
//     h - y0 - 1), and the B fragments of a tap row are read once and kept for the next halo row -- 30 + 18 fragment reads per
//     108 MFMAs (0.44 per MFMA instead of 0.83); halo overhead 1.9x instead of 2.5x.
template <int DT>
__global__ __launch_bounds__(512, 2) void conv_toeplitz9_kernel(const SmallNArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TR = 16, HR = TR + 8, HC = 40;
  constexpr int W_BYTES = 18 * 16 * 128;                 // [kh][group][16 rows][64 ci], chunks XOR-swizzled by row
  constexpr int HALO = HR * HC * 64;                     // one K-block of the halo: 61,440 B = 60 DMA pieces
  constexpr int NP = HALO / 1024;
  constexpr int T_PITCH = 17, T_WAVE = 2 * 48 * T_PITCH; // floats per wave: [group][48 p][17]
  static_assert(8 * T_WAVE * 4 <= HALO, "the T slabs of the 8 waves alias one halo stage");
  unsigned char* sW = smem;
  unsigned char* sX = smem + W_BYTES;                    // two halo stages
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int per_img = a.tiles_y * a.tiles_x;
  const unsigned short* __restrict__ Wg = reinterpret_cast<const unsigned short*>(a.w);

  // weights: LDS row (kh, G, n = 3d + co) <- packed slice [tap = 9kh + kwbase + d][row co]; unused rows are zero
  for (int i = tid; i < 18 * 16 * 8; i += 512) {
    const int ch = i & 7, row = (i >> 3) & 15, kg = i >> 7;
    const int kh = kg >> 1, G = kg & 1;
    const int d = row / 3, co = row - 3 * d;
    const int kw = 5 * G + d;
    const bool ok = row < 15 && kw < 9 && co < a.cout;
    U4 v = U4{0u, 0u, 0u, 0u};
    if (ok) v = reinterpret_cast<const U4*>(Wg)[((kh * 9 + kw) * a.NB + co) * 8 + ch];
    *reinterpret_cast<U4*>(sW + (kg * 16 + row) * 128 + ((ch ^ (row & 7)) << 4)) = v;
  }
  // B fragments (row r16, k-chunk 4 kb + g of K-block kb): ALL 36 of them stay in registers for the life of the block (144
  // VGPRs) -- every wave needs the same 18 per K-block, and read from LDS they were 18 of a wave's 48 fragment reads per 108 MFMAs
  __syncthreads();
  U4 fw[2][9][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int kh = 0; kh < 9; ++kh)
#pragma unroll
      for (int G = 0; G < 2; ++G)
        fw[kb][kh][G] = *reinterpret_cast<const U4*>(sW + (kh * 2 + G) * 16 * 128 + r16 * 128 + (((4 * kb + g) ^ (r16 & 7)) << 4));
  int a_off[3];                                          // A fragment: input column p = 16f + r16 (clamped to the halo row), chunk g
#pragma unroll
  for (int f = 0; f < 3; ++f) a_off[f] = (16 * f + r16 < HC ? 16 * f + r16 : HC - 1) * 64 + 16 * g;

  // ---- loader: wave w issues the pieces w, w + 8, ... (< 60); lane l of piece p fills slot 16 p + l / 4 (= halo pixel), chunk l % 4
  // (the slot -> (halo row, column) decode is redone per piece: a dozen integer instructions, against 10 registers that the
  //  register-resident weights need)
  constexpr int NU = (NP + 7) / 8;                       // 8
  const int slot0 = 16 * wave + (lane >> 2);
  const int chunk_b = (lane & 3) * 16;
  const unsigned img_bytes = (unsigned)(a.IH * a.IW * 128);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  struct TileXY {
    int n, ty, tx;
  };
  // (tile order: image, tile row, column (fastest), blocks in launch order.  An XCD-aware order with the tile row fastest -- the
  //  32 blocks of an XCD on one column strip, halo overlap served by that XCD's L2 -- measured 7 % SLOWER: 0.617 vs 0.579 ms)
  auto decomp = [&](int t) {
    TileXY c;
    c.n = t / per_img;
    const int rem = t - c.n * per_img;
    c.ty = rem / a.tiles_x;
    c.tx = rem - c.ty * a.tiles_x;
    return c;
  };
  auto fetch = [&](const TileXY& tc, int kb, int buf) {
    const int oy0 = tc.ty * TR - 4, ox0 = tc.tx * 32 - 4;
    // per-image resource: halo rows above / below the image fall outside it and read as zeros
    const BufSrd xsrd = make_srd(reinterpret_cast<const unsigned char*>(a.x) + (size_t)tc.n * img_bytes, img_bytes);
    const int sbase = (oy0 * a.IW + ox0) * 128 + kb * 64;
    const bool interior = ox0 >= 0 && ox0 + HC <= a.IW;
    unsigned char* dst = sX + buf * HALO;
    int s0 = slot0;
    asm volatile("" : "+v"(s0));                         // (opaque: keeps the decode below from being hoisted out of the tile loop)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (wave + 8 * u >= NP) continue;                  // (wave-uniform; only the last round is partial)
      const int slot = s0 + 128 * u;
      const int hr = slot / HC, hc = slot - hr * HC;
      unsigned off = (unsigned)((hr * a.IW + hc) * 128 + chunk_b + sbase);
      if (!interior && (unsigned)(ox0 + hc) >= (unsigned)a.IW) off = OOB;
      lds_dma16(xsrd, dst + (wave + 8 * u) * 1024, off);
    }
  };
  const float slope = a.prelu ? a.prelu[0] : a.slope;

  const int tstep = gridDim.x;
  int t = blockIdx.x;
  if (t >= a.ntiles) return;
  TileXY cur = decomp(t);
  fetch(cur, 0, 0);
  int buf = 0;
  const bool late = wave >= 4;                           // (one wave's DMA issue runs under its SIMD partner's MFMAs)
  for (; t < a.ntiles; t += tstep) {
    const bool has_next = t + tstep < a.ntiles;
    const TileXY nxt = has_next ? decomp(t + tstep) : cur;
    f32x4 acc[2][3][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int f = 0; f < 3; ++f) acc[r][f][0] = acc[r][f][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb, buf ^= 1) {
      // my DMA of this K-block has landed (first step: and my part of sW is written); after the barrier everyone's has, and
      // every wave is done with the other stage (its T reads of the previous tile included).  Raw barrier: the DMA is issued
      // by inline asm, the waits are explicit.
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      auto prefetch = [&]() {
        if (kb == 0)
          fetch(cur, 1, buf ^ 1);
        else if (has_next)
          fetch(nxt, 0, buf ^ 1);
      };
      if (!late) prefetch();
      const unsigned char* rowp = sX + buf * HALO + (2 * wave) * HC * 64;
#pragma unroll
      for (int hh = 0; hh < 10; ++hh) {
        U4 fa[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) fa[f] = *reinterpret_cast<const U4*>(rowp + hh * HC * 64 + a_off[f]);
        if (hh < 9) {                                    // row y0: tap row hh
#pragma unroll
          for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int G = 0; G < 2; ++G) acc[0][f][G] = mfma16<DT>(fa[f], fw[kb][hh][G], acc[0][f][G]);
        }
        if (hh >= 1) {                                   // row y0 + 1: tap row hh - 1
#pragma unroll
          for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int G = 0; G < 2; ++G) acc[1][f][G] = mfma16<DT>(fa[f], fw[kb][hh - 1][G], acc[1][f][G]);
        }
        if (hh == 2 && late) prefetch();
      }
    }
    // ---- every wave is done with the stage of K-block 1 (buf ^ 1 after the loop's flip): it becomes T; the DMA in flight
    // targets the other stage
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* sT = reinterpret_cast<float*>(sX + (buf ^ 1) * HALO) + wave * T_WAVE;
    const int ox0 = cur.tx * 32;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      // lane (g, r16) holds T_G[p = 16f + 4g + j][n = r16]
#pragma unroll
      for (int G = 0; G < 2; ++G)
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) sT[(G * 48 + 16 * f + 4 * g + j) * T_PITCH + r16] = acc[r][f][G][j];
      // (the wave reads back only what it wrote itself: LDS operations of one wave complete in order)
      const int oy = cur.ty * TR + 2 * wave + r;
      if (lane < 32 && oy < a.OH && ox0 + lane < a.OW) {
        const int x = lane;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          if (co < a.cout) {
            float s = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int d = 0; d < 5; ++d) s += sT[(x + d) * T_PITCH + 3 * d + co];
#pragma unroll
            for (int d = 0; d < 4; ++d) s += sT[(48 + x + 5 + d) * T_PITCH + 3 * d + co];
            const float o = act_apply(a.act, s, slope);
            if (a.out_f32)
              a.out_f32[(((size_t)cur.n * a.cout + co) * a.OH + oy) * a.OW + ox0 + x] = o;
            else
              reinterpret_cast<unsigned short*>(a.y)[((size_t)(cur.n * a.OH + oy) * a.OW + ox0 + x) * a.CoutP + co] = f2h<DT>(o);
          }
        }
      }
    }
    cur = nxt;
  }
}

// zero the pad channels [cout, CoutP) of a 16-bit NHWC output written column-wise by conv_smalln_kernel
template <int DT>
__global__ void zero_pad_channels_kernel(unsigned short* __restrict__ y, size_t P, int cout, int CoutP) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  for (int c = cout; c < CoutP; ++c) y[p * CoutP + c] = 0;
}

static LdsOptIn g_smalln_optin[4];

int dsr_launch_conv_smalln(SmallNArgs& a, int N, int dtype, hipStream_t st) {
  a.tiles_y = (a.OH + 7) / 8;
  a.tiles_x = (a.OW + 31) / 32;
  const int HR = 8 + a.KH - 1, HC = (32 + a.KW - 1 + 7) & ~7;
  const size_t lds = (size_t)HR * HC * 128 + (size_t)a.KH * a.KW * ((a.cout + 3) & ~3) * 128;
  if (lds > 160 * 1024 || (a.KW != 9 && a.KW != 3) || a.KH > 9 || a.CinP != 64) return 0;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.IH * a.IW * 128);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  dim3 grid(a.ntiles < 256 * per_cu ? a.ntiles : 256 * per_cu), block(512);   // persistent 8-wave blocks
  const int di = (dtype == DSR_DTYPE_BF16 ? 0 : 1) + (a.KW == 9 ? 0 : 2);
  const void* fn = di == 0   ? (const void*)conv_smalln_kernel<DSR_DTYPE_BF16, 9>
                   : di == 1 ? (const void*)conv_smalln_kernel<DSR_DTYPE_F16, 9>
                   : di == 2 ? (const void*)conv_smalln_kernel<DSR_DTYPE_BF16, 3>
                             : (const void*)conv_smalln_kernel<DSR_DTYPE_F16, 3>;
  g_smalln_optin[di].ensure(fn, 160 * 1024);   // > 64 KB of dynamic LDS needs the opt-in once per kernel and device (not a stream operation)
  const size_t P = (size_t)N * a.OH * a.OW;
  const bool zpad = !a.out_f32 && a.cout < a.CoutP;
  if (a.KW == 9 && a.KH == 9 && a.cout <= 3 && a.pad == 4 && !a.flip) {   // the generator's tail: Toeplitz mapping
    static LdsOptIn toeplitz_optin[2];
    const int dj = dtype == DSR_DTYPE_BF16 ? 0 : 1;
    const size_t tl = 18 * 16 * 128 + 2 * 24 * 40 * 64;      // weights + two one-K-block halo stages
    const void* tf = dj == 0 ? (const void*)conv_toeplitz9_kernel<DSR_DTYPE_BF16> : (const void*)conv_toeplitz9_kernel<DSR_DTYPE_F16>;
    toeplitz_optin[dj].ensure(tf, 160 * 1024);
    a.tiles_y = (a.OH + 15) / 16;                              // (this kernel's tiles are 16 rows x 32 columns)
    a.ntiles = N * a.tiles_y * a.tiles_x;
    dim3 tg(a.ntiles < 256 ? a.ntiles : 256);
    if (dj == 0) {
      hipLaunchKernelGGL((conv_toeplitz9_kernel<DSR_DTYPE_BF16>), tg, block, tl, st, a);
      if (zpad)
        hipLaunchKernelGGL((zero_pad_channels_kernel<DSR_DTYPE_BF16>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                           (unsigned short*)a.y, P, a.cout, a.CoutP);
    } else {
      hipLaunchKernelGGL((conv_toeplitz9_kernel<DSR_DTYPE_F16>), tg, block, tl, st, a);
      if (zpad)
        hipLaunchKernelGGL((zero_pad_channels_kernel<DSR_DTYPE_F16>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                           (unsigned short*)a.y, P, a.cout, a.CoutP);
    }
    return 1;
  }
  if (dtype == DSR_DTYPE_BF16) {
    if (a.KW == 9)
      hipLaunchKernelGGL((conv_smalln_kernel<DSR_DTYPE_BF16, 9>), grid, block, lds, st, a);
    else
      hipLaunchKernelGGL((conv_smalln_kernel<DSR_DTYPE_BF16, 3>), grid, block, lds, st, a);
    if (zpad)
      hipLaunchKernelGGL((zero_pad_channels_kernel<DSR_DTYPE_BF16>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                         (unsigned short*)a.y, P, a.cout, a.CoutP);
  } else {
    if (a.KW == 9)
      hipLaunchKernelGGL((conv_smalln_kernel<DSR_DTYPE_F16, 9>), grid, block, lds, st, a);
    else
      hipLaunchKernelGGL((conv_smalln_kernel<DSR_DTYPE_F16, 3>), grid, block, lds, st, a);
    if (zpad)
      hipLaunchKernelGGL((zero_pad_channels_kernel<DSR_DTYPE_F16>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                         (unsigned short*)a.y, P, a.cout, a.CoutP);
  }
  return 1;
}

// ------------------------------------------------------------------ weight gradient, wide tap rows, few channels
__device__ __forceinline__ s16x4 tr_read_sn(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ U4 tr_frag_sn(const unsigned char* base, int p1, int p2, int chunk, int within) {
  s16x4 lo = tr_read_sn(base + p1 * 128 + ((chunk ^ (p1 & 7)) << 4) + within);
  s16x4 hi = tr_read_sn(base + p2 * 128 + ((chunk ^ (p2 & 7)) << 4) + within);
  return __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// KW taps per row, KHB tap rows per block (blockIdx.y = tap-row group), 2 x 32 output pixels per tile, stride 1.
// CT co-tiles x IT ci-tiles of 16 per wave; the 4 waves own taps w, w+4, w+8, ... of the block's KHB*KW taps.
template <int DT, int KW, int KHB, int CT, int IT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_taps_kernel(const WgradTileArgs a, int KH) {
  constexpr int R = 2;
  constexpr int HR = R - 1 + KHB, HC = 31 + KW;
  static_assert((HR * HC) % 32 == 0 && HC % 8 == 0, "halo slots must fill whole DMA instructions");
  constexpr int NTB = KHB * KW;
  constexpr int NW = (NTB + 3) / 4;
  constexpr int XB = HR * HC * 128, YB = R * 32 * 128, STAGE = XB + YB;
  // one LDS object, two stages: the next tile's halo and dY tile arrive by LDS-DMA under this tile's MFMAs
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const int kh0 = blockIdx.y * KHB;

  f32x4 acc[NW][CT][IT];
#pragma unroll
  for (int t = 0; t < NW; ++t)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < IT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA role: slot position (lane & 7) of pixel slot q = 32u + 8*wave + (lane >> 3) holds chunk (lane & 7) ^ (q & 7)
  const int c = (tid & 7) ^ ((tid >> 3) & 7), pbase = tid >> 3;
  const bool yc_ok = (c * 8) < a.CoutP && c * 8 < CT * 16, xc_ok = (c * 8) < a.CinP && c * 8 < IT * 16;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int t_end = (blockIdx.z + 1) * a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  const int per_img = a.tiles_y * a.tiles_x;
  auto dma = [&](int t, int buf) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * R, ox0 = (rem % a.tiles_x) * 32;
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int u = 0; u < (HR * HC) / 32; ++u) {
      const int q = pbase + 32 * u;
      const int hr = q / HC, hc = q - hr * HC;
      const int iy = oy0 + hr + kh0 - a.pad, ix = ox0 + hc - a.pad;
      const bool ok = xc_ok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(st + (32 * u + 8 * wave) * 128), 16,
                                               ok ? (unsigned)((((n * a.IH + iy) * a.IW + ix) * a.CinP + c * 8) * 2) : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < (R * 32) / 32; ++u) {
      const int p = pbase + 32 * u;
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
      const bool ok = yc_ok && oy < a.OH && ox < a.OW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, (lds_ptr)(st + XB + (32 * u + 8 * wave) * 128), 16,
                                               ok ? (unsigned)((((n * a.OH + oy) * a.OW + ox) * a.CoutP + c * 8) * 2) : OOB, 0, 0, 0);
    }
  };

  int t = blockIdx.z * a.tiles_per_block;
  if (t < t_end) dma(t, 0);
  int buf = 0;
  for (; t < t_end; ++t, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // this tile landed; nobody still reads the other stage
    asm volatile("" ::: "memory");
    if (t + 1 < t_end) dma(t + 1, buf ^ 1);
    const unsigned char* sX = smem + buf * STAGE;
    const unsigned char* sY = sX + XB;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      U4 fa[CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int ch = i * 16 + cc;
        fa[i] = tr_frag_sn(sY, r * 32 + 4 * g + q4, r * 32 + 16 + 4 * g + q4, ch >> 3, (ch & 7) * 2);
      }
#pragma unroll
      for (int ti = 0; ti < NW; ++ti) {
        const int tt = wave + 4 * ti;          // wave-uniform
        if (tt < NTB) {
          const int khl = tt / KW, kw = tt - khl * KW;
          const int qb = (r + khl) * HC + kw;
          const int q1 = qb + 4 * g + q4, q2 = q1 + 16;
#pragma unroll
          for (int j = 0; j < IT; ++j) {
            const int ch = j * 16 + cc;
            U4 fb = tr_frag_sn(sX, q1, q2, ch >> 3, (ch & 7) * 2);
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[ti][i][j] = mfma16<DT>(fa[i], fb, acc[ti][i][j]);
          }
        }
      }
    }
  }
  const int ntaps = KH * KW;
  float* P = a.partial + (size_t)blockIdx.z * ntaps * a.CoutP * a.CinP;
#pragma unroll
  for (int ti = 0; ti < NW; ++ti) {
    const int tt = wave + 4 * ti;
    const int khl = tt / KW, kw = tt - khl * KW;
    const int kh = kh0 + khl;
    if (tt < NTB && kh < KH) {
      const int tap = kh * KW + kw;
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = i * 16 + 4 * g + r, ci = j * 16 + l16;
            if (co < a.CoutP && ci < a.CinP) P[((size_t)tap * a.CoutP + co) * a.CinP + ci] = acc[ti][i][j][r];
          }
    }
  }
}

// returns the number of partial slabs, 0 if the shape is not handled (9x9, stride 1, one side <= 16 channels,
// the other <= 64)
// ------------------------------------------------------------------ 9x9 weight gradient, <= 3 output channels: Toeplitz rows
// dW[kh][kw][co][ci] = sum_{y,x} dy[y][x][co] * in[y+kh-4][x+kw-4][ci].  With p = x + kw - 4 (input column) this is, for
// every pair (input row r, tap row kh), one GEMM over the pixels p of the row:
//     D_kh[(kw,co)][ci] += A_y[(kw,co)][p] * B_r[p][ci],   y = r - kh + 4,
//     A_y[(kw,co)][p] = dy[y][p - kw + 4][co]   (27 rows: the dy row shifted once per kw),   B_r[p][ci] = in[r][p][ci].
// All 9 kw share one MFMA row block instead of one MFMA each, and an input row is staged ONCE for its 9 tap rows
// (the tap-per-MFMA kernel above re-stages it 7x and uses 3 of 16 MFMA rows).
// A block walks the input rows of one (image, 64-column strip, row band): per row it DMAs B_r and the raw dy strip
// of row r+6, builds A_{r+5} in LDS (a ring of 12 images), and its 3 waves (3 tap rows each) issue 48 MFMAs.
// The pixel order inside an MFMA k-step is the permutation of the transposing read used for B
// ({4g..4g+3} U {16+4g..16+4g+3} for lane group g); A is built in that same order.
template <int DT>
__global__ __launch_bounds__(192, 2) void conv_wgrad_toeplitz9_kernel(const WgradTileArgs a) {
  constexpr int SLOTS = 12, A_IMG = 32 * 128, B_IMG = 64 * 128, D_RAW = 128 * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SLOTS * A_IMG + 2 * B_IMG + 2 * D_RAW];
  unsigned char* sA = smem;
  unsigned char* sB = smem + SLOTS * A_IMG;
  unsigned char* sD = sB + 2 * B_IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  // work item: image n, strip (64 input columns from p0), band of input rows [rb0, rb1)
  const int strips = a.tiles_x, bands = a.tiles_y, rows_per_band = a.tiles_per_block;
  const int item = blockIdx.x;
  const int n = item / (strips * bands);
  const int rem = item - n * strips * bands;
  const int band = rem / strips, strip = rem - band * strips;
  const int p0 = strip * 64;
  const int rb0 = band * rows_per_band;
  const int rb1 = rb0 + rows_per_band < a.IH ? rb0 + rows_per_band : a.IH;
  const int H = a.IH, W = a.IW;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;

  f32x4 acc[3][2][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[k][mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA of input row r (64 pixels x 8 chunks = 8 wave-instructions: waves 0,1 issue 3, wave 2 issues 2) and of the raw
  // dy strip of row yd (72 pixels x 16 B: one full and one mostly-masked wave-instruction, waves 0 and 1)
  auto dma_rows = [&](int r, int yd) {
    unsigned char* dstB = sB + (r & 1) * B_IMG;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int inst = wave + 3 * u;                     // wave-uniform
      if (inst < 8) {
        const int q = 8 * inst + (lane >> 3);            // pixel slot
        const int ch = (lane & 7) ^ (q & 7);
        const int ix = p0 + q;
        const bool ok = (unsigned)r < (unsigned)H && ix < W;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(dstB + inst * 1024), 16,
                                                 ok ? (unsigned)((((n * H + r) * W + ix) * 64 + ch * 8) * 2) : OOB, 0, 0, 0);
      }
    }
    if (wave < 2) {
      const int q = 64 * wave + lane;                    // raw slot: dy column p0 - 4 + q
      const int x = p0 - 4 + q;
      const bool ok = q < 72 && (unsigned)yd < (unsigned)H && (unsigned)x < (unsigned)W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, (lds_ptr)(sD + (yd & 1) * D_RAW + wave * 1024), 16,
                                               ok ? (unsigned)((((n * H + yd) * W + x) * 8) * 2) : OOB, 0, 0, 0);
    }
  };
  // A_y from the raw strip: row m = 3kw + co, chunk pc = 4ks + gg holds pixels 32ks + {4gg..4gg+3, 16+4gg..16+4gg+3}
  auto build_A = [&](int y) {
    const unsigned char* raw = sD + (y & 1) * D_RAW;
    unsigned char* img = sA + (((y % SLOTS) + SLOTS) % SLOTS) * A_IMG;
    for (int idx = tid; idx < 256; idx += 192) {
      const int m = idx >> 3, pc = idx & 7;
      const int kw = m / 3, co = m - 3 * kw;
      unsigned short e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = 32 * (pc >> 2) + (i < 4 ? 4 * (pc & 3) + i : 16 + 4 * (pc & 3) + (i - 4));
        const int q = p - kw + 8;                        // raw slot of dy column p0 + p - kw + 4
        e[i] = (m < 27) ? *reinterpret_cast<const unsigned short*>(raw + q * 16 + co * 2) : (unsigned short)0;
      }
      U4 v;
      v.x = e[0] | ((unsigned)e[1] << 16);
      v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16);
      v.w = e[6] | ((unsigned)e[7] << 16);
      *reinterpret_cast<U4*>(img + m * 128 + ((pc ^ (m & 7)) << 4)) = v;
    }
  };

  // rows rb0-9 .. rb0-1 only fill the ring (A_{rb0-4} .. A_{rb0+4}); from rb0 on every row also computes
  const int r_start = rb0 - 9;
  dma_rows(r_start, r_start + 5);
  for (int r = r_start; r < rb1; ++r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // B_r and raw dy(r+5) landed; A_{r+4} (built last step) is visible
    asm volatile("" ::: "memory");
    if (r + 1 < rb1) dma_rows(r + 1, r + 6);
    build_A(r + 5);
    if (r >= rb0) {
      const unsigned char* bimg = sB + (r & 1) * B_IMG;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        U4 fb[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const int ch = nf * 16 + cc;
          fb[nf] = tr_frag_sn(bimg, 32 * ks + 4 * g + q4, 32 * ks + 16 + 4 * g + q4, ch >> 3, (ch & 7) * 2);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int y = r - (3 * wave + k) + 4;
          if ((unsigned)y < (unsigned)H) {               // wave-uniform
            const unsigned char* img = sA + (y % SLOTS) * A_IMG;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
              const int m = 16 * mf + l16;
              const U4 fa = *reinterpret_cast<const U4*>(img + m * 128 + (((4 * ks + g) ^ (m & 7)) << 4));
#pragma unroll
              for (int nf = 0; nf < 4; ++nf) acc[k][mf][nf] = mfma16<DT>(fa, fb[nf], acc[k][mf][nf]);
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's LDS writes (A_{r+5}) are done before the barrier above
  }

  // D[m = 3kw + co][n = ci] -> partial[item][tap = 9kh + kw][co][ci]
  float* P = a.partial + (size_t)item * 81 * 8 * 64;
  for (int i = tid; i < 81 * 8 * 64 / 4; i += 192) reinterpret_cast<float4*>(P)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int kh = 3 * wave + k;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 16 * mf + 4 * g + j;
        if (m < 27) {
          const int kw = m / 3, co = m - 3 * kw;
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            P[((size_t)(kh * 9 + kw) * 8 + co) * 64 + 16 * nf + l16] = acc[k][mf][nf][j];
        }
      }
  }
}

// plan: items = N x strips x bands (each writes one partial slab); returns the number of slabs
int dsr_wgrad_toeplitz_plan(int N, int H, int W, WgradTileArgs* a) {
  const int strips = (W + 63) / 64;
  long long per = (long long)N * strips;
  int bands = (int)(512 / per);
  if (bands < 1) bands = 1;
  const int maxb = (H + 31) / 32;
  if (bands > maxb) bands = maxb;
  a->tiles_x = strips;
  a->tiles_y = bands;
  a->tiles_per_block = (H + bands - 1) / bands;          // input rows per band
  a->tiles_y = (H + a->tiles_per_block - 1) / a->tiles_per_block;
  a->ntiles = N * strips * a->tiles_y;
  a->tiles_co = a->tiles_ci = 1;
  return a->ntiles;
}

// ------------------------------------------------------------------ 9x9 input gradient, <= 3 output channels: Toeplitz rows
// dIn[r][p][ci] = sum_{kh,kw,co} dy[r-kh+4][p-kw+4][co] * W[kh][kw][co][ci]: with the same shifted image as above, stored
// pixel-major ( A_y[p][k = 3kw + co] ), every (row r, tap row kh) is one MFMA K-step of 32 over k = (kw, co) -- 9 K-steps per
// output row instead of the 21 of the tap-per-8-channels gather kernel (dy has 3 of 8 channels populated).
// Weights (the [tap][ci][co] image) stay in registers for the life of the block as A' operands (m = ci), so the
// accumulator holds 4 consecutive ci of one pixel and the 128-byte NHWC rows leave through a small per-wave LDS slab.
struct Tail9DgradArgs {
  const void* dy;   // [N][H][W][8]
  const void* w;    // dgrad weight image [81][64][8]
  void* dx;         // [N][H][W][64]
  int N, H, W;
  int strips, bands, rows_per_band;
  unsigned dy_bytes, dx_bytes;
  // PS (dsr_conv_dgrad_ps): the input x of this convolution is PReLU(PixelShuffle(2)(conv)) (generator.py:37-39 in front of :78)
  const void* act_out;   // x itself [N][H][W][64]: the activation OUTPUT whose sign selects the PReLU branch
  const float* prelu;    // the PReLU weight (one value)
  void* dyu;             // [N][H/2][W/2][256]: gradient of the shuffle conv's output, channel 4c + 2i + j <- pixel (2h+i, 2w+j)
  float* ps_partial;     // [blocks][2][256]: column sums of dyu (that conv's bias gradient) | PReLU weight gradient terms
};

// PS = true: the backward of the activation in front of this convolution rides in the epilogue.  dx = the gradient of
// PReLU(PixelShuffle(conv)) is never written: the accumulators are multiplied by the PReLU derivative (sign of the activation
// output, whose 64-pixel row strip arrives by LDS-DMA one row ahead), the PReLU-weight gradient terms d * o / slope are summed
// per thread, two consecutive rows of masked gradient are kept in the wave's slab and leave UN-SHUFFLED -- dyu[h][w][4c + 2i + j]
// from pixel (2h + i, 2w + j), sixteen-byte vectors of (2 channels x 4 sub-pixels) assembled from four 4-byte slab reads --
// with their column sums (the shuffle conv's bias gradient) accumulated on the way out.  That is all of act_bwd_kernel's
// PixelShuffle branch, which otherwise reads dx and the activation and writes dyu: three passes over a 1.07 GB tensor at
// config 3.  Bands start and end on even rows.
template <int DT, bool PS>
__global__ __launch_bounds__(256, 2) void conv_dgrad_toeplitz9_kernel(const Tail9DgradArgs a) {
  constexpr int SLOTS = PS ? 10 : 12, A_IMG = 64 * 64, D_RAW = 128 * 16, C_SLAB = 16 * 128;
  constexpr int NSLAB = PS ? 2 : 1, O_STRIP = 64 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SLOTS * A_IMG + 2 * D_RAW + 4 * NSLAB * C_SLAB + (PS ? 2 * O_STRIP : 0)];
  unsigned char* sA = smem;
  unsigned char* sD = smem + SLOTS * A_IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  unsigned char* sC = sD + 2 * D_RAW + wave * NSLAB * C_SLAB;
  [[maybe_unused]] unsigned char* sO = sD + 2 * D_RAW + 4 * NSLAB * C_SLAB;
  const int item = blockIdx.x;
  const int n = item / (a.strips * a.bands);
  const int rem = item - n * a.strips * a.bands;
  const int band = rem / a.strips, strip = rem - band * a.strips;
  const int p0 = strip * 64;
  const int rb0 = band * a.rows_per_band;
  const int rb1 = rb0 + a.rows_per_band < a.H ? rb0 + a.rows_per_band : a.H;
  const int H = a.H, W = a.W;
  const unsigned short* __restrict__ Wg = reinterpret_cast<const unsigned short*>(a.w);

  // ---- weights -> registers: fw[kh][mf] = A'[m = ci 16mf + l16][k = 8g..8g+7], k = 3kw + co (k >= 27: zero)
  U4 fw[9][4];
#pragma unroll
  for (int kh = 0; kh < 9; ++kh) {
    __syncthreads();
    for (int i = tid; i < 9 * 64; i += 256)              // slice [9 kw][64 ci][8 co] of this tap row -> LDS scratch (the ring area)
      reinterpret_cast<U4*>(sA)[i] = reinterpret_cast<const U4*>(Wg)[kh * 9 * 64 + i];
    __syncthreads();
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      unsigned short e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * g + i;
        const int kw = k / 3, co = k - 3 * kw;
        e[i] = k < 27 ? *reinterpret_cast<const unsigned short*>(sA + ((kw * 64 + 16 * mf + l16) * 8 + co) * 2) : (unsigned short)0;
      }
      fw[kh][mf].x = e[0] | ((unsigned)e[1] << 16);
      fw[kh][mf].y = e[2] | ((unsigned)e[3] << 16);
      fw[kh][mf].z = e[4] | ((unsigned)e[5] << 16);
      fw[kh][mf].w = e[6] | ((unsigned)e[7] << 16);
    }
  }
  __syncthreads();

  // (LDS-DMA by inline asm, dsr_common.h: the compiler's wait-count pass then knows nothing of it and every wait below is explicit)
  const BufSrd yrsrc = make_srd(a.dy, a.dy_bytes);
  [[maybe_unused]] const BufSrd orsrc = make_srd(a.act_out, a.dx_bytes);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  auto dma_raw = [&](int yd) {                           // raw dy strip of row yd: columns p0-4 .. p0+67 (waves 0 and 1)
    if (wave < 2) {
      const int q = 64 * wave + lane;
      const int x = p0 - 4 + q;
      const bool ok = q < 72 && (unsigned)yd < (unsigned)H && (unsigned)x < (unsigned)W;
      lds_dma16(yrsrc, sD + (yd & 1) * D_RAW + wave * 1024, ok ? (unsigned)((((n * H + yd) * W + x) * 8) * 2) : OOB);
    }
  };
  // PS: the activation-output strip of row yo (64 pixels x 128 B): a wave fetches its own 16 pixels (two pieces of 8 pixels x 8
  // chunks); slot (pixel, c) receives channel chunk c ^ (pixel & 7), so that the 8-byte reads below spread over the banks
  [[maybe_unused]] auto dma_out = [&](int yo) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int px = 16 * wave + 8 * u + (lane >> 3), c = lane & 7;
      const int x = p0 + px;
      const bool ok = yo < H && x < W;
      lds_dma16(orsrc, sO + (yo & 1) * O_STRIP + (16 * wave + 8 * u) * 128,
                ok ? (unsigned)((((n * H + yo) * W + x) * 64 + ((c ^ (px & 7)) << 3)) * 2) : OOB);
    }
  };
  auto build_A = [&](int y) {                            // A_y[p][k] = dy[y][p0 + p - kw + 4][co]; one 16-byte chunk per thread
    const unsigned char* raw = sD + (y & 1) * D_RAW;
    unsigned char* img = sA + (((y % SLOTS) + SLOTS) % SLOTS) * A_IMG;
    const int p = tid >> 2, c = tid & 3;
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * c + i;
      const int kw = k / 3, co = k - 3 * kw;
      e[i] = k < 27 ? *reinterpret_cast<const unsigned short*>(raw + (p - kw + 8) * 16 + co * 2) : (unsigned short)0;
    }
    U4 v;
    v.x = e[0] | ((unsigned)e[1] << 16);
    v.y = e[2] | ((unsigned)e[3] << 16);
    v.z = e[4] | ((unsigned)e[5] << 16);
    v.w = e[6] | ((unsigned)e[7] << 16);
    *reinterpret_cast<U4*>(img + p * 64 + ((c ^ ((p >> 2) & 3)) << 4)) = v;
  };

  const int px = 16 * wave + l16;                        // this lane's pixel (B' operand column) inside the strip
  const int b_off = px * 64 + ((g ^ ((px >> 2) & 3)) << 4);
  const int r_start = rb0 - 9;
  // output rows leave by range-checked buffer stores: exactly two per thread and row (PS: four per thread and row PAIR), so
  // that "at most that many outstanding" at the top of the next row means the strips requested BEFORE them have landed (vmcnt
  // retires in order) while the stores themselves stay in flight -- a vmcnt(0) there exposed the full store latency per row
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(PS ? a.dyu : a.dx, 0, a.dx_bytes, 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const unsigned sC_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)sC);
  [[maybe_unused]] float sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float sp = 0.f;
  [[maybe_unused]] float slope = 0.f, inv_slope = 0.f, poison = 0.f;
  if constexpr (PS) {
    slope = a.prelu[0];
    inv_slope = 1.f / slope;
    poison = !(slope > 0.f) ? __uint_as_float(0x7fc00000u) : 0.f;      // (see act_bwd_kernel: a slope that crossed zero poisons the launch)
  }
  dma_raw(r_start + 5);
  if constexpr (PS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler's own wait for `slope` would come later, behind DMAs it cannot see)
  for (int r = r_start; r < rb1; ++r) {
    if constexpr (PS) {
      if (r > rb0 && !(r & 1))                           // (uniform) the previous iteration (an odd row) issued four stores
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (r > rb0)                                       // (uniform) the previous iteration stored a row
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (r + 1 < rb1) dma_raw(r + 6);
    if constexpr (PS)
      if (r + 1 >= rb0 && r + 1 < rb1) dma_out(r + 1);
    build_A(r + 5);
    if (r >= rb0) {
      f32x4 acc[4];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (PS: as in conv_c64_kernel, everything outside the MFMA phase runs at priority 1 -- the other block's MFMA stream otherwise
      //  gets two issue slots for every one of this wave's mask / un-shuffle instructions)
      if constexpr (PS) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int kh = 0; kh < 9; ++kh) {
        const int y = r - kh + 4;
        if ((unsigned)y < (unsigned)H) {                 // block-uniform
          const U4 fb = *reinterpret_cast<const U4*>(sA + (y % SLOTS) * A_IMG + b_off);
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) acc[mf] = mfma16<DT>(fw[kh][mf], fb, acc[mf]);
        }
      }
      if constexpr (PS) __builtin_amdgcn_s_setprio(1);
      if constexpr (PS) {
        // acc[mf][j] = d(channel 16 mf + 4 g + j, pixel l16): g = d * PReLU'(o), PReLU-weight terms d * o / slope where o < 0
        const unsigned char* so = sO + (r & 1) * O_STRIP + px * 128 + (g & 1) * 8;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const uint2 ov = *reinterpret_cast<const uint2*>(so + (((2 * mf + (g >> 1)) ^ (px & 7)) << 4));
          const float o[4] = {h2f<DT>((unsigned short)(ov.x & 0xffff)), h2f<DT>((unsigned short)(ov.x >> 16)),
                              h2f<DT>((unsigned short)(ov.y & 0xffff)), h2f<DT>((unsigned short)(ov.y >> 16))};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = acc[mf][j];
            sp = __builtin_fmaf(d, __builtin_fminf(o[j], 0.f), sp);          // (times 1 / slope once, at the end)
            acc[mf][j] = o[j] >= 0.f ? d : d * slope;
          }
        }
        if (poison != 0.f) {                             // (uniform, never in a healthy run)
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mf][j] += poison;
        }
      }
      // acc[mf][j] = dIn[ci = 16mf + 4g + j][pixel l16 of this wave] -> slab -> 16-byte NHWC stores
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        uint2 h;
        h.x = (unsigned)f2h<DT>(acc[mf][0]) | ((unsigned)f2h<DT>(acc[mf][1]) << 16);
        h.y = (unsigned)f2h<DT>(acc[mf][2]) | ((unsigned)f2h<DT>(acc[mf][3]) << 16);
        const int c16 = 2 * mf + (g >> 1);
        // inline asm: behind an in-flight LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of an 8-byte LDS store it can see
        // (the raw strip requested at the top of this row would be waited for here, every row)
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const u32x2 hv = {h.x, h.y};
        asm volatile("ds_write_b64 %0, %1" ::"v"(sC_lds + (unsigned)((PS ? (r & 1) * C_SLAB : 0) + l16 * 128 + ((c16 ^ (l16 & 7)) << 4) + (g & 1) * 8)), "v"(hv) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the slab is this wave's own: no barrier)
      if constexpr (PS) {
        if (r & 1) {
          // rows r - 1 (i = 0) and r (i = 1) of this wave's 16 pixels -> 8 low-resolution pixels x 256 channels, un-shuffled:
          // vector q = lane + 64 v: pixel w' = q >> 5, channels 8u .. 8u+7 (u = q & 31) = shuffle channels 2u, 2u + 1 at the
          // four sub-pixels (element k: channel 2u + (k >> 2), i = (k >> 1) & 1, j = k & 1)
          int lo = lane;
          asm volatile("" : "+v"(lo));      // (opaque: the slab addresses below are re-derived per row pair -- hoisted out of the row
          const int u = lo & 31;            //  loop they were spilled to scratch, and every reload waited for the strips in flight)
          const int LH = H >> 1, LW = W >> 1;
          const int hh = (r - 1) >> 1;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int wq = (lo >> 5) + 2 * v;
            unsigned rd[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int pp = 2 * wq + j;
                rd[i][j] = *reinterpret_cast<const unsigned*>(sC + i * C_SLAB + pp * 128 + (((u >> 2) ^ (pp & 7)) << 4) + (u & 3) * 4);
              }
            u32x4 o4;
            o4.x = (rd[0][0] & 0xffffu) | (rd[0][1] << 16);
            o4.y = (rd[1][0] & 0xffffu) | (rd[1][1] << 16);
            o4.z = (rd[0][0] >> 16) | (rd[0][1] & 0xffff0000u);
            o4.w = (rd[1][0] >> 16) | (rd[1][1] & 0xffff0000u);
            const int xw = (p0 >> 1) + 8 * wave + wq;
            float f[8];
            unpack8<DT>(__builtin_bit_cast(U4, o4), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) sb[k] += xw < LW ? f[k] : 0.f;     // (a strip may reach past the right edge of the image)
            const unsigned off = (unsigned)((((n * LH + hh) * LW + xw) * 256 + 8 * u) * 2);
            __builtin_amdgcn_raw_buffer_store_b128(o4, xrsrc, xw < LW ? off : OOB, 0, 0);
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = lane + 64 * i;
        const int pp = idx >> 3, c16 = idx & 7;
        const U4 v = *reinterpret_cast<const U4*>(sC + pp * 128 + ((c16 ^ (pp & 7)) << 4));
        const int x = p0 + 16 * wave + pp;
        const unsigned off = (unsigned)((((n * H + r) * W + x) * 64 + c16 * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrsrc, x < W ? off : OOB, 0, 0);
      }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if constexpr (PS) {
    // one partial row [2][256] per block: column sums of dyu over the 8 threads that share a vector column (lane & 31: the two
    // half-waves of four waves), and the block's sum of PReLU-weight terms in column 0 of the second slice (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(sA);           // [8][256] + [256]
    const int part = 2 * wave + (lane >> 5), u = lane & 31;
#pragma unroll
    for (int k = 0; k < 8; ++k) red[part * 256 + 8 * u + k] = sb[k];
    red[8 * 256 + tid] = sp * inv_slope + poison;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q * 256 + tid];
    a.ps_partial[((size_t)blockIdx.x * 2 + 0) * 256 + tid] = s;
    float t = 0.f;
    if (tid == 0)
      for (int q = 0; q < 256; ++q) t += red[8 * 256 + q];
    a.ps_partial[((size_t)blockIdx.x * 2 + 1) * 256 + tid] = t;
  }
}

// plan of the 9x9 input gradient: returns the number of blocks (= partial rows of the PS form)
static int dgrad_toeplitz_plan(int N, int H, int W, bool ps, Tail9DgradArgs* a) {
  WgradTileArgs t;
  memset(&t, 0, sizeof(t));
  dsr_wgrad_toeplitz_plan(N, H, W, &t);
  a->strips = t.tiles_x;
  a->rows_per_band = t.tiles_per_block;
  if (ps && (a->rows_per_band & 1)) ++a->rows_per_band;        // bands start and end on even rows (H is even)
  a->bands = (H + a->rows_per_band - 1) / a->rows_per_band;
  return N * a->strips * a->bands;
}
int dsr_dgrad_toeplitz_ps_blocks(int N, int H, int W) {
  Tail9DgradArgs a;
  return dgrad_toeplitz_plan(N, H, W, true, &a);
}

void dsr_launch_dgrad_toeplitz(const void* dy, const void* w_dgrad, void* dx, int N, int H, int W, int dtype, hipStream_t st,
                               const void* act_out, const float* prelu, void* dyu, float* ps_partial) {
  Tail9DgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy;
  a.w = w_dgrad;
  a.dx = dx;
  a.N = N;
  a.H = H;
  a.W = W;
  const bool ps = act_out != nullptr;
  const int blocks = dgrad_toeplitz_plan(N, H, W, ps, &a);
  a.dy_bytes = (unsigned)((size_t)N * H * W * 16);
  a.dx_bytes = (unsigned)((size_t)N * H * W * 128);      // (check_desc refuses tensors of 2 GiB or more; dyu is the same size)
  a.act_out = act_out;
  a.prelu = prelu;
  a.dyu = dyu;
  a.ps_partial = ps_partial;
  if (ps) {
    static LdsOptIn optin[2];   // 76 KB of static LDS: above the 64 KB default
    if (dtype == DSR_DTYPE_BF16)
      hipLaunchKernelGGL((conv_dgrad_toeplitz9_kernel<DSR_DTYPE_BF16, true>), dim3(blocks), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv_dgrad_toeplitz9_kernel<DSR_DTYPE_F16, true>), dim3(blocks), dim3(256), 0, st, a);
    return;
  }
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_dgrad_toeplitz9_kernel<DSR_DTYPE_BF16, false>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_dgrad_toeplitz9_kernel<DSR_DTYPE_F16, false>), dim3(blocks), dim3(256), 0, st, a);
}

void dsr_launch_wgrad_toeplitz(const WgradTileArgs& a, int dtype, hipStream_t st) {
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_wgrad_toeplitz9_kernel<DSR_DTYPE_BF16>), dim3(a.ntiles), dim3(192), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_toeplitz9_kernel<DSR_DTYPE_F16>), dim3(a.ntiles), dim3(192), 0, st, a);
}

int dsr_wgrad_taps_plan(int KH, int KW, int stride, int N, int OH, int OW, int CinP, int CoutP, WgradTileArgs* a) {
  if (!(KH == 9 && KW == 9 && stride == 1)) return 0;
  const bool small_out = CoutP <= 16 && CinP <= 64, small_in = CinP <= 16 && CoutP <= 64;
  if (!small_out && !small_in) return 0;
  a->tiles_y = (OH + 1) / 2;
  a->tiles_x = (OW + 31) / 32;
  a->ntiles = N * a->tiles_y * a->tiles_x;
  a->tiles_co = a->tiles_ci = 1;
  long long want = 512 / 3;                     // 3 tap-row groups per spatial chunk
  if (want > a->ntiles) want = a->ntiles;
  a->tiles_per_block = (int)((a->ntiles + want - 1) / want);
  return (a->ntiles + a->tiles_per_block - 1) / a->tiles_per_block;
}

void dsr_launch_wgrad_taps(const WgradTileArgs& a, int KH, int ychunks, int dtype, hipStream_t st) {
  dim3 grid(1, 3, ychunks), block(256);
  const bool small_out = a.CoutP <= 16;
  if (dtype == DSR_DTYPE_BF16) {
    if (small_out)
      hipLaunchKernelGGL((conv_wgrad_taps_kernel<DSR_DTYPE_BF16, 9, 3, 1, 4>), grid, block, 0, st, a, KH);
    else
      hipLaunchKernelGGL((conv_wgrad_taps_kernel<DSR_DTYPE_BF16, 9, 3, 4, 1>), grid, block, 0, st, a, KH);
  } else {
    if (small_out)
      hipLaunchKernelGGL((conv_wgrad_taps_kernel<DSR_DTYPE_F16, 9, 3, 1, 4>), grid, block, 0, st, a, KH);
    else
      hipLaunchKernelGGL((conv_wgrad_taps_kernel<DSR_DTYPE_F16, 9, 3, 4, 1>), grid, block, 0, st, a, KH);
  }
}
