// Weight gradient, tile-resident form, for the convolutions that dominate the step (3x3 stride 1|2, 1x1):
//   dW[tap][co][ci] = sum_p dY[p][co] * X[p*s + tap][ci]
// A block owns one (64 co x 64 ci) pair and walks a list of spatial tiles (R output rows x 32 columns of one
// image).  Per tile it stages the dY tile and the X HALO tile ((R-1)s+KH rows x 31s+KW columns) once in LDS and
// derives ALL KH*KW taps from them -- a tap is just an address offset into the halo image -- so every staged byte
// feeds 9x more MFMAs than in the one-tap-per-block kernel (conv_wgrad.hip) and the operands are no longer
// re-fetched from L2 once per tap.  As there, pixels are the contraction index and both operands are read with
// the transposing ds_read_b64_tr_b16; the k order inside a fragment is the same permutation for A and B.
// 4 waves = 2 (co halves) x 2 (ci halves); each keeps taps x 2 x 2 accumulator tiles (144 VGPRs for 3x3).
#include "dsr_common.h"
#include "dsr_kernels.h"

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ U4 tr_frag(const unsigned char* base, int p1, int p2, int chunk, int within) {
  s16x4 lo = tr_read(base + p1 * 128 + ((chunk ^ (p1 & 7)) << 4) + within);
  s16x4 hi = tr_read(base + p2 * 128 + ((chunk ^ (p2 & 7)) << 4) + within);
  return __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int DT, int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile_kernel(const WgradTileArgs a) {
  constexpr int R = (S == 1) ? 4 : 2;
  constexpr int HR = (R - 1) * S + KH, HC = 31 * S + KW;
  constexpr int NT = KH * KW;
  __shared__ __attribute__((aligned(16))) unsigned char sX[HR * HC * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sY[R * 32 * 128];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it (and what derives from it) in SGPRs
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const int pair = blockIdx.x;
  const int co0 = (pair / a.tiles_ci) * 64, ci0 = (pair % a.tiles_ci) * 64;
  const unsigned short* __restrict__ X = reinterpret_cast<const unsigned short*>(a.x);
  const unsigned short* __restrict__ DY = reinterpret_cast<const unsigned short*>(a.dy);

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int c = tid & 7, pbase = tid >> 3;
  const bool yc_ok = (co0 + c * 8) < a.CoutP, xc_ok = (ci0 + c * 8) < a.CinP;
  int t_end = (blockIdx.y + 1) * a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  const int per_img = a.tiles_y * a.tiles_x;

  for (int t = blockIdx.y * a.tiles_per_block; t < t_end; ++t) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * R, ox0 = (rem % a.tiles_x) * 32;
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < R * 32 * 8 / 256; ++i) {
      const int p = pbase + 32 * i;
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
      const bool ok = yc_ok && oy < a.OH && ox < a.OW;
      U4 v = load16_or_zero(DY, ((size_t)(n * a.OH + oy) * a.OW + ox) * a.CoutP + co0 + c * 8, ok);
      *reinterpret_cast<U4*>(sY + p * 128 + ((c ^ (p & 7)) << 4)) = v;
    }
    {
      constexpr int NV = (HR * HC + 31) / 32;     // 7 (3x3 s1), 11 (3x3 s2), 4 (1x1): all loads first, then the writes
      U4 v[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int q = pbase + 32 * u;
        const int hr = q / HC, hc = q - hr * HC;
        bool ok = xc_ok && q < HR * HC;
        const int iy = pad_index(oy0 * S + hr - a.pad, a.IH, a.pad_mode, ok);
        const int ix = pad_index(ox0 * S + hc - a.pad, a.IW, a.pad_mode, ok);
        v[u] = load16_or_zero(X, ((size_t)(n * a.IH + iy) * a.IW + ix) * a.CinP + ci0 + c * 8, ok);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int q = pbase + 32 * u;
        if (q < HR * HC) *reinterpret_cast<U4*>(sX + q * 128 + ((c ^ (q & 7)) << 4)) = v[u];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < R; ++r) {   // not unrolled: 144 accumulator registers leave no room for hoisted addresses
      U4 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = wr * 32 + i * 16 + cc;
        fa[i] = tr_frag(sY, r * 32 + 4 * g + q4, r * 32 + 16 + 4 * g + q4, ch >> 3, (ch & 7) * 2);
      }
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int qb = (r * S + kh) * HC + kw;
          const int q1 = qb + (4 * g + q4) * S, q2 = qb + (16 + 4 * g + q4) * S;
          U4 fb[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ch = wc * 32 + j * 16 + cc;
            fb[j] = tr_frag(sX, q1, q2, ch >> 3, (ch & 7) * 2);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[kh * KW + kw][i][j] = mfma16<DT>(fa[i], fb[j], acc[kh * KW + kw][i][j]);
        }
    }
  }

  float* P = a.partial + (size_t)blockIdx.y * NT * a.CoutP * a.CinP;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 32 + i * 16 + 4 * g + r, ci = ci0 + wc * 32 + j * 16 + l16;
          if (co < a.CoutP && ci < a.CinP) P[((size_t)t * a.CoutP + co) * a.CinP + ci] = acc[t][i][j][r];
        }
}

// ------------------------------------------------------------------ 3x3 / stride 1 with an LDS-DMA operand pipeline
// Same decomposition as above (block = one 64 co x 64 ci pair, pixels are the contraction index, all 9 taps from one
// staged halo), rebuilt around `buffer_load ... lds`:
//   * tiles are 2 output rows x 32 columns; the dY tile (8 KB) and the X halo (4 x 40 pixels, 20 KB) of the NEXT
//     tile stream HBM -> LDS under the 72 MFMAs per wave of the current one (two stages, one barrier per tile, no
//     VGPR staging, no ds_write pass);
//   * the halo row pitch is 40 pixels (a multiple of 8), so the XOR swizzle key of a fragment read,
//     (pixel index & 7), depends on the lane and the tap COLUMN only: every read address is one per-lane register
//     plus an immediate -- the 9-tap loop carries no address arithmetic (the older kernel spent ~6 VALU
//     instructions per MFMA on it).
template <int DT>
__device__ __forceinline__ void conv_wgrad_dma_body(const WgradTileArgs& a, const int pair, const int ychunk) {
  constexpr int R = 2, HC = 40, HR = R + 2;
  constexpr int XB = HR * HC * 128, YB = R * 32 * 128, STAGE = XB + YB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it (and what derives from it) in SGPRs
  // A wave owns all 64 output channels (4 m-tiles) of a 16-input-channel strip (one n-tile): an X fragment -- the operand that
  // changes with every tap column and halo row -- then feeds 4 MFMAs per output row instead of 2, and a tile costs a wave
  // 8 dY + 12 X fragment reads for its 72 MFMAs instead of 4 + 24 (the 2 x 2 wave grid of 32 x 32 quadrants): the kernel is
  // bound by the LDS port (taking the DMA wait out altogether changed nothing), so reads per MFMA are what counts.  Every
  // accumulator still sums the same products in the same order: bit-identical.
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const int co0 = (pair / a.tiles_ci) * 64, ci0 = (pair % a.tiles_ci) * 64;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets (bytes inside a stage): rows lp = 4g + q4 (and lp + 16 at +2048)
  const int lp = 4 * g + q4;
  int offA[4], offB[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = i * 16 + cc;
    offA[i] = XB + lp * 128 + (((ch >> 3) ^ (lp & 7)) << 4) + (ch & 7) * 2;
  }
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int ch = wave * 16 + cc;
    offB[kw] = lp * 128 + (((ch >> 3) ^ ((kw + lp) & 7)) << 4) + (ch & 7) * 2;
  }

  // ---- DMA loader: slot position (lane & 7) of pixel slot q = 32u + 8*wave + (lane >> 3) holds chunk (lane&7)^(q&7)
  const int chunk = (tid & 7) ^ ((tid >> 3) & 7), pb = tid >> 3;
  const bool yc_ok = (co0 + chunk * 8) < a.CoutP, xc_ok = (ci0 + chunk * 8) < a.CinP;
  const BufSrd xrsrc = make_srd(const_cast<void*>(a.x), a.x_bytes);
  const BufSrd yrsrc = make_srd(const_cast<void*>(a.dy), a.dy_bytes);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int per_img = a.tiles_y * a.tiles_x;
  // Per-tile vector work of the loader is what the MFMA phase of the partner block pays for (conv_c64.hip): addresses
  // are a tile-invariant per-lane part + one scalar per tile, image rows are range-checked by a per-image buffer
  // resource, the column check runs only for tiles at the left / right image edge, and the tile coordinates are
  // advanced, not divided out.  Zero padding only; reflect / replicate (DIP) keep the general per-lane mapping.
  constexpr int NX = (HR * HC) / 32, NY = (R * 32) / 32;
  constexpr int FAR = 0x7FFFFF00;
  int xpart[NX], ypart[NY];
  unsigned hc_pack = 0;
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int q = pb + 32 * u;
    const int hr = q / HC, hc = q - hr * HC;
    xpart[u] = (xc_ok && hc < 34) ? ((hr * a.IW + hc) * a.CinP + ci0 + chunk * 8) * 2 : FAR;
    hc_pack |= (unsigned)hc << (6 * u);
  }
#pragma unroll
  for (int u = 0; u < NY; ++u) {
    const int p = pb + 32 * u;
    ypart[u] = yc_ok ? (((p >> 5) * a.OW + (p & 31)) * a.CoutP + co0 + chunk * 8) * 2 : FAR;
  }
  const unsigned ximg = (unsigned)(a.IH * a.IW * a.CinP * 2), yimg = (unsigned)(a.OH * a.OW * a.CoutP * 2);
  const bool fast = a.pad_mode == DSR_PAD_ZERO;
  struct TileXY {
    int n, ty, tx;
  };
  auto dma = [&](const TileXY& tc, int buf) {
    const int n = tc.n;
    const int oy0 = tc.ty * R, ox0 = tc.tx * 32;
    unsigned char* st = smem + buf * STAGE;
    if (fast) {
      const BufSrd xr = make_srd(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x)) + (size_t)n * ximg, ximg);
      const BufSrd yr = make_srd(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.dy)) + (size_t)n * yimg, yimg);
      const int xs = ((oy0 - a.pad) * a.IW + ox0 - a.pad) * a.CinP * 2;       // negative above the image: out of range
      const int ys = (oy0 * a.OW + ox0) * a.CoutP * 2;
      unsigned char* dx = st + 8 * wave * 128;
      if (ox0 - a.pad >= 0 && ox0 - a.pad + 34 <= a.IW) {                        // interior columns
#pragma unroll
        for (int u = 0; u < NX; ++u)
          lds_dma16(xr, dx + 32 * u * 128, (unsigned)(xpart[u] + xs));
      } else {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const int ix = ox0 - a.pad + (int)((hc_pack >> (6 * u)) & 63);
          lds_dma16(xr, dx + 32 * u * 128, (unsigned)ix < (unsigned)a.IW ? (unsigned)(xpart[u] + xs) : OOB);
        }
      }
      if (ox0 + 32 <= a.OW) {
#pragma unroll
        for (int u = 0; u < NY; ++u)
          lds_dma16(yr, dx + XB + 32 * u * 128, (unsigned)(ypart[u] + ys));
      } else {
#pragma unroll
        for (int u = 0; u < NY; ++u)
          lds_dma16(yr, dx + XB + 32 * u * 128, ox0 + (pb & 31) < a.OW ? (unsigned)(ypart[u] + ys) : OOB);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < (HR * HC) / 32; ++u) {      // 5 wave-instructions per wave: the X halo
      const int q = pb + 32 * u;
      const int hr = q / HC, hc = q - hr * HC;
      bool ok = xc_ok && hc < 34;
      const int iy = pad_index(oy0 + hr - a.pad, a.IH, a.pad_mode, ok);
      const int ix = pad_index(ox0 + hc - a.pad, a.IW, a.pad_mode, ok);
      lds_dma16(xrsrc, st + (32 * u + 8 * wave) * 128, ok ? (unsigned)((((n * a.IH + iy) * a.IW + ix) * a.CinP + ci0 + chunk * 8) * 2) : OOB);
    }
#pragma unroll
    for (int u = 0; u < (R * 32) / 32; ++u) {       // 2 per wave: the dY tile
      const int p = pb + 32 * u;
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
      const bool ok = yc_ok && oy < a.OH && ox < a.OW;
      lds_dma16(yrsrc, st + XB + (32 * u + 8 * wave) * 128, ok ? (unsigned)((((n * a.OH + oy) * a.OW + ox) * a.CoutP + co0 + chunk * 8) * 2) : OOB);
    }
  };
  auto next_tile = [&](TileXY c) {                   // tiles of a block are consecutive
    if (++c.tx == a.tiles_x) {
      c.tx = 0;
      if (++c.ty == a.tiles_y) {
        c.ty = 0;
        ++c.n;
      }
    }
    return c;
  };

  int t = ychunk * a.tiles_per_block;
  int t_end = t + a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  TileXY nxt;
  nxt.n = t / per_img;
  nxt.ty = (t - nxt.n * per_img) / a.tiles_x;
  nxt.tx = t - nxt.n * per_img - nxt.ty * a.tiles_x;
  __builtin_amdgcn_s_setprio(1);
  if (t < t_end) dma(nxt, 0);
  int buf = 0;
  for (; t < t_end; ++t, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // this tile has landed; nobody still reads the other stage
    asm volatile("" ::: "memory");
    nxt = next_tile(nxt);
    if (t + 1 < t_end) dma(nxt, buf ^ 1);
    // The co-resident wave of the other block on this SIMD is, most of the time, in its MFMA phase.  Vector issue is arbitrated
    // by priority, then age: at equal priority the ~50 instructions of a fetch (7 LDS-DMA requests among them) got a slot
    // now and then between the partner's MFMAs -- s_memtime stamps: a third of a tile's time.  They run at priority 1 (set at
    // the end of the MFMA phase below); the MFMA stream fills the slots that leaves (conv_c64.hip).
    __builtin_amdgcn_s_setprio(0);
    const unsigned char* st = smem + buf * STAGE;
    // An X fragment (16 pixels of halo row h at tap column kw) serves BOTH tile rows: row r = h - kh for the tap rows kh that
    // keep r inside the tile -- 12 fragment pairs per tile instead of 18 (row, tap) pairs, a third fewer LDS reads per MFMA.
    // Every accumulator still receives row 0 before row 1: bit-identical to the (row, tap) order.  Software pipeline: the pair
    // of unit u + 2 is requested before the MFMAs of unit u (three rolling sets); hipcc otherwise issues
    // `4 reads; s_waitcnt lgkmcnt(0); 4 MFMAs` per tap.
    auto frag = [&](int off) {
      const s16x4 lo = tr_read(st + off);
      const s16x4 hi = tr_read(st + off + 2048);
      return __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    constexpr int NU = 3 * HR;                         // units u = kw * HR + h
    auto load_b = [&](int u) {
      const int kw = u / HR, h = u - kw * HR;
      return frag(offB[kw] + (h * HC + kw) * 128);
    };
    U4 fa[R][4], fb[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = frag(offA[i]);
    fb[0] = load_b(0);
    fb[1] = load_b(1);
#pragma unroll
    for (int r = 1; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[r][i] = frag(offA[i] + r * 32 * 128);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 2 < NU) fb[(u + 2) % 3] = load_b(u + 2);
      __builtin_amdgcn_sched_barrier(0);
      const int kw = u / HR, h = u - kw * HR;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int kh = h - r;
        if (kh < 0 || kh > 2) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[kh * 3 + kw][i] = mfma16<DT>(fa[r][i], fb[u % 3], acc[kh * 3 + kw][i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(1);
  }

  float* P = a.partial + (size_t)ychunk * 9 * a.CoutP * a.CinP;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + i * 16 + 4 * g + r, ci = ci0 + wave * 16 + l16;
        if (co < a.CoutP && ci < a.CinP) P[((size_t)tp * a.CoutP + co) * a.CinP + ci] = acc[tp][i][r];
      }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_kernel(const WgradTileArgs a) {
  conv_wgrad_dma_body<DT>(a, blockIdx.x, blockIdx.y);
}

// Grouped form: up to DSR_WGRAD_BATCH_MAX problems (layers) in ONE launch.  A 64 -> 64 layer has a single 64 x 64 output
// tile pair, so a launch of its own can only fill the chip by cutting the pixel range into ~256 chunks, each of which
// writes a 147 KB fp32 partial slab (37 MB per layer, read back by a two-pass reduction): the generator's 33 trunk layers
// spent 62 us in the contraction and 19 us in reductions EACH.  With all layers of a backward pass in one grid the
// parallelism comes from the layers and ~30 chunks per layer suffice.  Block -> (problem, pair, chunk) by a uniform scan
// of the first-block table (scalar code, <= 36 steps).
template <int DT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_batch_kernel(const WgradBatchArgs b) {
  int pi = 0;
  while (pi + 1 < b.count && (int)blockIdx.x >= b.first_block[pi + 1]) ++pi;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const WgradTileArgs a = b.p[pi];
  const int local = (int)blockIdx.x - b.first_block[pi];
  const int pairs = a.tiles_co * a.tiles_ci;
  conv_wgrad_dma_body<DT>(a, local % pairs, local / pairs);
}

// ------------------------------------------------------------------ 3x3 / stride 2, same LDS-DMA pipeline
// Tiles are 1 output row x 32 columns; the halo is 3 rows x 65 pixels at a pitch of 80.  A fragment's pixel slots
// are 2 apart (q = kh*80 + kw + 2*lp), so the swizzle key is taken from (q >> 1): with an 80-pixel pitch it reduces to
// (lp + (kw >> 1)) & 7 -- per-lane, two variants -- and 16 consecutive fragment rows still spread over all 8 slots.
// COH = 2 (round 3): 8 waves, the block owns 128 output channels (two 64-channel halves, waves 4..7 the second) of its
// 64-input-channel block: the X halo -- the big operand: D.b2's input is 537 MB and every 64-channel output block re-reads it
// -- is fetched once per 128 instead of once per 64 output channels.  One block per CU (acc alone is 144 registers per wave).
template <int DT, int COH = 1>
__global__ __launch_bounds__(256 * COH, 2) void conv_wgrad_dma_s2_kernel(const WgradTileArgs a) {
  constexpr int HC = 80, HR = 3, NSLOT = HR * HC;
  constexpr int XB = NSLOT * 128, YB = 32 * 128, STAGE = XB + COH * YB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);   // wave-uniform: keep it (and what derives from it) in SGPRs
  const int hsel = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);          // output-channel half of this wave (0 when COH == 1)
  // (a wave owns all 64 output channels of a 16-input-channel strip, as in conv_wgrad_dma_body: 4 + 9 fragment reads per
  //  tile instead of 2 + 18)
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const int pair = blockIdx.x;
  const int co0 = (pair / a.tiles_ci) * 64 * COH + hsel * 64, ci0 = (pair % a.tiles_ci) * 64;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lp = 4 * g + q4;
  int offA[4], offB[2];                    // offB[kw >> 1]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = i * 16 + cc;
    offA[i] = XB + hsel * YB + lp * 128 + (((ch >> 3) ^ (lp & 7)) << 4) + (ch & 7) * 2;
  }
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int ch = wave * 16 + cc;
    offB[v] = 2 * lp * 128 + (((ch >> 3) ^ ((lp + v) & 7)) << 4) + (ch & 7) * 2;
  }

  const int pb = tid >> 3;
  const int xchunk = (tid & 7) ^ ((4 * wave + (lane >> 4)) & 7);     // key of slot q = 32u + 8*wave + (lane>>3): (q>>1)&7
  const int ychunk = (tid & 7) ^ ((tid >> 3) & 7);
  const bool yc_ok = (co0 + ychunk * 8) < a.CoutP, xc_ok = (ci0 + xchunk * 8) < a.CinP;
  const BufSrd xrsrc = make_srd(const_cast<void*>(a.x), a.x_bytes);
  const BufSrd yrsrc = make_srd(const_cast<void*>(a.dy), a.dy_bytes);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int per_img = a.tiles_y * a.tiles_x;
  // loader addresses as in conv_wgrad_dma_kernel: per-lane tile-invariant part + per-tile scalar (zero padding only)
  constexpr int NX = (NSLOT + 31) / 32;           // 8 (the last pass only has slots for waves 0 and 1)
  constexpr int FAR = 0x7FFFFF00;
  int xpart[NX];
  unsigned hc_lo = 0, hc_hi = 0;                  // halo column of this lane's slot in pass u, 7 bits each
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int q = pb + 32 * u;
    const int hr = q / HC, hc = q - hr * HC;
    xpart[u] = (xc_ok && hc < 65 && hr < HR) ? ((hr * a.IW + hc) * a.CinP + ci0 + xchunk * 8) * 2 : FAR;
    if (u < 4) hc_lo |= (unsigned)hc << (7 * u);
    else hc_hi |= (unsigned)hc << (7 * (u - 4));
  }
  const int ypart = yc_ok ? (pb * a.CoutP + co0 + ychunk * 8) * 2 : FAR;
  const unsigned ximg = (unsigned)(a.IH * a.IW * a.CinP * 2), yimg = (unsigned)(a.OH * a.OW * a.CoutP * 2);
  const bool fast = a.pad_mode == DSR_PAD_ZERO;
  struct TileXY {
    int n, ty, tx;
  };
  auto dma = [&](const TileXY& tc, int buf) {
    const int n = tc.n, oy0 = tc.ty, ox0 = tc.tx * 32;
    unsigned char* st = smem + buf * STAGE;
    if (fast) {
      const BufSrd xr = make_srd(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x)) + (size_t)n * ximg, ximg);
      const BufSrd yr = make_srd(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.dy)) + (size_t)n * yimg, yimg);
      const int ix0 = ox0 * 2 - a.pad;
      const int xs = ((oy0 * 2 - a.pad) * a.IW + ix0) * a.CinP * 2;
      const int ys = (oy0 * a.OW + ox0) * a.CoutP * 2;
      unsigned char* dx = st + 8 * wave * 128;
      const bool interior = ix0 >= 0 && ix0 + 65 <= a.IW;
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        if (COH == 2 && (u & 1) != hsel) continue;   // (the two halves share the halo fetch: even / odd passes)
        if (32 * u + 8 * wave < NSLOT) {             // wave-uniform
          unsigned off = (unsigned)(xpart[u] + xs);
          if (!interior) {
            const int hc = (int)(((u < 4 ? hc_lo >> (7 * u) : hc_hi >> (7 * (u - 4)))) & 127);
            if (!((unsigned)(ix0 + hc) < (unsigned)a.IW)) off = OOB;
          }
          lds_dma16(xr, dx + 32 * u * 128, off);
        }
      }
      lds_dma16(yr, dx + XB + hsel * YB, (ox0 + 32 <= a.OW || ox0 + pb < a.OW) ? (unsigned)(ypart + ys) : OOB);
      return;
    }
#pragma unroll
    for (int u = 0; u < (NSLOT + 31) / 32; ++u) {
      if (COH == 2 && (u & 1) != hsel) continue;
      if (32 * u + 8 * wave < NSLOT) {               // wave-uniform: the last pass only has slots for waves 0 and 1
        const int q = pb + 32 * u;
        const int hr = q / HC, hc = q - hr * HC;
        bool ok = xc_ok && hc < 65;
        const int iy = pad_index(oy0 * 2 + hr - a.pad, a.IH, a.pad_mode, ok);
        const int ix = pad_index(ox0 * 2 + hc - a.pad, a.IW, a.pad_mode, ok);
        lds_dma16(xrsrc, st + (32 * u + 8 * wave) * 128, ok ? (unsigned)((((n * a.IH + iy) * a.IW + ix) * a.CinP + ci0 + xchunk * 8) * 2) : OOB);
      }
    }
    {
      const int ox = ox0 + pb;
      const bool ok = yc_ok && ox < a.OW;
      lds_dma16(yrsrc, st + XB + hsel * YB + 8 * wave * 128, ok ? (unsigned)((((n * a.OH + oy0) * a.OW + ox) * a.CoutP + co0 + ychunk * 8) * 2) : OOB);
    }
  };

  int t = blockIdx.y * a.tiles_per_block;
  int t_end = t + a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  TileXY nxt;
  nxt.n = t / per_img;
  nxt.ty = (t - nxt.n * per_img) / a.tiles_x;
  nxt.tx = t - nxt.n * per_img - nxt.ty * a.tiles_x;
  __builtin_amdgcn_s_setprio(1);
  if (t < t_end) dma(nxt, 0);
  int buf = 0;
  for (; t < t_end; ++t, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (++nxt.tx == a.tiles_x) {                     // tiles of a block are consecutive
      nxt.tx = 0;
      if (++nxt.ty == a.tiles_y) {
        nxt.ty = 0;
        ++nxt.n;
      }
    }
    if (t + 1 < t_end) dma(nxt, buf ^ 1);
    __builtin_amdgcn_s_setprio(0);                   // (fetch at priority 1, MFMA phase at 0: see conv_wgrad_dma_body)
    const unsigned char* st = smem + buf * STAGE;
    // the 9 taps as a software pipeline (see conv_wgrad_dma_body): X fragments of tap u + 2 requested before the MFMAs of tap u
    U4 fa[4], fb[3];
    auto load_b = [&](int u) {
      const int kh = u / 3, kw = u % 3;
      const s16x4 lo = tr_read(st + offB[kw >> 1] + (kh * HC + kw) * 128);
      const s16x4 hi = tr_read(st + offB[kw >> 1] + (kh * HC + kw) * 128 + 32 * 128);
      return __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s16x4 lo = tr_read(st + offA[i]);
      const s16x4 hi = tr_read(st + offA[i] + 2048);
      fa[i] = __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    fb[0] = load_b(0);
    fb[1] = load_b(1);
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      if (u + 2 < 9) fb[(u + 2) % 3] = load_b(u + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[u][i] = mfma16<DT>(fa[i], fb[u % 3], acc[u][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(1);
  }

  float* P = a.partial + (size_t)blockIdx.y * 9 * a.CoutP * a.CinP;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + i * 16 + 4 * g + r, ci = ci0 + wave * 16 + l16;
        if (co < a.CoutP && ci < a.CinP) P[((size_t)tp * a.CoutP + co) * a.CinP + ci] = acc[tp][i][r];
      }
}

// stride-2 weight gradient with 128 output channels per (8-wave) block: taken where the re-read of X per output-channel block
// is what bounds the launch -- D.b2 (537 MB input, two 64-channel blocks): 0.296 -> 0.226 ms; neutral at 268 MB (D.b4), slower
// at 67 MB (D.b6: 0.169 -> 0.196 ms, one synchronised 8-wave block per CU hides less than two 4-wave blocks), so by input size.
// DSR_WGRAD_S2_CO128: 0 = never, 1 (default) = inputs of 400 MB or more, 2 = whenever Cout % 128 == 0 (tests).
static bool dsr_wgrad_s2_co128(int KH, int stride, int CoutP, long long x_bytes) {
  const char* e = getenv("DSR_WGRAD_S2_CO128");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0 || KH != 3 || stride != 2 || CoutP < 128 || CoutP % 128 != 0) return false;
  return mode == 2 || x_bytes >= 400ll * 1000 * 1000;
}

int dsr_wgrad_tile_plan(int KH, int KW, int stride, int N, int OH, int OW, int CinP, int CoutP, WgradTileArgs* a) {
  const bool k3 = KH == 3 && KW == 3 && (stride == 1 || stride == 2);
  const bool k1 = KH == 1 && KW == 1 && stride == 1;
  if (!k3 && !k1) return 0;
  const int R = k1 ? 4 : (stride == 1 ? 2 : 1);  // 3x3 runs on the LDS-DMA kernels: 2-row (stride 1) / 1-row (stride 2) tiles
  a->tiles_y = (OH + R - 1) / R;
  a->tiles_x = (OW + 31) / 32;
  a->ntiles = N * a->tiles_y * a->tiles_x;
  const bool co128 = dsr_wgrad_s2_co128(KH, stride, CoutP, (long long)N * (2 * OH) * (2 * OW) * CinP * 2);
  a->tiles_co = co128 ? CoutP / 128 : (CoutP + 63) / 64;
  a->tiles_ci = (CinP + 63) / 64;
  long long pairs = (long long)a->tiles_co * a->tiles_ci;
  long long want = ((co128 ? 256 : 512) + pairs - 1) / pairs;   // 2 resident 4-wave blocks per CU (VGPR-limited) or one 8-wave block, one wave of blocks
  if (want > a->ntiles) want = a->ntiles;
  if (want < 1) want = 1;
  a->tiles_per_block = (int)((a->ntiles + want - 1) / want);
  return (a->ntiles + a->tiles_per_block - 1) / a->tiles_per_block;
}

template <int DT>
static void launch_dt(const WgradTileArgs& a, int KH, int stride, dim3 grid, hipStream_t st) {
  if (KH == 3 && stride == 1)
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<DT>), grid, dim3(256), 0, st, a);
  else if (KH == 3 && a.tiles_co * 128 == a.CoutP && a.tiles_co * 64 != a.CoutP)     // (the plan chose 128-channel blocks)
    hipLaunchKernelGGL((conv_wgrad_dma_s2_kernel<DT, 2>), grid, dim3(512), 0, st, a);
  else if (KH == 3)
    hipLaunchKernelGGL((conv_wgrad_dma_s2_kernel<DT>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_tile_kernel<DT, 1, 1, 1>), grid, dim3(256), 0, st, a);
}

void dsr_launch_wgrad_dma_batch(const WgradBatchArgs& b, int dtype, hipStream_t st) {
  const int blocks = b.first_block[b.count];
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_wgrad_dma_batch_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL((conv_wgrad_dma_batch_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, b);
}

void dsr_launch_wgrad_tile(const WgradTileArgs& a, int KH, int stride, int ychunks, int dtype, hipStream_t st) {
  dim3 grid(a.tiles_co * a.tiles_ci, ychunks);
  if (dtype == DSR_DTYPE_BF16)
    launch_dt<DSR_DTYPE_BF16>(a, KH, stride, grid, st);
  else
    launch_dt<DSR_DTYPE_F16>(a, KH, stride, grid, st);
}
