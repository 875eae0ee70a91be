// Persistent form of the fast-path gather convolution (conv_gemm.hip: zero padding, Cin % 64 == 0, LDS-DMA loader,
// 128 x 64 tiles, 4 waves, two or three resident blocks per CU) for launches with many tiles.
//
// Why: timing the one-tile-per-block kernel with the K loop cut short shows a fixed cost per launch of about twice the
// time its output takes to write at the HBM rate (D.b1 forward, 537 MB out: 0.23 of 0.46 ms) -- all resident blocks
// reach their epilogue together, the stores drain while no MFMA runs, then all start K loops while HBM idles.  Here a
// block walks tiles t, t + grid, ...; at a tile boundary it
//   1. writes the C tile into the LDS stage it computed last (so the other stage is free),
//   2. starts the LDS-DMA of the NEXT tile's first K-step into that free stage,
//   3. issues the output stores as unconditional buffer stores (masked lanes use an out-of-range offset, which
//      the hardware drops) -- a fixed NUMBER of store instructions per wave,
//   4. enters the next K loop waiting with `s_waitcnt vmcnt(NSTORE)`: vmcnt retires in issue order, the DMA was
//      issued before the stores, so this waits for the DMA only and the stores drain under the next tile's MFMAs.
// Everything else (fragment layout, swizzle, epilogue semantics, statistics rows, PixelShuffle store) is the
// contract of conv_gemm_kernel; tests run both kernels on the same shapes.
#include <stdlib.h>

#include <type_traits>

#include "dsr_common.h"
#include "dsr_kernels.h"

template <int DT, int BN, bool SWAP>
__global__ __launch_bounds__(256, 2) void conv_gemm_persist_kernel(const ConvGemmArgs a) {
  constexpr int BM = 128, WGM = 2, WGN = 2, NT = 256, RPP = 32;
  constexpr int WM = 64, WN = BN / WGN, TM = 4, TN = WN / 16;
  constexpr int RA = BM / RPP, RB = BN / RPP;
  constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128, STAGE = A_STAGE + B_STAGE;
  constexpr bool C_SWZ = BN == 128;                      // 128 x 256 B C tile = exactly one stage: swizzle, no row pad
  constexpr int C_STRIDE = C_SWZ ? 256 : BN * 2 + 16;
  static_assert(BM * C_STRIDE <= STAGE, "the C tile must fit in one stage");
  constexpr int CH = BN / 8;                             // 16-byte chunks per C row
  constexpr int NSTORE = BM * CH / NT;                   // output store instructions per wave and tile (8 | 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // stage 0 | stage 1 | statistics | taps
  float* sStat = reinterpret_cast<float*>(smem + 2 * STAGE);
  int* sTaps = reinterpret_cast<int*>(smem + 2 * STAGE + WGM * 2 * BN * 4);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPRs (LDS-DMA base in M0 without v_readfirstlane)
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  const int j = tid & 7, rb = tid >> 3;
  const int jc = j ^ (rb & 7);                           // source-side swizzle of the lane-linear DMA image
  const int sw = r16 & 7;

  for (int i = tid; i < a.ntaps; i += NT) sTaps[i] = a.taps[i];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int cu8 = a.CU >> 3;
  const int ks = a.ksteps;
  const int total = a.tiles_m * a.tiles_n;
  const float slope = (a.flags & DSR_F_PRELU_PTR) ? a.prelu[0] : a.slope;
  const bool do_stats = (a.flags & DSR_F_STATS) != 0;

  // ---- per-tile loader state
  int tile_m = 0, m0 = 0, n0 = 0;
  int a_iy0[RA], a_ix0[RA], a_base[RA], b_base[RB];
  auto setup_tile = [&](int t) {
    const int bid = xcd_remap(t, total);
    int tile_n = 0;
    tile_m = bid;
    if (a.tiles_n != 1) {                      // (one column tile is the common case: skip the scalar division chain)
      tile_n = bid % a.tiles_n;
      tile_m = bid / a.tiles_n;
    }
    m0 = tile_m * BM;
    n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + rb + RPP * i;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int n = fd_div(a.fd_ghw, mm);
      const int rem = mm - n * (a.GH * a.GW);
      const int gy = fd_div(a.fd_gw, rem);
      const int gx = rem - gy * a.GW;
      a_iy0[i] = ok ? gy * a.isy : -(1 << 20);           // rows past M: every tap lands out of range
      a_ix0[i] = gx * a.isx;
      a_base[i] = ((n * a.IH * a.IW + (ok ? gy * a.isy : 0) * a.IW + a_ix0[i]) * a.CinP + jc * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = rb + RPP * i;
      const int co = (n0 + row) < a.NB ? (n0 + row) : a.NB - 1;   // rows beyond NB feed columns that are never stored
      b_base[i] = (co * a.CinP + jc * 8) * 2;
    }
  };
  struct TapStep {
    int dy, dx, toff, woff;
  };
  auto decode_step = [&](int s) {
    const int t = fd_div(a.fd_cu8, s);
    const int cbase = (s - t * cu8) * 64;
    const int tp = sTaps[t];
    TapStep d;
    d.dy = (int)(signed char)(tp & 0xff);
    d.dx = (int)(signed char)((tp >> 8) & 0xff);
    const int widx = (tp >> 16) & 0xffff;
    d.toff = ((d.dy * a.IW + d.dx) * a.CinP + cbase) * 2;
    d.woff = (widx * a.NB * a.CinP + cbase) * 2;
    return d;
  };
  auto dma_issue = [&](const TapStep& d, int stage) {
    unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const bool inb = (unsigned)(a_iy0[i] + d.dy) < (unsigned)a.IH && (unsigned)(a_ix0[i] + d.dx) < (unsigned)a.IW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(st + (wave * 8 + RPP * i) * 128), 16,
                                               inb ? (unsigned)(a_base[i] + d.toff) : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(st + A_STAGE + (wave * 8 + RPP * i) * 128), 16,
                                               (unsigned)(b_base[i] + d.woff), 0, 0, 0);
  };
  auto c_off = [](int row, int colbyte) {
    if constexpr (C_SWZ)
      return row * 256 + ((((colbyte >> 4) ^ (row & 15))) << 4) + (colbyte & 15);
    else
      return row * C_STRIDE + colbyte;
  };

  // stride-1 output grid (forward, stride-1 dgrad): grid pixel m is output pixel m
  const bool dense_out = a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.GH == a.OH && a.GW == a.OW;
  const unsigned st_lpart = (unsigned)(((tid / CH) * a.CoutP + (tid % CH) * 8) * 2);
  const unsigned st_step = (unsigned)((NT / CH) * a.CoutP * 2);
  const TapStep d0 = decode_step(0);
  int t = blockIdx.x;
  setup_tile(t);
  int st0 = 0;                       // stage that holds K-step 0 of the current tile
  dma_issue(d0, st0);
  bool first = true;

  for (;;) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int k = 0; k < TN; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias of this tile's columns: requested now, so that the K loop hides the load latency (in the epilogue each of
    // these loads would stall the wave for a full L2 round trip -- measured 3-4k cycles per tile)
    float bias_r[TN][SWAP ? 4 : 1];
#pragma unroll
    for (int k = 0; k < TN; ++k)
#pragma unroll
      for (int jj = 0; jj < (SWAP ? 4 : 1); ++jj) {
        const int col = n0 + wn * WN + 16 * k + (SWAP ? 4 * g + jj : r16);
        bias_r[k][jj] = ((a.flags & DSR_F_BIAS) && col < a.cout) ? a.bias[col] : 0.f;
      }
    TapStep nd = decode_step(ks > 1 ? 1 : 0);
    for (int s = 0; s < ks; ++s) {
      // own DMA of this step done.  At a tile boundary the previous tile's NSTORE output stores were issued AFTER
      // this step's DMA: vmcnt retires in order, so vmcnt(NSTORE) waits for the DMA and lets the stores drain.
      if (s == 0 && !first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int cur = st0 ^ (s & 1);
      const unsigned char* pa = smem + cur * STAGE + (wm * WM + r16) * 128;
      const unsigned char* pb = smem + cur * STAGE + A_STAGE + (wn * WN + r16) * 128;
      U4 fa[TM], fb[TN];
      {
        const int slot = (g ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const U4*>(pa + i * 16 * 128 + slot);
#pragma unroll
        for (int k = 0; k < TN; ++k) fb[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot);
      }
      if (s + 1 < ks) dma_issue(nd, cur ^ 1);
      // second half-step's fragments before the first half's MFMAs (second register set), as in conv_gemm.hip
      U4 fa2[TM], fb2[TN];
      {
        const int slot = ((4 + g) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa2[i] = *reinterpret_cast<const U4*>(pa + i * 16 * 128 + slot);
#pragma unroll
        for (int k = 0; k < TN; ++k) fb2[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < TN; ++k) acc[i][k] = SWAP ? mfma16<DT>(fb[k], fa[i], acc[i][k]) : mfma16<DT>(fa[i], fb[k], acc[i][k]);
      if (s + 2 < ks) nd = decode_step(s + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < TN; ++k) acc[i][k] = SWAP ? mfma16<DT>(fb2[k], fa2[i], acc[i][k]) : mfma16<DT>(fa2[i], fb2[k], acc[i][k]);
    }
    const int L = st0 ^ ((ks - 1) & 1);                  // stage of the last K-step: becomes the C tile
    unsigned char* sC = smem + L * STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every wave is done reading stage L
    asm volatile("" ::: "memory");

    // ------------------------------------------------------------------ epilogue: accumulators -> C tile (+ statistics)
    const int cm0 = m0, cn0 = n0, ctile_m = tile_m;      // this tile's coordinates (setup_tile below overwrites them)
    auto epilogue = [&](auto actf, auto stats_tag) {
      constexpr bool ST = decltype(stats_tag)::value;
      if constexpr (SWAP) {
        // acc[i][k][jj] = out[row = wm*WM + 16i + r16][col = wn*WN + 16k + 4g + jj]
#pragma unroll
        for (int k = 0; k < TN; ++k) {
          const int ct0 = wn * WN + 16 * k + 4 * g;
          float bv[4];
          bool cok[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            cok[jj] = cn0 + ct0 + jj < a.cout;
            bv[jj] = bias_r[k][jj];
          }
          const int cbase = c_off(wm * WM + r16, ct0 * 2);   // + 16 rows per i: the swizzle key (row & 15) is unchanged
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float o[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = cok[jj] ? actf(acc[i][k][jj] + bv[jj]) : 0.f;
            uint2 h;
            h.x = (unsigned)f2h<DT>(o[0]) | ((unsigned)f2h<DT>(o[1]) << 16);
            h.y = (unsigned)f2h<DT>(o[2]) | ((unsigned)f2h<DT>(o[3]) << 16);
            *reinterpret_cast<uint2*>(sC + cbase + i * 16 * C_STRIDE) = h;
          }
        }
      } else {
        // acc[i][k][r] = out[row = wm*WM + 16i + 4g + r][col = wn*WN + 16k + r16]
#pragma unroll
        for (int k = 0; k < TN; ++k) {
          const int ct = wn * WN + 16 * k + r16;
          const int col = cn0 + ct;
          const bool colok = col < a.cout;
          const float bv = bias_r[k][0];
          float s1 = 0.f, s2 = 0.f;
          int cbase[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) cbase[r] = c_off(wm * WM + 4 * g + r, ct * 2);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = wm * WM + 16 * i + 4 * g + r;
              const float v = acc[i][k][r] + bv;
              if constexpr (ST) {
                const float vm = (cm0 + row < a.M && colok) ? v : 0.f;   // statistics ignore tail rows / pad columns
                s1 += vm;
                s2 += vm * vm;
              }
              const float o = colok ? actf(v) : 0.f;
              *reinterpret_cast<unsigned short*>(sC + cbase[r] + i * 16 * C_STRIDE) = f2h<DT>(o);
            }
          }
          if constexpr (ST) {
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (g == 0) {
              sStat[(wm * 2 + 0) * BN + ct] = s1;
              sStat[(wm * 2 + 1) * BN + ct] = s2;
            }
          }
        }
      }
    };
    auto run_epilogue = [&](auto actf) {
      if constexpr (SWAP)
        epilogue(actf, std::false_type{});
      else
        epilogue(actf, std::true_type{});
    };
    if (a.act == DSR_ACT_NONE)
      run_epilogue([](float v) { return v; });
    else if (a.act == DSR_ACT_RELU)
      run_epilogue([](float v) { return v > 0.f ? v : 0.f; });
    else if (a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU)
      run_epilogue([slope](float v) { return v >= 0.f ? v : v * slope; });
    else
      run_epilogue([&](float v) { return act_apply(a.act, v, slope); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // C tile and statistics complete
    asm volatile("" ::: "memory");

    if (!SWAP && do_stats) {                             // statistics rows first: plain stores, issued BEFORE the DMA
      for (int c = tid; c < 2 * BN; c += NT) {
        const int which = c / BN, ct = c % BN;
        const int col = cn0 + ct;
        if (col < a.cout)
          a.stats[((size_t)ctile_m * 2 + which) * a.stats_stride + col] =
              sStat[(0 * 2 + which) * BN + ct] + sStat[(1 * 2 + which) * BN + ct];
      }
    }

    // ---- next tile: loader state + the DMA of its first K-step into the free stage
    const int tn = t + gridDim.x;
    const bool has_next = tn < total;
    if (has_next) {
      setup_tile(tn);
      dma_issue(d0, L ^ 1);
    }

    // ---- output stores: exactly NSTORE buffer stores per wave (masked lanes: out-of-range offset, dropped)
    if (!(a.flags & DSR_F_PIXSHUF) && dense_out && cm0 + BM <= a.M && cn0 + BN <= a.CoutP) {
      // dense output, tile inside the matrix: vector idx sits at (cm0 + idx / CH) * CoutP + cn0 + (idx % CH) * 8
      const unsigned sorg = (unsigned)((cm0 * a.CoutP + cn0) * 2);
#pragma unroll
      for (int q = 0; q < NSTORE; ++q) {
        const U4 v = *reinterpret_cast<const U4*>(sC + c_off(tid / CH + (NT / CH) * q, (tid % CH) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yrsrc,
                                               sorg + st_lpart + q * st_step, 0, 0);
      }
    } else if (!(a.flags & DSR_F_PIXSHUF)) {
#pragma unroll 2
      for (int q = 0; q < NSTORE; ++q) {
        const int idx = tid + NT * q;
        const int row = idx / CH, ch = idx % CH;
        const int m = cm0 + row, col0 = cn0 + ch * 8;
        const bool ok = m < a.M && col0 < a.CoutP;
        const int mm = ok ? m : 0;
        const int n = fd_div(a.fd_ghw, mm);
        const int rem = mm - n * (a.GH * a.GW);
        const int gy = fd_div(a.fd_gw, rem);
        const int gx = rem - gy * a.GW;
        const int oy = gy * a.osy + a.ooy, ox = gx * a.osx + a.oox;
        const unsigned off = (unsigned)((((n * a.OH + oy) * a.OW + ox) * a.CoutP + col0) * 2);
        const U4 v = *reinterpret_cast<const U4*>(sC + c_off(row, ch * 16));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yrsrc,
                                               ok ? off : OOB, 0, 0);
      }
    } else {
      // PixelShuffle(2): conv channel 4c+2i+j of grid pixel (h,w) -> channel c of pixel (2h+i, 2w+j)
      constexpr int CQ = BN / 32;   // 8-channel output chunks per sub-pixel in this tile
      static_assert(BM * 4 * CQ / NT == NSTORE, "store count");
#pragma unroll 2
      for (int q = 0; q < NSTORE; ++q) {
        const int idx = tid + NT * q;
        const int row = idx / (4 * CQ);
        const int rem2 = idx % (4 * CQ);
        const int sub = rem2 / CQ, cq = rem2 % CQ;
        const int m = cm0 + row;
        const int oc0 = cn0 / 4 + cq * 8;
        const bool ok = m < a.M && oc0 < a.CoutP;
        const int mm = ok ? m : 0;
        const int n = fd_div(a.fd_ghw, mm);
        const int rem = mm - n * (a.GH * a.GW);
        const int gy = fd_div(a.fd_gw, rem);
        const int gx = rem - gy * a.GW;
        const int oy = 2 * gy + (sub >> 1), ox = 2 * gx + (sub & 1);
        unsigned short vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          vv[e] = *reinterpret_cast<const unsigned short*>(sC + c_off(row, (4 * (cq * 8 + e) + sub) * 2));
        U4 o;
        o.x = vv[0] | ((unsigned)vv[1] << 16);
        o.y = vv[2] | ((unsigned)vv[3] << 16);
        o.z = vv[4] | ((unsigned)vv[5] << 16);
        o.w = vv[6] | ((unsigned)vv[7] << 16);
        const unsigned off = (unsigned)((((n * a.OH + oy) * a.OW + ox) * a.CoutP + oc0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), yrsrc,
                                               ok ? off : OOB, 0, 0);
      }
    }
    if (!has_next) break;
    t = tn;
    st0 = L ^ 1;
    first = false;
  }
}

template <int DT, int BN, bool SWAP>
static void launch_p(const ConvGemmArgs& b, int blocks, hipStream_t st) {
  constexpr int LDS = 2 * (128 * 128 + BN * 128) + 2 * 2 * BN * 4 + DSR_MAX_TAPS * 4;
  auto* fn = conv_gemm_persist_kernel<DT, BN, SWAP>;
  if constexpr (LDS > 64 * 1024) {
    static LdsOptIn optin;
    optin.ensure((const void*)fn, LDS);
  }
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), LDS, st, b);
}

// returns false when the launch does not qualify (caller falls through to the one-tile-per-block kernel)
bool dsr_launch_conv_gemm_persist(const ConvGemmArgs& a, int dtype, hipStream_t st) {
  static const bool enabled = [] {
    const char* e = getenv("DSR_CONV_PERSIST");          // tuning switch, default on
    return !(e && e[0] == '0');
  }();
  const bool fast = a.pad_mode == DSR_PAD_ZERO && (a.CU & 7) == 0 && a.ntaps > 0;
  if (!enabled || !fast || a.NB <= 16 || (a.flags & DSR_F_OUT_NCHW_F32)) return false;
  // Measured on one box, persistent vs one-tile-per-block (profiles/r01_persist_ab.txt): 64-wide tiles gain (PixelShuffle
  // conv dgrad 256->64: +27 %, 64->64 stride 2: +8 %), 128-wide tiles do not (-1..-2 %: their fixed cost is epilogue
  // arithmetic, which persistence does not overlap).  So: BN = 64 only.
  if (a.NB > 64) return false;
  constexpr int BN = 64;
  ConvGemmArgs b = a;
  b.tiles_m = (a.M + 127) / 128;
  b.tiles_n = (a.NB + BN - 1) / BN;
  const long long total = (long long)b.tiles_m * b.tiles_n;
  const int blocks = a.ksteps >= 24 ? 512 : 768;         // resident blocks per CU: 2 for long K loops, 3 otherwise; multiples of 8
  if (total < 2 * blocks) return false;                  // fewer than two tiles per resident block: nothing to overlap
  const bool swap = !(a.flags & DSR_F_STATS);
  if (dtype == DSR_DTYPE_BF16) {
    if (swap) launch_p<DSR_DTYPE_BF16, BN, true>(b, blocks, st);
    else launch_p<DSR_DTYPE_BF16, BN, false>(b, blocks, st);
  } else {
    if (swap) launch_p<DSR_DTYPE_F16, BN, true>(b, blocks, st);
    else launch_p<DSR_DTYPE_F16, BN, false>(b, blocks, st);
  }
  return true;
}
