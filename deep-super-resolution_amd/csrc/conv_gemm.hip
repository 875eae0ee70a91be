// Implicit-GEMM "gather" convolution for gfx950: out[pixel][co] = sum_{tap,ci} in[pixel(+tap)][ci] * W[tap][co][ci]
//
// One kernel serves conv forward, input-grad (dgrad: stride-1 directly, stride-2 as 4 output-parity
// classes) and any k x k / stride / zero|reflect|replicate padding the reference's layers use
// (models/GAN/generator.py:7,11,30,47,52,62; models/GAN/discriminator.py:7,25; models/DIP/utils.py:83-105),
// because the tap list (dy, dx, weight slice) is data: see conv_api.hip for how each case fills it.
//
//  * NHWC 16-bit activations, channels padded to 8: one 16-byte vector = 8 channels of one pixel.
//  * K is walked in "units" of 8 channels; a K-step is 8 units (64 k).  A-tile rows are pixels
//    (gathered, coalesced 128 B per pixel when Cin>=64), B-tile rows are output channels.
//  * global -> registers -> LDS (XOR-swizzled 128-B rows, conflict-free ds_read_b128), double-buffered
//    LDS, one barrier per K-step; the next step's global loads are issued before the MFMAs of this one.
//  * mfma_f32_16x16x32 (bf16 or f16), fp32 accumulate; 4 waves; wave tile (BM/WGM) x (BN/WGN).
//  * epilogue: +bias, activation, per-channel sum / sum-of-squares partials for BatchNorm (from the fp32
//    accumulators, one partial row per M-tile: deterministic, no atomics), staged through LDS so that
//    global stores are 16-byte NHWC vectors; optional PixelShuffle(2) store
//    (out[n,2h+i,2w+j,c] = y[n,h,w,4c+2i+j], generator.py:32,38) or fp32 NCHW store for the last layer.
#include <stdlib.h>

#include <type_traits>

#include "dsr_common.h"
#include "dsr_kernels.h"

template <int BM, int BN, int WGM, int NSTAGE>
struct ConvGemmLds {
  static constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128;
  static constexpr int MAIN = NSTAGE * (A_STAGE + B_STAGE);
  static constexpr int C = BM * (BN * 2 + 16);
  static constexpr int TILE = MAIN > C ? MAIN : C;
  static constexpr int TOTAL = TILE + WGM * 2 * BN * 4 + DSR_MAX_TAPS * 4;
};

// PADX: the LDS-DMA fast path for reflect / replicate padding (the DIP encoder / decoder convs, models/DIP/utils.py:83-105 with
// pad = 'reflection'): the padded coordinate of every A row is recomputed per K-step (a dozen VALU instructions per piece)
// instead of falling back to the register-staged generic loader -- 10.6 us against 23.6 us per 128 -> 128 layer at 128^2.
template <int DT, int BM, int BN, int WGM, int WGN, bool FAST, bool DMA, int NSTAGE, bool SWAP, bool PADX = false>
__global__ __launch_bounds__(64 * WGM * WGN, (NSTAGE == 3 || BM * BN >= 224 * 256) ? 2 : (WGM * WGN) / 2) void conv_gemm_kernel(const ConvGemmArgs a) {
  static_assert(!DMA || FAST, "the LDS-DMA loader exists for the fast path only");
  static_assert(!PADX || (DMA && NSTAGE == 2 && WGM * WGN == 4), "padded-coordinate form: 4-wave two-stage DMA kernels only");
  static_assert(NSTAGE == 2 || (NSTAGE == 3 && DMA), "three stages: DMA ring only");
  constexpr int NW = WGM * WGN;                 // 4 or 8 waves; two blocks per CU either way
  constexpr int NT = 64 * NW;
  constexpr int RPP = NT / 8;                   // tile rows covered by one pass of the loader (8 lanes per row)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  // (BM = 224: the 8-wave tile with 7 instead of 8 m-tiles per wave; its loader's last pass covers 32 rows only -- waves 4..7
  //  sit it out, wave-uniformly: a DMA piece writes its 8 rows whatever the offsets are, and theirs lie behind the stage)
  constexpr int RA = (BM + RPP - 1) / RPP;
  constexpr int RB = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0 || (DMA && BM % 8 == 0), "partial loader pass: LDS-DMA variants only");
  static_assert(BM % 128 == 0 || BM == 224 || BM == 64, "tile heights");
  constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128;
  constexpr int LDS_MAIN = NSTAGE * (A_STAGE + B_STAGE);
  constexpr int C_STRIDE = BN * 2 + 16;
  constexpr int LDS_C = BM * C_STRIDE;
  constexpr int LDS_BYTES = LDS_MAIN > LDS_C ? LDS_MAIN : LDS_C;
  // one dynamic LDS object (size ConvGemmLds<...>::TOTAL, passed at launch): stages | statistics | tap table
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(LDS_BYTES == ConvGemmLds<BM, BN, WGM, NSTAGE>::TILE, "LDS layout");
  unsigned char* sA = smem;
  unsigned char* sB = smem + NSTAGE * A_STAGE;
  float* sStat = reinterpret_cast<float*>(smem + LDS_BYTES);
  int* sTaps = reinterpret_cast<int*>(smem + LDS_BYTES + WGM * 2 * BN * 4);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPRs (LDS-DMA base in M0 without v_readfirstlane)
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  for (int i = tid; i < a.ntaps; i += NT) sTaps[i] = a.taps[i];

  // ---- loader role: unit j of the K-step, rows rb + 32*i
  const int j = tid & 7, rb = tid >> 3;
  int a_iy0[RA], a_ix0[RA], a_nb[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + rb + RPP * i;
    a_ok[i] = m < a.M;
    int mm = a_ok[i] ? m : 0;
    int n = fd_div(a.fd_ghw, mm);
    int rem = mm - n * (a.GH * a.GW);
    int gy = fd_div(a.fd_gw, rem);
    int gx = rem - gy * a.GW;
    a_iy0[i] = a_ok[i] ? gy * a.isy : -(1 << 20);   // rows past M: every tap lands out of range
    a_ix0[i] = gx * a.isx;
    a_nb[i] = n * a.IH * a.IW;
  }
  __syncthreads();   // sTaps visible

  U4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];   // two register sets: operand tiles 2 K-steps ahead of the MFMAs
  // All operand loads are raw buffer loads: a 32-bit byte offset per lane, and the hardware range check returns
  // zeros for the offset OOB -- so "outside the image / padding / K tail / M tail" costs one v_cndmask on the
  // offset instead of a 64-bit address select plus four selects on the data.
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  // fast path: zero padding and Cin a multiple of 64 -> the 8 units of a K-step share one tap (wave-uniform
  // decode) and a row's offset is (precomputed row base) + (per-step tap offset).
  // (FAST is a template parameter: the generic path's per-row state would otherwise cost ~20 VGPRs here)
  // DMA loader: the LDS image of one wave-instruction is lane-linear (row rb, 16-byte slot j), so the XOR swizzle
  // goes on the SOURCE side: slot j of row r holds channel chunk j ^ (r & 7).
  const int jc = DMA ? (j ^ (rb & 7)) : j;
  int a_base[RA], b_base[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) a_base[i] = ((a_nb[i] + a_iy0[i] * a.IW + a_ix0[i]) * a.CinP + jc * 8) * 2;   // bytes
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    // weight rows beyond NB only feed output columns that are never stored: clamp the row instead of predicating
    const int row = rb + RPP * i;
    const int co = (n0 + row) < a.NB ? (n0 + row) : a.NB - 1;
    b_base[i] = (co * a.CinP + jc * 8) * 2;
  }

  const int cu8 = a.CU >> 3;

  auto load_step = [&](int s, U4 (&ra)[RA], U4 (&rbv)[RB]) {
    if constexpr (FAST) {
      const int t = fd_div(a.fd_cu8, s);             // uniform; the K loop never runs past the last tap here
      const int cbase = (s - t * cu8) * 64;          // first channel of this step
      const int tp = sTaps[t];
      const int dy = (int)(signed char)(tp & 0xff);
      const int dx = (int)(signed char)((tp >> 8) & 0xff);
      const int widx = (tp >> 16) & 0xffff;
      const int toff = ((dy * a.IW + dx) * a.CinP + cbase) * 2;
      const int woff = (widx * a.NB * a.CinP + cbase) * 2;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const bool inb = (unsigned)(a_iy0[i] + dy) < (unsigned)a.IH && (unsigned)(a_ix0[i] + dx) < (unsigned)a.IW;
        ra[i] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(
                                           xrsrc, inb ? (unsigned)(a_base[i] + toff) : OOB, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
        if (rb + RPP * i < BN)
          rbv[i] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(b_base[i] + woff), 0, 0));
    } else {
    int u = s * 8 + j;
    bool uok = u < a.U;
    int uu = uok ? u : 0;
    int t = fd_div(a.fd_cu, uu);
    int c8 = uu - t * a.CU;
    int tp = sTaps[t];
    int dy = (int)(signed char)(tp & 0xff);
    int dx = (int)(signed char)((tp >> 8) & 0xff);
    int widx = (tp >> 16) & 0xffff;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      bool inb = a_ok[i] && uok;
      int iy = pad_index(a_iy0[i] + dy, a.IH, a.pad_mode, inb);
      int ix = pad_index(a_ix0[i] + dx, a.IW, a.pad_mode, inb);
      unsigned off = (unsigned)(((a_nb[i] + iy * a.IW + ix) * a.CinP + c8 * 8) * 2);
      ra[i] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, inb ? off : OOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int row = rb + RPP * i;
      int co = (n0 + row) < a.NB ? (n0 + row) : a.NB - 1;
      unsigned off = (unsigned)(((widx * a.NB + co) * a.CinP + c8 * 8) * 2);
      if (row < BN) rbv[i] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, uok ? off : OOB, 0, 0));
    }
    }
  };
  // LDS-DMA variant of load_step: `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 B = 8 tile rows straight into
  // the stage (an out-of-range offset writes zeros), so the operand tiles never pass through VGPRs.
  struct TapStep {   // wave-uniform decode of one K-step: tap offsets into the image and into the weight image
    int dy, dx, toff, woff, coff;
  };
  [[maybe_unused]] auto decode_step = [&](int s) {
    const int t = fd_div(a.fd_cu8, s);
    const int cbase = (s - t * cu8) * 64;
    const int tp = sTaps[t];
    TapStep d;
    d.dy = (int)(signed char)(tp & 0xff);
    d.dx = (int)(signed char)((tp >> 8) & 0xff);
    const int widx = (tp >> 16) & 0xffff;
    d.toff = ((d.dy * a.IW + d.dx) * a.CinP + cbase) * 2;
    d.woff = (widx * a.NB * a.CinP + cbase) * 2;
    d.coff = cbase * 2;
    return d;
  };
  [[maybe_unused]] auto dma_issue = [&](const TapStep& d, int stage) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      if (RPP * (i + 1) > BM && wave * 8 + RPP * i >= BM) continue;   // (compile-time false for whole passes)
      unsigned off;
      if constexpr (PADX) {
        bool ok = a_ok[i];
        const int iy = pad_index(a_iy0[i] + d.dy, a.IH, a.pad_mode, ok);
        const int ix = pad_index(a_ix0[i] + d.dx, a.IW, a.pad_mode, ok);
        off = ok ? (unsigned)(a_base[i] + (((iy - a_iy0[i]) * a.IW + (ix - a_ix0[i])) * a.CinP) * 2 + d.coff) : OOB;
      } else {
        const bool inb = (unsigned)(a_iy0[i] + d.dy) < (unsigned)a.IH && (unsigned)(a_ix0[i] + d.dx) < (unsigned)a.IW;
        off = inb ? (unsigned)(a_base[i] + d.toff) : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(sA + stage * A_STAGE + (wave * 8 + RPP * i) * 128), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      if (RPP * i < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + stage * B_STAGE + (wave * 8 + RPP * i) * 128), 16,
                                                 (unsigned)(b_base[i] + d.woff), 0, 0, 0);
  };
  // LDS-DMA variant of load_step: `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 B = 8 tile rows straight into
  // the stage (an out-of-range offset writes zeros), so the operand tiles never pass through VGPRs.
  [[maybe_unused]] auto dma_step = [&](int s, int stage) { dma_issue(decode_step(s), stage); };
  int st_off[RA > RB ? RA : RB];
#pragma unroll
  for (int i = 0; i < (RA > RB ? RA : RB); ++i) {
    const int row = rb + RPP * i;
    st_off[i] = row * 128 + ((j ^ (row & 7)) << 4);
  }
  auto store_step = [&](int stage, const U4 (&ra)[RA], const U4 (&rbv)[RB]) {
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<U4*>(sA + stage * A_STAGE + st_off[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      if (rb + RPP * i < BN) *reinterpret_cast<U4*>(sB + stage * B_STAGE + st_off[i]) = rbv[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int k = 0; k < TN; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline, depth 2 in registers + 2 LDS stages: at K-step s the tile of step s sits in LDS[s&1], the
  // tile of s+1 is in one register set (loaded two iterations ago) and the loads of s+2 are in flight in the
  // other.  An iteration = {write set(s+1) -> LDS[(s+1)&1]; issue loads of s+3 into that set; 32 MFMAs on
  // LDS[s&1]; barrier}: every global load gets two full iterations of MFMA work to land.
  // bias of this tile's columns: the DMA variants (which have registers to spare) request it before the K loop, so
  // its latency does not sit in the epilogue; the register-staged variants are at their occupancy edge (159 VGPRs, three
  // blocks per CU for the 128x64 tile) and read it where it is used
  constexpr bool PRELOAD_BIAS = DMA && NW == 4;   // (the 8-wave 256x256 variant has no registers to spare: 128 accumulators)
  float bias_r[TN][SWAP ? 4 : 1];
  auto bias_at = [&](int k, int jj) {
    const int col = n0 + wn * WN + 16 * k + (SWAP ? 4 * g + jj : r16);
    return ((a.flags & DSR_F_BIAS) && col < a.cout) ? a.bias[col] : 0.f;
  };
  if constexpr (PRELOAD_BIAS) {
#pragma unroll
    for (int k = 0; k < TN; ++k)
#pragma unroll
      for (int jj = 0; jj < (SWAP ? 4 : 1); ++jj) bias_r[k][jj] = bias_at(k, jj);
  }
  [[maybe_unused]] auto bias_of = [&](int k, int jj) { return PRELOAD_BIAS ? bias_r[k][jj] : bias_at(k, jj); };
  const int ks = a.ksteps;
  if constexpr (!DMA) {
    load_step(0, ra0, rb0);
    store_step(0, ra0, rb0);
    if constexpr (BN <= 128) {
      if (ks > 1) load_step(1, ra1, rb1);
      if (ks > 2) load_step(2, ra0, rb0);
    }
    __syncthreads();
  }

  const int sw = r16 & 7;
  auto compute = [&](int cur) {
    const unsigned char* pa = sA + cur * A_STAGE + (wm * WM + r16) * 128;
    const unsigned char* pb = sB + cur * B_STAGE + (wn * WN + r16) * 128;
#pragma unroll 1
    for (int kk = 0; kk < 2; ++kk) {   // rolled: one live set of 8 fragments, so that the depth-2 operand pipeline fits in
                                       // 256 VGPRs (unrolled it spills inside the loop and measures 10 % slower)
      const int slot = ((4 * kk + g) ^ sw) << 4;
      U4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const U4*>(pa + i * 16 * 128 + slot);
#pragma unroll
      for (int k = 0; k < TN; ++k) fb[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < TN; ++k) acc[i][k] = SWAP ? mfma16<DT>(fb[k], fa[i], acc[i][k]) : mfma16<DT>(fa[i], fb[k], acc[i][k]);
    }
  };
  if constexpr (DMA) {
    // One barrier per K-step: {wait for my own DMA of step s; barrier: step s has landed for every wave and every
    // wave is done reading the other stage; start the DMA of step s+1 into that stage; 32 MFMAs on step s}.
    auto compute_flat = [&](int cur) {
      const unsigned char* pa = sA + cur * A_STAGE + (wm * WM + r16) * 128;
      const unsigned char* pb = sB + cur * B_STAGE + (wn * WN + r16) * 128;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int slot = ((4 * kk + g) ^ sw) << 4;
        U4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const U4*>(pa + i * 16 * 128 + slot);
#pragma unroll
        for (int k = 0; k < TN; ++k) fb[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int k = 0; k < TN; ++k) acc[i][k] = SWAP ? mfma16<DT>(fb[k], fa[i], acc[i][k]) : mfma16<DT>(fa[i], fb[k], acc[i][k]);
      }
    };
    if constexpr (NSTAGE == 2 && NW == 8) {
      // 256x256 tile, 8 waves of 128x64, ONE resident block per CU (two waves per SIMD, 256 registers each: 128 of them
      // accumulators).  Per K-step and wave: 24 ds_read_b128 and 64 MFMAs (0.375 reads per MFMA; the 64x64 wave tile
      // of the 128x128 variant needs 0.5 and saturates the LDS at full MFMA rate) and 8 DMA pieces per 64 MFMAs (half
      // the 128x128 tile's).  The two waves of a SIMD would run in lock-step (same program, one barrier per K-step), so
      // waves 4..7 issue their DMA between the two MFMA groups: one wave's DMA issue (60-100 cycles per piece) runs under
      // its partner's MFMAs.  One live B set (4 fragments) and a rolling A fragment keep the loop inside 256 VGPRs.
      const bool late_dma = wave >= NW / 2;
      dma_step(0, 0);
      TapStep nd = decode_step(ks > 1 ? 1 : 0);
      for (int s = 0; s < ks; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int cur = s & 1;
        const unsigned char* pa = sA + cur * A_STAGE + (wm * WM + r16) * 128;
        const unsigned char* pb = sB + cur * B_STAGE + (wn * WN + r16) * 128;
        const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
        // The 2 * TM (A fragment, k-half) units of a step run as a software pipeline: the A fragment of unit u + 2 is
        // requested before the TN MFMAs of unit u (three rolling registers sets), and the second k-half's B fragments during
        // the first half's last units -- an LDS read then has 2 * TN MFMAs (128 cycles) to land.  Left to itself hipcc waits
        // for each A fragment right before its MFMAs (`ds_read_b128; s_waitcnt lgkmcnt(0); 4 x v_mfma`): with two waves per
        // SIMD that caps the MFMA pipe at ~2 * 64 / (64 + LDS latency).  The sched_barriers pin the order.
        constexpr int NU = 2 * TM;
        U4 fb0[TN], fb1[TN], fa[3];
        auto a_frag = [&](int u) { return *reinterpret_cast<const U4*>(pa + (u % TM) * 16 * 128 + (u < TM ? slot0 : slot1)); };
#pragma unroll
        for (int k = 0; k < TN; ++k) fb0[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot0);
        fa[0] = a_frag(0);
        fa[1] = a_frag(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!late_dma && s + 1 < ks) dma_issue(nd, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u + 2 < NU) fa[(u + 2) % 3] = a_frag(u + 2);
          if (u == TM - 3) {
#pragma unroll
            for (int k = 0; k < TN; ++k) fb1[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < TN; ++k) {
            const U4& fbk = u < TM ? fb0[k] : fb1[k];
            acc[u % TM][k] = SWAP ? mfma16<DT>(fbk, fa[u % 3], acc[u % TM][k]) : mfma16<DT>(fa[u % 3], fbk, acc[u % TM][k]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (u == TM - 1) {
            if (late_dma && s + 1 < ks) dma_issue(nd, cur ^ 1);
            if (s + 2 < ks) nd = decode_step(s + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else if constexpr (NSTAGE == 2) {
      dma_step(0, 0);
      TapStep nd = decode_step(ks > 1 ? 1 : 0);      // the tap table read of the NEXT step is kept out of the loop body's
      for (int s = 0; s < ks; ++s) {                 // head: it shares the LDS counter with the fragment reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the first half-step's fragments are requested BEFORE the address arithmetic of the next DMA, which then runs
        // under their LDS latency instead of in front of it
        const int cur = s & 1;
        const unsigned char* pa = sA + cur * A_STAGE + (wm * WM + r16) * 128;
        const unsigned char* pb = sB + cur * B_STAGE + (wn * WN + r16) * 128;
        U4 fa[TM], fb[TN];
        {
          const int slot = (g ^ sw) << 4;
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const U4*>(pa + i * 16 * 128 + slot);
#pragma unroll
          for (int k = 0; k < TN; ++k) fb[k] = *reinterpret_cast<const U4*>(pb + k * 16 * 128 + slot);
        }
        if (s + 1 < ks) dma_issue(nd, cur ^ 1);
        // the second half-step's fragments are requested before the first half's MFMAs (a second register set): their
        // LDS latency runs under 16 MFMAs instead of between the two MFMA groups
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
    } else {
      // three-stage ring, one resident block per CU: the DMA of step s+2 is issued before the MFMAs of step s, so a
      // tile has two full compute phases to land.  At the top of step s the groups of steps s and s+1 are in
      // flight; vmcnt(NDMA) retires the older one only (a wave issues exactly NDMA DMA instructions per step).
      constexpr int NDMA = RA + (BN + RPP - 1) / RPP;
      dma_step(0, 0);
      if (ks > 1) dma_step(1, 1);
      int cur = 0, nxt2 = 2;
      for (int s = 0; s < ks; ++s) {
        if (s + 1 < ks)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < ks) dma_step(s + 2, nxt2);
        compute_flat(cur);
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
      }
    }
    __syncthreads();   // the epilogue reuses the stages as its C tile
  } else if constexpr (BN <= 128) {
    for (int s = 0; s < ks; s += 2) {
      // even step: LDS[0] = s, set1 = s+1, set0 = s+2 (in flight)
      if (s + 1 < ks) store_step(1, ra1, rb1);
      if (s + 3 < ks) load_step(s + 3, ra1, rb1);
      compute(0);
      __syncthreads();
      if (s + 1 >= ks) break;
      // odd step: LDS[1] = s+1, set0 = s+2, set1 = s+3 (in flight)
      if (s + 2 < ks) store_step(0, ra0, rb0);
      if (s + 4 < ks) load_step(s + 4, ra0, rb0);
      compute(1);
      __syncthreads();
    }
  } else {
    // 128x128 tile: 64 accumulator + 32 fragment registers leave room for ONE register set under the 256-VGPR
    // budget of two resident blocks (a second set spills inside the loop and measures slower): depth-1 pipeline.
    for (int s = 0; s < ks; ++s) {
      const int cur = s & 1;
      const bool more = (s + 1) < ks;
      if (more) load_step(s + 1, ra0, rb0);
      compute(cur);
      if (more) store_step(cur ^ 1, ra0, rb0);
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ epilogue
  const float slope = (a.flags & DSR_F_PRELU_PTR) ? a.prelu[0] : a.slope;
  const bool do_stats = (a.flags & DSR_F_STATS) != 0;
  const bool nchw = (a.flags & DSR_F_OUT_NCHW_F32) != 0;
  unsigned char* sC = smem;

  if constexpr (SWAP) {
    // Accumulator layout: the MFMA is issued with the WEIGHT fragment as its A operand, so D[m = channel][n = pixel]:
    // acc[i][k][j] = out[row = wm*WM + 16i + r16][col = wn*WN + 16k + 4g + j].  A lane therefore owns 4 consecutive
    // channels of one pixel: one packed 8-byte LDS write per 4 values (the pixel-major layout needs four 2-byte ones).
    if (nchw) {
      // fp32 NCHW store straight from the accumulators (last layers: Cout <= 16, so this is a small tensor)
  #pragma unroll
      for (int k = 0; k < TN; ++k) {
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + wn * WN + 16 * k + 4 * g + j;
          const bool colok = col < a.cout;
          const float bv = bias_of(k, SWAP ? j : 0);
  #pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + 16 * i + r16;
            const float o = act_apply(a.act, acc[i][k][j] + bv, slope);
            if (m < a.M && colok) {
              int n = fd_div(a.fd_ghw, m);
              int rem = m - n * (a.GH * a.GW);
              int gy = fd_div(a.fd_gw, rem);
              int gx = rem - gy * a.GW;
              int oy = gy * a.osy + a.ooy, ox = gx * a.osx + a.oox;
              a.out_f32[(((size_t)n * a.cout + col) * a.OH + oy) * a.OW + ox] = o;
            }
          }
        }
      }
      return;
    }

    // the activation (and whether statistics are wanted) is selected ONCE, wave-uniformly, and the per-element loop is
    // instantiated per choice, so the hot loop carries no per-element branches and no dead statistics arithmetic
    auto epilogue = [&](auto actf, auto stats_tag) {
      constexpr bool ST = decltype(stats_tag)::value;
  #pragma unroll
      for (int k = 0; k < TN; ++k) {
        const int ct0 = wn * WN + 16 * k + 4 * g;   // first of this lane's 4 columns inside the block tile
        float bv[4], s1[4], s2[4];
        bool cok[4];
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          cok[j] = n0 + ct0 + j < a.cout;
          bv[j] = bias_of(k, SWAP ? j : 0);
          s1[j] = s2[j] = 0.f;
        }
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WM + 16 * i + r16;
          const bool rok = m0 + row < a.M;
          float o[4];
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[i][k][j] + bv[j];
            if constexpr (ST) {
              const float vm = (rok && cok[j]) ? v : 0.f;   // statistics ignore tail rows / pad columns
              s1[j] += vm;
              s2[j] = __builtin_fmaf(vm, vm, s2[j]);
            }
            o[j] = cok[j] ? actf(v) : 0.f;
          }
          uint2 h;
          h.x = (unsigned)f2h<DT>(o[0]) | ((unsigned)f2h<DT>(o[1]) << 16);
          h.y = (unsigned)f2h<DT>(o[2]) | ((unsigned)f2h<DT>(o[3]) << 16);
          *reinterpret_cast<uint2*>(sC + row * C_STRIDE + ct0 * 2) = h;
        }
        if constexpr (ST) {
  #pragma unroll
          for (int j = 0; j < 4; ++j) {
              s1[j] = row16_sum(s1[j]);          // over the 16 pixels (lanes r16) of the fragment: DPP, dsr_common.h
            s2[j] = row16_sum(s2[j]);
          }
          if (r16 == 0) {
  #pragma unroll
            for (int j = 0; j < 4; ++j) {
              sStat[(wm * 2 + 0) * BN + ct0 + j] = s1[j];
              sStat[(wm * 2 + 1) * BN + ct0 + j] = s2[j];
            }
          }
        }
      }
    };
    auto run_epilogue = [&](auto actf) {
      if (do_stats)
        epilogue(actf, std::true_type{});
      else
        epilogue(actf, std::false_type{});
    };
    if (a.act == DSR_ACT_NONE)
      run_epilogue([](float v) { return v; });
    else if (a.act == DSR_ACT_RELU)
      run_epilogue([](float v) { return v > 0.f ? v : 0.f; });
    else if (a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU)
      run_epilogue([slope](float v) { return v >= 0.f ? v : v * slope; });
    else
      run_epilogue([&](float v) { return act_apply(a.act, v, slope); });
  } else {
    if (nchw) {
      // fp32 NCHW store straight from the accumulators (last layers: Cout <= 16, so this is a small tensor)
  #pragma unroll
      for (int k = 0; k < TN; ++k) {
        const int col = n0 + wn * WN + 16 * k + r16;
        const bool colok = col < a.cout;
        const float bv = bias_of(k, 0);
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + 16 * i + 4 * g + r;
            const float o = act_apply(a.act, acc[i][k][r] + bv, slope);
            if (m < a.M && colok) {
              int n = fd_div(a.fd_ghw, m);
              int rem = m - n * (a.GH * a.GW);
              int gy = fd_div(a.fd_gw, rem);
              int gx = rem - gy * a.GW;
              int oy = gy * a.osy + a.ooy, ox = gx * a.osx + a.oox;
              a.out_f32[(((size_t)n * a.cout + col) * a.OH + oy) * a.OW + ox] = o;
            }
          }
        }
      }
      return;
    }

    // the activation is selected ONCE (wave-uniform) and the 16-element-per-tile loop is instantiated per choice,
    // so the hot loop carries no per-element branches
    auto epilogue = [&](auto actf) {
  #pragma unroll
      for (int k = 0; k < TN; ++k) {
        const int ct = wn * WN + 16 * k + r16;   // column inside the block tile
        const int col = n0 + ct;
        const bool colok = col < a.cout;
        const float bv = bias_of(k, 0);
        float s1 = 0.f, s2 = 0.f;
  #pragma unroll
        for (int i = 0; i < TM; ++i) {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = wm * WM + 16 * i + 4 * g + r;
            const float v = acc[i][k][r] + bv;
            const float vm = (m0 + row < a.M && colok) ? v : 0.f;   // statistics ignore tail rows / pad columns
            s1 += vm;
            s2 += vm * vm;
            const float o = colok ? actf(v) : 0.f;
            *reinterpret_cast<unsigned short*>(sC + row * C_STRIDE + ct * 2) = f2h<DT>(o);
          }
        }
        if (do_stats) {
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
    };
    if (a.act == DSR_ACT_NONE)
      epilogue([](float v) { return v; });
    else if (a.act == DSR_ACT_RELU)
      epilogue([](float v) { return v > 0.f ? v : 0.f; });
    else if (a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU)
      epilogue([slope](float v) { return v >= 0.f ? v : v * slope; });
    else
      epilogue([&](float v) { return act_apply(a.act, v, slope); });
  }
  __syncthreads();

  if (BM % 128 == 0 && do_stats) {       // (the 224-row tile is only launched without statistics)
    for (int c = tid; c < 2 * BN; c += NT) {
      int which = c / BN, ct = c % BN;
      int col = n0 + ct;
      if (col < a.cout) {
        // one statistics row per 128 tile rows (dsr_conv_stats_rows), whatever BM is
        constexpr int WPR = 128 / WM;                 // wave rows per statistics row
#pragma unroll
        for (int h = 0; h < BM / 128; ++h) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < WPR; ++w) s += sStat[((h * WPR + w) * 2 + which) * BN + ct];
          if (BM == 128 || m0 + 128 * h < a.M)
            a.stats[((size_t)(tile_m * (BM / 128) + h) * 2 + which) * a.stats_stride + col] = s;
        }
      }
    }
  }
  unsigned short* __restrict__ Y = reinterpret_cast<unsigned short*>(a.y);
  if (!(a.flags & DSR_F_PIXSHUF)) {
    constexpr int CH = BN / 8;
    // Dense output (stride-1 forward / dgrad: grid pixel m IS output pixel m) and a tile fully inside the matrix:
    // vector idx lives at (m0 + idx / CH) * CoutP + n0 + (idx % CH) * 8 -- one scalar per tile, one per-lane constant and a
    // scalar step per pass, instead of two magic divisions and a 64-bit address per store.
    if (a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.GH == a.OH && a.GW == a.OW && m0 + BM <= a.M &&
        n0 + BN <= a.CoutP && (BM * CH) % NT == 0 && (size_t)a.M * a.CoutP * 2 < 0xFFFFFF00ull) {
      const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)a.M * a.CoutP * 2), 0x00020000);
      const unsigned sorg = (unsigned)(((size_t)m0 * a.CoutP + n0) * 2);
      const unsigned lpart = (unsigned)((tid / CH) * a.CoutP * 2 + (tid % CH) * 16);
      const unsigned step = (unsigned)((NT / CH) * a.CoutP * 2);
      const unsigned char* src = sC + (tid / CH) * C_STRIDE + (tid % CH) * 16;
      if (a.mask_x) {       // (uniform) dgrad with the upstream activation's backward mask folded into the stores
        const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.mask_x), 0, (unsigned)((size_t)a.M * a.CoutP * 2), 0x00020000);
        // (the mask's activation as a constant per branch: with the run-time value every element went through
        //  act_grad_from_out's chain of compares)
        auto masked_stores = [&](auto maskf) {
#pragma unroll
          for (int it = 0; it < (BM * CH) / NT; ++it) {
            const U4 v = *reinterpret_cast<const U4*>(src + it * (NT / CH) * C_STRIDE);
            const U4 o = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(mr, sorg + lpart + it * step, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, maskf(v, o)), yr,
                                                   sorg + lpart + it * step, 0, 0);
          }
        };
        if (a.mask_act == DSR_ACT_RELU)
          masked_stores([](const U4& v, const U4& o) { return act_mask8<DT>(v, o, DSR_ACT_RELU, 0.f); });
        else if (a.mask_act == DSR_ACT_LEAKY || a.mask_act == DSR_ACT_PRELU)
          masked_stores([&](const U4& v, const U4& o) { return act_mask8<DT>(v, o, DSR_ACT_LEAKY, a.mask_slope); });
        else
          masked_stores([&](const U4& v, const U4& o) { return act_mask8<DT>(v, o, a.mask_act, a.mask_slope); });
        return;
      }
#pragma unroll
      for (int it = 0; it < (BM * CH) / NT; ++it) {
        const U4 v = *reinterpret_cast<const U4*>(src + it * (NT / CH) * C_STRIDE);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yr,
                                               sorg + lpart + it * step, 0, 0);
      }
      return;
    }
    for (int idx = tid; idx < BM * CH; idx += NT) {
      int row = idx / CH, ch = idx % CH;
      int m = m0 + row, col0 = n0 + ch * 8;
      if (m < a.M && col0 < a.CoutP) {
        int n = fd_div(a.fd_ghw, m);
        int rem = m - n * (a.GH * a.GW);
        int gy = fd_div(a.fd_gw, rem);
        int gx = rem - gy * a.GW;
        int oy = gy * a.osy + a.ooy, ox = gx * a.osx + a.oox;
        size_t off = ((size_t)(n * a.OH + oy) * a.OW + ox) * a.CoutP + col0;
        U4 v = *reinterpret_cast<const U4*>(sC + row * C_STRIDE + ch * 16);
        if (a.mask_x)
          v = act_mask8<DT>(v, *reinterpret_cast<const U4*>(reinterpret_cast<const unsigned short*>(a.mask_x) + off), a.mask_act, a.mask_slope);
        *reinterpret_cast<U4*>(Y + off) = v;
      }
    }
  } else {
    // PixelShuffle(2): conv channel 4c+2i+j of grid pixel (h,w) -> channel c of pixel (2h+i, 2w+j)
    if constexpr (BN >= 32) {
      constexpr int CQ = BN / 32;   // 8-channel output chunks per sub-pixel in this tile
      for (int idx = tid; idx < BM * 4 * CQ; idx += NT) {
        int row = idx / (4 * CQ);
        int rem2 = idx % (4 * CQ);
        int sub = rem2 / CQ, cq = rem2 % CQ;
        int m = m0 + row;
        int oc0 = n0 / 4 + cq * 8;
        if (m < a.M && oc0 < a.CoutP) {
          int n = fd_div(a.fd_ghw, m);
          int rem = m - n * (a.GH * a.GW);
          int gy = fd_div(a.fd_gw, rem);
          int gx = rem - gy * a.GW;
          int oy = 2 * gy + (sub >> 1), ox = 2 * gx + (sub & 1);
          const unsigned short* src = reinterpret_cast<const unsigned short*>(sC + row * C_STRIDE);
          unsigned short v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = src[4 * (cq * 8 + q) + sub];
          U4 o;
          o.x = v[0] | ((unsigned)v[1] << 16);
          o.y = v[2] | ((unsigned)v[3] << 16);
          o.z = v[4] | ((unsigned)v[5] << 16);
          o.w = v[6] | ((unsigned)v[7] << 16);
          size_t off = ((size_t)(n * a.OH + oy) * a.OW + ox) * a.CoutP + oc0;
          *reinterpret_cast<U4*>(Y + off) = o;
        }
      }
    }
  }
}

template <int DT, int BM, int BN, int WGM, int WGN, bool FAST, bool DMA, int NSTAGE, bool SWAP, bool PADX = false>
static void launch_swap(dim3 grid, const ConvGemmArgs& b, hipStream_t st) {
  constexpr int LDS = ConvGemmLds<BM, BN, WGM, NSTAGE>::TOTAL;
  auto* fn = conv_gemm_kernel<DT, BM, BN, WGM, WGN, FAST, DMA, NSTAGE, SWAP, PADX>;
  if constexpr (LDS > 64 * 1024) {   // more than 64 KB of dynamic LDS needs the opt-in, once per kernel (not a stream op)
    static LdsOptIn optin;
    optin.ensure((const void*)fn, LDS);
  }
  hipLaunchKernelGGL(fn, grid, dim3(64 * WGM * WGN), LDS, st, b);
}

// Launches that want BatchNorm statistics keep the pixel-major accumulator layout (a channel's column sum is then an
// in-lane sum plus two shuffles); all others use the channel-major one (SWAP: packed 8-byte C-tile writes).
template <int DT, int BM, int BN, int WGM, int WGN, bool FAST, bool DMA, int NSTAGE, bool PADX = false>
static void launch_variant(dim3 grid, const ConvGemmArgs& b, hipStream_t st) {
  if constexpr (DMA) {
    // (the 256x256 tile has no register room for the pixel-major epilogue: it takes its statistics from the
    //  channel-major accumulators -- 16-lane shuffle sums -- as well)
    if (!(b.flags & DSR_F_STATS) || BM >= 224) {
      launch_swap<DT, BM, BN, WGM, WGN, FAST, DMA, NSTAGE, true, PADX>(grid, b, st);
      return;
    }
  }
  // statistics launches, and the register-staged variants (measured 10 % slower with the channel-major epilogue)
  if constexpr (BM < 224) launch_swap<DT, BM, BN, WGM, WGN, FAST, DMA, NSTAGE, false, PADX>(grid, b, st);
}

static bool env_on(const char* name) {   // tuning switches, default on ("0" turns one off)
  const char* e = getenv(name);
  return !(e && e[0] == '0');
}

template <int DT, int BM, int BN, int WGM, int WGN, int NSTAGE = 2>
static void launch_one(const ConvGemmArgs& a, hipStream_t st) {
  ConvGemmArgs b = a;
  b.tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.NB + BN - 1) / BN;
  dim3 grid(b.tiles_m * b.tiles_n);
  const bool fast = a.pad_mode == DSR_PAD_ZERO && (a.CU & 7) == 0 && a.ntaps > 0;
  if constexpr (NSTAGE == 3) {
    launch_variant<DT, BM, BN, WGM, WGN, true, true, 3>(grid, b, st);
    return;
  } else {
    if constexpr (BN >= 64) {
      static const bool use_dma = env_on("DSR_CONV_DMA");      // 0 = register-staged loader
      if (fast && use_dma) {
        launch_variant<DT, BM, BN, WGM, WGN, true, true, 2>(grid, b, st);
        return;
      }
      if constexpr (BM == 128 && BN == 128) {                 // reflect / replicate padding on the DMA path (the 128-channel DIP layers)
        const bool padx = env_on("DSR_CONV_PADX");            // 0 = the register-staged generic loader (read per call: a test compares the two)
        if (padx && use_dma && a.pad_mode != DSR_PAD_ZERO && (a.CU & 7) == 0 && a.ntaps > 0) {
          launch_variant<DT, BM, BN, WGM, WGN, true, true, 2, true>(grid, b, st);
          return;
        }
      }
    }
    if constexpr (BM < 224) {       // (the 8-wave tiles exist as the LDS-DMA fast path only; dispatch_dt guarantees it)
      if (fast)
        launch_variant<DT, BM, BN, WGM, WGN, true, false, 2>(grid, b, st);
      else
        launch_variant<DT, BM, BN, WGM, WGN, false, false, 2>(grid, b, st);
    }
  }
}

template <int DT>
static void dispatch_dt(const ConvGemmArgs& a, hipStream_t st) {
  if (a.NB > 64) {
    // big problems: 256x128 tiles, 8 waves, three-stage DMA ring, one block per CU (needs >= 2 blocks per CU of work)
    // measured: 5-12 % SLOWER than two resident 128x128 blocks on every config-3 layer, so it is off unless asked for
    const bool use_big = true;
    const bool fast = a.pad_mode == DSR_PAD_ZERO && (a.CU & 7) == 0 && a.ntaps > 0;
    const long long big_tiles = (long long)((a.M + 255) / 256) * ((a.NB + 127) / 128);
    // 256x256 tile, 8 waves of 128x64 (2 x 4), two stages, one block per CU: per MFMA half the LDS fragment reads
    // (0.375 ds_read_b128 per MFMA instead of 0.5) and half the DMA pieces of the 128x128 tile
    const int big_mode = dsr_conv_big_mode();
    if (dsr_conv_gemm_use_224(a.M, a.NB, fast && env_on("DSR_CONV_DMA"), a.flags))
      launch_one<DT, 224, 256, 2, 4>(a, st);
    else if (dsr_conv_gemm_use_256(a.M, a.NB, fast && env_on("DSR_CONV_DMA"), (a.flags & DSR_F_STATS) != 0))
      launch_one<DT, 256, 256, 2, 4>(a, st);
    else if (use_big && big_mode == 1 && fast && big_tiles >= 512)
      launch_one<DT, 256, 128, 4, 2, 3>(a, st);
    else if (dsr_conv_gemm_use_64(a.M, a.NB, fast && env_on("DSR_CONV_DMA"), a.flags))
      launch_one<DT, 64, 128, 1, 4>(a, st);        // few 128-row tiles (VGG conv5_x at batch 32: 196 blocks on 256 CUs): twice as many half-height tiles
    else
      launch_one<DT, 128, 128, 2, 2>(a, st);      // (8 waves of 64x32 were tried: LDS-bound, 35 % slower)
  } else if (a.NB > 16)
    launch_one<DT, 128, 64, 2, 2>(a, st);
  else
    launch_one<DT, 128, 16, 4, 1>(a, st);
}

void dsr_launch_conv_gemm(const ConvGemmArgs& a, int dtype, hipStream_t st) {
  if (!a.mask_x) {          // (the persistent kernels keep a DMA in flight across their epilogue: no masked form)
    if (dsr_launch_conv_gemm_persist(a, dtype, st)) return;   // many-tile fast-path launches: persistent kernel
  }
  if (dtype == DSR_DTYPE_BF16)
    dispatch_dt<DSR_DTYPE_BF16>(a, st);
  else
    dispatch_dt<DSR_DTYPE_F16>(a, st);
}
