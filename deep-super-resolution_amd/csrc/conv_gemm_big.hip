// Persistent form of the 256 x 256 tile of the gather convolution (conv_gemm.hip: 8 waves of 128 x 64, LDS-DMA fast path,
// two 64 KB stages, ONE resident block per CU) for dense-output launches with more tiles than CUs.
//
// Why: with one block per CU nothing overlaps a tile's fixed cost -- measured on VGG 256->512 / 512->512 at 28x28 (same M
// and N, K = 2304 vs 4608, one tile per CU): 71 vs 122 us, i.e. 1.4 us per K-step and ~20 us per tile that are not K loop
// (launch, first-step DMA latency, epilogue arithmetic, 128 KB of output stores draining while no MFMA runs).  Here a block
// walks tiles t, t + grid, ...; at a tile boundary it
//   1. starts the LDS-DMA of the NEXT tile's first K-step into the stage the last K-step did not read (free already),
//   2. writes its C tile in two halves of 128 rows x 512 B (= exactly one stage, XOR-swizzled 16-byte chunks) through the
//      other stage and issues the output as unconditional buffer stores (masked lanes: out-of-range offset, dropped),
//   3. enters the next K loop with `s_waitcnt vmcnt(NSTORE)`: vmcnt retires in issue order and the DMA was issued before
//      the stores, so this waits for the DMA only and the stores drain under the next tile's MFMAs.
// Fragment layout, swizzle, K order (=> bit-identical outputs), epilogue semantics and statistics rows are the contract
// of conv_gemm_kernel<256,256,...,SWAP = true>; tests run both kernels on the same shapes.
// MEASURED SLOWER than one tile per block (see dsr_launch_conv_gemm_big below): kept as an opt-in experiment.
#include <stdlib.h>

#include <type_traits>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int BM = 256, BN = 256, WGM = 2, WGN = 4, NW = 8, NT = 512, RPP = NT / 8;
constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
constexpr int RA = BM / RPP, RB = BN / RPP;
constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128, STAGE = A_STAGE + B_STAGE;     // 64 KB
constexpr int CH = BN / 8;                                                           // 16-byte chunks per C row (32)
constexpr int NSTORE_HALF = 128 * CH / NT;                                           // 8 stores per thread and half tile
constexpr int NSTAT = 2;                                                             // statistics-row stores per thread and tile
constexpr int MAX_NB = 1024;                                                         // bias table held in LDS for the whole launch
constexpr int LDS_TOTAL = 2 * STAGE + WGM * 2 * BN * 4 + DSR_MAX_TAPS * 4 + MAX_NB * 4;
static_assert(128 * BN * 2 == STAGE, "half a C tile is exactly one stage");
}  // namespace

template <int DT>
__global__ __launch_bounds__(NT, 2) void conv_gemm_big_kernel(const ConvGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // stage 0 | stage 1 | statistics | taps | bias
  float* sStat = reinterpret_cast<float*>(smem + 2 * STAGE);
  int* sTaps = reinterpret_cast<int*>(smem + 2 * STAGE + WGM * 2 * BN * 4);
  // the bias of every output column, once per launch: a global load inside the epilogue would be waited for with vmcnt, which
  // retires in order -- behind the next tile's DMA
  float* sBias = reinterpret_cast<float*>(smem + 2 * STAGE + WGM * 2 * BN * 4 + DSR_MAX_TAPS * 4);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  const int j = tid & 7, rb = tid >> 3;
  const int jc = j ^ (rb & 7);                           // source-side swizzle of the lane-linear DMA image
  const int sw = r16 & 7;

  for (int i = tid; i < a.ntaps; i += NT) sTaps[i] = a.taps[i];
  for (int i = tid; i < a.NB; i += NT) sBias[i] = ((a.flags & DSR_F_BIAS) && i < a.cout) ? a.bias[i] : 0.f;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.stats, 0, (a.flags & DSR_F_STATS) ? (unsigned)((size_t)a.tiles_m * 2 * 2 * a.stats_stride * 4) : 0u, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
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
    const int tile_n = bid % a.tiles_n;
    tile_m = bid / a.tiles_n;
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
    for (int i = 0; i < RB; ++i) b_base[i] = ((n0 + rb + RPP * i) * a.CinP + jc * 8) * 2;     // NB % 256 == 0: every row exists
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
  // half a C tile: 128 rows x 512 B, 16-byte chunk c of row r at chunk position c ^ (r & 31)
  auto c_off = [](int row, int colbyte) { return row * 512 + ((((colbyte >> 4) ^ (row & 31))) << 4) + (colbyte & 15); };

  const unsigned st_lpart = (unsigned)(((tid / CH) * a.CoutP + (tid % CH) * 8) * 2);
  const unsigned st_step = (unsigned)((NT / CH) * a.CoutP * 2);
  const unsigned st_wrow = (unsigned)(128 * a.CoutP * 2);          // tile rows of wave row 1 start 128 rows further down
  const bool late_dma = wave >= NW / 2;
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
    TapStep nd = decode_step(ks > 1 ? 1 : 0);
    for (int s = 0; s < ks; ++s) {
      // own DMA of this step done.  At a tile boundary the previous tile's output / statistics stores were issued AFTER this
      // step's DMA: vmcnt retires in order, so vmcnt(number of those stores) waits for the DMA and lets the stores drain.
      if (s == 0 && !first)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSTORE_HALF + NSTAT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int cur = st0 ^ (s & 1);
      const unsigned char* pa = smem + cur * STAGE + (wm * WM + r16) * 128;
      const unsigned char* pb = smem + cur * STAGE + A_STAGE + (wn * WN + r16) * 128;
      const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
      // the software pipeline of conv_gemm.hip's 8-wave loop: A fragment of unit u + 2 requested before the MFMAs of unit u
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
        for (int k = 0; k < TN; ++k) acc[u % TM][k] = mfma16<DT>(u < TM ? fb0[k] : fb1[k], fa[u % 3], acc[u % TM][k]);
        __builtin_amdgcn_sched_barrier(0);
        if (u == TM - 1) {
          if (late_dma && s + 1 < ks) dma_issue(nd, cur ^ 1);
          if (s + 2 < ks) nd = decode_step(s + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    const int L = st0 ^ ((ks - 1) & 1);                  // stage of the last K-step: becomes the C staging area
    const int cm0 = m0, cn0 = n0, ctile_m = tile_m;      // this tile's coordinates (setup_tile below overwrites them)

    // ---- next tile: loader state + the DMA of its first K-step into stage L ^ 1.  Every wave passed the barrier of the last
    // K-step, so all reads of stage L ^ 1 (K-step ks - 2) are complete: the stage is free without another barrier.
    const int tn = t + gridDim.x;
    const bool has_next = tn < total;
    if (has_next) {
      setup_tile(tn);
      dma_issue(d0, L ^ 1);
    }
    unsigned char* sC = smem + L * STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every wave is done reading stage L
    asm volatile("" ::: "memory");

    // ---- epilogue: acc[i][k][jj] = out[row = wm*128 + 16i + r16][col = wn*64 + 16k + 4g + jj]; half h = rows of wave row wm = h.
    // One copy of the code (activation and statistics selected by wave-uniform branches): the tile loop around it leaves no
    // room for the per-activation instantiations of conv_gemm.hip.
    const int act = a.act;
    auto actf = [&](float v) {
      if (act == DSR_ACT_NONE) return v;
      if (act == DSR_ACT_RELU) return v > 0.f ? v : 0.f;
      if (act == DSR_ACT_LEAKY || act == DSR_ACT_PRELU) return v >= 0.f ? v : v * slope;
      return act_apply(act, v, slope);
    };
    // Two passes of 128 rows through the ONE free stage; every wave takes part in both: pass h holds fragment rows
    // i = 4h .. 4h+3 of both wave rows, i.e. tile rows wm*128 + 64h + (0..63) at staging rows wm*64 + (0..63).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int k = 0; k < TN; ++k) {                     // (unrolled: the accumulators must stay statically indexed)
        const int ct0 = wn * WN + 16 * k + 4 * g;
        float bv[4], s1[4], s2[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          bv[jj] = sBias[cn0 + ct0 + jj];
          s1[jj] = s2[jj] = 0.f;
        }
        // staging rows wm*64 + 16 ii + r16: the swizzle key (row & 31) has two values, everything else is an immediate
        const int cb0 = c_off(wm * 64 + r16, ct0 * 2), cb1 = c_off(wm * 64 + 16 + r16, ct0 * 2);
#pragma unroll
        for (int ii = 0; ii < TM / 2; ++ii) {
          const int i = 4 * h + ii;
          float o[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {               // (the launcher takes M % 256 == 0 and cout == NB only: no tail masks)
            const float v = acc[i][k][jj] + bv[jj];
            s1[jj] += v;
            s2[jj] = __builtin_fmaf(v, v, s2[jj]);
            o[jj] = actf(v);
          }
          uint2 hh;
          hh.x = (unsigned)f2h<DT>(o[0]) | ((unsigned)f2h<DT>(o[1]) << 16);
          hh.y = (unsigned)f2h<DT>(o[2]) | ((unsigned)f2h<DT>(o[3]) << 16);
          *reinterpret_cast<uint2*>(sC + ((ii & 1) ? cb1 : cb0) + (ii >> 1) * 32 * 512) = hh;
        }
        if (do_stats) {                                  // this pass's 64 rows: pass 0 sets the wave row's sums, pass 1 adds to them
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int x = 1; x < 16; x <<= 1) {           // over the 16 pixels (lanes r16) of the fragment
              s1[jj] += __shfl_xor(s1[jj], x, 64);
              s2[jj] += __shfl_xor(s2[jj], x, 64);
            }
          }
          if (r16 == 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              float* p1 = &sStat[(wm * 2 + 0) * BN + ct0 + jj];
              float* p2 = &sStat[(wm * 2 + 1) * BN + ct0 + jj];
              *p1 = h == 0 ? s1[jj] : *p1 + s1[jj];
              *p2 = h == 0 ? s2[jj] : *p2 + s2[jj];
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // 128 rows (and, after pass 1, the statistics) complete
      asm volatile("" ::: "memory");
      // exactly NSTORE_HALF buffer stores per thread: staging row r -> tile row (r >> 6) * 128 + 64 h + (r & 63)
      const unsigned sorg = (unsigned)(((size_t)(cm0 + 64 * h) * a.CoutP + cn0) * 2);     // (scalar) origin of this pass
#pragma unroll
      for (int q = 0; q < NSTORE_HALF; ++q) {
        const int row = tid / CH + (NT / CH) * q;        // staging row: 16 q + (0..15); rows >= 64 belong to wave row 1
        const U4 v = *reinterpret_cast<const U4*>(sC + c_off(row, (tid % CH) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrsrc,
                                               sorg + st_lpart + (q & 3) * st_step + (q >> 2) * st_wrow, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // the staging area has been read: the next pass may overwrite it
      asm volatile("" ::: "memory");
    }
    // ---- statistics rows: one per 128 tile rows (dsr_conv_stats_rows), exactly NSTAT buffer stores per thread
#pragma unroll
    for (int h = 0; h < NSTAT; ++h) {
      const int which = tid / BN, ct = tid % BN;         // 512 threads = 2 x 256 columns
      const int col = cn0 + ct;
      const bool ok = do_stats;
      const float sv = sStat[(h * 2 + which) * BN + ct];
      const unsigned off = (unsigned)((((size_t)(ctile_m * 2 + h) * 2 + which) * a.stats_stride + col) * 4);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv), srsrc, ok ? off : OOB, 0, 0);
    }
    if (!has_next) break;
    t = tn;
    st0 = L ^ 1;
    first = false;
  }
}

// returns false when the launch does not qualify (caller falls through to the one-tile-per-block kernel)
bool dsr_launch_conv_gemm_big(const ConvGemmArgs& a, int dtype, hipStream_t st) {
  // OFF unless DSR_CONV_BIG_PERSIST=1 (read per call: the test flips it).  Measured on D.b3 / b4 / b5 forward with statistics
  // (profiles/r02_persistent_256.txt): 0.405 / 0.177 / 0.317 ms against 0.322 / 0.148 / 0.259 ms for one tile per block.
  // The C tile of this shape is both LDS stages, so the epilogue here runs in two passes through one stage (four extra
  // barriers, statistics reduced twice), and vmcnt retires in order: the second K-step's DMA cannot be waited for without
  // the 16 output stores in front of it, so the drain is not hidden either.
  const char* e = getenv("DSR_CONV_BIG_PERSIST");
  if (!(e && e[0] == '1')) return false;
  const bool fast = a.pad_mode == DSR_PAD_ZERO && (a.CU & 7) == 0 && a.ntaps > 0;
  if (!dsr_conv_gemm_use_256(a.M, a.NB, fast, (a.flags & DSR_F_STATS) != 0)) return false;
  if (a.flags & (DSR_F_OUT_NCHW_F32 | DSR_F_PIXSHUF)) return false;
  const bool dense_out = a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 && a.GH == a.OH && a.GW == a.OW;
  if (!dense_out || a.NB % 256 != 0 || a.NB > MAX_NB || a.CoutP < a.NB) return false;
  if (a.M % BM != 0 || a.cout != a.NB) return false;     // no tail rows / pad columns: the epilogue carries no masks
  if ((size_t)a.M * a.CoutP * 2 >= 0xFFFFFF00ull) return false;
  ConvGemmArgs b = a;
  b.tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = a.NB / BN;
  const long long total = (long long)b.tiles_m * b.tiles_n;
  if (total <= 256) return false;                        // one tile per CU: nothing to overlap
  if ((a.flags & DSR_F_STATS) && (size_t)b.tiles_m * 2 * 2 * a.stats_stride * 4 >= 0xFFFFFF00ull) return false;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_big_kernel<DSR_DTYPE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    (void)hipFuncSetAttribute((const void*)conv_gemm_big_kernel<DSR_DTYPE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_done = true;
  }
  const int blocks = 256;
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_gemm_big_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(NT), LDS_TOTAL, st, b);
  else
    hipLaunchKernelGGL((conv_gemm_big_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(NT), LDS_TOTAL, st, b);
  return true;
}
