// 3x3 / stride 1 / zero-pad convolution with 64 input and 64 output channels -- the SRGAN trunk
// (models/GAN/generator.py:7,11,52: 33 forward + 33 input-grad launches per generator pass) and every other
// 64->64 layer (VGG conv1_2).  The generic gather kernel spends this shape's short K loop (9 steps) mostly in its
// prologue/epilogue and re-reads weights and input per K-step; here
//   * the block is persistent and its waves keep ALL weight fragments of their 32-channel half in registers
//     (9 taps x 2 k-halves x 2 n-tiles = 36 fragments = 144 VGPRs) for the whole launch,
//   * the 2-row x 32-column pixel tile's input HALO (4 x 34 pixels) is staged once in LDS and every tap is an
//     address offset into it (A fragments: one conflict-free ds_read_b128 per 2 MFMAs),
//   * the NEXT tile's halo travels HBM -> LDS by LDS-DMA (no VGPRs) under the current tile's 72 MFMAs per wave,
// so the MFMA phase touches LDS only.  Same epilogue contract as conv_gemm (bias, activation, BatchNorm
// sum / sum-of-squares rows -- one row per spatial TILE here --, 16-byte NHWC stores).  dgrad = the same kernel on the
// [tap][ci][co] weight image with mirrored tap offsets.
#include <cstdlib>
#include <type_traits>

#include "dsr_common.h"
#include "dsr_kernels.h"

// Development aid (-DDSR_C64_STAMPS): s_memtime deltas of the phases of a tile, accumulated per wave and written at the
// end of the kernel to a buffer of their own (dsr_debug_c64 copies it out); no output value depends on them.  The stamps
// themselves cost ~100 cycles each, so use the build for the SPLIT between phases, not for absolute speed.
#ifdef DSR_C64_STAMPS
__device__ unsigned long long g_c64_stamps[16];
extern "C" int dsr_debug_c64(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c64_stamps), sizeof(g_c64_stamps));
}
#define STAMP(k) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define STAMP(k)
#endif
// one v_max_f32 (fmaxf adds a canonicalising v_max(x, x) per operand)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// MODE 0: pixel-major accumulators, can emit BatchNorm statistics; 1: swapped accumulators (no statistics);
// 2: swapped + FOLD, the inference epilogue (eval-mode BatchNorm scale/shift, residual); 3: swapped + residual only, with
// the residual tile REQUESTED AT THE TOP of the tile, in front of the next tile's DMA (dsr_conv_dgrad_add: vmcnt retires in
// order, so a load issued in the epilogue would wait for that DMA -- measured 82 us per trunk dgrad against 42 + 32 us for
// dgrad and a separate add; mode 2 does the same since its scale / shift moved to LDS).  Separate instantiations: the
// layouts need different per-lane constants and the kernel has no registers to spare.
// Tile = TR rows x 32 columns.  A wave owns a 16-COLUMN strip (wq) of all TR rows and a 32-channel half (wc): its m-tiles are
// the tile's rows.  An A fragment (16 pixels of one halo row at one tap column) then serves up to three output rows
// (tap rows 0, 1, 2), so a tile costs 6 (TR + 2) fragment reads for 36 TR MFMAs per wave -- 0.25 ds_read_b128 per MFMA at
// TR = 4, against 0.5 when a wave owned one row of two 16-pixel halves and every fragment fed one row only.  (s_memtime
// stamps of that form: the MFMA phase ran at 59 % of the single-wave MFMA rate with the LDS ~saturated by two blocks.)
template <int DT, int MODE, int TR>
__global__ __launch_bounds__(256, 2) void conv_c64_kernel(const C64Args a) {
  constexpr int HC = 40;                       // halo row pitch in pixels (34 used; multiple of 8 keeps the swizzle row-free)
  constexpr int HRW = TR + 2;                  // halo rows
  constexpr int X_BYTES = HRW * HC * 128;      // 20,480 (TR = 2) | 30,720 (TR = 4)
  constexpr int C_STRIDE = 64 * 2 + 16;
  // one LDS object (a second one beside an LDS-DMA target makes hipcc drain the DMA before every LDS read):
  // two halo stages | C tile | statistics (dynamic: 80,896 bytes at TR = 4, two blocks per CU = 161,792 of 163,840)
  constexpr int C_BYTES = TR * 32 * C_STRIDE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sC = smem + 2 * X_BYTES;
  float (*sStat)[2][64] = reinterpret_cast<float (*)[2][64]>(smem + 2 * X_BYTES + C_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it (and what derives from it) in SGPRs
  const int g = lane >> 4, r16 = lane & 15;
  const int wq = wave >> 1, wc = wave & 1;     // 16-column strip of the tile, channel half (32 co)
  const int c0 = blockIdx.y * 64;              // this block's slice of the output channels (Cout = 64 * gridDim.y)
  const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(a.w);
  unsigned short* __restrict__ Y = reinterpret_cast<unsigned short*>(a.y);

  // Two accumulator layouts (uniform per launch).  Launches that emit BatchNorm statistics keep pixels on the MFMA's M
  // axis: a lane then holds 4 PIXELS of one channel and the per-channel sums need two cross-row shuffles.  All other
  // launches swap the operands (weights on M): a lane holds 4 consecutive CHANNELS of one pixel, which leave as one
  // 8-byte LDS write after two packed conversions instead of four 2-byte writes.  For PixelShuffle(2) launches the
  // weight rows of each 16-channel tile are taken in the order m -> 4 (m & 3) + (m >> 2), so that those 4 values are
  // the 4 consecutive output channels of ONE sub-pixel (conv channel 4c + s -> sub-pixel s, channel c).
  constexpr bool FOLD = MODE == 2;
  constexpr bool RES_EARLY = MODE == 3 || MODE == 2;   // residual tile requested at the top of the tile (see above)
  constexpr bool swp = MODE != 0;
  const bool do_stats = MODE == 0 && (a.flags & DSR_F_STATS) != 0;
  const bool pixshuf = (a.flags & DSR_F_PIXSHUF) != 0;
  const int wrow = (swp && pixshuf) ? 4 * (r16 & 3) + (r16 >> 2) : r16;
  // ---- weights: registers, once (rows = output channels wc*32 + nt*16 + wrow).  Requested FIRST: they are the long pole of
  // the prologue (36 x 16 B per lane, 2 us through the L1), and the integer set-up below runs under their latency (requested
  // behind the first tile's DMA instead, every launch of a few tiles per block measured 2 us slower)
  U4 fw[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        fw[t][kk][nt] = *reinterpret_cast<const U4*>(W + ((size_t)(t * a.CoutP + c0 + wc * 32 + nt * 16 + wrow)) * 64 + kk * 32 + g * 8);

  // A-fragment LDS offsets: halo column (16 wq + tx + r16) -> *128 + swizzled chunk (key = column & 7); + halo row * HC * 128
  int lds_off[3][2];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) lds_off[tx][kk] = (16 * wq + tx + r16) * 128 + (((4 * kk + g) ^ ((tx + r16) & 7)) << 4);

  // loader: LDS-DMA (buffer_load ... lds): wave-instruction u of wave w fills halo slots 32u + 8w .. +7 (8 pixels x
  // 128 B, lane-linear), slot position (lane & 7) of pixel q holding channel chunk (lane & 7) ^ (q & 7); an
  // out-of-range offset writes zeros.  The tile never passes through VGPRs.
  // Everything outside the MFMA phase competes with the partner block's MFMAs for the SIMD's issue port (s_memtime
  // stamps: MFMA phase 30 % of a tile, fetch 17 %, epilogue + stores 50 %), so per-tile vector work is kept minimal:
  // every address is a tile-invariant per-lane part (computed once per launch) plus a per-tile SCALAR part, the image
  // ROWS outside [0, H) are rejected by the buffer range check of a per-image resource, and the COLUMN check runs only
  // for tiles that touch the left / right image edge.
  constexpr int NV = (HRW * HC + 31) / 32;     // 5 | 8 (the last one covers 16 slots only at TR = 4: waves 2, 3 sit it out)
  constexpr int LAST_WAVES = ((HRW * HC) % 32 == 0) ? 4 : ((HRW * HC) % 32) / 8;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int FAR = 0x7FFFFF00;              // lane part of a slot that is never fetched: any scalar part leaves it out of range
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  const int per_img = a.tiles_y * a.tiles_x;
  const unsigned img_bytes = (unsigned)(a.H * a.W * 128);
  int ld_part[NV];                             // ((hr * W + hc) * 64 + chunk * 8) * 2 of this lane's slot
  typedef typename std::conditional<(6 * NV > 32), unsigned long long, unsigned>::type hc_pack_t;
  hc_pack_t hc_pack = 0;                       // its halo column, 6 bits per wave-instruction
  {
    const int c = (tid & 7) ^ ((tid >> 3) & 7), pb = tid >> 3;
    int hr = pb / HC, hc = pb - hr * HC;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      ld_part[u] = (hr < HRW && hc < 34) ? ((hr * a.W + hc) * 64 + c * 8) * 2 : FAR;   // pad columns 34..39 are never read
      hc_pack |= (hc_pack_t)hc << (6 * u);
      hc += 32;
      if (hc >= HC) {
        hc -= HC;
        ++hr;
      }
    }
  }
  // Tile coordinates (image, tile row, tile column) are carried along and advanced by the decomposed grid stride:
  // four integer divisions per tile were ~240 dependent scalar instructions, a sixth of a tile's time.
  struct TileXY {
    int n, ty, tx;
  };
  auto decomp = [&](int t) {
    TileXY c;
    c.n = t / per_img;
    const int rem = t - c.n * per_img;
    c.ty = rem / a.tiles_x;
    c.tx = rem - c.ty * a.tiles_x;
    return c;
  };
  const TileXY tstep_xy = decomp((int)gridDim.x);
  auto advance = [&](TileXY c) {
    c.tx += tstep_xy.tx;
    if (c.tx >= a.tiles_x) {
      c.tx -= a.tiles_x;
      ++c.ty;
    }
    c.ty += tstep_xy.ty;
    if (c.ty >= a.tiles_y) {
      c.ty -= a.tiles_y;
      ++c.n;
    }
    c.n += tstep_xy.n;
    return c;
  };
  auto fetch = [&](const TileXY& tc, int buf) {
    const int n = tc.n;
    const int oy0 = tc.ty * TR - 1, ox0 = tc.tx * 32 - 1;                         // all scalar
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x)) + (size_t)n * img_bytes, 0, img_bytes, 0x00020000);
    const int sbase = (oy0 * a.W + ox0) * 128;                                    // may be negative (row -1): out of range
    unsigned char* dst = smem + buf * X_BYTES + 8 * wave * 128;
    // (a DMA piece writes its 8 slots whatever the offsets are -- zeros for out-of-range ones --, so the waves whose slots of
    //  the last piece lie behind the stage must not issue it: wave-uniform)
    if (ox0 >= 0 && ox0 + 34 <= a.W) {                                            // interior columns: no per-lane check
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (u + 1 < NV || wave < LAST_WAVES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + 32 * u * 128), 16, (unsigned)(ld_part[u] + sbase), 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int ix = ox0 + (int)((hc_pack >> (6 * u)) & 63);
        if (u + 1 < NV || wave < LAST_WAVES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + 32 * u * 128), 16,
                                                   (unsigned)ix < (unsigned)a.W ? (unsigned)(ld_part[u] + sbase) : OOB, 0, 0, 0);
      }
    }
  };

  const float slope = (a.flags & DSR_F_PRELU_PTR) ? a.prelu[0] : a.slope;
  // channel of accumulator register r of n-tile nt: pixel-major layout: wc*32 + nt*16 + r16 (all r);
  // swapped: wc*32 + nt*16 + (4g + r), through the PixelShuffle row order: wc*32 + nt*16 + 4r + g
  auto chan = [&](int nt, int r) {
    return wc * 32 + nt * 16 + (!swp ? r16 : (pixshuf ? 4 * r + g : 4 * g + r));
  };
  constexpr int NB = swp ? 4 : 1;              // distinct channels among a lane's 4 accumulator registers
  // (FOLD launches keep no bias registers: the bias is folded into the shift of the affine map, see below)
  float bias_v[2][NB];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < NB; ++r) bias_v[nt][r] = (!FOLD && (a.flags & DSR_F_BIAS)) ? a.bias[c0 + chan(nt, r)] : 0.f;
  if constexpr (FOLD) {
    // inference: eval-mode BatchNorm folded in, y = act((conv + bias) * scale + shift) = act(conv * scale + shift') with
    // shift' = shift + bias * scale.  The per-channel pair lives in the (otherwise unused) statistics area of LDS: a load
    // from global memory in the epilogue would wait, in order, for the next tile's DMA.  Visible to
    // every wave after the first tile's barrier.
    if (tid < 64) {
      const bool aff = (a.flags & DSR_F_AFFINE) != 0;
      const float sc = aff ? a.scale[c0 + tid] : 1.f;
      const float bv = (a.flags & DSR_F_BIAS) ? a.bias[c0 + tid] : 0.f;
      sStat[0][0][tid] = sc;
      sStat[0][1][tid] = __builtin_fmaf(bv, sc, aff ? a.shift[c0 + tid] : 0.f);
    }
  }

  // C tile in LDS: pixel rows of 64 channels.  PixelShuffle(2) launches store channel 4c + s at byte s*32 + c*2, i.e.
  // already grouped by sub-pixel, so that the store loop reads whole 16-byte vectors in both layouts.
  int cw_base[2];                              // this lane's write address in the C tile for n-tile nt (tile pixel = 32 row + column):
#pragma unroll                                 //   pixel-major: pixel (16 wq + 4g [+ 32 i + r]), one channel;
  for (int nt = 0; nt < 2; ++nt) {             //   swapped: pixel (16 wq + r16 [+ 32 i]), 4 channels (8 bytes)
    const int col = wc * 32 + nt * 16 + r16;
    if constexpr (!swp)
      cw_base[nt] = (wq * 16 + 4 * g) * C_STRIDE + (pixshuf ? (col & 3) * 32 + (col >> 2) * 2 : col * 2);
    else
      cw_base[nt] = (wq * 16 + r16) * C_STRIDE + (pixshuf ? g * 32 + (wc * 8 + nt * 4) * 2 : (wc * 32 + nt * 16 + 4 * g) * 2);
  }
  // store loop: thread tid moves the 16-byte vectors idx = tid + 256 it of the tile (pixel idx >> 3, vector idx & 7): row `it`
  const int cr_base = (tid >> 3) * C_STRIDE + (tid & 7) * 16;          // + it * 32 * C_STRIDE
  int st_part;                                                          // byte offset of vector `tid` relative to the tile origin
  int st_step;                                                          // ... and of vector tid + 256 relative to vector tid
  if (!pixshuf) {
    st_part = ((tid >> 3) & 31) * a.CoutP * 2 + (tid & 7) * 16;       // prow = tid >> 3 (0..31): row 0 of the tile
    st_step = a.W * a.CoutP * 2;                                       // prow + 32: row 1
  } else {
    const int OCp = a.CoutP / 4;
    const int sub = (tid >> 1) & 3, cq = tid & 1, pxc = (tid >> 3) & 31;
    st_part = ((sub >> 1) * 2 * a.W + 2 * pxc + (sub & 1)) * OCp * 2 + cq * 16;
    st_step = 2 * (2 * a.W) * OCp * 2;                                 // conv row + 1 = output rows + 2
  }

  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tstep = gridDim.x;
  TileXY cur = decomp(t), nxt = advance(cur);
  if (t < a.ntiles) fetch(cur, 0);
  int buf = 0;
  // The weights and the first tile have to be there before the first MFMA anyway: wait HERE, with the builtin (an S_WAITCNT
  // the compiler's wait-count pass sees, unlike inline asm).  Otherwise the pass has to cover the weight loads itself and, not
  // knowing how many DMA pieces follow them, may put a vmcnt(0) in front of an MFMA INSIDE the tile loop -- where it also waits
  // for the next tile's DMA, every tile (it did in the inference instantiation).  0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef DSR_C64_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  float stat_acc = 0.f;
  for (; t < a.ntiles; t += tstep, buf ^= 1) {
    STAMP(0);
    // my DMA of this tile is done; after the barrier everyone's is, and every wave is past the previous tile.
    // The previous tile's TR output stores per thread (always issued: range-checked buffer stores) are younger than
    // this tile's DMA and stay in flight through the MFMA phase: vmcnt retires in order, so "at most 2 outstanding"
    // "at most TR outstanding" already means the DMA has landed.  A vmcnt(0) here exposed the full store latency per tile.
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TR) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(1);
    [[maybe_unused]] U4 rres[TR];
    if (RES_EARLY && (MODE == 3 || (a.flags & DSR_F_RESIDUAL))) {
      // this tile's residual vectors (the two this thread will store over), requested BEFORE the next tile's DMA
      const int rn = cur.n, roy0 = cur.ty * TR, rox0 = cur.tx * 32;
      const bool rfull = roy0 + TR <= a.H && rox0 + 32 <= a.W;
      const unsigned rorg = (unsigned)(((rn * a.H + roy0) * a.W + rox0) * a.CoutP * 2 + c0 * 2);
      const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res), 0, a.y_bytes, 0x00020000);
#pragma unroll
      for (int it = 0; it < TR; ++it) {
        unsigned off = rorg + (unsigned)(st_part + it * st_step);
        if (!rfull) {
          const int prow = (tid >> 3) + it * 32;
          if (!(roy0 + (prow >> 5) < a.H && rox0 + (prow & 31) < a.W)) off = OOB;
        }
        rres[it] = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
      }
    }
    if (t + tstep < a.ntiles) fetch(nxt, buf ^ 1);
    const unsigned char* sX = smem + buf * X_BYTES;
    STAMP(2);

    f32x4 acc[TR][2];                          // m-tile = tile row i (this wave's 16 columns of it) ; n-tile nt
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k)                // bias: free here
        acc[i][k] = f32x4{bias_v[k][0], bias_v[k][1 % NB], bias_v[k][2 % NB], bias_v[k][3 % NB]};
    // Fragment f = (tap column tx, k-half kk, halo row h), h innermost: 6 (TR + 2) of them per tile.  The fragment of halo row h
    // feeds output rows i = h - ty for the tap rows ty = 0, 1, 2 that stay inside the tile (2 MFMAs per row: two n-tiles).
    // Tap order is a compile-time property of each branch (forward: weight tap (ty, tx); dgrad: mirrored), so every fragment
    // address is a per-lane base register + an immediate.  Software pipeline: the fragment f + 2 is requested before the
    // MFMAs of fragment f (three rolling registers), pinned with sched_barriers -- hipcc otherwise reuses one register and
    // exposes the LDS latency per fragment.
    auto mfma_phase = [&](auto mirror, auto swapped) {
      constexpr bool MIR = decltype(mirror)::value;
      constexpr bool SWP = decltype(swapped)::value;
      constexpr int NF = 6 * HRW;
      U4 fa[3];
      auto load = [&](int f) {
        const int u = f / HRW, h = f - u * HRW;  // u = tx * 2 + kk
        return *reinterpret_cast<const U4*>(sX + h * HC * 128 + lds_off[u >> 1][u & 1]);
      };
      fa[0] = load(0);
      fa[1] = load(1);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + 2 < NF) fa[(f + 2) % 3] = load(f + 2);
        __builtin_amdgcn_sched_barrier(0);       // (the machine scheduler would sink the reads back to their use)
        const int u = f / HRW, h = f - u * HRW, tx = u >> 1, kk = u & 1;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int i = h - ty;                  // output row fed through tap row ty
          if (i < 0 || i >= TR) continue;
          const int tp = MIR ? (2 - ty) * 3 + (2 - tx) : ty * 3 + tx;     // weight tap whose halo offset is (ty, tx)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[i][nt] = SWP ? mfma16<DT>(fw[tp][kk][nt], fa[f % 3], acc[i][nt]) : mfma16<DT>(fa[f % 3], fw[tp][kk][nt], acc[i][nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // The co-resident wave of the other block on this SIMD is, most of the time, in its own MFMA phase.  Vector issue is
    // arbitrated by priority, then age: at equal priority this wave's ~75 epilogue / store / fetch instructions got about
    // one issue slot per two partner MFMAs (s_memtime stamps: 2,400 cycles for them against 1,150 cycles of MFMA work).
    // Everything outside the MFMA phase therefore runs at priority 1; the MFMA stream fills the slots that leaves.
    __builtin_amdgcn_s_setprio(0);
    if (a.tap_y[0] == 0)
      mfma_phase(std::false_type{}, std::integral_constant<bool, swp>{});
    else
      mfma_phase(std::true_type{}, std::integral_constant<bool, swp>{});
    __builtin_amdgcn_s_setprio(1);

    STAMP(3);
    // ---- epilogue
    const int n = cur.n;
    const int oy0 = cur.ty * TR, ox0 = cur.tx * 32;
    cur = nxt;
    nxt = advance(nxt);
    const bool full = oy0 + TR <= a.H && ox0 + 32 <= a.W;      // uniform: ragged tiles only at the bottom / right edge
    // inference: the per-channel scale / shift' pair of this lane's 2 x 4 channels, from LDS (never with PixelShuffle:
    // channels 4g .. 4g + 3 of each n-tile are one 16-byte read)
    [[maybe_unused]] float sc_v[2][NB], sh_v[2][NB];
    if constexpr (FOLD) {
      // (inline asm, the wait inside: an LDS read the compiler can see behind an in-flight LDS-DMA gets a vmcnt(0) in front
      //  of it -- the next tile's halo would be waited for here, before the epilogue instead of under it)
      typedef __attribute__((ext_vector_type(4))) float f4;
      f4 q[4];
      const unsigned a_sc = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&sStat[0][0][wc * 32 + 4 * g]);
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:256\n\t"
                   "ds_read_b128 %3, %4 offset:320\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(a_sc) : "memory");   // [1][c] is 64 floats behind [0][c]
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc_v[nt][r] = q[nt][r];
          sh_v[nt][r] = q[2 + nt][r];
        }
    }
    // SM: 0 no statistics, 1 statistics of a full tile, 2 statistics of a ragged tile (out-of-image pixels masked)
    static_assert(32 * C_STRIDE == 4608, "immediates of the ds_write_b64 below");
    const unsigned sC_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)sC);
    const unsigned cw_lds[2] = {sC_lds + (unsigned)cw_base[0], sC_lds + (unsigned)cw_base[1]};
    auto epilogue = [&](auto actf, auto smode) {
      constexpr int SM = decltype(smode)::value;
      if constexpr (SM == 3) {                 // swapped layout (never with statistics): 4 channels of one pixel per lane
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int i = 0; i < TR; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = actf(FOLD ? acc[i][nt][r] * sc_v[nt][r] + sh_v[nt][r] : acc[i][nt][r]);
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 pk;
            if constexpr (DT == DSR_DTYPE_BF16) {
              typedef __attribute__((ext_vector_type(2))) __bf16 h2;
              pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, h2));
              pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, h2));
            } else {
              typedef __attribute__((ext_vector_type(2))) _Float16 h2;
              pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, h2));
              pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, h2));
            }
            // inline asm: behind an in-flight LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of an 8-byte LDS store
            // it can see (the next tile's halo and the previous tile's output stores would be drained here)
            if (i == 0)                       // (i is a constant after unrolling: one of these survives)
              asm volatile("ds_write_b64 %0, %1" ::"v"(cw_lds[nt]), "v"(pk) : "memory");
            else if (i == 1)
              asm volatile("ds_write_b64 %0, %1 offset:4608" ::"v"(cw_lds[nt]), "v"(pk) : "memory");    // i * 32 * C_STRIDE
            else if (i == 2)
              asm volatile("ds_write_b64 %0, %1 offset:9216" ::"v"(cw_lds[nt]), "v"(pk) : "memory");
            else
              asm volatile("ds_write_b64 %0, %1 offset:13824" ::"v"(cw_lds[nt]), "v"(pk) : "memory");
            static_assert(TR <= 4, "offsets above");
          }
        return;
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float val = acc[i][nt][r];             // (FOLD launches are always swapped)
            if constexpr (SM != 0) {
              float vm = val;
              if constexpr (SM == 2) vm = (oy0 + i < a.H && ox0 + wq * 16 + 4 * g + r < a.W) ? val : 0.f;
              s1 += vm;
              s2 += vm * vm;
            }
            *reinterpret_cast<unsigned short*>(sC + cw_base[nt] + (i * 32 + r) * C_STRIDE) = f2h<DT>(actf(val));
          }
        }
        if constexpr (SM != 0) {
          s1 += __shfl_xor(s1, 16, 64);
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 16, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (g == 0) {
            sStat[wq][0][wc * 32 + nt * 16 + r16] = s1;
            sStat[wq][1][wc * 32 + nt * 16 + r16] = s2;
          }
        }
      }
    };
    auto epilogue_s = [&](auto actf) {
      if constexpr (swp)
        epilogue(actf, std::integral_constant<int, 3>{});
      else if (!do_stats)
        epilogue(actf, std::integral_constant<int, 0>{});
      else if (full)
        epilogue(actf, std::integral_constant<int, 1>{});
      else
        epilogue(actf, std::integral_constant<int, 2>{});
    };
    if (a.act == DSR_ACT_NONE)
      epilogue_s([](float x) { return x; });
    else if (a.act == DSR_ACT_RELU)
      epilogue_s([](float x) { return vmax(x, 0.f); });
    else if ((a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU) && slope >= 0.f && slope <= 1.f)
      epilogue_s([slope](float x) { return vmax(x, x * slope); });      // == x >= 0 ? x : x * slope for 0 <= slope <= 1
    else if (a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU)
      epilogue_s([slope](float x) { return x >= 0.f ? x : x * slope; });
    else
      epilogue_s([&](float x) { return act_apply(a.act, x, slope); });
    STAMP(4);
    // raw barrier: a __syncthreads() here would also drain the next tile's DMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(5);
    if (do_stats && tid < 128) {
      // BatchNorm statistics: summed over this block's tiles in a register, ONE partial row per block at the end (512
      // rows per launch instead of one per tile -- 8192 at config 3 -- so the finalize kernel needs no compaction pass)
      const int which = tid >> 6, col = tid & 63;
      stat_acc += sStat[0][which][col] + sStat[1][which][col];
    }
    {
      // exactly TR stores per thread (see the wait above).  Scalar part: byte offset of the tile origin in y.
      const unsigned sorg = pixshuf ? (unsigned)(((n * 2 * a.H + 2 * oy0) * (2 * a.W) + 2 * ox0) * (a.CoutP / 4) * 2 + (c0 / 4) * 2)
                                    : (unsigned)(((n * a.H + oy0) * a.W + ox0) * a.CoutP * 2 + c0 * 2);
#pragma unroll
      for (int it = 0; it < TR; ++it) {
        U4 v = *reinterpret_cast<const U4*>(sC + cr_base + it * 32 * C_STRIDE);
        unsigned off = sorg + (unsigned)(st_part + it * st_step);
        if (!full) {                                                    // (uniform) ragged tile: per-lane range check
          const int prow = (tid >> 3) + it * 32;
          if (!(oy0 + (prow >> 5) < a.H && ox0 + (prow & 31) < a.W)) off = OOB;
        }
        if constexpr (MODE == 3) {
          if (a.flags & DSR_F_MASK) {           // (uniform) the prefetched tile is an activation output: y = conv * act'(o)
            v = a.mask_act == DSR_ACT_RELU ? act_mask8<DT>(v, rres[it], DSR_ACT_RELU, 0.f)      // (constants per branch: no compare chain per element)
                                           : act_mask8<DT>(v, rres[it], a.mask_act == DSR_ACT_NONE ? DSR_ACT_NONE : DSR_ACT_LEAKY, a.mask_slope);
          } else {
            float f[8], rr[8];
            unpack8<DT>(v, f);
            unpack8<DT>(rres[it], rr);
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] += rr[q];
            v = pack8<DT>(f);
          }
        }
        if (FOLD && (a.flags & DSR_F_RESIDUAL)) {   // skip connection (generator.py:24,74): added after the activation
          float f[8], rr[8];
          unpack8<DT>(v, f);
          unpack8<DT>(rres[it], rr);
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] += rr[q];
          v = pack8<DT>(f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrsrc, off, 0, 0);
      }
    }
    STAMP(6);
  }
  if (do_stats && tid < 128)      // (every block owns at least one tile: grid.x <= ntiles)
    a.stats[((size_t)blockIdx.x * 2 + (tid >> 6)) * a.CoutP + c0 + (tid & 63)] = stat_acc;
#ifdef DSR_C64_STAMPS
  if (blockIdx.x == 37 && blockIdx.y == 0 && (tid == 0 || tid == 192))
    for (int q = 0; q < 8; ++q) g_c64_stamps[(tid ? 8 : 0) + q] = st_acc[q];
#endif
}

// tile rows per mode -- as many as the 256-register budget of two resident blocks allows WITHOUT a single spill (a spill is a
// VMEM instruction: it would break the "exactly TR stores outstanding" count the tile loop waits on): plain 4, statistics 3
// (the pixel-major epilogue needs more temporaries) and residual prefetch 3, folded inference epilogue 2
static constexpr int c64_tile_rows(int mode) { return mode == 1 ? 4 : (mode == 2 ? 2 : 3); }   // (mode 2: 2 or 3, chosen per launch)
static constexpr int c64_lds_bytes(int tr) { return 2 * (tr + 2) * 40 * 128 + tr * 32 * (64 * 2 + 16) + 2 * 2 * 64 * 4; }

// spatial tiles of a launch whose tiles are `tr` rows x 32 columns
int dsr_c64_tiles(int N, int H, int W, int tr) {
  return N * ((H + tr - 1) / tr) * ((W + 31) / 32);
}
// BatchNorm statistics rows written by one launch (always mode 0): one per persistent block of a 64-channel output slice
int dsr_c64_stat_rows(int N, int H, int W, int CoutP) {
  const int ntiles = dsr_c64_tiles(N, H, W, c64_tile_rows(0)), per_slice = 512 / (CoutP / 64);
  return ntiles < per_slice ? ntiles : per_slice;
}

template <int DT, int MODE, int TR = c64_tile_rows(MODE)>
static void c64_launch(const C64Args& a, dim3 grid, hipStream_t st) {
  constexpr int LDS = c64_lds_bytes(TR);
  auto* fn = conv_c64_kernel<DT, MODE, TR>;
  if constexpr (LDS > 64 * 1024) {
    static LdsOptIn optin;          // more than 64 KB of dynamic LDS needs the opt-in, once per kernel and device (not a stream operation)
    optin.ensure((const void*)fn, LDS);
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), LDS, st, a);
}

void dsr_launch_conv_c64(C64Args& a, int N, int dtype, hipStream_t st) {
  const int slices = a.CoutP / 64;                            // blockIdx.y: 64-channel slice of the output
  const bool fold = (a.flags & (DSR_F_AFFINE | DSR_F_RESIDUAL)) != 0;
  // residual alone, no activation, no PixelShuffle, one 64-channel slice (the input gradient of a residual block): mode 3
  // (DSR_F_MASK rides on the same prefetch: the tile is then an activation output whose derivative multiplies the result)
  const bool res_only = (a.flags & DSR_F_RESIDUAL) && !(a.flags & (DSR_F_AFFINE | DSR_F_PIXSHUF | DSR_F_STATS)) &&
                        a.act == DSR_ACT_NONE && slices == 1;
  const int mode = res_only ? 3 : (fold ? 2 : ((a.flags & DSR_F_STATS) ? 0 : 1));
  int tr = c64_tile_rows(mode);
  if (mode == 2) {
    // inference images are often a few tiles per block only: take the tile height (2 or 3 rows) with the shorter longest
    // block, counting a tile as its rows + 1 (halo rows and per-tile fixed work); 3 on a tie (fewer halo rows overall)
    auto cost = [&](int r) {
      const long long nt = (long long)N * ((a.H + r - 1) / r) * ((a.W + 31) / 32);
      const long long per = 512 / slices;
      return ((nt + per - 1) / per) * (r + 1);
    };
    tr = cost(3) <= cost(2) ? 3 : 2;
  }
  a.tiles_y = (a.H + tr - 1) / tr;
  a.tiles_x = (a.W + 31) / 32;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.H * a.W * 128);
  a.y_bytes = (unsigned)((size_t)N * a.H * a.W * a.CoutP * 2);   // (PixelShuffle: [N][2H][2W][CoutP/4] is the same size)
  int per_slice = 512 / slices;                               // 2 resident blocks per CU over all slices
#ifdef DSR_C64_STAMPS
  if (const char* e = getenv("DSR_C64_BLOCKS")) per_slice = atoi(e) / slices;
#endif
  dim3 grid(a.ntiles < per_slice ? a.ntiles : per_slice, slices);
#define C64_LAUNCH(DTV)                              \
  do {                                               \
    if (mode == 3) c64_launch<DTV, 3>(a, grid, st);   \
    else if (mode == 2 && tr == 3) c64_launch<DTV, 2, 3>(a, grid, st); \
    else if (mode == 2) c64_launch<DTV, 2, 2>(a, grid, st); \
    else if (mode == 1) c64_launch<DTV, 1>(a, grid, st); \
    else c64_launch<DTV, 0>(a, grid, st);            \
  } while (0)
  if (dtype == DSR_DTYPE_BF16)
    C64_LAUNCH(DSR_DTYPE_BF16);
  else
    C64_LAUNCH(DSR_DTYPE_F16);
#undef C64_LAUNCH
}
