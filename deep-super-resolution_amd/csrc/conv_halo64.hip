// 3x3 / stride 1 / zero-pad 1 convolution with 64 (or 128: two slices of 64, see `slices`) OUTPUT channels and a multiple of 32
// input channels (128, 256): the input
// gradients of the layers that widen 64 channels -- the PixelShuffle convs 64 -> 256 (models/GAN/generator.py:30: 620
// GFLOP per batch-32 pass at 256x256), D's 64 -> 128 block (models/GAN/discriminator.py:31), VGG conv2_1 (utils/GAN.py:24).
//
// On the gather kernel (conv_gemm_persist_kernel<64>) these ran at 590-780 TFLOP/s: with only 64 output columns per A row an
// MFMA-rate K loop needs 64 B per clock and CU of A operand, and the gather kernel fetches a tap-shifted A tile per (tap,
// 64-channel block) -- the input passes through the vector-memory path NINE times.  Here:
//   * the block owns an 8-row x 64-column pixel tile and walks the input channels in K-blocks of 32; per K-block the
//     10 x 66 pixel HALO (64 bytes per pixel) arrives ONCE by LDS-DMA and every tap is an address offset into it, and the
//     K-block's 9 x 64 x 32 weights (36 KB, L2-resident) arrive by LDS-DMA beside it: 1.3x + the weights instead of 9x;
//   * 8 waves = 4 column strips of 16 x 2 halves of 32 output channels; a wave owns ALL 8 tile rows of its strip: an A
//     fragment (16 pixels of one halo row at one tap column) feeds up to three output rows, a B fragment (one tap, 16 output
//     channels) all 8 rows -- 30 + 18 fragment reads for 144 MFMAs per K-block and wave (0.33 ds_read_b128 per MFMA; the
//     gather kernel's 128x64 tile needs 0.75);
//   * 64-byte pixel / weight rows: a wave's fragment read covers 16 consecutive rows = 1 KB contiguous, conflict-free without
//     a swizzle, and a K-block is exactly one mfma_f32_16x16x32;
//   * two stages (halo + weights each), one barrier per K-block; the next K-block's (or next tile's first) DMA is issued
//     right after the barrier and lands under the 144 MFMAs; persistent blocks, one per CU (160 KB of LDS);
//   * weights are the MFMA's A operand (a lane ends up with 4 consecutive channels of one pixel); the C tile leaves through
//     the halo stage that has just been consumed, four rows at a time, as full 128-byte lines.
// dgrad = the same kernel on the [tap][ci][co] weight image with mirrored taps (as conv_c64.hip).
#include <stdlib.h>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int H64_TR = 8, H64_TC = 64;                 // pixel tile
constexpr int H64_HR = H64_TR + 2, H64_HP = H64_TC + 2;  // halo rows, halo row pitch in pixels (no pad columns)
constexpr int H64_SLOTS = H64_HR * H64_HP;             // 660 pixel slots of 64 B
constexpr int H64_HPIECES = (H64_SLOTS + 15) / 16;     // 42 DMA pieces of 16 slots (1 KB)
constexpr int H64_HALO = H64_HPIECES * 1024;           // 43,008 B per halo stage
constexpr int H64_WPIECES = 9 * 64 / 16;               // 36 pieces: 9 taps x 64 output channels x 64 B
constexpr int H64_WTS = H64_WPIECES * 1024;            // 36,864 B per weight stage
constexpr int H64_CSTRIDE = 64 * 2 + 16;
constexpr int H64_LDS = 2 * H64_HALO + 2 * H64_WTS;    // 159,744 B
static_assert(4 * H64_TC * H64_CSTRIDE <= H64_HALO, "half a C tile is staged in one halo stage");
}   // namespace

template <int DT, bool MIR>
__global__ __launch_bounds__(512, 2) void conv_halo64_kernel(const Halo64Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sHalo = smem;
  unsigned char* const sWts = smem + 2 * H64_HALO;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wc = wave >> 2;            // 16-column strip, 32-channel half
  const int g = lane >> 4, r16 = lane & 15;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int FAR = 0x7FFFFF00;

  // ---- loader, halo: wave w issues the pieces w, w + 8, ... (< 42); lane l of piece p fills slot 16 p + l / 4, chunk l % 4
  constexpr int NHU = (H64_HPIECES + 7) / 8;          // 6
  int h_part[NHU];                                    // ((hr * W + hc) * CinP + chunk * 8) * 2 of this lane's slot, FAR if none
  unsigned long long hc_pack = 0;                     // its halo column, 7 bits per piece
#pragma unroll
  for (int u = 0; u < NHU; ++u) {
    const int slot = 16 * (wave + 8 * u) + (lane >> 2);
    const int hr = slot / H64_HP, hc = slot - hr * H64_HP;
    h_part[u] = slot < H64_SLOTS ? ((hr * a.W + hc) * a.CinP + (lane & 3) * 8) * 2 : FAR;
    hc_pack |= (unsigned long long)hc << (7 * u);
  }
  // ---- loader, weights: piece q = rows 16 q .. 16 q + 15 of the [9 * 64] x 32 K-block slice; wave w issues q = w, w + 8, ...
  constexpr int NWU = (H64_WPIECES + 7) / 8;          // 5
  int w_part[NWU];
#pragma unroll
  for (int v = 0; v < NWU; ++v) {
    const int row = 16 * (wave + 8 * v) + (lane >> 2);      // tap * 64 + channel of the slice
    w_part[v] = (((row >> 6) * a.cout_full + (row & 63)) * a.CinP + (lane & 3) * 8) * 2;     // + slice * 64 * CinP * 2 per tile
  }
  const BufSrd wsrd = make_srd(a.w, a.w_bytes);
  const unsigned img_bytes = (unsigned)(a.H * a.W * a.CinP * 2);
  const int per_img = a.tiles_y * a.tiles_x;

  const int slices = a.cout_full >> 6;                 // 64-channel output slices: tile t = (spatial tile, slice), slices of one
                                                      // spatial tile on neighbouring blocks (the second read of its halo hits L2)
  const int w_slice = 64 * a.CinP * 2, y_pix = a.cout_full * 2;
  struct TileXY {
    int n, ty, tx, sl;
  };
  auto decomp = [&](int t) {
    TileXY c;
    const int ts = t / slices;
    c.sl = t - ts * slices;
    c.n = ts / per_img;
    const int rem = ts - c.n * per_img;
    c.ty = rem / a.tiles_x;
    c.tx = rem - c.ty * a.tiles_x;
    return c;
  };
  auto fetch = [&](const TileXY& tc, int kb, int buf) {
    const int oy0 = tc.ty * H64_TR - 1, ox0 = tc.tx * H64_TC - 1;
    // per-image resource: halo rows above / below the image fall outside it and read as zeros
    const BufSrd xsrd = make_srd(reinterpret_cast<const unsigned char*>(a.x) + (size_t)tc.n * img_bytes, img_bytes);
    const int sbase = (oy0 * a.W + ox0) * a.CinP * 2 + kb * 64;
    const bool interior = ox0 >= 0 && ox0 + H64_HP <= a.W;
    unsigned char* hdst = sHalo + buf * H64_HALO;
#pragma unroll
    for (int u = 0; u < NHU; ++u) {
      if (wave + 8 * u >= H64_HPIECES) continue;      // (wave-uniform; only the last round is partial)
      unsigned off = (unsigned)(h_part[u] + sbase);
      if (!interior) {
        const int ix = ox0 + (int)((hc_pack >> (7 * u)) & 127);
        if ((unsigned)ix >= (unsigned)a.W) off = OOB;
      }
      lds_dma16(xsrd, hdst + (wave + 8 * u) * 1024, off);
    }
    unsigned char* wdst = sWts + buf * H64_WTS;
#pragma unroll
    for (int v = 0; v < NWU; ++v) {
      if (wave + 8 * v >= H64_WPIECES) continue;
      lds_dma16(wsrd, wdst + (wave + 8 * v) * 1024, (unsigned)(w_part[v] + tc.sl * w_slice + kb * 64));
    }
  };

  // fragment addresses: A (pixels) at halo slot (h, 16 wq + tx + r16), chunk g; B (weights) at row tap * 64 + 32 wc + 16 nt + r16
  const int a_off = ((16 * wq + r16) * 64) + 16 * g;
  const int b_off = ((32 * wc + r16) * 64) + 16 * g;

  // store loop: thread tid moves the 16-byte vectors tid + 512 it of a 4-row half tile (pixel = vector / 8)
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.mask_x ? a.mask_x : a.y), 0, a.y_bytes, 0x00020000);
  const int cw_base = (16 * wq + r16) * H64_CSTRIDE + (32 * wc + 4 * g) * 2;      // + (row & 3) * 64 * CSTRIDE + nt * 32
  const int cr_base = (tid >> 3) * H64_CSTRIDE + (tid & 7) * 16;                   // + it * 64 * CSTRIDE: row `it` of the half
  const int st_part = ((tid >> 3) & 63) * y_pix + (tid & 7) * 16;                   // byte offset inside an output row of the tile

  const int tstep = gridDim.x;
  int t = blockIdx.x;
  if (t >= a.ntiles) return;
  TileXY cur = decomp(t);
  const int KB = a.kblocks;
  fetch(cur, 0, 0);
  int buf = 0;
  const float slope = a.slope;

  for (; t < a.ntiles; t += tstep) {
    const bool has_next = t + tstep < a.ntiles;
    const TileXY nxt = has_next ? decomp(t + tstep) : cur;
    f32x4 acc[H64_TR][2];
#pragma unroll
    for (int i = 0; i < H64_TR; ++i)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.flags & DSR_F_BIAS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = a.bias[cur.sl * 64 + 32 * wc + 16 * nt + 4 * g + r];
        }
        acc[i][nt] = f32x4{bv[0], bv[1], bv[2], bv[3]};
      }
    for (int kb = 0; kb < KB; ++kb, buf ^= 1) {
      // my DMA of this K-block has landed; after the barrier everyone's has, and every wave is done with the other stage
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // The two waves of a SIMD (w and w + 4) would run in lock-step: both would issue their ~10 DMA pieces (60-100 cycles
      // each) and only then start their MFMAs.  Waves 4..7 issue theirs after their first tap column instead: one wave's DMA
      // issue runs under its partner's MFMAs (as in conv_gemm_kernel<256x256>).
      auto prefetch = [&]() {
        if (kb + 1 < KB)
          fetch(cur, kb + 1, buf ^ 1);
        else if (has_next)
          fetch(nxt, 0, buf ^ 1);
      };
      const bool late = wave >= 4;
      if (!late) prefetch();
      const unsigned char* sX = sHalo + buf * H64_HALO + a_off;
      const unsigned char* sW = sWts + buf * H64_WTS + b_off;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        U4 fb[3][2];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int tap = MIR ? (2 - ty) * 3 + (2 - tx) : ty * 3 + tx;   // weight slice whose halo offset is (ty, tx): an immediate
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) fb[ty][nt] = *reinterpret_cast<const U4*>(sW + tap * 64 * 64 + nt * 16 * 64);
        }
        // halo rows as a software pipeline: fragment h + 2 requested before the MFMAs of fragment h
        U4 fa[3];
        auto load = [&](int h) { return *reinterpret_cast<const U4*>(sX + (h * H64_HP + tx) * 64); };
        fa[0] = load(0);
        fa[1] = load(1);
#pragma unroll
        for (int h = 0; h < H64_HR; ++h) {
          if (h + 2 < H64_HR) fa[(h + 2) % 3] = load(h + 2);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const int i = h - ty;
            if (i < 0 || i >= H64_TR) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[i][nt] = mfma16<DT>(fb[ty][nt], fa[h % 3], acc[i][nt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tx == 0 && late) prefetch();
      }
    }
    // ---- epilogue: the stage just consumed (buf ^ 1 after the loop's flip) becomes the C staging area; the DMA in flight
    // targets the other stage.  Two halves of 4 rows x 64 pixels.
    unsigned char* sC = sHalo + (buf ^ 1) * H64_HALO;
    const int oy0 = cur.ty * H64_TR, ox0 = cur.tx * H64_TC;
    const unsigned sorg = (unsigned)(((cur.n * a.H + oy0) * a.W + ox0) * y_pix + cur.sl * 128);
    const bool full = oy0 + H64_TR <= a.H && ox0 + H64_TC <= a.W;
    // (the activation and the mask's activation enter as callables chosen by wave-uniform branches OUTSIDE the store loops: with
    //  the run-time values inside, every element went through act_apply's / act_grad_from_out's select chains)
    auto epilogue = [&](auto actf, auto maskf) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_s_barrier();          // every wave is done reading the stage (half 0) / the staged half (half 1)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = half * 4 + ii;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = actf(acc[i][nt][r]);
          uint2 pk;
          pk.x = (unsigned)f2h<DT>(v[0]) | ((unsigned)f2h<DT>(v[1]) << 16);
          pk.y = (unsigned)f2h<DT>(v[2]) | ((unsigned)f2h<DT>(v[3]) << 16);
          *reinterpret_cast<uint2*>(sC + cw_base + ii * 64 * H64_CSTRIDE + nt * 32) = pk;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const U4 v = *reinterpret_cast<const U4*>(sC + cr_base + it * 64 * H64_CSTRIDE);
        const int row = half * 4 + it;
        unsigned off = sorg + (unsigned)(row * a.W * y_pix + st_part);
        if (!full && !(oy0 + row < a.H && ox0 + ((tid >> 3) & 63) < a.W)) off = OOB;
        if (a.mask_x) {       // (uniform) dsr_conv_dgrad_masked: the stored gradient is multiplied by act'(mask_x), the activation OUTPUT
                              // this gradient is taken with respect to (same shape as y) -- as conv_gemm.hip's masked stores
          const U4 o = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, off, 0, 0));
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, maskf(v, o)), yrsrc, off, 0, 0);
          continue;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yrsrc, off, 0, 0);
      }
    }
    };
    auto with_mask = [&](auto actf) {
      if (a.mask_act == DSR_ACT_RELU)
        epilogue(actf, [](const U4& v, const U4& o) { return act_mask8<DT>(v, o, DSR_ACT_RELU, 0.f); });
      else if (a.mask_act == DSR_ACT_LEAKY || a.mask_act == DSR_ACT_PRELU)
        epilogue(actf, [&](const U4& v, const U4& o) { return act_mask8<DT>(v, o, DSR_ACT_LEAKY, a.mask_slope); });
      else
        epilogue(actf, [&](const U4& v, const U4& o) { return act_mask8<DT>(v, o, a.mask_act, a.mask_slope); });
    };
    if (a.act == DSR_ACT_NONE)
      with_mask([](float x) { return x; });
    else if (a.act == DSR_ACT_RELU)
      with_mask([](float x) { return x > 0.f ? x : 0.f; });
    else if (a.act == DSR_ACT_LEAKY)
      with_mask([slope](float x) { return x >= 0.f ? x : x * slope; });
    else
      with_mask([&](float x) { return act_apply(a.act, x, slope); });
    cur = nxt;
  }
}

bool dsr_halo64_supported(int KH, int KW, int stride, int pad, int pad_mode, int H, int W, int CinP, int CoutP) {
  const char* e = getenv("DSR_CONV_HALO64");         // 0 = these layers stay on the gather kernel (read per call: a test compares the two)
  const bool on = !(e && e[0] == '0');
  // (more than 64 outputs = several 64-channel slices per spatial tile: opt-in, DSR_CONV_HALO64=2 / 3 / 4 admit 128 / 256 / 512
  //  outputs.  Measured: 128 outputs are 7-17 % faster than the gather kernel's 128x128 tile launch by launch (VGG conv2_2 0.77 ->
  //  0.72 ms, D.b3's input gradient 0.33 -> 0.27) and the two-stream step 0.27 ms SLOWER with them (34.90 / 34.67 against 34.62 /
  //  34.41 ms, same box, alternating) -- a persistent block that fills a CU's LDS leaves the other stream's kernels nothing to
  //  run beside; 256 outputs are slower already launch by launch (0.63 vs 0.57 ms).  DESIGN.md 8.23)
  const int max_out = (e && e[0] == '2') ? 128 : ((e && e[0] == '3') ? 256 : ((e && e[0] == '4') ? 512 : 64));
  const bool two = CoutP % 64 == 0 && CoutP > 64 && CoutP <= max_out;
  return on && KH == 3 && KW == 3 && stride == 1 && pad == 1 && pad_mode == DSR_PAD_ZERO && (CoutP == 64 || two) && CinP % 32 == 0 &&
         CinP >= 128 && CinP <= 1024 && H >= 8 && W >= 32 && (size_t)H * W * CinP * 2 < 0x7FFFFF00ull &&
         (size_t)H * W * CoutP * 2 < 0x7FFFFF00ull;
}

void dsr_launch_conv_halo64(Halo64Args& a, int N, int dtype, hipStream_t st) {
  a.kblocks = a.CinP / 32;
  a.tiles_y = (a.H + H64_TR - 1) / H64_TR;
  a.tiles_x = (a.W + H64_TC - 1) / H64_TC;
  if (a.cout_full < 64) a.cout_full = 64;
  a.ntiles = N * a.tiles_y * a.tiles_x * (a.cout_full / 64);
  a.w_bytes = (unsigned)(9 * a.cout_full * a.CinP * 2);
  a.y_bytes = (unsigned)((size_t)N * a.H * a.W * a.cout_full * 2);
  const int blocks = a.ntiles < 256 ? a.ntiles : 256;         // persistent: one 8-wave block per CU (156 KB of LDS each)
  static LdsOptIn optin[4];
#define H64_LAUNCH(I, DTV, MIRV)                                                                      \
  do {                                                                                                \
    optin[I].ensure((const void*)conv_halo64_kernel<DTV, MIRV>, H64_LDS);                             \
    hipLaunchKernelGGL((conv_halo64_kernel<DTV, MIRV>), dim3(blocks), dim3(512), H64_LDS, st, a);      \
  } while (0)
  if (dtype == DSR_DTYPE_BF16) {
    if (a.mirror) H64_LAUNCH(0, DSR_DTYPE_BF16, true);
    else H64_LAUNCH(1, DSR_DTYPE_BF16, false);
  } else {
    if (a.mirror) H64_LAUNCH(2, DSR_DTYPE_F16, true);
    else H64_LAUNCH(3, DSR_DTYPE_F16, false);
  }
#undef H64_LAUNCH
}
