// The discriminator's first two convolutions as ONE forward kernel (models/GAN/discriminator.py:25-29,60-63):
//     a0 = LeakyReLU(Conv2d(3 -> 64, 3x3, stride 1, pad 1)(image))           -- 3 % of D's FLOPs, 1.07 GB of output at 512^2 x 32
//     y1 = Conv2d(64 -> 64, 3x3, stride 2, pad 1)(a0) (+ BatchNorm statistics) -- reads those 1.07 GB again
// As two launches the pair is bound by that tensor: the first layer is a pure store (0.27 ms), the second reads it at 3.4 TB/s
// (0.32 ms for 155 GFLOP).  Here a block owns a 4 x 32 tile of y1, recomputes the 9 x 65 pixels of a0 it needs from the image
// halo (11 x 67 pixels of 16 bytes) straight into LDS, and runs the stride-2 convolution from there with its weights in
// registers; a0 goes to HBM only when the caller wants it (training: the weight gradient of the second layer reads it), and is
// never read back by the forward pass.
//   stage 1  D[co][pixel] = W0[co][(tap, ci)] x image[(pixel + tap)][ci]: three k-steps of 4 taps x 8 channels per 16 pixels
//            (conv_cin8.hip's mapping); + bias, LeakyReLU, ZERO outside the image (that is the second layer's padding), 16-bit,
//            8-byte LDS writes.  The a0 halo is stored as two column-parity planes per row, so that the stride-2 reads of stage 2
//            touch 16 CONSECUTIVE 128-byte slots (conflict-free with the usual 16-byte-chunk XOR swizzle).
//   stage 2  wave (output row, 32-channel half): 9 taps x 2 k-halves x 2 column groups x 2 n-tiles = 72 MFMAs from 36 register-
//            resident weight fragments (as conv_c64.hip) and 36 fragment reads.
//   epilogue BatchNorm sum / sum-of-squares from the fp32 accumulators (ONE partial row per block, as conv_c64.hip), C tile
//            through LDS, full 128-byte lines.
// One 8-wave block per CU (120 KB of LDS), persistent; the next tile's image halo arrives by LDS-DMA under this tile's work.
#include <stdlib.h>

#include "../../include/dsr_hip.h"
#include <type_traits>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int F2_TR = 4, F2_TC = 32;                        // y1 tile
constexpr int F2_AR = 2 * F2_TR + 1, F2_AC = 2 * F2_TC + 1; // a0 halo: 9 x 65
constexpr int F2_AP = 72;                                   // a0 halo row pitch in slots: even columns at 0..32, odd at 40..71
constexpr int F2_XR = F2_AR + 2, F2_XC = F2_AC + 2;         // image halo: 11 x 67
constexpr int F2_XPX = F2_XR * F2_XC;                       // 737 pixels of 16 B
constexpr int F2_XPIECES = (F2_XPX + 63) / 64;              // 12 DMA pieces of 64 pixels
constexpr int F2_X = F2_XPIECES * 1024;                     // 12,288 B per image-halo stage
constexpr int F2_A = F2_AR * F2_AP * 128;                   // 82,944 B
constexpr int F2_W0 = 3 * 64 * 64;                          // [k-step][co][4 taps x 8 ch]: 12,288 B
constexpr int F2_CSTRIDE = 64 * 2 + 16;
constexpr int F2_OFF_A = 2 * F2_X, F2_OFF_W0 = F2_OFF_A + F2_A, F2_OFF_B0 = F2_OFF_W0 + F2_W0, F2_OFF_STAT = F2_OFF_B0 + 256;
constexpr int F2_LDS = F2_OFF_STAT + 8 * 2 * 32 * 4;        // 122,112 B
constexpr int F2_GROUPS = F2_AR * 4 + 1;                    // stage-1 pixel groups: 4 x 16 columns per a0 row + the 65th column of all rows
static_assert(F2_TR * F2_TC * F2_CSTRIDE <= F2_A, "the C tile is staged over the a0 halo");
}   // namespace

// one v_max_f32 (fmaxf adds a canonicalising v_max(x, x) per operand)
__device__ __forceinline__ float f2_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#ifdef DSR_F2_STAMPS
// Diagnostic build only (tools/diag_first2.cpp): per-block cycle sums of the phases of a tile, stamped by s_memtime on wave 0; the
// values go to a buffer nothing else reads.
__device__ unsigned long long f2_stamps[256][12];
#define F2_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); f2_acc[k] += now_ - f2_last; f2_last = now_; } while (0)
#else
#define F2_STAMP(k)
#endif

template <int DT>
__global__ __launch_bounds__(512, 2) void conv_first2_kernel(const First2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sX = smem;
  unsigned char* const sA = smem + F2_OFF_A;
  unsigned char* const sW0 = smem + F2_OFF_W0;
  float* const sB0 = reinterpret_cast<float*>(smem + F2_OFF_B0);
  float* const sStat = reinterpret_cast<float*>(smem + F2_OFF_STAT);     // [8 waves][2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int orow = wave & 3, half = wave >> 2;               // stage 2: output row of the tile, 32-channel half
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // ---- second layer's weights -> registers (rows = output channels half*32 + nt*16 + r16 of the [9][64][64] forward image)
  const unsigned short* __restrict__ W1 = reinterpret_cast<const unsigned short*>(a.w1);
  U4 fw[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        fw[t][kk][nt] = *reinterpret_cast<const U4*>(W1 + ((size_t)(t * 64 + half * 32 + nt * 16 + r16)) * 64 + kk * 32 + g * 8);
  // ---- first layer's weights -> LDS, [k-step ks][co][tap 4ks + q][8 ch] (taps 9..11: zero), and its bias
  {
    const unsigned short* __restrict__ W0 = reinterpret_cast<const unsigned short*>(a.w0);
    for (int i = tid; i < 3 * 64 * 4; i += 512) {
      const int q = i & 3, co = (i >> 2) & 63, ks = i >> 8;
      const int tap = 4 * ks + q;
      U4 v = U4{0u, 0u, 0u, 0u};
      if (tap < 9) v = reinterpret_cast<const U4*>(W0)[tap * 64 + co];
      *reinterpret_cast<U4*>(sW0 + (ks * 64 + co) * 64 + q * 16) = v;
    }
    if (tid < 64) sB0[tid] = a.b0 ? a.b0[tid] : 0.f;
  }
  float bias1[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias1[nt][j] = a.b1 ? a.b1[half * 32 + nt * 16 + 4 * g + j] : 0.f;
  // the first layer's 12 weight fragments (3 k-steps x 4 n-tiles) stay in registers too: read from LDS inside stage 1 they put an
  // LDS round trip in front of every one of its MFMAs (measured: the fused kernel then ran no faster than the two launches)
  __syncthreads();
  U4 fw0[3][4];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fw0[ks][nt] = *reinterpret_cast<const U4*>(sW0 + (ks * 64 + 16 * nt + r16) * 64 + 16 * g);

  // ---- image-halo loader: piece p = pixels 64 p .. 64 p + 63 of the 11 x 67 halo; wave w issues pieces w and w + 8
  const unsigned img_bytes = (unsigned)(a.H * a.W * 16);
  const int per_img = a.tiles_y * a.tiles_x;
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
  auto fetch = [&](const TileXY& tc, int buf) {
    const int iy0 = 2 * tc.ty * F2_TR - 2, ix0 = 2 * tc.tx * F2_TC - 2;
    const BufSrd xsrd = make_srd(reinterpret_cast<const unsigned char*>(a.x) + (size_t)tc.n * img_bytes, img_bytes);
    int l0 = lane;
    asm volatile("" : "+v"(l0));                             // (keeps the decode below out of the tile loop's live registers)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = wave + 8 * u;
      if (p >= F2_XPIECES) continue;                         // (wave-uniform)
      const int q = 64 * p + l0;
      const int xr = q / F2_XC, xc = q - xr * F2_XC;
      const int ix = ix0 + xc;
      // rows outside the image fall outside the per-image resource; columns need the explicit check
      unsigned off = (unsigned)(((iy0 + xr) * a.W + ix) * 16);
      if (q >= F2_XPX || (unsigned)ix >= (unsigned)a.W) off = OOB;
      lds_dma16(xsrd, sX + buf * F2_X + p * 1024, off);
    }
  };

  // ---- stage-2 fragment addresses: a0 slot (2 orow + kh) * AP + (kw & 1) * 40 + 16 pg + r16 + (kw >> 1); the swizzle key
  // (slot & 7) only depends on r16 + (kw >> 1)
  int a2_off[2][2];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a2_off[v][kk] = (2 * orow * F2_AP + r16 + v) * 128 + (((4 * kk + g) ^ ((r16 + v) & 7)) << 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y1, 0, a.y1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(a.a0 ? a.a0 : a.y1, 0, a.a0 ? a.a0_bytes : 0u, 0x00020000);

  const int tstep = gridDim.x;
  int t = blockIdx.x;
  if (t >= a.ntiles) return;
  TileXY cur = decomp(t);
  fetch(cur, 0);
  int buf = 0;
  float stat_acc = 0.f;
#ifdef DSR_F2_STAMPS
  unsigned long long f2_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long f2_begin = __builtin_amdgcn_s_memtime();
  unsigned long long f2_last = f2_begin;
#endif
  for (; t < a.ntiles; t += tstep, buf ^= 1) {
    const bool has_next = t + tstep < a.ntiles;
    const TileXY nxt = has_next ? decomp(t + tstep) : cur;
    // this tile's image halo has landed (first tile: and sW0 / sB0 are written); everyone is done with the previous tile
    F2_STAMP(8);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    F2_STAMP(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    F2_STAMP(1);
    if (has_next) fetch(nxt, buf ^ 1);
    const int ay0 = 2 * cur.ty * F2_TR - 1, ax0 = 2 * cur.tx * F2_TC - 1;   // a0 coordinates of halo slot (0, 0)
    // ================================================================ stage 1: a0 halo -> LDS
    {
      const unsigned char* sXc = sX + buf * F2_X;
      int l1 = lane;
      asm volatile("" : "+v"(l1));
      const int gg = l1 >> 4, rr = l1 & 15;
      const bool interior = ay0 >= 0 && ay0 + F2_AR <= a.H && ax0 >= 0 && ax0 + F2_AC <= a.W;     // (uniform)
      const bool small_slope = a.slope0 <= 1.f;
      int toff[3];                                           // byte offset of this lane's tap in k-step ks (taps >= 9: weights are zero)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int tap = 4 * ks + gg;
        toff[ks] = tap < 9 ? ((tap / 3) * F2_XC + tap % 3) * 16 : 0;
      }
      // (one instantiation per (interior tile, slope <= 1): the common one has neither the per-element border select nor the
      //  compare + select form of LeakyReLU; chosen by a wave-uniform branch)
      auto stage1 = [&](auto interior_c, auto small_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value, SMALL = decltype(small_c)::value;
      for (int j = wave; j < F2_GROUPS; j += 8) {
        int ar, ac;
        bool lane_ok = true;
        if (j < F2_AR * 4) {
          ar = j >> 2;
          ac = 16 * (j & 3) + rr;
        } else {                                             // the 65th column of every row: lanes 0..8
          ar = rr < F2_AR ? rr : F2_AR - 1;
          ac = F2_AC - 1;
          lane_ok = rr < F2_AR;
        }
        const unsigned char* px = sXc + (ar * F2_XC + ac) * 16;
        U4 fb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fb[ks] = *reinterpret_cast<const U4*>(px + toff[ks]);
        const int ay = ay0 + ar, ax = ax0 + ac;
        // (tiles whose whole halo lies inside the image -- all but the border ones -- skip the per-element select; uniform)
        [[maybe_unused]] const bool inside = lane_ok && (unsigned)ay < (unsigned)a.H && (unsigned)ax < (unsigned)a.W;
        const int slot = ar * F2_AP + (ac & 1) * 40 + (ac >> 1);
        f32x4 acc4[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc4[nt] = *reinterpret_cast<const f32x4*>(sB0 + 16 * nt + 4 * gg);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc4[nt] = mfma16<DT>(fw0[ks][nt], fb[ks], acc4[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4 acc = acc4[nt];
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // LeakyReLU: max(x, x * slope) for 0 <= slope <= 1 (one instruction less than compare + select)
            float o;
            if constexpr (SMALL)
              o = f2_vmax(acc[q], acc[q] * a.slope0);
            else
              o = acc[q] >= 0.f ? acc[q] : acc[q] * a.slope0;
            v[q] = INTERIOR ? o : (inside ? o : 0.f);
          }
          uint2 h;
          h.x = (unsigned)f2h<DT>(v[0]) | ((unsigned)f2h<DT>(v[1]) << 16);
          h.y = (unsigned)f2h<DT>(v[2]) | ((unsigned)f2h<DT>(v[3]) << 16);
          if (lane_ok) *reinterpret_cast<uint2*>(sA + slot * 128 + (((2 * nt + (gg >> 1)) ^ (slot & 7)) << 4) + (gg & 1) * 8) = h;
        }
      }
      };
      if (interior && small_slope)
        stage1(std::true_type{}, std::true_type{});
      else if (small_slope)
        stage1(std::false_type{}, std::true_type{});
      else
        stage1(std::false_type{}, std::false_type{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F2_STAMP(2);
    __builtin_amdgcn_s_barrier();                            // the a0 halo is complete
    asm volatile("" ::: "memory");
    F2_STAMP(3);
    // ---- a0 to HBM (training): the tile's own 8 x 64 pixels (halo slots (1..8, 1..64)), full 128-byte lines
    if (a.a0) {
      int tv = tid;
      asm volatile("" : "+v"(tv));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int vec = tv + 512 * i;
        const int pxl = vec >> 3, ch = vec & 7;
        const int row = pxl >> 6, col = pxl & 63;
        const int ar = row + 1, ac = col + 1;
        const int slot = ar * F2_AP + (ac & 1) * 40 + (ac >> 1);
        const U4 v = *reinterpret_cast<const U4*>(sA + slot * 128 + ((ch ^ (slot & 7)) << 4));
        const int ay = ay0 + ar, ax = ax0 + ac;
        const unsigned off = (ay < a.H && ax < a.W) ? (unsigned)((((cur.n * a.H + ay) * a.W + ax) * 64 + ch * 8) * 2) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), arsrc, off, 0, 0);
      }
    }
    F2_STAMP(4);
    // ================================================================ stage 2: 3x3 stride 2 from the LDS halo
    f32x4 acc[2][2];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[pg][nt] = f32x4{bias1[nt][0], bias1[nt][1], bias1[nt][2], bias1[nt][3]};
    {
      // 36 fragments f = (kh, kw, kk, pg), pg innermost; requested two ahead of their MFMAs
      auto load = [&](int f) {
        const int pg = f & 1, kk = (f >> 1) & 1, tap = f >> 2;
        const int kh = tap / 3, kw = tap - 3 * kh;
        return *reinterpret_cast<const U4*>(sA + a2_off[kw >> 1][kk] + (kh * F2_AP + (kw & 1) * 40 + 16 * pg) * 128);
      };
      U4 fa[3];
      fa[0] = load(0);
      fa[1] = load(1);
#pragma unroll
      for (int f = 0; f < 36; ++f) {
        if (f + 2 < 36) fa[(f + 2) % 3] = load(f + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int pg = f & 1, kk = (f >> 1) & 1, tap = f >> 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[pg][nt] = mfma16<DT>(fw[tap][kk][nt], fa[f % 3], acc[pg][nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    F2_STAMP(5);
    __builtin_amdgcn_s_barrier();                            // every wave is done reading the a0 halo: it becomes the C tile
    asm volatile("" ::: "memory");
    F2_STAMP(6);
    // ================================================================ epilogue
    const int oy0 = cur.ty * F2_TR, ox0 = cur.tx * F2_TC;
    unsigned char* sC = sA;
    {
      const bool rowok = oy0 + orow < a.OH;
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[nt][j] = s2[nt][j] = 0.f;
#pragma unroll
      for (int pg = 0; pg < 2; ++pg) {
        const bool pok = rowok && ox0 + 16 * pg + r16 < a.OW;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 v = acc[pg][nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float vm = pok ? v[j] : 0.f;
            s1[nt][j] += vm;
            s2[nt][j] = __builtin_fmaf(vm, vm, s2[nt][j]);
          }
          uint2 h;
          h.x = (unsigned)f2h<DT>(v[0]) | ((unsigned)f2h<DT>(v[1]) << 16);
          h.y = (unsigned)f2h<DT>(v[2]) | ((unsigned)f2h<DT>(v[3]) << 16);
          *reinterpret_cast<uint2*>(sC + (orow * F2_TC + 16 * pg + r16) * F2_CSTRIDE + (half * 32 + nt * 16 + 4 * g) * 2) = h;
        }
      }
      if (a.stats) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // over the 16 pixels (lanes r16) of the fragment: four DPP adds (lane ^ 1, lane ^ 2 inside a quad, then the mirrored
            // quad pair and the mirrored half row -- every lane of a quad holds the quad's sum by then) instead of four
            // ds_bpermute round trips
            s1[nt][j] = row16_sum(s1[nt][j]);
            s2[nt][j] = row16_sum(s2[nt][j]);
          }
        if (r16 == 0) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              sStat[(wave * 2 + 0) * 32 + nt * 16 + 4 * g + j] = s1[nt][j];
              sStat[(wave * 2 + 1) * 32 + nt * 16 + 4 * g + j] = s2[nt][j];
            }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F2_STAMP(7);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    F2_STAMP(6);
    if (a.stats && tid < 128) {
      // channel c = tid & 63 lives in half c >> 5, i.e. waves 4 (c >> 5) + 0..3 (the tile's four rows), fixed order
      const int which = tid >> 6, c = tid & 63, hb = (c >> 5) * 4, ci = c & 31;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += sStat[((hb + w) * 2 + which) * 32 + ci];
      stat_acc += s;
    }
    {
      int tv = tid;
      asm volatile("" : "+v"(tv));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int vec = tv + 512 * i;
        const int pxl = vec >> 3, ch = vec & 7;
        const int row = pxl >> 5, col = pxl & 31;
        const U4 v = *reinterpret_cast<const U4*>(sC + pxl * F2_CSTRIDE + ch * 16);
        const int oy = oy0 + row, ox = ox0 + col;
        const unsigned off = (oy < a.OH && ox < a.OW) ? (unsigned)((((cur.n * a.OH + oy) * a.OW + ox) * 64 + ch * 8) * 2) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yrsrc, off, 0, 0);
      }
    }
    cur = nxt;
    F2_STAMP(9);
#ifdef DSR_F2_STAMPS
    f2_acc[10] += 1;
#endif
  }
#ifdef DSR_F2_STAMPS
  if (tid == 0) {
    f2_acc[11] = __builtin_amdgcn_s_memtime() - f2_begin;
    for (int i = 0; i < 12; ++i) f2_stamps[blockIdx.x][i] = f2_acc[i];
  }
#endif
  if (a.stats && tid < 128) a.stats[((size_t)blockIdx.x * 2 + (tid >> 6)) * 64 + (tid & 63)] = stat_acc;
}

static int first2_blocks(long long ntiles) { return (int)(ntiles < 256 ? ntiles : 256); }

extern "C" int dsr_conv_first2_supported(const dsr_conv_desc* d0, const dsr_conv_desc* d1) {
  if (!d0 || !d1) return 0;
  const char* e = getenv("DSR_CONV_FIRST2");                 // 0 = the two layers run as two launches (read per call: tests compare)
  if (e && e[0] == '0') return 0;
  return d0->Cin <= 8 && d0->Cout == 64 && d0->KH == 3 && d0->KW == 3 && d0->stride == 1 && d0->pad == 1 && d0->pad_mode == DSR_PAD_ZERO &&
         d1->Cin == 64 && d1->Cout == 64 && d1->KH == 3 && d1->KW == 3 && d1->stride == 2 && d1->pad == 1 && d1->pad_mode == DSR_PAD_ZERO &&
         d1->N == d0->N && d1->H == d0->H && d1->W == d0->W && d1->dtype == d0->dtype &&
         (size_t)d0->N * d0->H * d0->W * 128 < (1ull << 31);
}
extern "C" int dsr_conv_first2_stats_rows(const dsr_conv_desc* d0) {
  if (!d0) return -1;
  const int OH = (d0->H - 1) / 2 + 1, OW = (d0->W - 1) / 2 + 1;
  const long long ntiles = (long long)d0->N * ((OH + F2_TR - 1) / F2_TR) * ((OW + F2_TC - 1) / F2_TC);
  return first2_blocks(ntiles);
}
// x: the image, NHWC 16-bit with 8 channels (3 used); w0 / w1: the packed forward images of the two layers
// (dsr_conv_pack_weight); bias0 / bias1 nullable; a0 nullable ([N][H][W][64]: the first layer's activation, for the backward
// pass); y1 [N][OH][OW][64] raw second-layer output; stats nullable: dsr_conv_first2_stats_rows() rows of [2][64].
extern "C" int dsr_conv_first2_fwd(const dsr_conv_desc* d0, const dsr_conv_desc* d1, const void* x, const void* w0, const float* bias0,
                                   float slope0, const void* w1, const float* bias1, void* a0, void* y1, float* stats,
                                   dsr_stream_t s) {
  if (!d0 || !d1 || !x || !w0 || !w1 || !y1) return dsr_fail(DSR_E_ARG, "conv_first2_fwd: null pointer");
  if (!dsr_conv_first2_supported(d0, d1)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_first2_fwd: not a (<=8 -> 64, 3x3 s1) + (64 -> 64, 3x3 s2) pair");
  if (!(slope0 >= 0.f)) return dsr_fail(DSR_E_ARG, "conv_first2_fwd: LeakyReLU slope %g", (double)slope0);
  First2Args a;
  a.x = x;
  a.w0 = w0;
  a.b0 = bias0;
  a.slope0 = slope0;
  a.w1 = w1;
  a.b1 = bias1;
  a.a0 = a0;
  a.y1 = y1;
  a.stats = stats;
  a.H = d0->H;
  a.W = d0->W;
  a.OH = (d0->H - 1) / 2 + 1;
  a.OW = (d0->W - 1) / 2 + 1;
  a.tiles_y = (a.OH + F2_TR - 1) / F2_TR;
  a.tiles_x = (a.OW + F2_TC - 1) / F2_TC;
  a.ntiles = d0->N * a.tiles_y * a.tiles_x;
  a.a0_bytes = (unsigned)((size_t)d0->N * a.H * a.W * 128);
  a.y1_bytes = (unsigned)((size_t)d0->N * a.OH * a.OW * 128);
  const int blocks = first2_blocks(a.ntiles);
  static LdsOptIn optin[2];
  if (d0->dtype == DSR_BF16) {
    optin[0].ensure((const void*)conv_first2_kernel<DSR_DTYPE_BF16>, F2_LDS);
    hipLaunchKernelGGL((conv_first2_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(512), F2_LDS, s, a);
  } else {
    optin[1].ensure((const void*)conv_first2_kernel<DSR_DTYPE_F16>, F2_LDS);
    hipLaunchKernelGGL((conv_first2_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(512), F2_LDS, s, a);
  }
  return dsr_launch_status("dsr_conv_first2_fwd");
}
