// Input gradient of a 3x3 / stride 2 / zero-pad 1 convolution (discriminator.py:29,31,33,35: the four stride-2 conv
// blocks) as ONE launch.
//
// dx[n][2y+ph][2x+pw][ci] = sum over the taps (kh,kw) with (ph+1-kh) and (pw+1-kw) even of
//                               dy[n][y + (ph+1-kh)/2][x + (pw+1-kw)/2][co] * W[co][ci][kh][kw]
// i.e. four output-parity classes c = 2ph+pw with 1, 2, 2 and 4 taps.  The gather kernel runs them as four launches whose
// K loops are 1-4 taps long (D.b0: 313 TF, D.b2: 483 TF: prologue and epilogue dominate).  Here a block owns 256 pixels of
// the dY grid and 64 input channels and forms ALL FOUR classes from the four shifted dY tiles it needs:
//
//     shift (0,0): dY[y][x]      feeds one tap of every class      (1,1) (1,2) (2,1) (2,2)
//     shift (0,1): dY[y][x+1]    classes pw = 1                    (1,0) (2,0)
//     shift (1,0): dY[y+1][x]    classes ph = 1                    (0,1) (0,2)
//     shift (1,1): dY[y+1][x+1]  class (1,1)                       (0,0)
//
// so a 64-channel block of dY is staged 4 times instead of 9, prologue and epilogue are paid once per 9 taps, and every wave
// carries the same work: 8 waves as 2 (pixel halves) x 4 (16-channel slices), a wave's accumulators being 128 pixels x
// {4 classes x 16 channels} -- per shift it issues MFMAs for the classes that shift feeds (4, 2, 2, 1 of them).
// Same machinery as conv_gemm_kernel<256x256>: LDS-DMA operand stages (swizzle on the source side), two stages, one raw
// barrier per step, mfma_f32_16x16x32 with the weight fragment as the A operand (a lane owns 4 consecutive channels of one
// pixel), waves 4..7 issue their DMA between MFMA groups, C tiles leave through LDS as full 128-byte lines.
// Bit-compatible with the four-launch form: every output element sums the same products in the same order (taps ascending
// in kh,kw inside a class, channel blocks inside a tap) -- tests/test_gpu_kernels.py compares the two bit for bit.
#include <stdlib.h>

#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int S2_BM = 256;                        // dY-grid pixels per block
constexpr int S2_A = S2_BM * 128;                 // A stage: 256 rows x 64 k x 2 B
constexpr int S2_BSLOT = 64 * 128;                // one tap slice: 64 input channels x 64 k
constexpr int S2_STAGE = S2_A + 4 * S2_BSLOT;     // 64 KB
constexpr int S2_CSTRIDE = 64 * 2 + 16;           // C tile row: 64 channels + pad
constexpr int S2_LDS = 2 * S2_STAGE;

// class c = 2*ph + pw, shift s = 2*sy + sx: the tap (kh*3 + kw) that shift s feeds into class c, or -1
__host__ __device__ constexpr int s2_tap(int c, int s) {
  const int ph = c >> 1, pw = c & 1, sy = s >> 1, sx = s & 1;
  if ((ph == 0 && sy == 1) || (pw == 0 && sx == 1)) return -1;
  const int kh = ph == 0 ? 1 : (sy == 1 ? 0 : 2);
  const int kw = pw == 0 ? 1 : (sx == 1 ? 0 : 2);
  return kh * 3 + kw;
}
}   // namespace

template <int DT>
__global__ __launch_bounds__(512, 2) void conv_dgrad_s2_kernel(const DgradS2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;           // pixel half, 16-channel slice
  const int g = lane >> 4, r16 = lane & 15;

  // ---- loader role: 16-byte slot j of tile row rb + 64*i (A), of weight row rb (B)
  const int j = tid & 7, rb = tid >> 3;
  const int jc = j ^ (rb & 7);                       // source-side swizzle (the LDS image of a DMA is lane-linear)
  int a_gy[4], a_gx[4], a_base[4], b_base = 0;
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, a.dx_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tap_stride = a.CinP * a.CoutP * 2;

  // the block's tile t = (pixel tile, 64-channel block of dx): loader addresses
  auto setup = [&](int t) {
    const int cib = t % a.ci_blocks, m0 = (t / a.ci_blocks) * S2_BM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + rb + 64 * i;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int n = fd_div(a.fd_ghw, mm);
      const int rem = mm - n * (a.OH * a.OW);
      const int gy = fd_div(a.fd_gw, rem);
      const int gx = rem - gy * a.OW;
      a_gy[i] = ok ? gy : (1 << 20);                 // rows past M: every shift lands out of range
      a_gx[i] = gx;
      a_base[i] = (((n * a.OH + gy) * a.OW + gx) * a.CoutP + jc * 8) * 2;
    }
    b_base = ((cib * 64 + rb) * a.CoutP + jc * 8) * 2;      // + tap * CinP * CoutP * 2 + kb * 128
  };
  auto dma_issue = [&](int s, int kb, int stage) {
    const int sy = s >> 1, sx = s & 1;
    const int toff = ((sy * a.OW + sx) * a.CoutP + kb * 64) * 2;
    unsigned char* st = smem + stage * S2_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool inb = a_gy[i] + sy < a.OH && a_gx[i] + sx < a.OW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dyr, (lds_ptr)(st + (wave * 8 + 64 * i) * 128), 16,
                                               inb ? (unsigned)(a_base[i] + toff) : OOB, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int tap = s == 0 ? s2_tap(c, 0) : (s == 1 ? s2_tap(c, 1) : (s == 2 ? s2_tap(c, 2) : s2_tap(c, 3)));
      if (tap >= 0)                                                     // (uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(st + S2_A + c * S2_BSLOT + wave * 8 * 128), 16,
                                                 (unsigned)(b_base + tap * tap_stride + kb * 128), 0, 0, 0);
    }
  };

  const int sw = r16 & 7;
  const bool late_dma = wave >= 4;
  const int ntiles = a.tiles_m * a.ci_blocks;
  // Persistent: a block walks tiles t, t + gridDim.x, ...  The first DMA of tile t+1 is issued during the LAST step of tile
  // t (into the stage that step is not reading), BEFORE tile t's 16 output stores per thread, so at the top of tile t+1
  // "s_waitcnt vmcnt(16)" waits for that DMA only (vmcnt retires in order) and the stores -- 128 KB per tile, the dominant
  // cost of the layers with few channels -- drain under the next tile's MFMAs.  Every thread issues exactly 16 stores
  // (range-checked buffer stores: rows past M go to an out-of-range offset).
  int t = xcd_remap(blockIdx.x, gridDim.x);
  int q = 0;               // global step counter: stage = q & 1
  bool first_tile = true;
  if (t < ntiles) {
    setup(t);
    dma_issue(3, 0, 0);
  }
  for (; t < ntiles; t += gridDim.x) {
    const int cib = t % a.ci_blocks, m0 = (t / a.ci_blocks) * S2_BM;     // (this tile's, for the stores)
    const int tn = t + (int)gridDim.x;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Steps run shift-major in the order 3, 2, 1, 0 with the 64-channel blocks of dY innermost: inside every class that is
    // the gather kernel's own order (taps ascending in kh,kw, channel blocks inside a tap), so the fp32 sums are identical.
    // One barrier per step: {my DMA of this step has landed; barrier; start the next step's DMA into the other stage; MFMAs}.
    // The shift loop is unrolled: the set of classes a step feeds is static.
#pragma unroll
    for (int s = 3; s >= 0; --s) {
      for (int kb = 0; kb < a.kblocks; ++kb, ++q) {
        if (s == 3 && kb == 0 && !first_tile)
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the previous tile's stores stay in flight
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int cur = q & 1;
        const unsigned char* pa = smem + cur * S2_STAGE + (wm * 128 + r16) * 128;
        const unsigned char* pb = smem + cur * S2_STAGE + S2_A + (wn * 16 + r16) * 128;
        const bool same = kb + 1 < a.kblocks;
        const bool last = !same && s == 0;                    // last step of this tile
        auto issue_next = [&]() {
          if (!last) {
            dma_issue(same ? s : s - 1, same ? kb + 1 : 0, cur ^ 1);
          } else if (tn < ntiles) {                           // (uniform) next tile's first step
            setup(tn);
            dma_issue(3, 0, cur ^ 1);
          }
        };
        // 16 (k-half, pixel fragment) units as a software pipeline: the A fragment of unit u + 2 is requested before the
        // MFMAs of unit u and the second k-half's weight fragments during the first half's last units (see the 8-wave loop
        // of conv_gemm.hip: hipcc otherwise waits for every fragment right in front of its MFMAs)
        const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
        U4 fb0[4], fb1[4], fa[3];
        auto a_frag = [&](int u) { return *reinterpret_cast<const U4*>(pa + (u & 7) * 16 * 128 + (u < 8 ? slot0 : slot1)); };
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (s2_tap(c, s) >= 0) fb0[c] = *reinterpret_cast<const U4*>(pb + c * S2_BSLOT + slot0);
        fa[0] = a_frag(0);
        fa[1] = a_frag(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!late_dma) issue_next();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (u + 2 < 16) fa[(u + 2) % 3] = a_frag(u + 2);
          if (u == 5) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (s2_tap(c, s) >= 0) fb1[c] = *reinterpret_cast<const U4*>(pb + c * S2_BSLOT + slot1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (s2_tap(c, s) >= 0) acc[u & 7][c] = mfma16<DT>(u < 8 ? fb0[c] : fb1[c], fa[u % 3], acc[u & 7][c]);
          __builtin_amdgcn_sched_barrier(0);
          if (u == 7) {
            if (late_dma) issue_next();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    first_tile = false;
    // ---- epilogue: class by class, 256 pixels x 64 channels through LDS (the stage the last step read: the other one is
    // receiving the next tile), then full 128-byte lines to dx.  Raw barriers: a __syncthreads() would drain the DMA.
    unsigned char* sC = smem + ((q - 1) & 1) * S2_STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave is done reading that stage
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ph = c >> 1, pw = c & 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + 16 * i + r16;
        uint2 h;
        h.x = (unsigned)f2h<DT>(acc[i][c][0]) | ((unsigned)f2h<DT>(acc[i][c][1]) << 16);
        h.y = (unsigned)f2h<DT>(acc[i][c][2]) | ((unsigned)f2h<DT>(acc[i][c][3]) << 16);
        *reinterpret_cast<uint2*>(sC + row * S2_CSTRIDE + (wn * 16 + 4 * g) * 2) = h;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + 512 * it;              // 256 rows x 8 chunks
        const int row = idx >> 3, ch = idx & 7;
        const int m = m0 + row;
        const int mm = m < a.M ? m : 0;
        const int n = fd_div(a.fd_ghw, mm);
        const int rem = mm - n * (a.OH * a.OW);
        const int gy = fd_div(a.fd_gw, rem);
        const int gx = rem - gy * a.OW;
        const unsigned off = (unsigned)((((n * a.H + 2 * gy + ph) * a.W + 2 * gx + pw) * a.CinP + cib * 64 + ch * 8) * 2);
        const U4 v = *reinterpret_cast<const U4*>(sC + row * S2_CSTRIDE + ch * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, dxr, m < a.M ? off : OOB, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                           // the C tile has been read: the next class may overwrite it
      asm volatile("" ::: "memory");
    }
  }
}

bool dsr_dgrad_s2_supported(int KH, int KW, int stride, int pad, int pad_mode, int H, int W, int CinP, int CoutP, int N) {
  const char* e = getenv("DSR_DGRAD_S2");          // tuning switch, read per call (tests flip it inside one process): 0 = four launches, 2 = always
  if (e && e[0] == '0') return false;
  if (KH != 3 || KW != 3 || stride != 2 || pad != 1 || pad_mode != DSR_PAD_ZERO) return false;
  if ((H & 1) || (W & 1) || CinP % 64 || CoutP % 64) return false;
  const long long M = (long long)N * (H / 2) * (W / 2);
  const long long blocks = ((M + S2_BM - 1) / S2_BM) * (CinP / 64);
  const bool force = e && e[0] == '2';              // (tests: take the kernel however small the grid)
  return (force || blocks >= 128) && (long long)N * H * W * CinP * 2 < (1ll << 31);
}

void dsr_launch_dgrad_s2(DgradS2Args& a, int N, int dtype, hipStream_t st) {
  a.OH = a.H / 2;
  a.OW = a.W / 2;
  a.M = N * a.OH * a.OW;
  a.ci_blocks = a.CinP / 64;
  a.kblocks = a.CoutP / 64;
  a.dy_bytes = (unsigned)((size_t)a.M * a.CoutP * 2);
  a.w_bytes = (unsigned)((size_t)9 * a.CinP * a.CoutP * 2);
  a.fd_ghw = fd_make((unsigned)(a.OH * a.OW));
  a.fd_gw = fd_make((unsigned)a.OW);
  a.tiles_m = (a.M + S2_BM - 1) / S2_BM;
  a.dx_bytes = (unsigned)((size_t)N * a.H * a.W * a.CinP * 2);
  const int ntiles = a.tiles_m * a.ci_blocks;
  const int blocks = ntiles < 256 ? ntiles : 256;          // persistent: one 8-wave block per CU (128 KB of LDS each)
  static LdsOptIn optin[2];   // more than 64 KB of dynamic LDS needs the opt-in, once per kernel and device (not a stream operation)
  optin[0].ensure((const void*)conv_dgrad_s2_kernel<DSR_DTYPE_BF16>, S2_LDS);
  optin[1].ensure((const void*)conv_dgrad_s2_kernel<DSR_DTYPE_F16>, S2_LDS);
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_dgrad_s2_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(512), S2_LDS, st, a);
  else
    hipLaunchKernelGGL((conv_dgrad_s2_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(512), S2_LDS, st, a);
}
