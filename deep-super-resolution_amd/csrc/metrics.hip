// Image-quality metric on the device: SSIM (Wang, Bovik, Sheikh, Simoncelli 2004) as the reference's scripts use it --
// torchmetrics StructuralSimilarityIndexMeasure(data_range=1.0) at train_GAN.py:31,111, DIP.py:74,158,184, eval_GAN.py:31,48:
// Gaussian 11x11 window, sigma 1.5, K1 = 0.01, K2 = 0.03, per channel, mean over all window positions that lie inside the
// image (torchmetrics pads by reflection and crops that border again, which is the same set of positions).
// torchmetrics is not installed here and cannot be fetched: "parity unpinned"; the oracle (oracle/metrics.py) restates
// the published formula with F.conv2d and is what the GPU test compares against.
//
// fp32 NCHW in, one partial sum per block out (deterministic two-stage reduction, no atomics).  A block computes a 32x8
// patch of window centres from a (32+10)x(8+10) tile of both images staged in LDS; 121 taps x 5 moments per thread.
#include "dsr_common.h"
#include "dsr_kernels.h"
#include "../../include/dsr_hip.h"

#define SSIM_K 11
#define SSIM_TW 32
#define SSIM_TH 8

struct SsimWindow {
  float g[SSIM_K];   // separable 1-D Gaussian, normalised to sum 1
};

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                   int tiles_x, int tiles_y, float c1, float c2, SsimWindow win,
                                                   float* __restrict__ partial) {
  constexpr int LW = SSIM_TW + SSIM_K - 1, LH = SSIM_TH + SSIM_K - 1;
  __shared__ float sa[LH][LW + 1], sb[LH][LW + 1];
  __shared__ float red[4];
  const int plane = blockIdx.x / (tiles_x * tiles_y);
  const int t = blockIdx.x % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * SSIM_TH, tx0 = (t % tiles_x) * SSIM_TW;
  const float* pa = a + (size_t)plane * H * W;
  const float* pb = b + (size_t)plane * H * W;
  for (int i = threadIdx.x; i < LH * LW; i += 256) {
    const int ly = i / LW, lx = i % LW;
    const int y = ty0 + ly, x = tx0 + lx;
    const bool ok = y < H && x < W;
    sa[ly][lx] = ok ? pa[(size_t)y * W + x] : 0.f;
    sb[ly][lx] = ok ? pb[(size_t)y * W + x] : 0.f;
  }
  __syncthreads();
  const int lx = threadIdx.x % SSIM_TW, ly = threadIdx.x / SSIM_TW;
  const int OH = H - SSIM_K + 1, OW = W - SSIM_K + 1;        // window centres fully inside the image
  float v = 0.f;
  if (ty0 + ly < OH && tx0 + lx < OW) {
    float ma = 0.f, mb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
    for (int dy = 0; dy < SSIM_K; ++dy) {
      float ra = 0.f, rb = 0.f, raa = 0.f, rbb = 0.f, rab = 0.f;
#pragma unroll
      for (int dx = 0; dx < SSIM_K; ++dx) {
        const float xa = sa[ly + dy][lx + dx], xb = sb[ly + dy][lx + dx], g = win.g[dx];
        ra += g * xa;
        rb += g * xb;
        raa += g * xa * xa;
        rbb += g * xb * xb;
        rab += g * xa * xb;
      }
      const float g = win.g[dy];
      ma += g * ra;
      mb += g * rb;
      saa += g * raa;
      sbb += g * rbb;
      sab += g * rab;
    }
    const float va = saa - ma * ma, vb = sbb - mb * mb, cab = sab - ma * mb;
    v = ((2.f * ma * mb + c1) * (2.f * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2));
  }
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int dsr_ssim_blocks(int planes, int H, int W) {
  if (planes < 1 || H < SSIM_K || W < SSIM_K) return 0;
  const int OH = H - SSIM_K + 1, OW = W - SSIM_K + 1;
  return planes * ((OH + SSIM_TH - 1) / SSIM_TH) * ((OW + SSIM_TW - 1) / SSIM_TW);
}

extern "C" int dsr_ssim_f32(const float* img1, const float* img2, int planes, int H, int W, float data_range, float* partial,
                            hipStream_t st) {
  DSR_REQUIRE(img1 && img2 && partial && planes > 0, "ssim: null pointer or no planes");
  DSR_REQUIRE(H >= SSIM_K && W >= SSIM_K, "ssim: image %dx%d smaller than the 11x11 window", H, W);
  DSR_REQUIRE(data_range > 0.f, "ssim: data_range must be positive");
  const int OH = H - SSIM_K + 1, OW = W - SSIM_K + 1;
  const int tiles_y = (OH + SSIM_TH - 1) / SSIM_TH, tiles_x = (OW + SSIM_TW - 1) / SSIM_TW;
  SsimWindow win;
  double s = 0.0, g[SSIM_K];
  for (int i = 0; i < SSIM_K; ++i) {
    const double d = i - (SSIM_K - 1) / 2.0;
    g[i] = exp(-d * d / (2.0 * 1.5 * 1.5));
    s += g[i];
  }
  for (int i = 0; i < SSIM_K; ++i) win.g[i] = (float)(g[i] / s);
  const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
  hipLaunchKernelGGL(ssim_kernel, dim3(planes * tiles_x * tiles_y), dim3(256), 0, st, img1, img2, H, W, tiles_x, tiles_y, c1, c2,
                     win, partial);
  return dsr_launch_status("dsr_ssim_f32");
}

// ---- measurement aid: (shader-clock cycles, 100 MHz real-time ticks) pairs, one per XCD.  Two samples around a region give
// the clock the chip actually held there: MI355X lowers its shader clock under load (tools/clock_probe.hip: a register-only
// MFMA loop on every CU runs at 1.22-1.28 GHz, not 2.4), which is what a fraction "of the 2.5 PFLOP/s peak" is really
// measured against.  s_memtime counters of different XCDs are not synchronised with each other, so every block files its
// pair under its own XCD (HW_REG_XCC_ID) and the host differences samples of the SAME XCD only.
__global__ void clock_sample_kernel(unsigned long long* __restrict__ out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7u;
    out[2 * xcc] = __builtin_amdgcn_s_memtime();
    out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
extern "C" int dsr_clock_sample(unsigned long long* out16, dsr_stream_t st) {
  DSR_REQUIRE(out16, "clock_sample: null pointer");
  hipLaunchKernelGGL(clock_sample_kernel, dim3(64), dim3(64), 0, st, out16);     // 64 blocks: every XCD gets some
  return dsr_launch_status("dsr_clock_sample");
}
