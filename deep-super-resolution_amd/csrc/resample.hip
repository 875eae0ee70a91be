// Resampling / data-movement kernels around the conv stacks (all HBM-bound, NHWC 16-bit unless noted):
//   MaxPool2d(2,2)                          utils/GAN.py:24,29,38,47 (VGG19 trunk)        fwd + bwd
//   Upsample(scale 2, bilinear)             models/DIP/skip.py:77 (align_corners=False)    fwd + bwd
//   antialiased resize + crop + normalise   utils/GAN.py:82-83 (torchvision transforms())  fwd + bwd
//   channel box copy                        models/DIP/utils.py:18-38 (Concat + centre crop) fwd + bwd
//   fixed-kernel strided downsampler        utils/downsampler.py:44-71 (Lanczos/Gauss/box, ReplicationPad2d)
#include "../../include/dsr_hip.h"
#include "dsr_common.h"
#include "dsr_kernels.h"

// ------------------------------------------------------------------ MaxPool 2x2 / stride 2 (floor)
template <int DT>
__global__ void maxpool2_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int H,
                                    int W, int Cp) {
  const int OH = H / 2, OW = W / 2, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * OH * OW * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float f[8];
      unpack8<DT>(*reinterpret_cast<const U4*>(x + ((size_t)(n * H + 2 * oy + i) * W + 2 * ox + j) * Cp + ch * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = f[k] > m[k] ? f[k] : m[k];
    }
  *reinterpret_cast<U4*>(y + pix * Cp + ch * 8) = pack8<DT>(m);
}

// dx at (y,x) = dy of its window iff (y,x) is the FIRST maximum of the window in scan order (ATen's choice)
template <int DT>
__global__ void maxpool2_bwd_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                    unsigned short* __restrict__ dx, int N, int H, int W, int Cp, int relu_mask) {
  const int OH = H / 2, OW = W / 2, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
  int oy = yy / 2, ox = xx / 2;
  float g[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = 0.f;
  if (oy < OH && ox < OW) {
    float v[4][8], d[8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        unpack8<DT>(*reinterpret_cast<const U4*>(x + ((size_t)(n * H + 2 * oy + i) * W + 2 * ox + j) * Cp + ch * 8),
                    v[i * 2 + j]);
    unpack8<DT>(*reinterpret_cast<const U4*>(dy + ((size_t)(n * OH + oy) * OW + ox) * Cp + ch * 8), d);
    const int me = (yy & 1) * 2 + (xx & 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int arg = 0;
      float m = v[0][k];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (v[q][k] > m) {
          m = v[q][k];
          arg = q;
        }
      // relu_mask: x is a ReLU output and its backward rides along: the routed gradient survives where the maximum is > 0
      g[k] = (arg == me && (!relu_mask || m > 0.f)) ? d[k] : 0.f;
    }
  }
  *reinterpret_cast<U4*>(dx + pix * Cp + ch * 8) = pack8<DT>(g);
}

// ------------------------------------------------------------------ 2x2 average pooling / nearest x2 (DIP options)
// nn.AvgPool2d(2, 2) (floor mode: an odd trailing row / column is dropped) and its adjoint
template <int DT>
__global__ void avgpool2_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int H,
                                    int W, int Cp) {
  const int OH = H / 2, OW = W / 2, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * OH * OW * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
  const unsigned short* base = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * Cp + ch * 8;
  float a[8], b[8], c[8], d[8], o[8];
  unpack8<DT>(*reinterpret_cast<const U4*>(base), a);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + Cp), b);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + (size_t)W * Cp), c);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + (size_t)W * Cp + Cp), d);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (a[k] + b[k] + c[k] + d[k]) * 0.25f;
  *reinterpret_cast<U4*>(y + pix * Cp + ch * 8) = pack8<DT>(o);
}
template <int DT>
__global__ void avgpool2_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int N,
                                    int H, int W, int Cp) {
  const int OH = H / 2, OW = W / 2, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
  float g[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = 0.f;
  if ((yy >> 1) < OH && (xx >> 1) < OW) {
    unpack8<DT>(*reinterpret_cast<const U4*>(dy + (((size_t)n * OH + (yy >> 1)) * OW + (xx >> 1)) * Cp + ch * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] *= 0.25f;
  }
  *reinterpret_cast<U4*>(dx + pix * Cp + ch * 8) = pack8<DT>(g);
}

// nn.Upsample(scale_factor=2, mode='nearest'): y[oy][ox] = x[oy/2][ox/2]; the adjoint sums each 2x2 block of dy
template <int DT>
__global__ void nearest2x_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int H,
                                     int W, int Cp) {
  const int OH = 2 * H, OW = 2 * W, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * OH * OW * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
  *reinterpret_cast<U4*>(y + pix * Cp + ch * 8) =
      *reinterpret_cast<const U4*>(x + (((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * Cp + ch * 8);
}
template <int DT>
__global__ void nearest2x_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int N,
                                     int H, int W, int Cp) {
  const int OW = 2 * W, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
  const unsigned short* base = dy + (((size_t)n * 2 * H + 2 * yy) * OW + 2 * xx) * Cp + ch * 8;
  float a[8], b[8], c[8], d[8], o[8];
  unpack8<DT>(*reinterpret_cast<const U4*>(base), a);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + Cp), b);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + (size_t)OW * Cp), c);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + (size_t)OW * Cp + Cp), d);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (a[k] + b[k]) + (c[k] + d[k]);
  *reinterpret_cast<U4*>(dx + pix * Cp + ch * 8) = pack8<DT>(o);
}

// ------------------------------------------------------------------ bilinear x2 (align_corners = False)
__device__ __forceinline__ void bil_src(int o, int n_in, int& i0, int& i1, float& l) {
  float s = (o + 0.5f) * 0.5f - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l = s - (float)i0;
}

template <int DT>
__global__ void bilinear2x_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int N, int H,
                                      int W, int Cp) {
  const int OH = 2 * H, OW = 2 * W, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * OH * OW * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
  int y0, y1, x0, x1;
  float ly, lx;
  bil_src(oy, H, y0, y1, ly);
  bil_src(ox, W, x0, x1, lx);
  float a[8], b[8], c[8], d[8], o[8];
  const unsigned short* base = x + (size_t)n * H * W * Cp + ch * 8;
  unpack8<DT>(*reinterpret_cast<const U4*>(base + ((size_t)y0 * W + x0) * Cp), a);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + ((size_t)y0 * W + x1) * Cp), b);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + ((size_t)y1 * W + x0) * Cp), c);
  unpack8<DT>(*reinterpret_cast<const U4*>(base + ((size_t)y1 * W + x1) * Cp), d);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    o[k] = (1.f - ly) * ((1.f - lx) * a[k] + lx * b[k]) + ly * ((1.f - lx) * c[k] + lx * d[k]);
  *reinterpret_cast<U4*>(y + pix * Cp + ch * 8) = pack8<DT>(o);
}

// adjoint in gather form: input pixel (y,x) collects from the <= 5x5 outputs whose stencil touches it
template <int DT>
__global__ void bilinear2x_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int N,
                                      int H, int W, int Cp) {
  const int OH = 2 * H, OW = 2 * W, cpr = Cp / 8;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int oy = 2 * yy - 2; oy <= 2 * yy + 2; ++oy) {
    if (oy < 0 || oy >= OH) continue;
    int y0, y1;
    float ly;
    bil_src(oy, H, y0, y1, ly);
    float wy = (y0 == yy ? 1.f - ly : 0.f) + (y1 == yy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = 2 * xx - 2; ox <= 2 * xx + 2; ++ox) {
      if (ox < 0 || ox >= OW) continue;
      int x0, x1;
      float lx;
      bil_src(ox, W, x0, x1, lx);
      float wx = (x0 == xx ? 1.f - lx : 0.f) + (x1 == xx ? lx : 0.f);
      if (wx == 0.f) continue;
      float d[8];
      unpack8<DT>(*reinterpret_cast<const U4*>(dy + ((size_t)(n * OH + oy) * OW + ox) * Cp + ch * 8), d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += wy * wx * d[k];
    }
  }
  *reinterpret_cast<U4*>(dx + pix * Cp + ch * 8) = pack8<DT>(acc);
}

// ------------------------------------------------------------------ separable resample (antialiased resize + crop) + normalise
// tables (host-built, device resident): for output index o: start[o], count[o], w[o*KT + i]
// forward : dst[n][oy][ox][c] (NHWC 16-bit, Cp=8) = (sum_ij wy wx src[n][c][ys+i][xs+j] - mean[c]) / std[c]
template <int DT>
__global__ void resize_norm_fwd_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int N, int C,
                                       int H, int W, int OH, int OW, const int* __restrict__ ys, const int* __restrict__ yc,
                                       const float* __restrict__ yw, const int* __restrict__ xs,
                                       const int* __restrict__ xc, const float* __restrict__ xw, int KT, float m0, float m1,
                                       float m2, float s0, float s1, float s2) {
  size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * OH * OW;
  if (pix >= total) return;
  int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((size_t)OW * OH));
  const float mean[3] = {m0, m1, m2}, istd[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = 0.f;
  const int y0 = ys[oy], ny = yc[oy], x0 = xs[ox], nx = xc[ox];
  // the three channel planes share every address and weight; the x weights of this output column stay in registers
  // (same summation order per channel as the plain triple loop: rows outer, columns inner)
  constexpr int KX = 16;
  float wxr[KX];
#pragma unroll
  for (int j = 0; j < KX; ++j) wxr[j] = j < nx ? xw[ox * KT + j] : 0.f;
  const size_t plane = (size_t)H * W;
  const float* p0 = src + (size_t)n * C * plane + (size_t)y0 * W + x0;
  float acc[3] = {0.f, 0.f, 0.f};
  if (nx <= KX) {
    for (int i = 0; i < ny; ++i) {
      const float wyi = yw[oy * KT + i];
      const float* r = p0 + (size_t)i * W;
      float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KX; ++j) {
        if (j < nx) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (c < C) row[c] += wxr[j] * r[(size_t)c * plane + j];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wyi * row[c];
    }
  } else {
    for (int i = 0; i < ny; ++i) {
      const float wyi = yw[oy * KT + i];
      const float* r = p0 + (size_t)i * W;
      float row[3] = {0.f, 0.f, 0.f};
      for (int j = 0; j < nx; ++j)
        for (int c = 0; c < 3 && c < C; ++c) row[c] += xw[ox * KT + j] * r[(size_t)c * plane + j];
      for (int c = 0; c < 3; ++c) acc[c] += wyi * row[c];
    }
  }
  for (int c = 0; c < C && c < 3; ++c) o[c] = (acc[c] - mean[c]) * istd[c];
  *reinterpret_cast<U4*>(dst + pix * 8) = pack8<DT>(o);
}

// backward with the TRANSPOSED tables: for input index i: the outputs o that read it and their weights
template <int DT>
__global__ void resize_norm_bwd_kernel(const unsigned short* __restrict__ dout, float* __restrict__ dsrc, int N, int C,
                                       int H, int W, int OH, int OW, const int* __restrict__ ty_s,
                                       const int* __restrict__ ty_c, const float* __restrict__ ty_w,
                                       const int* __restrict__ tx_s, const int* __restrict__ tx_c,
                                       const float* __restrict__ tx_w, int KT, float s0, float s1, float s2) {
  size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W;
  if (pix >= total) return;
  int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
  const float istd[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  float acc[3] = {0.f, 0.f, 0.f};
  const int oy0 = ty_s[yy], ny = ty_c[yy], ox0 = tx_s[xx], nx = tx_c[xx];
  for (int i = 0; i < ny; ++i) {
    const float wy = ty_w[yy * KT + i];
    for (int j = 0; j < nx; ++j) {
      const float w = wy * tx_w[xx * KT + j];
      float d[8];
      unpack8<DT>(*reinterpret_cast<const U4*>(dout + ((size_t)(n * OH + oy0 + i) * OW + ox0 + j) * 8), d);
      acc[0] += w * d[0];
      acc[1] += w * d[1];
      acc[2] += w * d[2];
    }
  }
  for (int c = 0; c < C && c < 3; ++c) dsrc[((size_t)n * C + c) * H * W + (size_t)yy * W + xx] = acc[c] * istd[c];
}

// ------------------------------------------------------------------ channel box copy (Concat / its adjoint)
// dst[n][dy0+y][dx0+x][cd0+c] = src[n][sy0+y][sx0+x][cs0+c]   for y<BH, x<BW, c<C
__global__ void box_copy_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int N, int BH,
                                int BW, int C, int SH, int SW, int SCp, int sy0, int sx0, int cs0, int DH, int DW,
                                int DCp, int dy0, int dx0, int cd0) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * BH * BW * C;
  if (idx >= total) return;
  int c = (int)(idx % C);
  size_t p = idx / C;
  int x = (int)(p % BW), y = (int)((p / BW) % BH), n = (int)(p / ((size_t)BW * BH));
  dst[((size_t)(n * DH + dy0 + y) * DW + dx0 + x) * DCp + cd0 + c] =
      src[((size_t)(n * SH + sy0 + y) * SW + sx0 + x) * SCp + cs0 + c];
}

// ------------------------------------------------------------------ fixed-kernel downsampler (fp32 NCHW, depthwise)
// out[n][c][oy][ox] = sum_ij K[i][j] x[n][c][clamp(oy*f + i - p)][clamp(ox*f + j - p)]   (ReplicationPad2d(p))
__global__ void downsample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ y,
                                      int NC, int H, int W, int OH, int OW, int k, int f, int p) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NC * OH * OW;
  if (idx >= total) return;
  int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), nc = (int)(idx / ((size_t)OW * OH));
  const float* xp = x + (size_t)nc * H * W;
  float acc = 0.f;
  for (int i = 0; i < k; ++i) {
    int yy = oy * f + i - p;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    for (int j = 0; j < k; ++j) {
      int xx = ox * f + j - p;
      xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
      acc += kern[i * k + j] * xp[(size_t)yy * W + xx];
    }
  }
  y[idx] = acc;
}

// adjoint, gather form: input (y,x) collects from every padded coordinate that replicates onto it
__global__ void downsample_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ kern,
                                      float* __restrict__ dx, int NC, int H, int W, int OH, int OW, int k, int f, int p) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)NC * H * W;
  if (idx >= total) return;
  int xx = (int)(idx % W), yy = (int)((idx / W) % H), nc = (int)(idx / ((size_t)W * H));
  const float* dp = dy + (size_t)nc * OH * OW;
  // padded coordinates a in [ya0, ya1] map onto row yy (a = unpadded coordinate, may be negative / >= H)
  const int ya0 = yy == 0 ? -p : yy, ya1 = yy == H - 1 ? H - 1 + p : yy;
  const int xa0 = xx == 0 ? -p : xx, xa1 = xx == W - 1 ? W - 1 + p : xx;
  float acc = 0.f;
  for (int a = ya0; a <= ya1; ++a) {
    for (int oy = 0; oy < OH; ++oy) {
      int i = a + p - oy * f;
      if (i < 0 || i >= k) continue;
      for (int b = xa0; b <= xa1; ++b) {
        for (int ox = 0; ox < OW; ++ox) {
          int j = b + p - ox * f;
          if (j < 0 || j >= k) continue;
          acc += kern[i * k + j] * dp[(size_t)oy * OW + ox];
        }
      }
    }
  }
  dx[idx] = acc;
}

// ================================================================== C ABI
#define DT_SWITCH2(dtype, CALL)          \
  if ((dtype) == DSR_BF16) {             \
    constexpr int DT = DSR_DTYPE_BF16;   \
    CALL;                                \
  } else {                               \
    constexpr int DT = DSR_DTYPE_F16;    \
    CALL;                                \
  }
static inline unsigned nb(size_t n) { return (unsigned)((n + 255) / 256); }

extern "C" int dsr_maxpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(x && y && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "maxpool2_fwd: null pointer or bad shape");
  DSR_REQUIRE(H >= 2 && W >= 2, "maxpool2_fwd: input smaller than the 2x2 window");
  size_t total = (size_t)N * (H / 2) * (W / 2) * (Cp / 8);
  DT_SWITCH2(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (unsigned short*)y, N, H, W, Cp));
  return dsr_launch_status("dsr_maxpool2_fwd");
}
extern "C" int dsr_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int Cp,
                                dsr_stream_t st) {
  DSR_REQUIRE(x && dy && dx && DSR_DTYPE_OK(dtype) && N > 0 && H >= 2 && W >= 2 && Cp >= 8 && Cp % 8 == 0, "maxpool2_bwd: null pointer or bad shape");
  size_t total = (size_t)N * H * W * (Cp / 8);
  DT_SWITCH2(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)dx, N, H, W,
                                       Cp, 0));
  return dsr_launch_status("dsr_maxpool2_bwd");
}
extern "C" int dsr_maxpool2_relu_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int Cp,
                                     dsr_stream_t st) {
  DSR_REQUIRE(x && dy && dx && DSR_DTYPE_OK(dtype) && N > 0 && H >= 2 && W >= 2 && Cp >= 8 && Cp % 8 == 0, "maxpool2_relu_bwd: null pointer or bad shape");
  size_t total = (size_t)N * H * W * (Cp / 8);
  DT_SWITCH2(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)dx, N, H, W,
                                       Cp, 1));
  return dsr_launch_status("dsr_maxpool2_relu_bwd");
}
extern "C" int dsr_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(x && y && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "avgpool2_fwd: null pointer or bad shape");
  size_t total = (size_t)N * (H / 2) * (W / 2) * (Cp / 8);
  if (total == 0) return 0;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((avgpool2_fwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (unsigned short*)y, N, H, W, Cp));
  return dsr_launch_status("dsr_avgpool2_fwd");
}
extern "C" int dsr_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(dy && dx && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "avgpool2_bwd: null pointer or bad shape");
  size_t total = (size_t)N * H * W * (Cp / 8);
  if (total == 0) return 0;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((avgpool2_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)dy, (unsigned short*)dx, N, H, W, Cp));
  return dsr_launch_status("dsr_avgpool2_bwd");
}
extern "C" int dsr_nearest2x_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(x && y && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "nearest2x_fwd: null pointer or bad shape");
  size_t total = (size_t)N * 4 * H * W * (Cp / 8);
  if (total == 0) return 0;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((nearest2x_fwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (unsigned short*)y, N, H, W, Cp));
  return dsr_launch_status("dsr_nearest2x_fwd");
}
extern "C" int dsr_nearest2x_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(dy && dx && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "nearest2x_bwd: null pointer or bad shape");
  size_t total = (size_t)N * H * W * (Cp / 8);
  if (total == 0) return 0;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((nearest2x_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)dy, (unsigned short*)dx, N, H, W, Cp));
  return dsr_launch_status("dsr_nearest2x_bwd");
}
extern "C" int dsr_bilinear2x_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(x && y && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "bilinear2x_fwd: null pointer or bad shape");
  size_t total = (size_t)N * 4 * H * W * (Cp / 8);
  DT_SWITCH2(dtype, hipLaunchKernelGGL((bilinear2x_fwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)x, (unsigned short*)y, N, H, W, Cp));
  return dsr_launch_status("dsr_bilinear2x_fwd");
}
extern "C" int dsr_bilinear2x_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t st) {
  DSR_REQUIRE(dy && dx && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && Cp >= 8 && Cp % 8 == 0, "bilinear2x_bwd: null pointer or bad shape");
  size_t total = (size_t)N * H * W * (Cp / 8);
  DT_SWITCH2(dtype, hipLaunchKernelGGL((bilinear2x_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)dy, (unsigned short*)dx, N, H, W, Cp));
  return dsr_launch_status("dsr_bilinear2x_bwd");
}
extern "C" int dsr_resize_norm_fwd(int dtype, const float* src, void* dst, int N, int C, int H, int W, int OH, int OW,
                                   const int* ys, const int* yc, const float* yw, const int* xs, const int* xc,
                                   const float* xw, int KT, const float* mean3, const float* std3, dsr_stream_t st) {
  DSR_REQUIRE(src && dst && ys && yc && yw && xs && xc && xw && mean3 && std3 && DSR_DTYPE_OK(dtype) && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && KT > 0, "resize_norm_fwd: null pointer or bad shape");
  if (C > 3) return dsr_fail(DSR_E_UNSUPPORTED, "resize_norm: C %d > 3", C);
  size_t total = (size_t)N * OH * OW;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((resize_norm_fwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st, src,
                                       (unsigned short*)dst, N, C, H, W, OH, OW, ys, yc, yw, xs, xc, xw, KT, mean3[0],
                                       mean3[1], mean3[2], std3[0], std3[1], std3[2]));
  return dsr_launch_status("dsr_resize_norm_fwd");
}
extern "C" int dsr_resize_norm_bwd(int dtype, const void* dout, float* dsrc, int N, int C, int H, int W, int OH, int OW,
                                   const int* ty_s, const int* ty_c, const float* ty_w, const int* tx_s,
                                   const int* tx_c, const float* tx_w, int KT, const float* std3, dsr_stream_t st) {
  DSR_REQUIRE(dout && dsrc && ty_s && ty_c && ty_w && tx_s && tx_c && tx_w && std3 && DSR_DTYPE_OK(dtype) && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && KT > 0, "resize_norm_bwd: null pointer or bad shape");
  if (C > 3) return dsr_fail(DSR_E_UNSUPPORTED, "resize_norm: C %d > 3", C);
  size_t total = (size_t)N * H * W;
  DT_SWITCH2(dtype, hipLaunchKernelGGL((resize_norm_bwd_kernel<DT>), dim3(nb(total)), dim3(256), 0, st,
                                       (const unsigned short*)dout, dsrc, N, C, H, W, OH, OW, ty_s, ty_c, ty_w, tx_s,
                                       tx_c, tx_w, KT, std3[0], std3[1], std3[2]));
  return dsr_launch_status("dsr_resize_norm_bwd");
}
extern "C" int dsr_box_copy(const void* src, void* dst, int N, int BH, int BW, int C, int SH, int SW, int SCp, int sy0,
                            int sx0, int cs0, int DH, int DW, int DCp, int dy0, int dx0, int cd0, dsr_stream_t st) {
  DSR_REQUIRE(src && dst && N > 0 && BH > 0 && BW > 0 && C > 0 && cs0 >= 0 && cd0 >= 0, "box_copy: null pointer or empty box");
  if (sy0 < 0 || sx0 < 0 || dy0 < 0 || dx0 < 0 || sy0 + BH > SH || sx0 + BW > SW || dy0 + BH > DH || dx0 + BW > DW ||
      cs0 + C > SCp || cd0 + C > DCp)
    return dsr_fail(DSR_E_ARG, "box_copy: box outside a tensor");
  size_t total = (size_t)N * BH * BW * C;
  hipLaunchKernelGGL(box_copy_kernel, dim3(nb(total)), dim3(256), 0, st, (const unsigned short*)src,
                     (unsigned short*)dst, N, BH, BW, C, SH, SW, SCp, sy0, sx0, cs0, DH, DW, DCp, dy0, dx0, cd0);
  return dsr_launch_status("dsr_box_copy");
}
extern "C" int dsr_downsample_fwd(const float* x, const float* kern, float* y, int NC, int H, int W, int k, int f,
                                  int p, dsr_stream_t st) {
  DSR_REQUIRE(x && kern && y && NC > 0 && H > 0 && W > 0 && k > 0 && f > 0 && p >= 0, "downsample_fwd: null pointer or bad shape");
  int OH = (H + 2 * p - k) / f + 1, OW = (W + 2 * p - k) / f + 1;
  if (OH < 1 || OW < 1) return dsr_fail(DSR_E_ARG, "downsample: empty output");
  size_t total = (size_t)NC * OH * OW;
  hipLaunchKernelGGL(downsample_fwd_kernel, dim3(nb(total)), dim3(256), 0, st, x, kern, y, NC, H, W, OH, OW, k, f, p);
  return dsr_launch_status("dsr_downsample_fwd");
}
extern "C" int dsr_downsample_bwd(const float* dy, const float* kern, float* dx, int NC, int H, int W, int k, int f,
                                  int p, dsr_stream_t st) {
  DSR_REQUIRE(dy && kern && dx && NC > 0 && H > 0 && W > 0 && k > 0 && f > 0 && p >= 0 && H + 2 * p >= k && W + 2 * p >= k, "downsample_bwd: null pointer or bad shape");
  int OH = (H + 2 * p - k) / f + 1, OW = (W + 2 * p - k) / f + 1;
  size_t total = (size_t)NC * H * W;
  hipLaunchKernelGGL(downsample_bwd_kernel, dim3(nb(total)), dim3(256), 0, st, dy, kern, dx, NC, H, W, OH, OW, k, f, p);
  return dsr_launch_status("dsr_downsample_bwd");
}
