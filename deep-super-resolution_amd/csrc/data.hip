// Data-side byte kernels: what the reference does per image on the host with Pillow / numpy before a
// training step (dataset.py:9-62,121-159; utils/degradation.py:5-20), on device-resident uint8 RGB images so that batches
// can be cut from a DIV2K set held in HBM.  All integer / byte arithmetic: results equal Pillow's and numpy's bit for bit
// (oracle/data.py is pinned against both).
//   resample_u8_kernel        one pass of Pillow's 8-bit resampler (libImaging/Resample.c): 22-bit fixed-point weights from
//                             the host (bounds + coefficient tables, float64 arithmetic as Pillow's precompute_coeffs),
//                             int32 accumulation from 1 << 21, arithmetic shift, clip to 0..255.  Image.resize = horizontal
//                             pass then vertical pass.
//   noise_gaussian_u8_kernel  utils/degradation.py:5-7: clip(image + noise, 0, 255) truncated to uint8 (float64 as in numpy
//                             when the noise was drawn there, float32 when drawn on the device)
//   salt_pepper_u8_kernel     utils/degradation.py:9-17
//   patch_batch_kernel        crop B patches out of B (different) images and convert: ToTensor (dataset.py:59-60) +
//                             scale_images (:149-159) -> fp32 NCHW batch, one launch
// These are HBM-bound byte movers on megabyte-sized images (a 2040 x 1356 DIV2K image is 8.3 MB); they are written for
// coalesced access (consecutive lanes touch consecutive bytes), not tuned further -- they are not in the timed step.
#include "../../include/dsr_hip.h"
#include "dsr_common.h"
#include "dsr_kernels.h"

namespace {
constexpr int PREC = 32 - 8 - 2;

// AXIS 1: dst[y][xx][c] from src[y][x0 + t][c];  AXIS 0: dst[yy][x][c] from src[y0 + t][x][c].  One thread per output BYTE:
// consecutive threads = consecutive bytes of a destination row (both passes read rows of the source contiguously).
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                          int H, int W, int C, int out_size, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize) {
  const int OH = AXIS == 0 ? out_size : H, OW = AXIS == 1 ? out_size : W;
  const size_t total = (size_t)OH * OW * C;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int rowbytes = OW * C;
  const int oy = (int)(idx / rowbytes);
  const int ob = (int)(idx - (size_t)oy * rowbytes);          // byte inside the destination row: ox * C + c
  const int o = AXIS == 0 ? oy : ob / C;                      // index along the resampled axis
  const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
  const int* __restrict__ k = kk + (size_t)o * ksize;
  int acc = 1 << (PREC - 1);
  if (AXIS == 1) {
    const int c = ob - o * C;
    const unsigned char* p = src + ((size_t)oy * W + x0) * C + c;
    for (int t = 0; t < n; ++t) acc += (int)p[(size_t)t * C] * k[t];
  } else {
    const unsigned char* p = src + (size_t)x0 * W * C + ob;
    for (int t = 0; t < n; ++t) acc += (int)p[(size_t)t * W * C] * k[t];
  }
  acc >>= PREC;                                               // arithmetic shift, then Pillow's clip8 table
  dst[idx] = (unsigned char)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

template <typename NT>
__global__ __launch_bounds__(256) void noise_gaussian_u8_kernel(const unsigned char* __restrict__ img, const NT* __restrict__ noise,
                                                                unsigned char* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  NT v = (NT)img[i] + noise[i];
  v = v < (NT)0 ? (NT)0 : (v > (NT)255 ? (NT)255 : v);        // np.clip
  out[i] = (unsigned char)v;                                  // astype(np.uint8): truncation
}

__global__ __launch_bounds__(256) void salt_pepper_u8_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ salt,
                                                             const unsigned char* __restrict__ pepper, unsigned char* __restrict__ out,
                                                             size_t pixels, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels * C) return;
  const size_t p = i / C;
  unsigned char v = img[i];
  if (salt[p]) v = 255;
  if (pepper[p]) v = 0;                                       // (pepper is applied after salt: it wins where both hit)
  out[i] = v;
}

struct PatchBatch {
  const unsigned char* img[DSR_PATCH_BATCH_MAX];
  int width[DSR_PATCH_BATCH_MAX];       // row pitch of the source image in pixels
  int top[DSR_PATCH_BATCH_MAX], left[DSR_PATCH_BATCH_MAX];
};

// out[b][c][y][x] (fp32) from img_b[top_b + y][left_b + x][c]; consecutive threads = consecutive x (4-byte stores coalesce,
// the 3-byte-pitch reads of a row share cache lines)
__global__ __launch_bounds__(256) void patch_batch_kernel(const PatchBatch t, int ph, int pw, int mode, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int per = 3 * ph * pw;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  const int c = i / (ph * pw);
  const int rem = i - c * ph * pw;
  const int y = rem / pw, x = rem - y * pw;
  const unsigned char u = t.img[b][((size_t)(t.top[b] + y) * t.width[b] + t.left[b] + x) * 3 + c];
  float v = (float)u / 255.0f;                                // torchvision ToTensor (dataset.py:59-60)
  if (mode == DSR_PATCH_LR_REF) {
    v = v / 255.0f;                                           // dataset.py:152 (a second division: the reference's behaviour)
  } else if (mode == DSR_PATCH_HR_REF) {
    v = v / 255.0f;                                           // :155
    v = v * 2.0f;                                             // :156
    v = v - 1.0f;                                             // :157
  } else if (mode == DSR_PATCH_HR_UNIT) {
    v = v * 2.0f;                                             // the scaling the reference's comments intend: [-1, 1]
    v = v - 1.0f;
  }
  out[(size_t)b * per + i] = v;
}
// dataset.py:149-159 in place on fp32 data that ToTensor already scaled: true IEEE divisions (ATen's device `x /= 255.0`
// multiplies by the reciprocal and lands one ulp off the host reference)
__global__ __launch_bounds__(256) void scale_images_kernel(float* __restrict__ x, size_t n, int mode) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i] / 255.0f;
  if (mode == DSR_PATCH_HR_REF) {
    v = v * 2.0f;
    v = v - 1.0f;
  }
  x[i] = v;
}
}  // namespace

extern "C" int dsr_scale_images_f32(float* x, size_t n, int mode, dsr_stream_t st) {
  DSR_REQUIRE(x && n > 0 && (mode == DSR_PATCH_LR_REF || mode == DSR_PATCH_HR_REF), "scale_images_f32: null pointer, empty tensor or bad mode");
  hipLaunchKernelGGL(scale_images_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, mode);
  return dsr_launch_status("dsr_scale_images_f32");
}

extern "C" int dsr_resample_u8(const unsigned char* src, unsigned char* dst, int H, int W, int C, int axis, int out_size,
                               const int* bounds, const int* kk, int ksize, dsr_stream_t st) {
  DSR_REQUIRE(src && dst && bounds && kk && H > 0 && W > 0 && C > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1),
              "resample_u8: null pointer or bad shape");
  const size_t total = (size_t)(axis == 0 ? out_size : H) * (axis == 1 ? out_size : W) * C;
  if (total > 0x7FFFFFFFull * 256) return dsr_fail(DSR_E_UNSUPPORTED, "resample_u8: image too large");
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (axis == 0)
    hipLaunchKernelGGL(resample_u8_kernel<0>, dim3(blocks), dim3(256), 0, st, src, dst, H, W, C, out_size, bounds, kk, ksize);
  else
    hipLaunchKernelGGL(resample_u8_kernel<1>, dim3(blocks), dim3(256), 0, st, src, dst, H, W, C, out_size, bounds, kk, ksize);
  return dsr_launch_status("dsr_resample_u8");
}

extern "C" int dsr_noise_gaussian_u8(const unsigned char* img, const void* noise, int noise_is_f64, unsigned char* out, size_t n,
                                     dsr_stream_t st) {
  DSR_REQUIRE(img && noise && out && n > 0, "noise_gaussian_u8: null pointer or empty image");
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (noise_is_f64)
    hipLaunchKernelGGL(noise_gaussian_u8_kernel<double>, dim3(blocks), dim3(256), 0, st, img, (const double*)noise, out, n);
  else
    hipLaunchKernelGGL(noise_gaussian_u8_kernel<float>, dim3(blocks), dim3(256), 0, st, img, (const float*)noise, out, n);
  return dsr_launch_status("dsr_noise_gaussian_u8");
}

extern "C" int dsr_salt_pepper_u8(const unsigned char* img, const unsigned char* salt, const unsigned char* pepper,
                                  unsigned char* out, int H, int W, int C, dsr_stream_t st) {
  DSR_REQUIRE(img && salt && pepper && out && H > 0 && W > 0 && C > 0, "salt_pepper_u8: null pointer or bad shape");
  const size_t pixels = (size_t)H * W;
  hipLaunchKernelGGL(salt_pepper_u8_kernel, dim3((unsigned)((pixels * C + 255) / 256)), dim3(256), 0, st, img, salt, pepper, out,
                     pixels, C);
  return dsr_launch_status("dsr_salt_pepper_u8");
}

extern "C" int dsr_patch_batch_u8(int count, const unsigned char* const* images, const int* heights, const int* widths,
                                  const int* tops, const int* lefts, int ph, int pw, int mode, float* out, dsr_stream_t st) {
  if (count <= 0 || !images || !heights || !widths || !tops || !lefts || !out || ph <= 0 || pw <= 0)
    return dsr_fail(DSR_E_ARG, "patch_batch_u8: null table or bad shape");
  if (mode < DSR_PATCH_UNIT || mode > DSR_PATCH_HR_UNIT) return dsr_fail(DSR_E_ARG, "patch_batch_u8: mode %d", mode);
  for (int i = 0; i < count; ++i) {
    if (!images[i]) return dsr_fail(DSR_E_ARG, "patch_batch_u8: null image %d", i);
    if (tops[i] < 0 || lefts[i] < 0 || tops[i] + ph > heights[i] || lefts[i] + pw > widths[i])
      return dsr_fail(DSR_E_ARG, "patch_batch_u8: patch %d (%d,%d)+(%d,%d) leaves its %dx%d image", i, tops[i], lefts[i], ph, pw,
                      heights[i], widths[i]);
  }
  const int per = 3 * ph * pw;
  for (int i0 = 0; i0 < count; i0 += DSR_PATCH_BATCH_MAX) {
    PatchBatch t;
    const int n = count - i0 < DSR_PATCH_BATCH_MAX ? count - i0 : DSR_PATCH_BATCH_MAX;
    for (int j = 0; j < n; ++j) {
      t.img[j] = images[i0 + j];
      t.width[j] = widths[i0 + j];
      t.top[j] = tops[i0 + j];
      t.left[j] = lefts[i0 + j];
    }
    hipLaunchKernelGGL(patch_batch_kernel, dim3((per + 255) / 256, n), dim3(256), 0, st, t, ph, pw, mode, out + (size_t)i0 * per);
  }
  return dsr_launch_status("dsr_patch_batch_u8");
}
