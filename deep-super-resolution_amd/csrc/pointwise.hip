// HBM-bound pointwise / reduction kernels around the convolutions (gfx950).
// Everything works on NHWC 16-bit tensors with channels padded to 8, one 16-byte vector
// (8 channels of one pixel) per access, fp32 math, and deterministic two-stage reductions
// (per-block partial rows written with plain stores, summed by a finalize kernel) -- no atomics.
//
// Reference semantics covered here:
//   nn.BatchNorm2d train/eval forward + backward      models/GAN/generator.py:8,12,53; discriminator.py:10;
//                                                      models/DIP/utils.py:79-80
//   PReLU(1) / LeakyReLU(0.2) / Tanh / Sigmoid / ReLU  generator.py:9,34,48,64; discriminator.py:12,27,41,45;
//                                                      models/DIP/utils.py:68; skip.py:94
//   residual add x + z                                 generator.py:23,74
//   PixelShuffle(2) backward (un-shuffle)              generator.py:32,38
#include <stdlib.h>

#include "dsr_common.h"
#include "dsr_kernels.h"
#include "../../include/dsr_hip.h"

// ------------------------------------------------------------------ layout conversion
template <int DT>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int N, int C, int H,
                                    int W, int Cp) {
  // one thread = one pixel x 8-channel chunk
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int cpr = Cp / 8;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int hw = (int)(pix % ((size_t)H * W));
  int n = (int)(pix / ((size_t)H * W));
  float f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int c = ch * 8 + k;
    f[k] = c < C ? src[((size_t)n * C + c) * H * W + hw] : 0.f;
  }
  *reinterpret_cast<U4*>(dst + pix * Cp + ch * 8) = pack8<DT>(f);
}

template <int DT>
__global__ void nhwc_to_nchw_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int N, int C, int H,
                                    int W, int Cp) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * C * H * W;
  if (idx >= total) return;
  int hw = (int)(idx % ((size_t)H * W));
  int c = (int)((idx / ((size_t)H * W)) % C);
  int n = (int)(idx / ((size_t)H * W * C));
  dst[idx] = h2f<DT>(src[((size_t)n * H * W + hw) * Cp + c]);
}

// ------------------------------------------------------------------ weight packing
// w [Cout][Cin][KH][KW] fp32 -> fwd [T][NBo][CinP] and dgrad [T][NBi][CoutP] (16-bit, zero padded)
template <int DT>
__global__ void pack_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                   unsigned short* __restrict__ wd, int Cout, int Cin, int T, int NBo, int CinP, int NBi,
                                   int CoutP) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t nf = (size_t)T * NBo * CinP, nd = (size_t)T * NBi * CoutP;
  if (idx < nf) {
    int ci = (int)(idx % CinP);
    int co = (int)((idx / CinP) % NBo);
    int t = (int)(idx / ((size_t)CinP * NBo));
    float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + t] : 0.f;
    wf[idx] = f2h<DT>(v);
  }
  if (wd != nullptr && idx < nd) {
    int co = (int)(idx % CoutP);
    int ci = (int)((idx / CoutP) % NBi);
    int t = (int)(idx / ((size_t)CoutP * NBi));
    float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + t] : 0.f;
    wd[idx] = f2h<DT>(v);
  }
}

// All conv weights of an optimiser re-packed in one launch per 48 tensors (after Adam rewrote them): the per-layer
// launches above are 45 x ~8 us per step for a few MB of data.
#define DSR_PACK_GROUP 48
struct PackGroup {
  const float* w[DSR_PACK_GROUP];
  unsigned short* wf[DSR_PACK_GROUP];
  unsigned short* wd[DSR_PACK_GROUP];
  int cout[DSR_PACK_GROUP], cin[DSR_PACK_GROUP], taps[DSR_PACK_GROUP];
  unsigned first_block[DSR_PACK_GROUP + 1];
  int count;
};
template <int DT>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackGroup a) {
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) ++t;      // block-uniform scan
  const int Cout = a.cout[t], Cin = a.cin[t], T = a.taps[t];
  const int CoutP = (Cout + 7) & ~7, CinP = (Cin + 7) & ~7;
  const float* __restrict__ w = a.w[t];
  const size_t idx = (size_t)(blockIdx.x - a.first_block[t]) * 256 + threadIdx.x;
  const size_t nf = (size_t)T * CoutP * CinP;                              // both images have T * CoutP * CinP elements
  if (idx >= nf) return;
  {
    const int ci = (int)(idx % CinP);
    const int co = (int)((idx / CinP) % CoutP);
    const int tp = (int)(idx / ((size_t)CinP * CoutP));
    a.wf[t][idx] = f2h<DT>((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + tp] : 0.f);
  }
  {
    const int co = (int)(idx % CoutP);
    const int ci = (int)((idx / CoutP) % CinP);
    const int tp = (int)(idx / ((size_t)CoutP * CinP));
    a.wd[t][idx] = f2h<DT>((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + tp] : 0.f);
  }
}
extern "C" int dsr_conv_pack_weight_multi(int dtype, int count, const float* const* w, void* const* wf, void* const* wd,
                                          const int* cout, const int* cin, const int* taps, hipStream_t st) {
  if (count < 0 || (count && (!w || !wf || !wd || !cout || !cin || !taps))) return dsr_fail(DSR_E_ARG, "pack_weight_multi: null table");
  for (int i0 = 0; i0 < count; i0 += DSR_PACK_GROUP) {
    PackGroup g;
    g.count = count - i0 < DSR_PACK_GROUP ? count - i0 : DSR_PACK_GROUP;
    unsigned blocks = 0;
    for (int j = 0; j < g.count; ++j) {
      g.w[j] = w[i0 + j];
      g.wf[j] = (unsigned short*)wf[i0 + j];
      g.wd[j] = (unsigned short*)wd[i0 + j];
      g.cout[j] = cout[i0 + j];
      g.cin[j] = cin[i0 + j];
      g.taps[j] = taps[i0 + j];
      g.first_block[j] = blocks;
      const size_t n = (size_t)taps[i0 + j] * ((cout[i0 + j] + 7) & ~7) * ((cin[i0 + j] + 7) & ~7);
      blocks += (unsigned)((n + 255) / 256);
    }
    g.first_block[g.count] = blocks;
    if (!blocks) continue;
    if (dtype == DSR_BF16)
      hipLaunchKernelGGL((pack_weight_multi_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((pack_weight_multi_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, g);
  }
  return dsr_launch_status("dsr_conv_pack_weight_multi");
}

// ------------------------------------------------------------------ parallel compaction of partial rows
// in: [rows][width] fp32 partial sums  ->  out: [nchunks][width], out[c] = sum of rows [c*rpc, (c+1)*rpc).
// 64 columns x 4 row-lanes per block; fixed summation order => deterministic.  The finalize kernels below then
// walk <= DSR_COMPACT_ROWS rows instead of up to 65,536 (one per conv M-tile).
#define DSR_COMPACT_ROWS 64
__global__ __launch_bounds__(256) void compact_rows_kernel(const float* __restrict__ in, int rows, int width, int rpc,
                                                           float* __restrict__ out) {
  __shared__ double red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rpc;
  int r1 = r0 + rpc;
  if (r1 > rows) r1 = rows;
  double s = 0.0;
  if (col < width) {
    int r = r0 + ry;
    for (; r + 28 < r1; r += 32) {          // 8 independent loads in flight, summed in row order (deterministic)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = in[(size_t)(r + 4 * u) * width + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; r < r1; r += 4) s += (double)in[(size_t)r * width + col];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < width) out[(size_t)blockIdx.y * width + col] = (float)(red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// returns the number of rows left (and the pointer to use); scratch = `partial + rows*width` (caller reserves
// DSR_COMPACT_ROWS extra rows behind every partial buffer, see dsr_pw_scratch_rows()).
static const float* compact_rows(const float* partial, int rows, int width, int* rows_out, hipStream_t st) {
  if (rows <= 32) {
    *rows_out = rows;
    return partial;
  }
  // few rows in: the finalize kernels walk them serially with one thread per channel (a latency chain)
  const int target = rows >= 4096 ? DSR_COMPACT_ROWS : 16;
  int rpc = (rows + target - 1) / target;
  int nch = (rows + rpc - 1) / rpc;
  float* out = const_cast<float*>(partial) + (size_t)rows * width;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((width + 63) / 64, nch), dim3(256), 0, st, partial, rows, width, rpc, out);
  *rows_out = nch;
  return out;
}

// ------------------------------------------------------------------ generic partial-row sum
// out[c] = sum_b partial[b*stride_b + c]   (fp64 accumulate), optional scale
__global__ void sum_rows_kernel(const float* __restrict__ partial, int rows, int row_stride, int col_offset, int C,
                                float scale, float* __restrict__ out, int accumulate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  int b = 0;
  for (; b + 8 <= rows; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(b + u) * row_stride + col_offset + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)t[u];
  }
  for (; b < rows; ++b) s += (double)partial[(size_t)b * row_stride + col_offset + c];
  float v = (float)(s * (double)scale);
  out[c] = accumulate ? out[c] + v : v;
}

// ------------------------------------------------------------------ BatchNorm statistics -> affine
// stats partial rows: [tiles][2][stride] (sum, sumsq) from the conv epilogue.
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int tiles, int stride, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches, float momentum, float eps, int updates,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out, int Cp) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  if (c >= C) {   // padded channels: identity-free zeros
    scale[c] = 0.f;
    shift[c] = 0.f;
    mean_out[c] = 0.f;
    rstd_out[c] = 0.f;
    return;
  }
  double s1 = 0.0, s2 = 0.0;
  int t = 0;
  for (; t + 8 <= tiles; t += 8) {          // 16 independent loads in flight; summed in tile order
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = partial[((size_t)(t + u) * 2 + 0) * stride + c];
      b[u] = partial[((size_t)(t + u) * 2 + 1) * stride + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += (double)a[u];
      s2 += (double)b[u];
    }
  }
  for (; t < tiles; ++t) {
    s1 += (double)partial[((size_t)t * 2 + 0) * stride + c];
    s2 += (double)partial[((size_t)t * 2 + 1) * stride + c];
  }
  double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean != nullptr) {
    double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    float rm = running_mean[c], rv = running_var[c];
    for (int u = 0; u < updates; ++u) {
      rm = (1.f - momentum) * rm + momentum * (float)mean;
      rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
  if (c == 0 && num_batches != nullptr) *num_batches += updates;
}

// ---- single-launch finalize for up to DSR_FINALIZE_PAR_ROWS partial rows: 64 channels x 4 row-lanes per block.  Row-lane
// q sums rows q, q+4, ... in that order with 8 loads in flight (double accumulators); the four partial sums are combined
// in a fixed order through LDS, so the result does not depend on timing.  Replaces compact_rows + the serial finalize
// (two launches and a boundary) wherever the producers emit few rows: the persistent c64 kernel (one row per block),
// and the small problems (config 2, DIP).
#define DSR_FINALIZE_PAR_ROWS 512      // measured (a sweep over the row count): one launch wins up to ~512 rows (6.6 us vs ~10 us for
                                       // compaction + finalize); at 2048 rows its latency chain loses (18.6 us vs 8 us)
#define DSR_FINALIZE_PAR_ROWS_BWD 64   // three slices per row: loses from 512 rows on (36.7 us)
// CW channels x (256 / CW) row-lanes per block: with CW = 16 a 64-channel layer is summed by four blocks of 16 lanes each
// (512 rows = 32 per lane = four rounds of 8 loads) instead of one block whose 4 lanes walk 128 rows each (16 rounds: the
// 9.8 us of the single-block form were that load chain).
template <int NS, int CW>
__device__ __forceinline__ void par_column_sums(const float* __restrict__ partial, int rows, int row_stride, int slice_stride,
                                                int c, bool active, double (&tot)[NS], double* red /* [NS][256 / CW][CW] */) {
  constexpr int LANES = 256 / CW;
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  double s[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) s[k] = 0.0;
  if (active) {
    int r = ry;
    for (; r + 7 * LANES < rows; r += 8 * LANES) {
      float v[NS][8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k][u] = partial[(size_t)(r + LANES * u) * row_stride + k * slice_stride + c];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += (double)v[k][u];
    }
    for (; r < rows; r += LANES)
#pragma unroll
      for (int k = 0; k < NS; ++k) s[k] += (double)partial[(size_t)r * row_stride + k * slice_stride + c];
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) red[(k * LANES + ry) * CW + cx] = s[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < LANES; ++q) t += red[(k * LANES + q) * CW + cx];      // fixed order: deterministic
    tot[k] = t;
  }
}

template <int CW>
__global__ __launch_bounds__(256) void bn_finalize_par_kernel(
    const float* __restrict__ partial, int tiles, int stride, int C, float count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ num_batches, float momentum, float eps, int updates, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out, int Cp) {
  __shared__ double red[2 * 256];
  const int c = blockIdx.x * CW + (threadIdx.x % CW);
  double tot[2];
  par_column_sums<2, CW>(partial, tiles, 2 * stride, stride, c, c < C, tot, red);
  if (threadIdx.x >= CW || c >= Cp) return;
  if (c >= C) {   // padded channels
    scale[c] = 0.f;
    shift[c] = 0.f;
    mean_out[c] = 0.f;
    rstd_out[c] = 0.f;
    return;
  }
  const double mean = tot[0] / count;
  double var = tot[1] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    float rm = running_mean[c], rv = running_var[c];
    for (int u = 0; u < updates; ++u) {
      rm = (1.f - momentum) * rm + momentum * (float)mean;
      rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
  if (c == 0 && num_batches != nullptr) *num_batches += updates;
}

// backward: partial rows [blocks][3][Cp] = (sum g, sum g*xhat, PReLU terms).  One block of 64 channels at a time; the PReLU
// slope gradient (a sum over channels) is only produced by the single-block case Cp <= 64 (the generator's layers).
__global__ __launch_bounds__(256) void bn_bwd_finalize_par_kernel(const float* __restrict__ partial, int blocks, int C, int Cp,
                                                                  float count, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ dprelu,
                                                                  float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double red[3 * 4 * 64];
  const int cx = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + cx;
  double tot[3];
  par_column_sums<3, 64>(partial, blocks, 3 * Cp, Cp, c, c < Cp, tot, red);
  if (threadIdx.x < 64 && c < Cp) {
    if (c < C) tot[1] = (double)rstd[c] * (tot[1] - (double)mean[c] * tot[0]);      // sum g*y -> sum g*xhat
    if (c < C) {
      if (dgamma) dgamma[c] = (float)tot[1];
      if (dbeta) dbeta[c] = (float)tot[0];
    }
    c1[c] = c < C ? (float)(tot[0] / count) : 0.f;
    c2[c] = c < C ? (float)(tot[1] / count) : 0.f;
  }
  if (dprelu) {        // (Cp <= 64: one block) channel sum in a fixed order
    __syncthreads();
    if (threadIdx.x < 64) red[cx] = c < C ? tot[2] : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int i = 0; i < 64; ++i) s += red[i];
      dprelu[0] = (float)s;
    }
  }
}

__global__ __launch_bounds__(256) void sum_rows_par_kernel(const float* __restrict__ partial, int rows, int row_stride,
                                                           int col_offset, int C, float scale, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ double red[4 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double tot[1];
  par_column_sums<1, 64>(partial + col_offset, rows, row_stride, 0, c, c < C, tot, red);
  if (threadIdx.x >= 64 || c >= C) return;
  const float v = (float)(tot[0] * (double)scale);
  out[c] = accumulate ? out[c] + v : v;
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, int C, int Cp, float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  if (c >= C) {
    scale[c] = 0.f;
    shift[c] = 0.f;
    if (mean_out) mean_out[c] = 0.f;
    if (rstd_out) rstd_out[c] = 0.f;
    return;
  }
  float rstd = 1.f / sqrtf(running_var[c] + eps);
  float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
  if (mean_out) mean_out[c] = running_mean[c];
  if (rstd_out) rstd_out[c] = rstd;
}

// channel statistics of a plain NHWC tensor (used where BN does not follow a conv: DIP's BN after Concat)
template <int DT>
__global__ __launch_bounds__(256) void channel_stats_kernel(const unsigned short* __restrict__ x, size_t P, int Cp,
                                                            int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[256 * 16];
  const int cpr = Cp / 8;
  const int tid = threadIdx.x;
  const int rpi = 256 / cpr;                 // rows per iteration
  const int ch = tid % cpr, rr = tid / cpr;
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
  size_t p0 = (size_t)blockIdx.x * rows_per_block;
  size_t p1 = p0 + rows_per_block;
  if (p1 > P) p1 = P;
  if (rr < rpi) {
    for (size_t p = p0 + rr; p < p1; p += rpi) {
      U4 v = *reinterpret_cast<const U4*>(x + p * Cp + ch * 8);
      float f[8];
      unpack8<DT>(v, f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s1[k] += f[k];
        s2[k] += f[k] * f[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[tid * 16 + k] = s1[k];
    red[tid * 16 + 8 + k] = s2[k];
  }
  __syncthreads();
  for (int c = tid; c < 2 * Cp; c += 256) {
    int which = c / Cp, cc = c % Cp;
    int chn = cc / 8, k = cc % 8;
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[(r * cpr + chn) * 16 + which * 8 + k];
    partial[((size_t)blockIdx.x * 2 + which) * Cp + cc] = s;
  }
}

// ------------------------------------------------------------------ BN-apply + activation (+ residual)
// Every thread owns one 8-channel chunk and walks pixels: the per-channel affine parameters are loaded ONCE into
// registers (a one-vector-per-thread version spends more load instructions on parameters than on payload and
// reaches only ~0.8 TB/s).
// 8 consecutive per-channel fp32 parameters as two 16-byte loads (eight scalar loads per array put a chain of ~30 dependent
// L2 round trips in front of a kernel that only streams for 20-30 us)
typedef __attribute__((ext_vector_type(4))) float PF4;
__device__ __forceinline__ void load8(const float* __restrict__ src, int c0, float (&dst)[8]) {
  const PF4 a = *reinterpret_cast<const PF4*>(src + c0), b = *reinterpret_cast<const PF4*>(src + c0 + 4);
  dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
  dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}
// streaming accesses: tensors far beyond the 256 MiB Infinity Cache stream fastest with nontemporal accesses, cache-sized
// ones with plain ones (tools/stream_probe.hip: 6.0-6.6 vs 5.2-5.9 TB/s at 537 MB, 6.0-6.4 vs 6.5-7.1 TB/s at 67 MB)
template <bool NT>
__device__ __forceinline__ U4 ld16(const unsigned short* p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(p)) : *reinterpret_cast<const U4*>(p);
}
template <bool NT>
__device__ __forceinline__ unsigned ld4(const unsigned short* p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p)) : *reinterpret_cast<const unsigned*>(p);
}
template <bool NT>
__device__ __forceinline__ void st16(unsigned short* p, const U4& v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<U4*>(p)); else *reinterpret_cast<U4*>(p) = v;
}
#define DSR_PW_NT_BYTES (192ull << 20)   // operand tensors of at least this size use the nontemporal path

// ACTC (this and the three kernels below): the activation as a compile-time constant (-1: the run-time `act`).  With the run-time
// value every element went through act_apply's / act_grad_from_out's chain of scalar compares and branches (114-589 s_cbranch
// per kernel): the launchers pick the instantiation for None / LeakyReLU / PReLU / ReLU.
template <int DT, bool NT, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const unsigned short* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const unsigned short* __restrict__ residual,
                                                         unsigned short* __restrict__ out, size_t P, int Cp, int act_rt,
                                                         float slope_v, const float* __restrict__ prelu) {
  const int act = ACTC >= 0 ? ACTC : act_rt;
  const int cpr = Cp / 8;
  const int rpi = 256 / cpr;
  const int ch = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  if (rr >= rpi) return;
  const float slope = prelu ? prelu[0] : slope_v;
  float sc[8], sh[8];
  if (scale) {
    load8(scale, ch * 8, sc);
    load8(shift, ch * 8, sh);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) sc[k] = 1.f, sh[k] = 0.f;
  }
  auto apply = [&](const U4& vy, const U4& vr) {
    float f[8], r[8];
    unpack8<DT>(vy, f);
    if (residual) unpack8<DT>(vr, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = act_apply(act, f[k] * sc[k] + sh[k], slope);
      if (residual) v += r[k];
      f[k] = v;
    }
    return pack8<DT>(f);
  };
  // two pixel rows per iteration: all loads of an iteration are issued before the first use
  const size_t stride = (size_t)gridDim.x * rpi;
  size_t p = (size_t)blockIdx.x * rpi + rr;
  const size_t coff = (size_t)ch * 8;
  for (; p + stride < P; p += 2 * stride) {
    const size_t o0 = p * Cp + coff, o1 = (p + stride) * Cp + coff;
    const U4 y0 = ld16<NT>(y + o0), y1 = ld16<NT>(y + o1);
    U4 r0 = y0, r1 = y1;
    if (residual) {
      r0 = ld16<NT>(residual + o0);
      r1 = ld16<NT>(residual + o1);
    }
    st16<NT>(out + o0, apply(y0, r0));
    st16<NT>(out + o1, apply(y1, r1));
  }
  if (p < P) {
    const size_t o0 = p * Cp + coff;
    const U4 y0 = ld16<NT>(y + o0);
    const U4 r0 = residual ? ld16<NT>(residual + o0) : y0;
    st16<NT>(out + o0, apply(y0, r0));
  }
}

// backward of out = act(scale*y + shift) [+ residual]; g = dout * act'(z).
// pass 1: per-channel partial sums of g, g*xhat and the PReLU slope gradient sum(dout * z * [z<0]).
// WITH_P: the launch also forms the PReLU slope gradient (8 more accumulators).  The per-channel scale / shift live in LDS (read
// per row pair) rather than in 16 registers: with both, the two-rows-in-flight form fits 5 waves per SIMD instead of 4.
template <int DT, bool NT, int UNR, bool WITH_P, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(
    const unsigned short* __restrict__ dout, const unsigned short* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd, size_t P, int Cp,
    int rows_per_block, int act_rt, float slope_v, const float* __restrict__ prelu, float* __restrict__ partial) {
  const int act = ACTC >= 0 ? ACTC : act_rt;
  __shared__ float red[256 * 24];
  const int cpr = Cp / 8;
  const int tid = threadIdx.x;
  const int rpi = 256 / cpr;
  const int ch = tid % cpr, rr = tid / cpr;
  const float slope = prelu ? prelu[0] : slope_v;
  // slot 1 holds sum g*y (the raw conv output), not sum g*xhat: xhat = (y - mean) * rstd is affine in y, so the finalize
  // kernel forms sum g*xhat = rstd * (sum g*y - mean * sum g) in double precision and this kernel keeps two parameter
  // arrays less in registers (134 -> under 100 VGPRs: 3 -> 5 waves per SIMD)
  float sg[8], sgx[8];
  [[maybe_unused]] float sp[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sg[k] = sgx[k] = sp[k] = 0.f;
  float* s_aff = red;                                   // [2][Cp] in front of the reduction rows (the block-end reduction starts behind a barrier)
  for (int c = tid; c < Cp; c += 256) {
    s_aff[c] = scale[c];
    s_aff[Cp + c] = shift[c];
  }
  __syncthreads();
  size_t p0 = (size_t)blockIdx.x * rows_per_block;
  size_t p1 = p0 + rows_per_block;
  if (p1 > P) p1 = P;
  auto accum = [&](const U4& vd, const U4& vy) {
    float d[8], f[8], csc[8], csh[8];
    unpack8<DT>(vd, d);
    unpack8<DT>(vy, f);
    load8(s_aff, ch * 8, csc);
    load8(s_aff + Cp, ch * 8, csh);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = f[k] * csc[k] + csh[k];
      const float o = act_apply(act, z, slope);
      const float gg = d[k] * act_grad_from_out(act, (act == DSR_ACT_LEAKY || act == DSR_ACT_PRELU) ? z : o, slope);
      sg[k] += gg;
      sgx[k] += gg * f[k];
      if constexpr (WITH_P)
        if (z < 0.f) sp[k] += d[k] * z;
    }
  };
  if (rr < rpi) {
    const size_t coff = (size_t)ch * 8;
    size_t p = p0 + rr;
    if constexpr (UNR == 2) {
      for (; p + rpi < p1; p += 2 * rpi) {       // two pixel rows in flight (summed in row order: deterministic)
        const size_t o0 = p * Cp + coff, o1 = (p + rpi) * Cp + coff;
        const U4 d0 = ld16<NT>(dout + o0), y0 = ld16<NT>(y + o0), d1 = ld16<NT>(dout + o1), y1 = ld16<NT>(y + o1);
        accum(d0, y0);
        accum(d1, y1);
      }
    } else {
      for (; p + rpi < p1; p += rpi) {
        const size_t o0 = p * Cp + coff;
        accum(ld16<NT>(dout + o0), ld16<NT>(y + o0));
      }
    }
    if (p < p1) {
      const size_t o0 = p * Cp + coff;
      accum(ld16<NT>(dout + o0), ld16<NT>(y + o0));
    }
  }
  __syncthreads();                                      // every thread is done with the scale / shift rows
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[tid * 24 + k] = sg[k];
    red[tid * 24 + 8 + k] = sgx[k];
    red[tid * 24 + 16 + k] = WITH_P ? sp[k] : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < 3 * Cp; c += 256) {
    int which = c / Cp, cc = c % Cp;
    int chn = cc / 8, k = cc % 8;
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[(r * cpr + chn) * 24 + which * 8 + k];
    partial[((size_t)blockIdx.x * 3 + which) * Cp + cc] = s;
  }
}

// finalize: dgamma = sum g*xhat, dbeta = sum g, dprelu = sum over channels; c1 = dbeta/n, c2 = dgamma/n
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int C, int Cp, float count,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ dprelu, float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double sp[256];
  int c = threadIdx.x;
  double acc_p = 0.0;
  for (int cc = c; cc < Cp; cc += blockDim.x) {
    double g = 0.0, gx = 0.0, p = 0.0;
    int b = 0;
    for (; b + 4 <= blocks; b += 4) {       // 12 independent loads in flight; summed in block order
      float t0[4], t1[4], t2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t0[u] = partial[((size_t)(b + u) * 3 + 0) * Cp + cc];
        t1[u] = partial[((size_t)(b + u) * 3 + 1) * Cp + cc];
        t2[u] = partial[((size_t)(b + u) * 3 + 2) * Cp + cc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g += (double)t0[u];
        gx += (double)t1[u];
        p += (double)t2[u];
      }
    }
    for (; b < blocks; ++b) {
      g += (double)partial[((size_t)b * 3 + 0) * Cp + cc];
      gx += (double)partial[((size_t)b * 3 + 1) * Cp + cc];
      p += (double)partial[((size_t)b * 3 + 2) * Cp + cc];
    }
    if (cc < C) {
      gx = (double)rstd[cc] * (gx - (double)mean[cc] * g);      // sum g*y -> sum g*xhat
      if (dgamma) dgamma[cc] = (float)gx;
      if (dbeta) dbeta[cc] = (float)g;
      acc_p += p;
    }
    c1[cc] = cc < C ? (float)(g / count) : 0.f;
    c2[cc] = cc < C ? (float)(gx / count) : 0.f;
  }
  sp[threadIdx.x] = acc_p;
  __syncthreads();
  if (threadIdx.x == 0 && dprelu) {
    double s = 0.0;
    for (int i = 0; i < (int)blockDim.x; ++i) s += sp[i];
    dprelu[0] = (float)s;
  }
}

// pass 2: dy = scale * (g - c1 - xhat * c2)      (scale = gamma * rstd), folded per channel into
//   dy = A*g + B*y + C,  A = scale, B = -scale*c2*rstd, C = scale*(c2*mean*rstd - c1);  eval mode: dy = scale*g.
template <int DT, bool NT, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const unsigned short* __restrict__ dout, const unsigned short* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ c1, const float* __restrict__ c2, unsigned short* __restrict__ dy, size_t P, int Cp,
    int act_rt, float slope_v, const float* __restrict__ prelu, int train) {
  const int act = ACTC >= 0 ? ACTC : act_rt;
  const int cpr = Cp / 8;
  const int rpi = 256 / cpr;
  const int ch = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  if (rr >= rpi) return;
  const float slope = prelu ? prelu[0] : slope_v;
  float sc[8], sh[8], cb[8], cc[8];
  load8(scale, ch * 8, sc);
  load8(shift, ch * 8, sh);
  if (train) {
    float k1[8], k2[8], mu[8], rs[8];
    load8(c1, ch * 8, k1);
    load8(c2, ch * 8, k2);
    load8(mean, ch * 8, mu);
    load8(rstd, ch * 8, rs);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      cb[k] = -sc[k] * k2[k] * rs[k];
      cc[k] = sc[k] * (k2[k] * mu[k] * rs[k] - k1[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) cb[k] = cc[k] = 0.f;
  }
  const bool lin = act == DSR_ACT_LEAKY || act == DSR_ACT_PRELU;
  auto apply = [&](const U4& vd, const U4& vy) {
    float d[8], f[8];
    unpack8<DT>(vd, d);
    unpack8<DT>(vy, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = f[k] * sc[k] + sh[k];
      const float gg = d[k] * act_grad_from_out(act, lin ? z : act_apply(act, z, slope), slope);
      f[k] = sc[k] * gg + cb[k] * f[k] + cc[k];
    }
    return pack8<DT>(f);
  };
  const size_t stride = (size_t)gridDim.x * rpi;
  size_t p = (size_t)blockIdx.x * rpi + rr;
  const size_t coff = (size_t)ch * 8;
  for (; p + stride < P; p += 2 * stride) {
    const size_t o0 = p * Cp + coff, o1 = (p + stride) * Cp + coff;
    const U4 d0 = ld16<NT>(dout + o0), y0 = ld16<NT>(y + o0), d1 = ld16<NT>(dout + o1), y1 = ld16<NT>(y + o1);
    st16<NT>(dy + o0, apply(d0, y0));
    st16<NT>(dy + o1, apply(d1, y1));
  }
  if (p < P) {
    const size_t o0 = p * Cp + coff;
    const U4 d0 = ld16<NT>(dout + o0), y0 = ld16<NT>(y + o0);
    st16<NT>(dy + o0, apply(d0, y0));
  }
}

// ------------------------------------------------------------------ activation backward for conv+act layers
// dy[conv layout] = dout * act'(out), where out is the stored activation OUTPUT (valid for slope > 0).
// pixshuf: out/dout are [N][2H][2W][Cq] and dy is [N][H][W][4*C] with channel 4c+2i+j <- pixel (2h+i,2w+j).
// Also emits per-block partial rows: [blocks][2][CyP] = (bias grad column sums, PReLU-slope grad terms).
template <int DT, bool NT, int ACTC = -1>
__global__ __launch_bounds__(256) void act_bwd_kernel(const unsigned short* __restrict__ dout,
                                                      const unsigned short* __restrict__ out,
                                                      unsigned short* __restrict__ dy, int N, int H, int W, int CyP,
                                                      int CoP, int pixshuf, int act_rt, float slope_v,
                                                      const float* __restrict__ prelu, int rows_per_block,
                                                      float* __restrict__ partial) {
  const int act = ACTC >= 0 ? ACTC : act_rt;
  __shared__ float red[256 * 16];
  const int cpr = CyP / 8;
  const int tid = threadIdx.x;
  const int rpi = 256 / cpr;
  const int ch = tid % cpr, rr = tid / cpr;
  const float slope = prelu ? prelu[0] : slope_v;
  // The sign of the pre-activation is read off the stored OUTPUT, which is only possible for a positive slope (with
  // slope <= 0 both branches of PReLU / LeakyReLU give outputs of one sign).  A learned PReLU slope that has crossed
  // zero must not produce silently wrong gradients: every gradient of this launch becomes NaN instead, which the
  // losses / parameters show on the next step (functional.check_prelu_slopes() names the offending module).
  const bool lin = act == DSR_ACT_LEAKY || act == DSR_ACT_PRELU;
  const float poison = (lin && !(slope > 0.f)) ? __uint_as_float(0x7fc00000u) : 0.f;
  float sb[8], sp[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sb[k] = sp[k] = 0.f;
  const size_t P = (size_t)N * H * W;
  size_t p0 = (size_t)blockIdx.x * rows_per_block;
  size_t p1 = p0 + rows_per_block;
  if (p1 > P) p1 = P;
  auto one = [&](const float (&d)[8], const float (&o)[8], size_t p) {
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      g[k] = d[k] * act_grad_from_out(act, o[k], slope) + poison;
      sb[k] += g[k];
      if (act == DSR_ACT_PRELU && o[k] < 0.f) sp[k] += d[k] * (o[k] / slope);
      sp[k] += poison;
    }
    st16<NT>(dy + p * CyP + ch * 8, pack8<DT>(g));
  };
  if (rr < rpi && !pixshuf) {
    size_t p = p0 + rr;
    for (; p + rpi < p1; p += 2 * rpi) {        // two pixel rows in flight
      const U4 d0 = ld16<NT>(dout + p * CoP + ch * 8), o0 = ld16<NT>(out + p * CoP + ch * 8);
      const U4 d1 = ld16<NT>(dout + (p + rpi) * CoP + ch * 8), o1 = ld16<NT>(out + (p + rpi) * CoP + ch * 8);
      float d[8], o[8];
      unpack8<DT>(d0, d);
      unpack8<DT>(o0, o);
      one(d, o, p);
      unpack8<DT>(d1, d);
      unpack8<DT>(o1, o);
      one(d, o, p + rpi);
    }
    if (p < p1) {
      float d[8], o[8];
      unpack8<DT>(ld16<NT>(dout + p * CoP + ch * 8), d);
      unpack8<DT>(ld16<NT>(out + p * CoP + ch * 8), o);
      one(d, o, p);
    }
  } else if (rr < rpi) {
    // PixelShuffle(2) un-shuffle: conv channels 8ch..8ch+7 = shuffle channels c0, c0+1 (c0 = 2ch) at the 4 sub-pixels: one
    // 4-byte load per sub-pixel and tensor instead of eight 2-byte gathers (a conv pixel's 32 lanes read 128 contiguous bytes
    // per sub-pixel).  Two conv pixels in flight per thread (16 loads), 32-bit index arithmetic (P < 2^31 here).
    const int c0 = ch * 2;
    const bool cok = c0 < CoP;
    auto gather = [&](int p, unsigned (&dv)[4], unsigned (&ov)[4]) {
      const int n = p / (W * H);
      const int rem = p - n * (W * H);
      const int h = rem / W, w = rem - h * W;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const size_t q = ((size_t)(n * 2 * H + 2 * h + (sub >> 1)) * (2 * W) + 2 * w + (sub & 1)) * CoP + c0;
        dv[sub] = cok ? ld4<NT>(dout + q) : 0u;
        ov[sub] = cok ? ld4<NT>(out + q) : 0u;
      }
    };
    auto finish = [&](const unsigned (&dv)[4], const unsigned (&ov)[4], size_t p) {
      float d[8], o[8];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        d[sub] = h2f<DT>((unsigned short)(dv[sub] & 0xffff));
        d[4 + sub] = h2f<DT>((unsigned short)(dv[sub] >> 16));
        o[sub] = h2f<DT>((unsigned short)(ov[sub] & 0xffff));
        o[4 + sub] = h2f<DT>((unsigned short)(ov[sub] >> 16));
      }
      one(d, o, p);
    };
    int p = (int)p0 + rr;
    const int pe = (int)p1;
    for (; p + rpi < pe; p += 2 * rpi) {
      unsigned dv0[4], ov0[4], dv1[4], ov1[4];
      gather(p, dv0, ov0);
      gather(p + rpi, dv1, ov1);
      finish(dv0, ov0, (size_t)p);
      finish(dv1, ov1, (size_t)(p + rpi));
    }
    if (p < pe) {
      unsigned dv0[4], ov0[4];
      gather(p, dv0, ov0);
      finish(dv0, ov0, (size_t)p);
    }
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[tid * 16 + k] = sb[k];
    red[tid * 16 + 8 + k] = sp[k];
  }
  __syncthreads();
  for (int c = tid; c < 2 * CyP; c += 256) {
    int which = c / CyP, cc = c % CyP;
    int chn = cc / 8, k = cc % 8;
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[(r * cpr + chn) * 16 + which * 8 + k];
    partial[((size_t)blockIdx.x * 2 + which) * CyP + cc] = s;
  }
}

// final layer (fp32 NCHW output through tanh/sigmoid): dy NHWC16 [N][H][W][Cp] from NCHW fp32 dout/out
template <int DT>
__global__ void act_bwd_nchw_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                    unsigned short* __restrict__ dy, int N, int C, int H, int W, int Cp, int act) {
  size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t P = (size_t)N * H * W;
  if (pix >= P) return;
  int hw = (int)(pix % ((size_t)H * W));
  int n = (int)(pix / ((size_t)H * W));
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int c = c0 + k;
      if (c < C) {
        size_t q = ((size_t)n * C + c) * H * W + hw;
        g[k] = dout[q] * act_grad_from_out(act, out[q], 0.f);
      } else {
        g[k] = 0.f;
      }
    }
    *reinterpret_cast<U4*>(dy + pix * Cp + c0) = pack8<DT>(g);
  }
}

// column sums of an NHWC tensor -> partial rows [blocks][Cp]  (bias gradients)
template <int DT>
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ x, size_t P, int Cp,
                                                     int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int cpr = Cp / 8;
  const int tid = threadIdx.x;
  const int rpi = 256 / cpr;
  const int ch = tid % cpr, rr = tid / cpr;
  float s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = 0.f;
  size_t p0 = (size_t)blockIdx.x * rows_per_block;
  size_t p1 = p0 + rows_per_block;
  if (p1 > P) p1 = P;
  if (rr < rpi) {
    for (size_t p = p0 + rr; p < p1; p += rpi) {
      float f[8];
      unpack8<DT>(*reinterpret_cast<const U4*>(x + p * Cp + ch * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s1[k] += f[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[tid * 8 + k] = s1[k];
  __syncthreads();
  for (int c = tid; c < Cp; c += 256) {
    int chn = c / 8, k = c % 8;
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[(r * cpr + chn) * 8 + k];
    partial[(size_t)blockIdx.x * Cp + c] = s;
  }
}

// out = a + b (16-bit NHWC), used for gradient joins of residual / skip branches
template <int DT>
__global__ void add_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                           unsigned short* __restrict__ out, size_t nvec) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nvec) return;
  float x[8], y[8];
  unpack8<DT>(*reinterpret_cast<const U4*>(a + idx * 8), x);
  unpack8<DT>(*reinterpret_cast<const U4*>(b + idx * 8), y);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] += y[k];
  *reinterpret_cast<U4*>(out + idx * 8) = pack8<DT>(x);
}

// out[i] = (a * x[i] + b * y[i]) * g[0]     fp32; y and g optional (y == nullptr -> 0, g == nullptr -> 1).
// The scalar glue of the step recipes (loss sums of utils/GAN.py:105,122, loss-gradient scaling by autograd's
// incoming scalar, the static loss scale of the fp16 DIP path) stays on the HIP path with this one kernel.
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a,
                                                    float b, const float* __restrict__ g, float* __restrict__ out,
                                                    size_t n) {
  const float gs = g ? g[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (a * x[i] + (y ? b * y[i] : 0.f)) * gs;
}

// ------------------------------------------------------------------ losses (fp32 NCHW, mean reduction)
// mode 0: L1, 1: MSE.  Writes grad = d(mean loss)/d(pred) * gscale and per-block partial loss sums.
__global__ __launch_bounds__(256) void diff_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        float* __restrict__ grad, size_t n, int mode, float inv_n,
                                                        float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float d = pred[i] - tgt[i];
    if (mode == 0) {
      s += fabsf(d);
      if (grad) grad[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_n;
    } else {
      s += d * d;
      if (grad) grad[i] = 2.f * d * inv_n;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// BCE against a constant target (0 or 1) on probabilities p[n]; log clamped at -100 (nn.BCELoss).
// loss = mean; grad[i] = dloss/dp[i].  Single block (n = batch size).
__global__ void bce_const_kernel(const float* __restrict__ p, int n, float target, float* __restrict__ loss,
                                 float* __restrict__ grad, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = p[i];
    float l1 = fmaxf(logf(v), -100.f), l0 = fmaxf(logf(1.f - v), -100.f);
    s += -(target * l1 + (1.f - target) * l0);
    if (grad) {
      // torch's binary_cross_entropy backward: (x - t) / max((1 - x) * x, 1e-12), mean reduction
      grad[i] = (v - target) / fmaxf((1.f - v) * v, 1e-12f) / (float)n;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    t /= (float)n;
    loss[0] = accumulate ? loss[0] + t : t;
  }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults, no weight decay)
// step counter lives on the device so the launch is graph-capturable.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                   float b1, float b2, float eps, const int* __restrict__ step,
                                                   float grad_scale, unsigned short* __restrict__ shadow16) {
  // A pure stream (28 B per parameter, nothing re-read): two 16-byte vectors per thread and tensor in flight and
  // nontemporal accesses measured 6.3 TB/s against 5.8 for the one-vector cached form (537 M parameters).
  typedef __attribute__((ext_vector_type(4))) float F4;
  typedef __attribute__((ext_vector_type(2))) unsigned U2;
  const AdamCoef co = adam_coef(step, lr, b1, b2, eps, grad_scale);
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * 512;
  for (size_t i0 = (size_t)blockIdx.x * 512 + threadIdx.x; i0 < n4; i0 += stride) {
    F4 pi[2], gi[2], mi[2], vi[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = i0 + u * 256;
      if (i < n4) {
        pi[u] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(p) + i);
        gi[u] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(g) + i);
        mi[u] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(m) + i);
        vi[u] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(v) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t i = i0 + u * 256;
      if (i >= n4) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = pi[u][k], mk = mi[u][k], vk = vi[u][k];
        adam_update(pk, gi[u][k], mk, vk, co);
        pi[u][k] = pk;
        mi[u][k] = mk;
        vi[u][k] = vk;
      }
      __builtin_nontemporal_store(pi[u], reinterpret_cast<F4*>(p) + i);
      __builtin_nontemporal_store(mi[u], reinterpret_cast<F4*>(m) + i);
      __builtin_nontemporal_store(vi[u], reinterpret_cast<F4*>(v) + i);
      if (shadow16) {   // bf16 image of the updated parameter (dense1's MFMA operand) without a separate cast pass
        U2 h;
        h.x = (unsigned)f2h<DSR_DTYPE_BF16>(pi[u].x) | ((unsigned)f2h<DSR_DTYPE_BF16>(pi[u].y) << 16);
        h.y = (unsigned)f2h<DSR_DTYPE_BF16>(pi[u].z) | ((unsigned)f2h<DSR_DTYPE_BF16>(pi[u].w) << 16);
        __builtin_nontemporal_store(h, reinterpret_cast<U2*>(shadow16) + i);
      }
    }
  }
  // tail (n not a multiple of 4): the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float pk = p[i], mk = m[i], vk = v[i];
    adam_update(pk, g[i], mk, vk, co);
    m[i] = mk;
    v[i] = vk;
    p[i] = pk;
    if (shadow16) shadow16[i] = f2h<DSR_DTYPE_BF16>(pk);
  }
}
__global__ void incr_kernel(int* step) { *step += 1; }

// ================================================================== host launchers
#define DT_SWITCH(dtype, CALL)                 \
  if ((dtype) == DSR_DTYPE_BF16) {             \
    constexpr int DT = DSR_DTYPE_BF16;         \
    CALL;                                      \
  } else {                                     \
    constexpr int DT = DSR_DTYPE_F16;          \
    CALL;                                      \
  }

// the activation as a template constant for the pointwise BatchNorm / activation kernels (AC; -1 = run-time value)
#define ACT_SWITCH(act, CALL)                  \
  switch (act) {                               \
    case DSR_ACT_NONE: {                       \
      constexpr int AC = DSR_ACT_NONE;         \
      CALL;                                    \
    } break;                                   \
    case DSR_ACT_LEAKY: {                      \
      constexpr int AC = DSR_ACT_LEAKY;        \
      CALL;                                    \
    } break;                                   \
    case DSR_ACT_PRELU: {                      \
      constexpr int AC = DSR_ACT_PRELU;        \
      CALL;                                    \
    } break;                                   \
    case DSR_ACT_RELU: {                       \
      constexpr int AC = DSR_ACT_RELU;         \
      CALL;                                    \
    } break;                                   \
    default: {                                 \
      constexpr int AC = -1;                   \
      CALL;                                    \
    }                                          \
  }

static inline unsigned nblk(size_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int dsr_pw_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, hipStream_t st) {
  DSR_REQUIRE(src && dst && DSR_DTYPE_OK(dtype) && N > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && Cp >= C, "nchw_to_nhwc: null pointer, empty shape or bad channel padding");
  size_t total = (size_t)N * H * W * (Cp / 8);
  DT_SWITCH(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<DT>), dim3(nblk(total, 256)), dim3(256), 0, st, src,
                                      (unsigned short*)dst, N, C, H, W, Cp));
  return dsr_launch_status("dsr_pw_nchw_to_nhwc");
}
extern "C" int dsr_pw_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, hipStream_t st) {
  DSR_REQUIRE(src && dst && DSR_DTYPE_OK(dtype) && N > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && Cp >= C, "nhwc_to_nchw: null pointer, empty shape or bad channel padding");
  size_t total = (size_t)N * C * H * W;
  DT_SWITCH(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<DT>), dim3(nblk(total, 256)), dim3(256), 0, st,
                                      (const unsigned short*)src, dst, N, C, H, W, Cp));
  return dsr_launch_status("dsr_pw_nhwc_to_nchw");
}
extern "C" int dsr_pw_pack_weight(int dtype, const float* w, void* wf, void* wd, int Cout, int Cin, int T, int NBo, int CinP,
                        int NBi, int CoutP, hipStream_t st) {
  DSR_REQUIRE(w && wf && DSR_DTYPE_OK(dtype) && Cout > 0 && Cin > 0 && T > 0 && NBo >= Cout && CinP >= Cin, "pack_weight: null pointer or bad shape");
  size_t nf = (size_t)T * NBo * CinP, nd = wd ? (size_t)T * NBi * CoutP : 0;
  size_t total = nf > nd ? nf : nd;
  DT_SWITCH(dtype, hipLaunchKernelGGL((pack_weight_kernel<DT>), dim3(nblk(total, 256)), dim3(256), 0, st, w,
                                      (unsigned short*)wf, (unsigned short*)wd, Cout, Cin, T, NBo, CinP, NBi, CoutP));
  return dsr_launch_status("dsr_pw_pack_weight");
}
extern "C" int dsr_pw_scratch_rows(void) { return DSR_COMPACT_ROWS; }
extern "C" int dsr_pw_sum_rows(const float* partial, int rows, int row_stride, int col_offset, int C, float scale, float* out,
                     int accumulate, int compact, hipStream_t st) {
  DSR_REQUIRE(partial && out && rows >= 0 && row_stride > 0 && col_offset >= 0 && C > 0, "sum_rows: null pointer or bad shape");
  if (rows > 32 && rows <= DSR_FINALIZE_PAR_ROWS && row_stride > 1) {      // one launch: parallel over 4 row-lanes
    hipLaunchKernelGGL(sum_rows_par_kernel, dim3(nblk(C, 64)), dim3(256), 0, st, partial, rows, row_stride, col_offset, C, scale,
                       out, accumulate);
    return dsr_launch_status("dsr_pw_sum_rows");
  }
  if (compact && row_stride > 1) partial = compact_rows(partial, rows, row_stride, &rows, st);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(nblk(C, 128)), dim3(128), 0, st, partial, rows, row_stride, col_offset, C,
                     scale, out, accumulate);
  return dsr_launch_status("dsr_pw_sum_rows");
}
extern "C" int dsr_pw_bn_finalize(const float* partial, int tiles, int stride, int C, int Cp, float count, const float* gamma,
                        const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, int updates,
                        float* scale, float* shift, float* mean, float* rstd, hipStream_t st) {
  DSR_REQUIRE(partial && gamma && beta && scale && shift && mean && rstd && tiles > 0 && C > 0 && Cp >= C && stride >= Cp && count > 0.f && updates >= 0, "bn_finalize: null pointer or bad shape");
  if (tiles > 32 && tiles <= DSR_FINALIZE_PAR_ROWS) {
    // 16 channels per block (16 row-lanes) once there are enough rows to share out; 64 channels x 4 lanes for short tables
    if (tiles > 128 && Cp % 16 == 0)
      hipLaunchKernelGGL(bn_finalize_par_kernel<16>, dim3(nblk(Cp, 16)), dim3(256), 0, st, partial, tiles, stride, C, count, gamma,
                         beta, rm, rv, nbt, momentum, eps, updates, scale, shift, mean, rstd, Cp);
    else
      hipLaunchKernelGGL(bn_finalize_par_kernel<64>, dim3(nblk(Cp, 64)), dim3(256), 0, st, partial, tiles, stride, C, count, gamma,
                         beta, rm, rv, nbt, momentum, eps, updates, scale, shift, mean, rstd, Cp);
    return dsr_launch_status("dsr_pw_bn_finalize");
  }
  partial = compact_rows(partial, tiles, 2 * stride, &tiles, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk(Cp, 64)), dim3(64), 0, st, partial, tiles, stride, C, count, gamma,
                     beta, rm, rv, nbt, momentum, eps, updates, scale, shift, mean, rstd, Cp);
  return dsr_launch_status("dsr_pw_bn_finalize");
}
extern "C" int dsr_pw_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                           int Cp, float* scale, float* shift, float* mean, float* rstd, hipStream_t st) {
  DSR_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0 && Cp >= C, "bn_eval_affine: null pointer or bad shape");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(nblk(Cp, 128)), dim3(128), 0, st, gamma, beta, rm, rv, eps, C, Cp,
                     scale, shift, mean, rstd);
  return dsr_launch_status("dsr_pw_bn_eval_affine");
}
extern "C" int dsr_pw_reduce_blocks(size_t P, int* rows_per_block) {
  if (!rows_per_block) return dsr_fail(DSR_E_ARG, "reduce_blocks: null rows_per_block");
  // 1024 blocks = the 4 blocks per CU that are resident at once: one round, no tail (measured on the BatchNorm backward
  // reduction, TB/s of its two operand tensors at 32 x 128 x 128 x 64: 512 blocks 2.8, 1024 4.4, 1536 3.7, 2048 4.2, 4096 3.8);
  // at least 64 rows each
  static const size_t target = [] { const char* e = getenv("DSR_PW_REDUCE_BLOCKS"); return (size_t)(e ? atoi(e) : 1024); }();
  size_t rpb = (P + target - 1) / target;
  if (rpb < 64) rpb = 64;
  *rows_per_block = (int)rpb;
  return (int)((P + rpb - 1) / rpb);
}
extern "C" int dsr_pw_channel_stats(int dtype, const void* x, size_t P, int Cp, int blocks, int rpb, float* partial,
                          hipStream_t st) {
  DSR_REQUIRE(x && partial && DSR_DTYPE_OK(dtype) && P > 0 && DSR_CP_OK(Cp) && blocks > 0 && rpb > 0, "channel_stats: null pointer or bad shape");
  DT_SWITCH(dtype, hipLaunchKernelGGL((channel_stats_kernel<DT>), dim3(blocks), dim3(256), 0, st,
                                      (const unsigned short*)x, P, Cp, rpb, partial));
  return dsr_launch_status("dsr_pw_channel_stats");
}
// grid of the row-walking pointwise kernels; env DSR_PW_BLOCKS / DSR_PW_NT override the policy (tuning).
// Measured (tools/microbench_pw.py, profiles/r02_pointwise_sweep.txt): tensors beyond the Infinity Cache stream best
// nontemporally from many short blocks (16384), cache-sized ones with plain accesses from ~4096 blocks.
static unsigned pw_grid(size_t P, int rpi, bool nt) {
  static const int forced = [] { const char* e = getenv("DSR_PW_BLOCKS"); return e ? atoi(e) : 0; }();
  if (forced > 0) return (unsigned)forced;
  const size_t cap = nt ? 16384 : 4096;
  size_t want = (P + (size_t)rpi - 1) / (size_t)rpi;      // at least one pixel row per thread
  return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}
static bool pw_nontemporal(size_t P, int Cp) {
  static const int forced = [] { const char* e = getenv("DSR_PW_NT"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  return P * (size_t)Cp * 2 >= DSR_PW_NT_BYTES;
}
extern "C" int dsr_pw_bn_act_fwd(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                       size_t P, int Cp, int act, float slope, const float* prelu, hipStream_t st) {
  DSR_REQUIRE(y && out && DSR_DTYPE_OK(dtype) && P > 0 && DSR_CP_OK(Cp) && (!scale == !shift), "bn_act_fwd: null pointer or bad shape");
  DSR_REQUIRE(act != DSR_ACT_PRELU || prelu, "bn_act_fwd: PReLU needs its weight pointer");
  const int rpi = 256 / (Cp / 8);
  const bool nt = pw_nontemporal(P, Cp);
  const unsigned blocks = pw_grid(P, rpi, nt);
  if (nt) {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((bn_act_fwd_kernel<DT, true, AC>), dim3(blocks), dim3(256), 0, st,
                                        (const unsigned short*)y, scale, shift, (const unsigned short*)residual,
                                        (unsigned short*)out, P, Cp, act, slope, prelu)));
  } else {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((bn_act_fwd_kernel<DT, false, AC>), dim3(blocks), dim3(256), 0, st,
                                        (const unsigned short*)y, scale, shift, (const unsigned short*)residual,
                                        (unsigned short*)out, P, Cp, act, slope, prelu)));
  }
  return dsr_launch_status("dsr_pw_bn_act_fwd");
}
extern "C" int dsr_pw_bn_act_bwd_reduce(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                              const float* mean, const float* rstd, size_t P, int Cp, int blocks, int rpb, int act,
                              float slope, const float* prelu, float* partial, hipStream_t st) {
  DSR_REQUIRE(dout && y && scale && shift && mean && rstd && partial && DSR_DTYPE_OK(dtype) && P > 0 && DSR_CP_OK(Cp) && blocks > 0 && rpb > 0, "bn_act_bwd_reduce: null pointer or bad shape");
  DSR_REQUIRE(act != DSR_ACT_PRELU || prelu, "bn_act_bwd_reduce: PReLU needs its weight pointer");
  static const int unr = [] { const char* e = getenv("DSR_PW_REDUCE_UNROLL"); return e ? atoi(e) : 2; }();
#define LAUNCH_RED2(NTV, U, WP)                                                                                             \
  ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<DT, NTV, U, WP, AC>), dim3(blocks), dim3(256), 0, st,           \
                                      (const unsigned short*)dout, (const unsigned short*)y, scale, shift, mean, rstd, P, \
                                      Cp, rpb, act, slope, prelu, partial)))
#define LAUNCH_RED(NTV, U)                              \
  do {                                                  \
    if (act == DSR_ACT_PRELU) { LAUNCH_RED2(NTV, U, true); } else { LAUNCH_RED2(NTV, U, false); } \
  } while (0)
  if (pw_nontemporal(P, Cp)) {
    if (unr == 2) { LAUNCH_RED(true, 2); } else { LAUNCH_RED(true, 1); }
  } else {
    if (unr == 2) { LAUNCH_RED(false, 2); } else { LAUNCH_RED(false, 1); }
  }
#undef LAUNCH_RED
#undef LAUNCH_RED2
  return dsr_launch_status("dsr_pw_bn_act_bwd_reduce");
}
extern "C" int dsr_pw_bn_bwd_finalize(const float* partial, int blocks, int C, int Cp, float count, const float* mean,
                            const float* rstd, float* dgamma, float* dbeta, float* dprelu, float* c1, float* c2,
                            hipStream_t st) {
  DSR_REQUIRE(partial && mean && rstd && c1 && c2 && blocks > 0 && C > 0 && Cp >= C && count > 0.f, "bn_bwd_finalize: null pointer or bad shape");
  if (blocks > 32 && blocks <= DSR_FINALIZE_PAR_ROWS_BWD && (!dprelu || Cp <= 64)) {
    hipLaunchKernelGGL(bn_bwd_finalize_par_kernel, dim3(nblk(Cp, 64)), dim3(256), 0, st, partial, blocks, C, Cp, count, mean,
                       rstd, dgamma, dbeta, dprelu, c1, c2);
    return dsr_launch_status("dsr_pw_bn_bwd_finalize");
  }
  partial = compact_rows(partial, blocks, 3 * Cp, &blocks, st);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, partial, blocks, C, Cp, count, mean, rstd, dgamma,
                     dbeta, dprelu, c1, c2);
  return dsr_launch_status("dsr_pw_bn_bwd_finalize");
}
extern "C" int dsr_pw_bn_act_bwd_apply(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                             const float* mean, const float* rstd, const float* c1, const float* c2, void* dy, size_t P,
                             int Cp, int act, float slope, const float* prelu, int train, hipStream_t st) {
  DSR_REQUIRE(dout && y && scale && shift && mean && rstd && dy && (!train || (c1 && c2)) && DSR_DTYPE_OK(dtype) && P > 0 && DSR_CP_OK(Cp), "bn_act_bwd_apply: null pointer or bad shape");
  DSR_REQUIRE(act != DSR_ACT_PRELU || prelu, "bn_act_bwd_apply: PReLU needs its weight pointer");
  const int rpi = 256 / (Cp / 8);
  const bool nt = pw_nontemporal(P, Cp);
  const unsigned blocks = pw_grid(P, rpi, nt);
  if (nt) {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DT, true, AC>), dim3(blocks), dim3(256), 0, st,
                                        (const unsigned short*)dout, (const unsigned short*)y, scale, shift, mean, rstd, c1,
                                        c2, (unsigned short*)dy, P, Cp, act, slope, prelu, train)));
  } else {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DT, false, AC>), dim3(blocks), dim3(256), 0, st,
                                        (const unsigned short*)dout, (const unsigned short*)y, scale, shift, mean, rstd, c1,
                                        c2, (unsigned short*)dy, P, Cp, act, slope, prelu, train)));
  }
  return dsr_launch_status("dsr_pw_bn_act_bwd_apply");
}
extern "C" int dsr_pw_act_bwd(int dtype, const void* dout, const void* out, void* dy, int N, int H, int W, int CyP, int CoP,
                    int pixshuf, int act, float slope, const float* prelu, int blocks, int rpb, float* partial,
                    hipStream_t st) {
  DSR_REQUIRE(dout && out && dy && DSR_DTYPE_OK(dtype) && N > 0 && H > 0 && W > 0 && DSR_CP_OK(CyP) && CoP > 0 && CoP % 2 == 0 && blocks > 0 && rpb > 0, "act_bwd: null pointer or bad shape");
  DSR_REQUIRE(act != DSR_ACT_PRELU || prelu, "act_bwd: PReLU needs its weight pointer");
  // the derivative is taken from the stored OUTPUT, which identifies the branch only for a positive slope
  DSR_REQUIRE(act != DSR_ACT_LEAKY || slope > 0.f, "act_bwd: LeakyReLU slope %g must be > 0 (the activation gradient is derived from the stored output)", (double)slope);
  if (pw_nontemporal((size_t)N * H * W, CyP)) {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((act_bwd_kernel<DT, true, AC>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dout,
                                        (const unsigned short*)out, (unsigned short*)dy, N, H, W, CyP, CoP, pixshuf, act,
                                        slope, prelu, rpb, partial)));
  } else {
    ACT_SWITCH(act, DT_SWITCH(dtype, hipLaunchKernelGGL((act_bwd_kernel<DT, false, AC>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dout,
                                        (const unsigned short*)out, (unsigned short*)dy, N, H, W, CyP, CoP, pixshuf, act,
                                        slope, prelu, rpb, partial)));
  }
  return dsr_launch_status("dsr_pw_act_bwd");
}
extern "C" int dsr_pw_act_bwd_nchw(int dtype, const float* dout, const float* out, void* dy, int N, int C, int H, int W, int Cp,
                         int act, hipStream_t st) {
  DSR_REQUIRE(dout && out && dy && DSR_DTYPE_OK(dtype) && N > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && Cp >= C, "act_bwd_nchw: null pointer or bad shape");
  size_t P = (size_t)N * H * W;
  DT_SWITCH(dtype, hipLaunchKernelGGL((act_bwd_nchw_kernel<DT>), dim3(nblk(P, 256)), dim3(256), 0, st, dout, out,
                                      (unsigned short*)dy, N, C, H, W, Cp, act));
  return dsr_launch_status("dsr_pw_act_bwd_nchw");
}
extern "C" int dsr_pw_colsum(int dtype, const void* x, size_t P, int Cp, int blocks, int rpb, float* partial, hipStream_t st) {
  DSR_REQUIRE(x && partial && DSR_DTYPE_OK(dtype) && P > 0 && DSR_CP_OK(Cp) && blocks > 0 && rpb > 0, "colsum: null pointer or bad shape");
  DT_SWITCH(dtype, hipLaunchKernelGGL((colsum_kernel<DT>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)x, P,
                                      Cp, rpb, partial));
  return dsr_launch_status("dsr_pw_colsum");
}
extern "C" int dsr_pw_add(int dtype, const void* a, const void* b, void* out, size_t nvec, hipStream_t st) {
  DSR_REQUIRE(a && b && out && DSR_DTYPE_OK(dtype) && nvec > 0, "add: null pointer or empty");
  DT_SWITCH(dtype, hipLaunchKernelGGL((add_kernel<DT>), dim3(nblk(nvec, 256)), dim3(256), 0, st,
                                      (const unsigned short*)a, (const unsigned short*)b, (unsigned short*)out, nvec));
  return dsr_launch_status("dsr_pw_add");
}
extern "C" int dsr_pw_axpby_f32(const float* x, const float* y, float a, float b, const float* g, float* out, size_t n,
                                hipStream_t st) {
  DSR_REQUIRE(x && out && n > 0, "axpby_f32: null pointer or empty");
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, st, x, y, a, b, g, out, n);
  return dsr_launch_status("dsr_pw_axpby_f32");
}
extern "C" int dsr_pw_diff_loss(const float* pred, const float* tgt, float* grad, size_t n, int mode, float* partial, int blocks,
                      hipStream_t st) {
  DSR_REQUIRE(pred && tgt && partial && n > 0 && blocks > 0 && (mode == 0 || mode == 1), "diff_loss: null pointer, empty tensor or bad mode");
  hipLaunchKernelGGL(diff_loss_kernel, dim3(blocks), dim3(256), 0, st, pred, tgt, grad, n, mode, 1.f / (float)n,
                     partial);
  return dsr_launch_status("dsr_pw_diff_loss");
}
extern "C" int dsr_pw_bce_const(const float* p, int n, float target, float* loss, float* grad, int accumulate, hipStream_t st) {
  DSR_REQUIRE(p && loss && n > 0, "bce_const: null pointer or empty");
  hipLaunchKernelGGL(bce_const_kernel, dim3(1), dim3(256), 0, st, p, n, target, loss, grad, accumulate);
  return dsr_launch_status("dsr_pw_bce_const");
}
extern "C" int dsr_pw_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                 const int* step, float grad_scale, void* shadow_bf16, hipStream_t st) {
  DSR_REQUIRE(p && g && m && v && step && n > 0, "adam: null pointer or empty tensor");
  size_t want = (n / 4 + 511) / 512;
  unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, step, grad_scale,
                     (unsigned short*)shadow_bf16);
  return dsr_launch_status("dsr_pw_adam");
}
// ---- multi-tensor Adam: the generator and discriminator hold ~100 small tensors each; one launch per 64 of them.
#define DSR_ADAM_GROUP 64
#define DSR_ADAM_CHUNK 4096   // elements per block
struct AdamGroup {
  float* p[DSR_ADAM_GROUP];
  const float* g[DSR_ADAM_GROUP];
  float* m[DSR_ADAM_GROUP];
  float* v[DSR_ADAM_GROUP];
  unsigned n[DSR_ADAM_GROUP];
  unsigned first_block[DSR_ADAM_GROUP + 1];
  int count;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamGroup a, float lr, float b1, float b2, float eps,
                                                         const int* __restrict__ step, float grad_scale) {
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) ++t;      // wave-uniform scan of <= 64 entries
  const unsigned base = (blockIdx.x - a.first_block[t]) * DSR_ADAM_CHUNK;
  const unsigned n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const AdamCoef co = adam_coef(step, lr, b1, b2, eps, grad_scale);
#pragma unroll 4
  for (unsigned i = base + threadIdx.x; i < base + DSR_ADAM_CHUNK && i < n; i += 256) {
    float pk = p[i], mk = m[i], vk = v[i];
    adam_update(pk, g[i], mk, vk, co);
    m[i] = mk;
    v[i] = vk;
    p[i] = pk;
  }
}
extern "C" int dsr_pw_adam_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                                 const size_t* n, float lr, float b1, float b2, float eps, const int* step,
                                 float grad_scale, hipStream_t st) {
  if (count < 0 || (count && (!p || !g || !m || !v || !n))) return dsr_fail(DSR_E_ARG, "adam_multi: null table");
  for (int i0 = 0; i0 < count; i0 += DSR_ADAM_GROUP) {
    AdamGroup a;
    a.count = count - i0 < DSR_ADAM_GROUP ? count - i0 : DSR_ADAM_GROUP;
    unsigned blocks = 0;
    for (int j = 0; j < a.count; ++j) {
      if (n[i0 + j] > 0xFFFFFFFFull - DSR_ADAM_CHUNK) return dsr_fail(DSR_E_UNSUPPORTED, "adam_multi: tensor too large");
      a.p[j] = p[i0 + j];
      a.g[j] = g[i0 + j];
      a.m[j] = m[i0 + j];
      a.v[j] = v[i0 + j];
      a.n[j] = (unsigned)n[i0 + j];
      a.first_block[j] = blocks;
      blocks += (unsigned)((n[i0 + j] + DSR_ADAM_CHUNK - 1) / DSR_ADAM_CHUNK);
    }
    a.first_block[a.count] = blocks;
    if (blocks) hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, st, a, lr, b1, b2, eps, step, grad_scale);
  }
  return dsr_launch_status("dsr_pw_adam_multi");
}
extern "C" int dsr_pw_incr(int* step, hipStream_t st) {
  DSR_REQUIRE(step, "incr: null pointer");
  hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(1), 0, st, step);
  return dsr_launch_status("dsr_pw_incr");
}
