// Weight gradient of a convolution on gfx950:
//   dW[tap][co][ci] = sum over output pixels p of dY[p][co] * X[p shifted by tap][ci]
// i.e. a GEMM whose contraction index is the PIXEL -- the non-contiguous index of both NHWC
// operands.  Both operand tiles are staged [pixel][channel] (coalesced 128-B rows) in LDS and read
// back with ds_read_b64_tr_b16, the CDNA4 transposing LDS read, which hands each lane 4 pixels of
// one channel: exactly the MFMA 16x16x32 A/B fragment (two reads per fragment).  The k index
// inside a fragment is permuted (pixels 4g..4g+3 and 16+4g..16+4g+3 for lane group g) -- identically
// for A and B, so the contraction is unaffected -- which makes the transposed reads bank-conflict
// free on XOR-swizzled 128-B rows.
//
// Work split: grid = (co-tile x ci-tile) x tap x split.  A block owns a 64x64 (co x ci) tile of one
// tap; its 4 waves each take a different 32-pixel slice of every 128-pixel step (wave-level
// split-K, 16 MFMAs per wave per step) and are summed through LDS at the end.  Each split writes a
// partial slab (plain stores, deterministic); dsr_wgrad_reduce sums the slabs into the fp32
// [Cout][Cin][KH][KW] gradient the reference's optimiser expects.
#include "dsr_common.h"
#include "dsr_kernels.h"

__device__ __forceinline__ s16x4 lds_tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}

template <int DT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  // per wave: A tile [32 px][64 co] (4 KB) + B tile [32 px][64 ci] (4 KB); 4 waves -> 32 KB.
  // the same memory is reused for the cross-wave reduction (3 x 16 KB needed -> 48 KB).
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 64 * 64 * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15;

  const int tile = blockIdx.x;
  const int tco = tile / a.tiles_ci, tci = tile % a.tiles_ci;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int tap = blockIdx.y;
  const int kh = tap / a.KW, kw = tap % a.KW;
  const int split = blockIdx.z;

  unsigned char* sAw = smem + wave * 8192;
  unsigned char* sBw = sAw + 4096;

  const unsigned short* __restrict__ X = reinterpret_cast<const unsigned short*>(a.x);
  const unsigned short* __restrict__ DY = reinterpret_cast<const unsigned short*>(a.dy);

  // loader role inside the wave: 16-B chunk c (8 channels), pixels pp + 8*i
  const int c = lane & 7, pp = lane >> 3;
  const bool a_cok = (co0 + c * 8) < a.CoutP;
  const bool b_cok = (ci0 + c * 8) < a.CinP;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int k_begin = split * a.chunk;
  int k_end = k_begin + a.chunk;
  if (k_end > a.M) k_end = a.M;

  // transposed-read addresses (constant per lane): block row q = l16>>2, 4 channels at 4*(l16&3)
  // inside the 16-channel group of fragment index f -> channel f*16 + 4*(l16&3)
  const int q = l16 >> 2, cc = 4 * (l16 & 3);

  for (int k0 = k_begin; k0 < k_end; k0 += 128) {
    const int kw0 = k0 + wave * 32;
    U4 va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int p = pp + 8 * i;
      int m = kw0 + p;
      bool ok = m < k_end;
      int mm = ok ? m : 0;
      int n = fd_div(a.fd_ohw, mm);
      int rem = mm - n * (a.OH * a.OW);
      int oy = fd_div(a.fd_ow, rem);
      int ox = rem - oy * a.OW;
      va[i] = load16_or_zero(DY, (size_t)mm * a.CoutP + co0 + c * 8, ok && a_cok);
      bool inb = ok && b_cok;
      int iy = pad_index(oy * a.stride + kh - a.pad, a.IH, a.pad_mode, inb);
      int ix = pad_index(ox * a.stride + kw - a.pad, a.IW, a.pad_mode, inb);
      size_t off = ((size_t)(n * a.IH + iy) * a.IW + ix) * a.CinP + ci0 + c * 8;
      vb[i] = load16_or_zero(X, off, inb);
    }
    __syncthreads();   // previous step's transposed reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int p = pp + 8 * i;
      int o = p * 128 + ((c ^ (p & 7)) << 4);
      *reinterpret_cast<U4*>(sAw + o) = va[i];
      *reinterpret_cast<U4*>(sBw + o) = vb[i];
    }
    __syncthreads();
    U4 fa[4], fb[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int ch = f * 16 + cc;          // first of this lane's 4 channels
      const int chunk = ch >> 3, within = (ch & 7) * 2;
      const int p1 = 4 * g + q, p2 = 16 + 4 * g + q;
      const int o1 = p1 * 128 + ((chunk ^ (p1 & 7)) << 4) + within;
      const int o2 = p2 * 128 + ((chunk ^ (p2 & 7)) << 4) + within;
      s16x4 a1 = lds_tr_read(sAw + o1), a2 = lds_tr_read(sAw + o2);
      s16x4 b1 = lds_tr_read(sBw + o1), b2 = lds_tr_read(sBw + o2);
      fa[f] = __builtin_bit_cast(U4, __builtin_shufflevector(a1, a2, 0, 1, 2, 3, 4, 5, 6, 7));
      fb[f] = __builtin_bit_cast(U4, __builtin_shufflevector(b1, b2, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i][k] = mfma16<DT>(fa[i], fb[k], acc[i][k]);
  }

  // ---- cross-wave sum: waves 1..3 park their 64x64 fp32 tiles in LDS, wave 0 adds and stores
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  if (wave > 0) {
    float* dst = red + (wave - 1) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(i * 16 + 4 * g + r) * 64 + k * 16 + l16] = acc[i][k][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* P = a.partial + ((size_t)split * gridDim.y + tap) * (size_t)a.CoutP * a.CinP;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = i * 16 + 4 * g + r, col = k * 16 + l16;
          float v = acc[i][k][r] + red[row * 64 + col] + red[4096 + row * 64 + col] + red[8192 + row * 64 + col];
          int co = co0 + row, ci = ci0 + col;
          if (co < a.CoutP && ci < a.CinP) P[(size_t)co * a.CinP + ci] = v;
        }
  }
}

void dsr_launch_wgrad(const WgradArgs& a, int dtype, hipStream_t st) {
  dim3 grid(a.tiles_co * a.tiles_ci, a.KH * a.KW, a.splits), block(256);
  if (dtype == DSR_DTYPE_BF16)
    hipLaunchKernelGGL((conv_wgrad_kernel<DSR_DTYPE_BF16>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<DSR_DTYPE_F16>), grid, block, 0, st, a);
}

// dw[co][ci][kh][kw] (fp32, PyTorch layout) = sum over splits of partial[split][tap][co][ci]
// grid.y slices the splits so that many loads are in flight; with more than one slice the first pass writes
// slice sums to `scratch` ([slices][slab]) and a second launch folds those (deterministic order).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           float* __restrict__ scratch, int splits, int per_slice,
                                                           int ntaps, int Cout, int Cin, int CoutP, int CinP) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = ntaps * CoutP * CinP;
  if (idx >= total) return;
  const size_t slab = (size_t)total;
  const int z0 = blockIdx.y * per_slice;
  int z1 = z0 + per_slice;
  if (z1 > splits) z1 = splits;
  const float* p = partial + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = z0;
  for (; z + 3 < z1; z += 4) {   // 4 independent loads in flight per thread
    s0 += p[(size_t)z * slab];
    s1 += p[(size_t)(z + 1) * slab];
    s2 += p[(size_t)(z + 2) * slab];
    s3 += p[(size_t)(z + 3) * slab];
  }
  for (; z < z1; ++z) s0 += p[(size_t)z * slab];
  const float s = (s0 + s1) + (s2 + s3);
  if (scratch != nullptr) {
    scratch[(size_t)blockIdx.y * slab + idx] = s;
    return;
  }
  const int ci = idx % CinP;
  const int co = (idx / CinP) % CoutP;
  const int tap = idx / (CinP * CoutP);
  if (ci < Cin && co < Cout) dw[((size_t)co * Cin + ci) * ntaps + tap] = s;
}

// `partial` must have room for splits + DSR_WGRAD_SCRATCH_SLABS slabs when splits > 32.
void dsr_launch_wgrad_reduce(const float* partial, float* dw, int splits, int ntaps, int Cout, int Cin, int CoutP,
                             int CinP, hipStream_t st) {
  const int total = ntaps * CoutP * CinP;
  const int bx = (total + 255) / 256;
  if (splits <= 32) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, 1), dim3(256), 0, st, partial, dw, (float*)nullptr, splits, splits,
                       ntaps, Cout, Cin, CoutP, CinP);
    return;
  }
  const int slices = DSR_WGRAD_SCRATCH_SLABS;
  const int per = (splits + slices - 1) / slices;
  const int used = (splits + per - 1) / per;
  float* scratch = const_cast<float*>(partial) + (size_t)splits * total;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, used), dim3(256), 0, st, partial, dw, scratch, splits, per, ntaps,
                     Cout, Cin, CoutP, CinP);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, 1), dim3(256), 0, st, (const float*)scratch, dw, (float*)nullptr, used,
                     used, ntaps, Cout, Cin, CoutP, CinP);
}

// ---- batched form (3x3): one launch reduces every problem of a dsr_conv_wgrad_batched call.  256 outputs per block;
// the slabs of an entry are summed in slab order with four independent partial sums (fixed order => deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceBatchArgs r) {
  int ei = 0;
  while (ei + 1 < r.count && (int)blockIdx.x >= r.e[ei + 1].first_block) ++ei;
  ei = __builtin_amdgcn_readfirstlane(ei);
  const WgradReduceBatchArgs::Entry e = r.e[ei];
  const int total = 9 * e.CoutP * e.CinP;
  const int idx = ((int)blockIdx.x - e.first_block) * 256 + threadIdx.x;
  if (idx >= total) return;
  const float* p = e.partial + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = 0;
  for (; z + 3 < e.splits; z += 4) {
    s0 += p[(size_t)z * total];
    s1 += p[(size_t)(z + 1) * total];
    s2 += p[(size_t)(z + 2) * total];
    s3 += p[(size_t)(z + 3) * total];
  }
  for (; z < e.splits; ++z) s0 += p[(size_t)z * total];
  const float s = (s0 + s1) + (s2 + s3);
  const int ci = idx % e.CinP;
  const int co = (idx / e.CinP) % e.CoutP;
  const int tap = idx / (e.CinP * e.CoutP);
  if (ci < e.Cin && co < e.Cout) e.dw[((size_t)co * e.Cin + ci) * 9 + tap] = s;
}

void dsr_launch_wgrad_reduce_batch(const WgradReduceBatchArgs& r, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(r.total_blocks), dim3(256), 0, st, r);
}
