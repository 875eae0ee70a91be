// Weight gradient, tile-resident form, for the convolutions that dominate the step (3x3 stride 1|2, 1x1):
//   dW[tap][co][ci] = sum_p dY[p][co] * X[p*s + tap][ci]
// A block owns one (64 co x 64 ci) pair and walks a list of spatial tiles (R output rows x 32 columns of one
// image).  Per tile it stages the dY tile and the X HALO tile ((R-1)s+KH rows x 31s+KW columns) once in LDS and
// derives ALL KH*KW taps from them -- a tap is just an address offset into the halo image -- so every staged byte
// feeds 9x more MFMAs than in the one-tap-per-block kernel (conv_wgrad.hip) and the operands are no longer
// re-fetched from L2 once per tap.  As there, pixels are the contraction index and both operands are read with
// the transposing ds_read_b64_tr_b16; the k order inside a fragment is the same permutation for A and B.
// 4 waves = 2 (co halves) x 2 (ci halves); each keeps taps x 2 x 2 accumulator tiles (144 VGPRs for 3x3).
#include "dsr_common.h"
#include "dsr_kernels.h"

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ U4 tr_frag(const unsigned char* base, int p1, int p2, int chunk, int within) {
  s16x4 lo = tr_read(base + p1 * 128 + ((chunk ^ (p1 & 7)) << 4) + within);
  s16x4 hi = tr_read(base + p2 * 128 + ((chunk ^ (p2 & 7)) << 4) + within);
  return __builtin_bit_cast(U4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int DT, int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile_kernel(const WgradTileArgs a) {
  constexpr int R = (S == 1) ? 4 : 2;
  constexpr int HR = (R - 1) * S + KH, HC = 31 * S + KW;
  constexpr int NT = KH * KW;
  __shared__ __attribute__((aligned(16))) unsigned char sX[HR * HC * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sY[R * 32 * 128];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, cc = 4 * (l16 & 3);
  const int pair = blockIdx.x;
  const int co0 = (pair / a.tiles_ci) * 64, ci0 = (pair % a.tiles_ci) * 64;
  const unsigned short* __restrict__ X = reinterpret_cast<const unsigned short*>(a.x);
  const unsigned short* __restrict__ DY = reinterpret_cast<const unsigned short*>(a.dy);

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int c = tid & 7, pbase = tid >> 3;
  const bool yc_ok = (co0 + c * 8) < a.CoutP, xc_ok = (ci0 + c * 8) < a.CinP;
  int t_end = (blockIdx.y + 1) * a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  const int per_img = a.tiles_y * a.tiles_x;

  for (int t = blockIdx.y * a.tiles_per_block; t < t_end; ++t) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * R, ox0 = (rem % a.tiles_x) * 32;
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < R * 32 * 8 / 256; ++i) {
      const int p = pbase + 32 * i;
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
      const bool ok = yc_ok && oy < a.OH && ox < a.OW;
      U4 v = load16_or_zero(DY, ((size_t)(n * a.OH + oy) * a.OW + ox) * a.CoutP + co0 + c * 8, ok);
      *reinterpret_cast<U4*>(sY + p * 128 + ((c ^ (p & 7)) << 4)) = v;
    }
    {
      constexpr int NV = (HR * HC + 31) / 32;     // 7 (3x3 s1), 11 (3x3 s2), 4 (1x1): all loads first, then the writes
      U4 v[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int q = pbase + 32 * u;
        const int hr = q / HC, hc = q - hr * HC;
        bool ok = xc_ok && q < HR * HC;
        const int iy = pad_index(oy0 * S + hr - a.pad, a.IH, a.pad_mode, ok);
        const int ix = pad_index(ox0 * S + hc - a.pad, a.IW, a.pad_mode, ok);
        v[u] = load16_or_zero(X, ((size_t)(n * a.IH + iy) * a.IW + ix) * a.CinP + ci0 + c * 8, ok);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int q = pbase + 32 * u;
        if (q < HR * HC) *reinterpret_cast<U4*>(sX + q * 128 + ((c ^ (q & 7)) << 4)) = v[u];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < R; ++r) {   // not unrolled: 144 accumulator registers leave no room for hoisted addresses
      U4 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = wr * 32 + i * 16 + cc;
        fa[i] = tr_frag(sY, r * 32 + 4 * g + q4, r * 32 + 16 + 4 * g + q4, ch >> 3, (ch & 7) * 2);
      }
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int qb = (r * S + kh) * HC + kw;
          const int q1 = qb + (4 * g + q4) * S, q2 = qb + (16 + 4 * g + q4) * S;
          U4 fb[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ch = wc * 32 + j * 16 + cc;
            fb[j] = tr_frag(sX, q1, q2, ch >> 3, (ch & 7) * 2);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[kh * KW + kw][i][j] = mfma16<DT>(fa[i], fb[j], acc[kh * KW + kw][i][j]);
        }
    }
  }

  float* P = a.partial + (size_t)blockIdx.y * NT * a.CoutP * a.CinP;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wr * 32 + i * 16 + 4 * g + r, ci = ci0 + wc * 32 + j * 16 + l16;
          if (co < a.CoutP && ci < a.CinP) P[((size_t)t * a.CoutP + co) * a.CinP + ci] = acc[t][i][j][r];
        }
}

int dsr_wgrad_tile_plan(int KH, int KW, int stride, int N, int OH, int OW, int CinP, int CoutP, WgradTileArgs* a) {
  const bool k3 = KH == 3 && KW == 3 && (stride == 1 || stride == 2);
  const bool k1 = KH == 1 && KW == 1 && stride == 1;
  if (!k3 && !k1) return 0;
  const int R = stride == 1 ? 4 : 2;
  a->tiles_y = (OH + R - 1) / R;
  a->tiles_x = (OW + 31) / 32;
  a->ntiles = N * a->tiles_y * a->tiles_x;
  a->tiles_co = (CoutP + 63) / 64;
  a->tiles_ci = (CinP + 63) / 64;
  long long pairs = (long long)a->tiles_co * a->tiles_ci;
  long long want = (512 + pairs - 1) / pairs;            // 2 resident blocks per CU (VGPR-limited), one wave of blocks
  if (want > a->ntiles) want = a->ntiles;
  if (want < 1) want = 1;
  a->tiles_per_block = (int)((a->ntiles + want - 1) / want);
  return (a->ntiles + a->tiles_per_block - 1) / a->tiles_per_block;
}

template <int DT>
static void launch_dt(const WgradTileArgs& a, int KH, int stride, dim3 grid, hipStream_t st) {
  if (KH == 3 && stride == 1)
    hipLaunchKernelGGL((conv_wgrad_tile_kernel<DT, 3, 3, 1>), grid, dim3(256), 0, st, a);
  else if (KH == 3)
    hipLaunchKernelGGL((conv_wgrad_tile_kernel<DT, 3, 3, 2>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_tile_kernel<DT, 1, 1, 1>), grid, dim3(256), 0, st, a);
}

void dsr_launch_wgrad_tile(const WgradTileArgs& a, int KH, int stride, int ychunks, int dtype, hipStream_t st) {
  dim3 grid(a.tiles_co * a.tiles_ci, ychunks);
  if (dtype == DSR_DTYPE_BF16)
    launch_dt<DSR_DTYPE_BF16>(a, KH, stride, grid, st);
  else
    launch_dt<DSR_DTYPE_F16>(a, KH, stride, grid, st);
}
