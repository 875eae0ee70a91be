// The discriminator's dense head on gfx950 (models/GAN/discriminator.py:37-45, 65-72):
//   flatten (C,H,W order, :65) -> Linear(K, 1024) -> LeakyReLU(0.2) -> Linear(1024, 1) -> Sigmoid.
// dense1 is a batch-32 x (K up to 524,288) x 1024 problem: every kernel here is bound by
// streaming the K x 1024 weight (or its gradient) once through HBM, so the design goal is
// full-line, fully coalesced 16-byte accesses with many bytes in flight; MFMA 16x16x32 is used only
// because it is the cheapest way to do the (tiny) arithmetic at that rate.
//
//   linear_fwd   : split-K partial slabs [S][B][O] (deterministic), W rows streamed straight to VGPRs
//   linear_dgrad : dx[b][k] = sum_o dy[b][o] W[o][k]  -- W tile through LDS, ds_read_b64_tr_b16 transposes
//   linear_wgrad : dW[o][k] = sum_b dy[b][o] x[b][k]  -- batch is the MFMA K; 16-byte fp32 stores
#include "../../include/dsr_hip.h"
#include <cstdlib>

#include "dsr_common.h"
#include "dsr_kernels.h"

// ------------------------------------------------------------------ fp32 -> 16-bit shadow copy of a weight
template <int DT>
__global__ void cast16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n8; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    reinterpret_cast<U4*>(dst)[i] = pack8<DT>(f);
  }
}

// ------------------------------------------------------------------ flatten NHWC -> [B][C*HW] (CHW order) and friends
// mode 0: flat[b][c*HW+p] = act[b][p][c]        (forward input of dense1)
// mode 1: flatT[c*HW+p][b] = act[b][p][c]       (batch-minor copy for wgrad; Bp columns, zero padded)
// mode 2: act[b][p][c] = flat[b][c*HW+p]        (gradient back to NHWC)
// All three are 2-byte transposes; a thread moves 8 channels (one 16-byte NHWC vector) and the 64 lanes of a wave sit
// on the index that is CONTIGUOUS ON THE 2-BYTE SIDE (pixels for modes 0/2, batch for mode 1), so every 2-byte-side
// access is one 128-byte line per wave; the 16-byte side is strided, but a wave walks the 8 vectors of each 128-byte
// line back to back (L1 hits).  The element-per-thread form touched one line per 2 bytes.
template <int DT>
__global__ __launch_bounds__(256) void flatten_kernel(const unsigned short* __restrict__ src,
                                                      unsigned short* __restrict__ dst, int B, int HW, int C, int Cp,
                                                      int Bp, int mode) {
  const int lane = threadIdx.x & 63;
  const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per (64 items, 64-channel group)
  const int cgroups = (Cp + 63) / 64;
  if (mode == 0 || mode == 2) {
    const int ptiles = (HW + 63) / 64;
    if (wv >= (size_t)B * ptiles * cgroups) return;
    const int cg = (int)(wv % cgroups);
    const int pt = (int)((wv / cgroups) % ptiles);
    const int b = (int)(wv / ((size_t)cgroups * ptiles));
    const int p = pt * 64 + lane;
    if (p >= HW) return;
    const size_t K = (size_t)C * HW;
    for (int c8 = 0; c8 < 8; ++c8) {
      const int c0 = cg * 64 + c8 * 8;
      if (c0 >= Cp) break;
      if (mode == 0) {                                   // flat[b][c*HW+p] = act[b][p][c]
        const U4 v = *reinterpret_cast<const U4*>(src + ((size_t)b * HW + p) * Cp + c0);
        const unsigned short e[8] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16),
                                     (unsigned short)(v.y & 0xffff), (unsigned short)(v.y >> 16),
                                     (unsigned short)(v.z & 0xffff), (unsigned short)(v.z >> 16),
                                     (unsigned short)(v.w & 0xffff), (unsigned short)(v.w >> 16)};
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c0 + i < C) dst[(size_t)b * K + (size_t)(c0 + i) * HW + p] = e[i];
      } else {                                           // act[b][p][c] = flat[b][c*HW+p]  (pad channels zero)
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (c0 + i < C) ? src[(size_t)b * K + (size_t)(c0 + i) * HW + p] : (unsigned short)0;
        U4 v;
        v.x = e[0] | ((unsigned)e[1] << 16);
        v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16);
        v.w = e[6] | ((unsigned)e[7] << 16);
        *reinterpret_cast<U4*>(dst + ((size_t)b * HW + p) * Cp + c0) = v;
      }
    }
  } else {                                               // mode 1: flatT[c*HW+p][b] = act[b][p][c]  (Bp columns, zero padded)
    const int btiles = (Bp + 63) / 64;
    if (wv >= (size_t)HW * btiles * cgroups) return;
    const int cg = (int)(wv % cgroups);
    const int bt = (int)((wv / cgroups) % btiles);
    const int p = (int)(wv / ((size_t)cgroups * btiles));
    const int b = bt * 64 + lane;
    if (b >= Bp) return;
    for (int c8 = 0; c8 < 8; ++c8) {
      const int c0 = cg * 64 + c8 * 8;
      if (c0 >= Cp) break;
      U4 v = U4{0u, 0u, 0u, 0u};
      if (b < B) v = *reinterpret_cast<const U4*>(src + ((size_t)b * HW + p) * Cp + c0);
      const unsigned short e[8] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16),
                                   (unsigned short)(v.y & 0xffff), (unsigned short)(v.y >> 16),
                                   (unsigned short)(v.z & 0xffff), (unsigned short)(v.z >> 16),
                                   (unsigned short)(v.w & 0xffff), (unsigned short)(v.w >> 16)};
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (c0 + i < C) dst[((size_t)(c0 + i) * HW + p) * Bp + b] = e[i];
    }
  }
}

// The same three copies through LDS for the shapes the dense head has (Cp a multiple of 64; HW, resp. Bp, a multiple of 8): a
// block moves one 64-item x 64-channel tile, 16 bytes per lane and full 128-byte lines on BOTH sides (the strided form above
// fetched every source line 4-5 times: 319 MB of reads per 67 MB tensor in the PMC pass).  mode 0 / 1: [item][channel] ->
// [channel][item] with item = pixel / batch sample; mode 2 the inverse.  Rows of 65 halfwords + 1: the 2-byte column walks of
// the transposing side fall on distinct banks.
template <int DT>
__global__ __launch_bounds__(256) void flatten_tile_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                           int B, int HW, int C, int Cp, int Bp, int mode) {
  constexpr int PITCH = 66;                                  // halfwords per LDS row (64 + 2: keeps 4-byte alignment of the rows)
  __shared__ unsigned short tile[64 * PITCH];
  const int tid = threadIdx.x;
  const int cgroups = Cp / 64;
  const int cg = blockIdx.x % cgroups;
  const int c0 = cg * 64;
  const size_t K = (size_t)C * HW;
  if (mode == 0 || mode == 2) {
    const int ptiles = (HW + 63) / 64;
    const int pt = (blockIdx.x / cgroups) % ptiles, b = blockIdx.x / (cgroups * ptiles);
    const int p0 = pt * 64;
    if (mode == 0) {
      // in: act[b][p0 + r][c0 + 8j ..]  ->  tile[r][8j ..]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i, j = tid & 7;
        U4 v = U4{0u, 0u, 0u, 0u};
        if (p0 + r < HW) v = *reinterpret_cast<const U4*>(src + ((size_t)b * HW + p0 + r) * Cp + c0 + 8 * j);
        unsigned* t32 = reinterpret_cast<unsigned*>(tile + r * PITCH + 8 * j);
        t32[0] = v.x, t32[1] = v.y, t32[2] = v.z, t32[3] = v.w;
      }
      __syncthreads();
      // out: flat[b][(c0 + c) * HW + p0 + 8v ..] = tile[8v + k][c], k = 0..7
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = (tid >> 3) + 32 * i, v8 = tid & 7;
        if (c0 + c < C && p0 + 8 * v8 < HW) {                // (HW % 8 == 0: a vector is inside or outside as a whole)
          unsigned short e[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) e[k] = tile[(8 * v8 + k) * PITCH + c];
          U4 o;
          o.x = e[0] | ((unsigned)e[1] << 16), o.y = e[2] | ((unsigned)e[3] << 16);
          o.z = e[4] | ((unsigned)e[5] << 16), o.w = e[6] | ((unsigned)e[7] << 16);
          *reinterpret_cast<U4*>(dst + (size_t)b * K + (size_t)(c0 + c) * HW + p0 + 8 * v8) = o;
        }
      }
    } else {
      // in: flat[b][(c0 + c) * HW + p0 + 8v ..]  ->  tile[c][8v ..]     (pad channels: zeros)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = (tid >> 3) + 32 * i, v8 = tid & 7;
        U4 v = U4{0u, 0u, 0u, 0u};
        if (c0 + c < C && p0 + 8 * v8 < HW) v = *reinterpret_cast<const U4*>(src + (size_t)b * K + (size_t)(c0 + c) * HW + p0 + 8 * v8);
        unsigned* t32 = reinterpret_cast<unsigned*>(tile + c * PITCH + 8 * v8);
        t32[0] = v.x, t32[1] = v.y, t32[2] = v.z, t32[3] = v.w;
      }
      __syncthreads();
      // out: act[b][p0 + r][c0 + 8j ..] = tile[8j + k][r]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i, j = tid & 7;
        if (p0 + r < HW) {
          unsigned short e[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) e[k] = tile[(8 * j + k) * PITCH + r];
          U4 o;
          o.x = e[0] | ((unsigned)e[1] << 16), o.y = e[2] | ((unsigned)e[3] << 16);
          o.z = e[4] | ((unsigned)e[5] << 16), o.w = e[6] | ((unsigned)e[7] << 16);
          *reinterpret_cast<U4*>(dst + ((size_t)b * HW + p0 + r) * Cp + c0 + 8 * j) = o;
        }
      }
    }
  } else {
    // mode 1: flatT[(c0 + c) * HW + p][b0 + 8v ..] = act[b0 + 8v + k][p][c0 + c]   (b >= B: zeros)
    const int btiles = (Bp + 63) / 64;
    const int bt = (blockIdx.x / cgroups) % btiles, p = blockIdx.x / (cgroups * btiles);
    const int b0 = bt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 3) + 32 * i, j = tid & 7;
      U4 v = U4{0u, 0u, 0u, 0u};
      if (b0 + r < B) v = *reinterpret_cast<const U4*>(src + ((size_t)(b0 + r) * HW + p) * Cp + c0 + 8 * j);
      unsigned* t32 = reinterpret_cast<unsigned*>(tile + r * PITCH + 8 * j);
      t32[0] = v.x, t32[1] = v.y, t32[2] = v.z, t32[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = (tid >> 3) + 32 * i, v8 = tid & 7;
      if (c0 + c < C && b0 + 8 * v8 < Bp) {                  // (Bp % 8 == 0)
        unsigned short e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = tile[(8 * v8 + k) * PITCH + c];
        U4 o;
        o.x = e[0] | ((unsigned)e[1] << 16), o.y = e[2] | ((unsigned)e[3] << 16);
        o.z = e[4] | ((unsigned)e[5] << 16), o.w = e[6] | ((unsigned)e[7] << 16);
        *reinterpret_cast<U4*>(dst + ((size_t)(c0 + c) * HW + p) * Bp + b0 + 8 * v8) = o;
      }
    }
  }
}

// ------------------------------------------------------------------ forward: split-K partials
// W is [O][K] row-major and the reduction runs along the contiguous axis, so an MFMA operand read straight from
// global memory puts adjacent lanes on rows 2K bytes apart (64 separate cache lines per load instruction: measured
// 2.2 TB/s).  Instead the block stages a 256-row x 64-k weight tile (and the 64 x 64 activation tile) through LDS with
// full-line loads -- 8 adjacent lanes per 128-byte row piece -- and the waves read XOR-swizzled fragments back.
// Register prefetch of the next stage stays in flight across the compute of the current one.
template <int DT, int MT>
__global__ __launch_bounds__(512) void linear_fwd_kernel(const unsigned short* __restrict__ x,
                                                         const unsigned short* __restrict__ w,
                                                         float* __restrict__ partial, int B, size_t K, int O,
                                                         size_t kchunk) {
  constexpr int WB = 256 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[WB + 64 * 128];
  unsigned char* sW = smem;
  unsigned char* sX = smem + WB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int ob = blockIdx.x * 256;
  const size_t k0 = (size_t)blockIdx.y * kchunk;
  size_t k1 = k0 + kchunk;
  if (k1 > K) k1 = K;
  const int lc = tid & 7, lr = tid >> 3;       // loader: chunk lc (8 k) of rows lr + 64 i
  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // KT 64-wide k tiles are requested together (a 256-byte run per weight row instead of 128-byte ones a step apart: the
  // rows are K * 2 bytes = 1 MB apart, so every request opens a DRAM page of its own) and consumed one after the other
  // through the same LDS tile.
  constexpr int KT = 2;                      // k tiles requested together (4: slower again, 292 vs 252 us -- registers)
  U4 vw[KT][4], vx[KT];
  auto gload = [&](size_t k) {
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      const size_t kk = k + 64 * h + lc * 8;
      const bool kok = kk < k1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = ob + lr + 64 * i;
        vw[h][i] = load16_or_zero(w, (size_t)o * K + kk, kok && o < O);
      }
      vx[h] = load16_or_zero(x, (size_t)lr * K + kk, kok && lr < B);
    }
  };
  gload(k0);
  for (size_t k = k0; k < k1; k += 64 * KT) {
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      if (k + 64 * h >= k1) break;                           // (uniform)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lr + 64 * i;
        *reinterpret_cast<U4*>(sW + row * 128 + ((lc ^ (row & 7)) << 4)) = vw[h][i];
      }
      *reinterpret_cast<U4*>(sX + lr * 128 + ((lc ^ (lr & 7)) << 4)) = vx[h];
      __syncthreads();
      if (h == KT - 1 && k + 64 * KT < k1) gload(k + 64 * KT);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = 4 * ks + g;
        U4 fb[2], fa[MT];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const int row = wave * 32 + nf * 16 + r;
          fb[nf] = *reinterpret_cast<const U4*>(sW + row * 128 + ((ch ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = 16 * m + r;
          fa[m] = *reinterpret_cast<const U4*>(sX + row * 128 + ((ch ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = mfma16<DT>(fa[m], fb[0], acc[m][0]);
          acc[m][1] = mfma16<DT>(fa[m], fb[1], acc[m][1]);
        }
      }
      __syncthreads();
    }
  }
  float* P = partial + (size_t)blockIdx.y * B * O;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = 16 * m + 4 * g + j, o = ob + wave * 32 + nf * 16 + r;
        if (b < B && o < O) P[(size_t)b * O + o] = acc[m][nf][j];
      }
}

// out[b][o] = act(sum_s partial[s][b][o] + bias[o])   (fp32)
__global__ void linear_reduce_kernel(const float* __restrict__ partial, int S, int B, int O,
                                     const float* __restrict__ bias, int act, float slope, float* __restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * O) return;
  double s = 0.0;
  const size_t stride = (size_t)B * O;
  int z = 0;
  for (; z + 8 <= S; z += 8) {              // 8 independent loads in flight (one at a time: 0.4 us per split slice), summed in order
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = partial[(size_t)(z + u) * stride + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v8[u];
  }
  for (; z < S; ++z) s += (double)partial[(size_t)z * stride + idx];
  float v = (float)s + (bias ? bias[idx % O] : 0.f);
  out[idx] = act_apply(act, v, slope);
}

// ------------------------------------------------------------------ dgrad: dx[b][k] = sum_o dy[b][o] W[o][k]
__device__ __forceinline__ s16x4 lds_tr_read16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// Block = 128 k-columns; the O rows stream through a double-buffered LDS stage of 32 rows (W: 32 x 256 B, XOR-swizzled
// 16-byte chunks; dy: 64 x 64 B padded to 80).  Global loads for stage s+1 are issued before stage s is computed and
// land in the other buffer afterwards: one barrier per stage, loads always in flight.
// k-slot mapping of one 32-row stage (fixed by the transposing read): slots 0-3 of lane group g <-> rows 4g..4g+3,
// slots 4-7 <-> rows 16+4g..16+4g+3.
// NFW n-fragments per wave: 2 = 128 k columns per block (256-byte runs per weight row), 4 = 256 columns (512-byte runs: the rows
// are 1 MB apart, every run opens a DRAM page of its own).
template <int DT, int MT, int NFW = 2>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ w,
                                                           unsigned short* __restrict__ dx, int B, int O, size_t K) {
  constexpr int COLS = 64 * NFW, RP = COLS * 2;            // k columns per block, bytes per staged weight row
  constexpr int CPR = RP / 16, RPP = 256 / CPR, NLD = 32 / RPP;   // 16-byte chunks per row, rows per loader pass, passes per stage
  constexpr int WB = 32 * RP, YB = 64 * 80;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (WB + YB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const size_t kb = (size_t)blockIdx.x * COLS;
  const int c = tid % CPR, rr = tid / CPR;      // W loader: 16-byte chunk c of rows rr, rr + RPP, ...
  const bool kok = (kb + c * 8) < K;
  const int yb = tid >> 2, yc = tid & 3;        // dy loader: batch row yb, chunk yc (8 outputs)
  const bool ybok = yb < B;
  f32x4 acc[MT][NFW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) acc[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int q = l16 >> 2, cc = 4 * (l16 & 3);
  U4 vw[NLD], vy;
  auto gload = [&](int o0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      vw[i] = load16_or_zero(w, (size_t)(o0 + rr + RPP * i) * K + kb + c * 8, kok && (o0 + rr + RPP * i) < O);
    vy = load16_or_zero(dy, (size_t)yb * O + o0 + yc * 8, ybok && (o0 + yc * 8) < O);
  };
  auto sstore = [&](int buf) {
    unsigned char* sW = smem + buf * (WB + YB);
    unsigned char* sY = sW + WB;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = rr + RPP * i;
      *reinterpret_cast<U4*>(sW + row * RP + ((c ^ (row & 15)) << 4)) = vw[i];
    }
    *reinterpret_cast<U4*>(sY + yb * 80 + yc * 16) = vy;
  };
  const int nstage = (O + 31) / 32;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstage) gload(32 * (s + 1));
    const unsigned char* sW = smem + buf * (WB + YB);
    const unsigned char* sY = sW + WB;
    U4 fb[NFW];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      const int col = wave * 16 * NFW + nf * 16 + cc;
      const int chunk = col >> 3, within = (col & 7) * 2;
      const int p1 = 4 * g + q, p2 = p1 + 16;
      s16x4 b1 = lds_tr_read16(sW + p1 * RP + ((chunk ^ (p1 & 15)) << 4) + within);
      s16x4 b2 = lds_tr_read16(sW + p2 * RP + ((chunk ^ (p2 & 15)) << 4) + within);
      fb[nf] = __builtin_bit_cast(U4, __builtin_shufflevector(b1, b2, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const unsigned char* row = sY + (16 * m + l16) * 80;
      const uint2 a1 = *reinterpret_cast<const uint2*>(row + 8 * g);
      const uint2 a2 = *reinterpret_cast<const uint2*>(row + 32 + 8 * g);
      U4 fa = {a1.x, a1.y, a2.x, a2.y};
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) acc[m][nf] = mfma16<DT>(fa, fb[nf], acc[m][nf]);
    }
    if (s + 1 < nstage) sstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      const size_t k = kb + wave * 16 * NFW + nf * 16 + l16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = 16 * m + 4 * g + j;
        if (b < B && k < K) dx[(size_t)b * K + k] = f2h<DT>(acc[m][nf][j]);
      }
    }
}

// ------------------------------------------------------------------ wgrad: dW[o][k] = sum_b dyT[o][b] xT[k][b]
// 32x32x16 MFMA with D[m = o][n = k]: the 32 lanes of a half-wave hold 32 consecutive k of one output row, so every
// store instruction writes two full 128-byte lines (a 16x16 tile would write 64-byte pieces, which the memory side
// turns into read-modify-write).  A wave keeps its 64 x BP slice of dyT in registers and streams xT.
template <int DT, int BP>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const unsigned short* __restrict__ dyT,
                                                           const unsigned short* __restrict__ xT,
                                                           float* __restrict__ dw, int O, size_t K, int ktiles_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int o0 = blockIdx.y * 256 + wave * 64;
  if (o0 >= O) return;
  constexpr int KS = BP / 16;
  U4 fa[2][KS];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int o = o0 + 32 * mf + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) fa[mf][s] = load16_or_zero(dyT, (size_t)o * BP + 16 * s + 8 * hf, o < O);
  }
  const size_t kt0 = (size_t)blockIdx.x * ktiles_per_block;
  U4 fb[KS], fn[KS];
  {
    const size_t krow = kt0 * 32 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) fb[s] = load16_or_zero(xT, krow * BP + 16 * s + 8 * hf, krow < K);
  }
  for (int t = 0; t < ktiles_per_block; ++t) {
    const size_t kbase = (kt0 + t) * 32;
    if (kbase >= K) break;
    {
      const size_t krow = kbase + 32 + r;
      const bool ok = (t + 1 < ktiles_per_block) && krow < K;
#pragma unroll
      for (int s = 0; s < KS; ++s) fn[s] = load16_or_zero(xT, krow * BP + 16 * s + 8 * hf, ok);
    }
    const size_t k = kbase + r;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma32<DT>(fa[mf][s], fb[s], acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = o0 + 32 * mf + 8 * (i >> 2) + 4 * hf + (i & 3);
        if (o < O && k < K) dw[(size_t)o * K + k] = acc[i];
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) fb[s] = fn[s];
  }
}

// ------------------------------------------------------------------ wgrad + Adam in one pass
// dense1 of the config-3 discriminator holds 537 M parameters: its gradient is a rank-64 product (64 samples), so
// "write 2.1 GB of dW, then read it back in the Adam launch" is 4.3 GB of HBM traffic for a tensor that never needs to
// exist.  Here a wave forms a 64 (o) x 64 (k) tile of dW in MFMA accumulators exactly as linear_wgrad_kernel /
// linear_wgrad_gathered_kernel do (same fragments, same order: the same bits), turns it through LDS so that a lane owns
// 4 consecutive k of one row, and applies Adam to p / m / v in place (16-byte nontemporal accesses, full 128-byte lines
// per row for the fp32 tensors AND for the bf16 shadow of p, which is why the tile is 64 k wide).
struct AdamFused {
  float* p;
  float* m;
  float* v;
  unsigned short* shadow;   // bf16 image of p (the MFMA operand of the next forward), may be null
  const int* step;
  float lr, b1, b2, eps, grad_scale;
};
template <int DT, int BP>
__global__ __launch_bounds__(256, 2) void linear_wgrad_adam_kernel(const unsigned short* __restrict__ dyT,
                                                                   const unsigned short* __restrict__ xT, int O, size_t K,
                                                                   int R, float scale, int kpairs_per_block,
                                                                   const AdamFused a) {
  typedef __attribute__((ext_vector_type(4))) float F4;
  typedef __attribute__((ext_vector_type(2))) unsigned U2;
  constexpr int RS = 72;                         // tile row pitch in floats: 4 * RS = 32 (mod 64 banks), so the two half-waves
                                                 // of a ds_write_b32 (rows 4 apart) land on disjoint banks
  __shared__ __attribute__((aligned(16))) float tile[4][32 * RS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  // the four waves of a block share 64 rows of o and take adjacent 64-wide k tiles: the block touches 1 KB contiguous
  // per row and tensor at a time (rows are K * 4 bytes = 2 MB apart)
  const int o0 = blockIdx.y * 64;
  constexpr int KS = BP / 16;
  const AdamCoef co = adam_coef(a.step, a.lr, a.b1, a.b2, a.eps, a.grad_scale);
  float* my = tile[wave];                        // (no block-level barrier below: a wave owns its slice of `tile`)
  const size_t kp0 = (size_t)blockIdx.x * kpairs_per_block;
  for (int t = 0; t < kpairs_per_block; ++t) {
    const size_t kbase = ((kp0 + t) * 4 + wave) * 64;
    if (kbase >= K) break;
    f32x16 acc[2][2];                            // [mf: 32-row half of the wave's 64 o][kt: 32-wide half of the 64 k]
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][kt][i] = 0.f;
    for (int q = 0; q < R; ++q) {
      U4 fa[2][KS], fb[2][KS];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const size_t krow = kbase + 32 * kt + r;
#pragma unroll
        for (int s = 0; s < KS; ++s) fb[kt][s] = load16_or_zero(xT, ((size_t)q * K + krow) * BP + 16 * s + 8 * hf, krow < K);
      }
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        const int o = o0 + 32 * mf + r;
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[mf][s] = load16_or_zero(dyT, ((size_t)q * O + o) * BP + 16 * s + 8 * hf, o < O);
      }
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < KS; ++s) acc[mf][kt] = mfma32<DT>(fa[mf][s], fb[kt][s], acc[mf][kt]);
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      // this lane's part of the 32 x 64 tile after the turn: rows 4 j + (lane >> 4), floats 4 (lane & 15) .. + 3
      const int rr = lane >> 4, kc = 4 * (lane & 15);
      F4 pv[8], mv[8], vv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {              // requested before the turn: 24 x 16 bytes per lane in flight
        const int o = o0 + 32 * mf + 4 * j + rr;
        const size_t e = (size_t)(o < O ? o : 0) * K + kbase + kc;
        pv[j] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(a.p + e));
        mv[j] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(a.m + e));
        vv[j] = __builtin_nontemporal_load(reinterpret_cast<const F4*>(a.v + e));
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          my[(8 * (i >> 2) + 4 * hf + (i & 3)) * RS + 32 * kt + r] = acc[mf][kt][i] * scale;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();           // (LDS serves one wave's requests in order; this only pins the compiler)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = o0 + 32 * mf + 4 * j + rr;
        const F4 g = *reinterpret_cast<const F4*>(my + (4 * j + rr) * RS + kc);
        F4 pn = pv[j], mn = mv[j], vn = vv[j];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float pk = pn[c], mk = mn[c], vk = vn[c];
          adam_update(pk, g[c], mk, vk, co);
          pn[c] = pk;
          mn[c] = mk;
          vn[c] = vk;
        }
        if (o < O) {
          const size_t e = (size_t)o * K + kbase + kc;
          __builtin_nontemporal_store(pn, reinterpret_cast<F4*>(a.p + e));
          __builtin_nontemporal_store(mn, reinterpret_cast<F4*>(a.m + e));
          __builtin_nontemporal_store(vn, reinterpret_cast<F4*>(a.v + e));
          if (a.shadow) {
            U2 h;
            h.x = (unsigned)f2h<DSR_DTYPE_BF16>(pn[0]) | ((unsigned)f2h<DSR_DTYPE_BF16>(pn[1]) << 16);
            h.y = (unsigned)f2h<DSR_DTYPE_BF16>(pn[2]) | ((unsigned)f2h<DSR_DTYPE_BF16>(pn[3]) << 16);
            __builtin_nontemporal_store(h, reinterpret_cast<U2*>(a.shadow + e));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();           // the tile is free again
    }
  }
}

// Data-parallel form: dW = scale * sum_r dyT_r xT_r over R gathered rank-local factor pairs ([R][O][BP], [R][K][BP]).
// A data-parallel step would otherwise all-reduce dW itself (2.1 GB for the config-3 dense1; one xGMI link between two
// GPUs moves that in ~30 ms) while the factors are 67 MB + 128 KB per rank: all-gather those and form the summed
// gradient here.  The dyT fragments no longer fit in registers for R ranks, so they are re-read (L2) per k tile.
template <int DT, int BP>
__global__ __launch_bounds__(256) void linear_wgrad_gathered_kernel(const unsigned short* __restrict__ dyT,
                                                                    const unsigned short* __restrict__ xT,
                                                                    float* __restrict__ dw, int O, size_t K, int R,
                                                                    float scale, int ktiles_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int o0 = blockIdx.y * 256 + wave * 64;
  if (o0 >= O) return;
  constexpr int KS = BP / 16;
  const size_t kt0 = (size_t)blockIdx.x * ktiles_per_block;
  for (int t = 0; t < ktiles_per_block; ++t) {
    const size_t kbase = (kt0 + t) * 32;
    if (kbase >= K) break;
    const size_t krow = kbase + r;
    f32x16 acc[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][i] = 0.f;
    for (int q = 0; q < R; ++q) {
      U4 fb[KS], fa[2][KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) fb[s] = load16_or_zero(xT, ((size_t)q * K + krow) * BP + 16 * s + 8 * hf, krow < K);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        const int o = o0 + 32 * mf + r;
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[mf][s] = load16_or_zero(dyT, ((size_t)q * O + o) * BP + 16 * s + 8 * hf, o < O);
      }
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int s = 0; s < KS; ++s) acc[mf] = mfma32<DT>(fa[mf][s], fb[s], acc[mf]);
    }
    const size_t k = kbase + r;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = o0 + 32 * mf + 8 * (i >> 2) + 4 * hf + (i & 3);
        if (o < O && k < K) dw[(size_t)o * K + k] = acc[mf][i] * scale;
      }
  }
}

// ------------------------------------------------------------------ the small fp32 tail: Linear(K1,1) + Sigmoid and its backward
// out[b] = sigmoid(sum_k h[b][k] w2[k] + b2)
__global__ __launch_bounds__(256) void dense2_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, int K1, float* __restrict__ out) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < K1; k += 256) s += h[(size_t)b * K1 + k] * w2[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0] + red[1] + red[2] + red[3] + b2[0];
    out[b] = 1.f / (1.f + expf(-v));
  }
}

// one thread per hidden unit k: everything the head's backward needs besides the two big GEMMs
//   dz[b]   = dout[b] * out[b] * (1 - out[b])
//   dw2[k]  = sum_b dz[b] * h[b][k];  db2 = sum_b dz[b]
//   dh[b][k]= dz[b] * w2[k] * leaky'(h[b][k])     (h is the post-LeakyReLU activation; slope > 0)
//   db1[k]  = sum_b dh[b][k];  dy16[b][k] and dyT16[k][b] = 16-bit copies of dh for the MFMA kernels
template <int DT>
__global__ void dense2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                  const float* __restrict__ h, const float* __restrict__ w2, int B, int K1, int BP,
                                  float slope, float* __restrict__ dw2, float* __restrict__ db2,
                                  float* __restrict__ db1, unsigned short* __restrict__ dy16,
                                  unsigned short* __restrict__ dyT16) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K1) return;
  float sw = 0.f, sb1 = 0.f, sz = 0.f;
  const float wk = w2[k];
  for (int b = 0; b < BP; ++b) {
    float d = 0.f;
    if (b < B) {
      const float o = out[b];
      const float dz = dout[b] * o * (1.f - o);
      const float hv = h[(size_t)b * K1 + k];
      sz += dz;
      sw += dz * hv;
      d = dz * wk * (hv >= 0.f ? 1.f : slope);
      sb1 += d;
      dy16[(size_t)b * K1 + k] = f2h<DT>(d);
    }
    dyT16[(size_t)k * BP + b] = f2h<DT>(d);
  }
  dw2[k] = sw;
  db1[k] = sb1;
  if (k == 0) db2[0] = sz;
}

// ================================================================== C ABI
extern "C" int dsr_cast16(int dtype, const float* src, void* dst, size_t n, dsr_stream_t st) {
  DSR_REQUIRE(src && dst && DSR_DTYPE_OK(dtype) && n > 0, "cast16: null pointer or empty");
  if (n % 8) return dsr_fail(DSR_E_ARG, "cast16: element count %zu not a multiple of 8", n);
  size_t n8 = n / 8;
  unsigned blocks = (unsigned)((n8 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == DSR_BF16)
    hipLaunchKernelGGL((cast16_kernel<DSR_DTYPE_BF16>), dim3(blocks), dim3(256), 0, st, src, (unsigned short*)dst, n8);
  else
    hipLaunchKernelGGL((cast16_kernel<DSR_DTYPE_F16>), dim3(blocks), dim3(256), 0, st, src, (unsigned short*)dst, n8);
  return dsr_launch_status("dsr_cast16");
}

extern "C" int dsr_flatten(int dtype, const void* src, void* dst, int B, int HW, int C, int Cp, int Bp, int mode,
                           dsr_stream_t st) {
  DSR_REQUIRE(src && dst && DSR_DTYPE_OK(dtype) && B > 0 && HW > 0 && C > 0 && Cp >= C && mode >= 0 && mode <= 2 && (mode != 1 || Bp >= B), "flatten: null pointer or bad shape");
  if (Cp % 8) return dsr_fail(DSR_E_ARG, "flatten: Cp %% 8 != 0");
  const size_t cgroups = (size_t)(Cp + 63) / 64;
  // DSR_FLATTEN_TILE (tuning switch, read per call: a test compares the two forms): 0 = the strided form for every shape
  const char* ft = getenv("DSR_FLATTEN_TILE");
  if (!(ft && ft[0] == '0') && Cp % 64 == 0 && (mode == 1 ? Bp % 8 == 0 : HW % 8 == 0)) {
    const size_t tiles = (mode == 1 ? (size_t)HW * ((Bp + 63) / 64) : (size_t)B * ((HW + 63) / 64)) * cgroups;
    if (tiles < (1ull << 31)) {
      if (dtype == DSR_BF16)
        hipLaunchKernelGGL((flatten_tile_kernel<DSR_DTYPE_BF16>), dim3((unsigned)tiles), dim3(256), 0, st, (const unsigned short*)src,
                           (unsigned short*)dst, B, HW, C, Cp, Bp, mode);
      else
        hipLaunchKernelGGL((flatten_tile_kernel<DSR_DTYPE_F16>), dim3((unsigned)tiles), dim3(256), 0, st, (const unsigned short*)src,
                           (unsigned short*)dst, B, HW, C, Cp, Bp, mode);
      return dsr_launch_status("dsr_flatten");
    }
  }
  const size_t waves = mode == 1 ? (size_t)HW * ((Bp + 63) / 64) * cgroups : (size_t)B * ((HW + 63) / 64) * cgroups;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (dtype == DSR_BF16)
    hipLaunchKernelGGL((flatten_kernel<DSR_DTYPE_BF16>), grid, block, 0, st, (const unsigned short*)src,
                       (unsigned short*)dst, B, HW, C, Cp, Bp, mode);
  else
    hipLaunchKernelGGL((flatten_kernel<DSR_DTYPE_F16>), grid, block, 0, st, (const unsigned short*)src,
                       (unsigned short*)dst, B, HW, C, Cp, Bp, mode);
  return dsr_launch_status("dsr_flatten");
}

static int linear_splits(size_t K, int O, size_t* kchunk) {
  long long otiles = (O + 255) / 256;
  long long want = (768 + otiles - 1) / otiles;          // three 8-wave blocks per CU
  long long maxs = (long long)((K + 1023) / 1024);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  size_t ch = (K + want - 1) / want;
  ch = (ch + 127) / 128 * 128;
  *kchunk = ch;
  return (int)((K + ch - 1) / ch);
}

extern "C" size_t dsr_linear_fwd_workspace(int B, size_t K, int O) {
  size_t ch;
  int S = linear_splits(K, O, &ch);
  return (size_t)S * B * O * sizeof(float);
}

extern "C" int dsr_linear_fwd(int dtype, const void* x, const void* w16, const float* bias, int act, float slope,
                              float* out, int B, size_t K, int O, void* workspace, size_t ws_bytes, dsr_stream_t st) {
  DSR_REQUIRE(x && w16 && out && DSR_DTYPE_OK(dtype) && K > 0 && O > 0, "linear_fwd: null pointer or bad shape");
  if (B < 1 || B > 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear: batch %d outside 1..64", B);
  if (K % 8) return dsr_fail(DSR_E_ARG, "linear: K %% 8 != 0");
  size_t ch;
  int S = linear_splits(K, O, &ch);
  if (!workspace || ws_bytes < (size_t)S * B * O * sizeof(float)) return dsr_fail(DSR_E_WORKSPACE, "linear_fwd: workspace");
  dim3 grid((O + 255) / 256, S), block(512);
  const unsigned short* X = (const unsigned short*)x;
  const unsigned short* W = (const unsigned short*)w16;
  float* P = (float*)workspace;
#define LAUNCH_FWD(DTV, MTV) \
  hipLaunchKernelGGL((linear_fwd_kernel<DTV, MTV>), grid, block, 0, st, X, W, P, B, K, O, ch)
  if (dtype == DSR_BF16) {
    if (B <= 32) LAUNCH_FWD(DSR_DTYPE_BF16, 2); else LAUNCH_FWD(DSR_DTYPE_BF16, 4);
  } else {
    if (B <= 32) LAUNCH_FWD(DSR_DTYPE_F16, 2); else LAUNCH_FWD(DSR_DTYPE_F16, 4);
  }
#undef LAUNCH_FWD
  hipLaunchKernelGGL(linear_reduce_kernel, dim3((B * O + 255) / 256), dim3(256), 0, st, P, S, B, O, bias, act, slope, out);
  return dsr_launch_status("dsr_linear_fwd");
}

extern "C" int dsr_linear_dgrad(int dtype, const void* dy16, const void* w16, void* dx, int B, int O, size_t K,
                                dsr_stream_t st) {
  DSR_REQUIRE(dy16 && w16 && dx && DSR_DTYPE_OK(dtype) && K > 0 && O > 0, "linear_dgrad: null pointer or bad shape");
  if (B < 1 || B > 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear: batch %d outside 1..64", B);
  if (K % 8 || O % 8) return dsr_fail(DSR_E_ARG, "linear_dgrad: K %% 8 or O %% 8");
  const char* ew = getenv("DSR_LINEAR_DGRAD_COLS");       // tuning switch, read per call: 128 | 256 k columns per block
  const bool wide = ew && atoi(ew) == 256 && K >= 256 * 512;
  dim3 grid((unsigned)((K + (wide ? 255 : 127)) / (wide ? 256 : 128))), block(256);
  const unsigned short* DY = (const unsigned short*)dy16;
  const unsigned short* W = (const unsigned short*)w16;
  unsigned short* DX = (unsigned short*)dx;
#define LAUNCH_DG(DTV, MTV)                                                                                     \
  do {                                                                                                          \
    if (wide) hipLaunchKernelGGL((linear_dgrad_kernel<DTV, MTV, 4>), grid, block, 0, st, DY, W, DX, B, O, K);   \
    else hipLaunchKernelGGL((linear_dgrad_kernel<DTV, MTV, 2>), grid, block, 0, st, DY, W, DX, B, O, K);        \
  } while (0)
  if (dtype == DSR_BF16) {
    if (B <= 32) LAUNCH_DG(DSR_DTYPE_BF16, 2); else LAUNCH_DG(DSR_DTYPE_BF16, 4);
  } else {
    if (B <= 32) LAUNCH_DG(DSR_DTYPE_F16, 2); else LAUNCH_DG(DSR_DTYPE_F16, 4);
  }
#undef LAUNCH_DG
  return dsr_launch_status("dsr_linear_dgrad");
}

extern "C" int dsr_linear_wgrad(int dtype, const void* dyT16, const void* xT16, float* dw, int Bp, int O, size_t K,
                                dsr_stream_t st) {
  DSR_REQUIRE(dyT16 && xT16 && dw && DSR_DTYPE_OK(dtype) && K > 0 && O > 0, "linear_wgrad: null pointer or bad shape");
  if (Bp != 32 && Bp != 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear_wgrad: padded batch must be 32 or 64");
  if (K % 4) return dsr_fail(DSR_E_ARG, "linear_wgrad: K %% 4");
  const int tpb = 16;   // 32-wide k tiles per block -> 512 k per block
  size_t ktiles = (K + 31) / 32;
  dim3 grid((unsigned)((ktiles + tpb - 1) / tpb), (O + 255) / 256), block(256);
  const unsigned short* DYT = (const unsigned short*)dyT16;
  const unsigned short* XT = (const unsigned short*)xT16;
#define LAUNCH_WG(DTV, BPV) hipLaunchKernelGGL((linear_wgrad_kernel<DTV, BPV>), grid, block, 0, st, DYT, XT, dw, O, K, tpb)
  if (dtype == DSR_BF16) {
    if (Bp == 32) LAUNCH_WG(DSR_DTYPE_BF16, 32); else LAUNCH_WG(DSR_DTYPE_BF16, 64);
  } else {
    if (Bp == 32) LAUNCH_WG(DSR_DTYPE_F16, 32); else LAUNCH_WG(DSR_DTYPE_F16, 64);
  }
#undef LAUNCH_WG
  return dsr_launch_status("dsr_linear_wgrad");
}

extern "C" int dsr_linear_wgrad_gathered(int dtype, const void* dyT16_all, const void* xT16_all, float* dw, int Bp, int O,
                                         size_t K, int R, float scale, dsr_stream_t st) {
  DSR_REQUIRE(dyT16_all && xT16_all && dw && DSR_DTYPE_OK(dtype) && K > 0 && O > 0, "linear_wgrad_gathered: null pointer or bad shape");
  if (Bp != 32 && Bp != 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear_wgrad_gathered: padded batch must be 32 or 64");
  if (R < 1) return dsr_fail(DSR_E_ARG, "linear_wgrad_gathered: R < 1");
  const int tpb = 16;
  size_t ktiles = (K + 31) / 32;
  dim3 grid((unsigned)((ktiles + tpb - 1) / tpb), (O + 255) / 256), block(256);
  const unsigned short* DYT = (const unsigned short*)dyT16_all;
  const unsigned short* XT = (const unsigned short*)xT16_all;
#define LAUNCH_WGG(DTV, BPV) \
  hipLaunchKernelGGL((linear_wgrad_gathered_kernel<DTV, BPV>), grid, block, 0, st, DYT, XT, dw, O, K, R, scale, tpb)
  if (dtype == DSR_BF16) {
    if (Bp == 32) LAUNCH_WGG(DSR_DTYPE_BF16, 32); else LAUNCH_WGG(DSR_DTYPE_BF16, 64);
  } else {
    if (Bp == 32) LAUNCH_WGG(DSR_DTYPE_F16, 32); else LAUNCH_WGG(DSR_DTYPE_F16, 64);
  }
#undef LAUNCH_WGG
  return dsr_launch_status("dsr_linear_wgrad_gathered");
}

extern "C" int dsr_linear_wgrad_adam(int dtype, const void* dyT16_all, const void* xT16_all, int Bp, int O, size_t K, int R,
                                     float scale, float* p, float* m, float* v, void* shadow_bf16, const int* step, float lr,
                                     float b1, float b2, float eps, float grad_scale, dsr_stream_t st) {
  DSR_REQUIRE(dyT16_all && xT16_all && p && m && v && step && DSR_DTYPE_OK(dtype) && K > 0 && O > 0,
              "linear_wgrad_adam: null pointer or bad shape");
  if (Bp != 32 && Bp != 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear_wgrad_adam: padded batch must be 32 or 64");
  if (R < 1) return dsr_fail(DSR_E_ARG, "linear_wgrad_adam: R < 1");
  if (K % 64) return dsr_fail(DSR_E_UNSUPPORTED, "linear_wgrad_adam: K %% 64 (use dsr_linear_wgrad + dsr_pw_adam)");
  const char* e = getenv("DSR_WGRAD_ADAM_KPB");      // tuning switch: 256-wide k groups per block
  const int kpb = e ? atoi(e) : 1;          // (1: 2.71 ms stand-alone and 0.1-0.15 ms per config-3 step better than 2; 4 worse)
  if (kpb < 1) return dsr_fail(DSR_E_ARG, "linear_wgrad_adam: DSR_WGRAD_ADAM_KPB < 1");
  const size_t kgroups = (K + 255) / 256;
  dim3 grid((unsigned)((kgroups + kpb - 1) / kpb), (O + 63) / 64), block(256);
  const unsigned short* DYT = (const unsigned short*)dyT16_all;
  const unsigned short* XT = (const unsigned short*)xT16_all;
  AdamFused a;
  a.p = p;
  a.m = m;
  a.v = v;
  a.shadow = (unsigned short*)shadow_bf16;
  a.step = step;
  a.lr = lr;
  a.b1 = b1;
  a.b2 = b2;
  a.eps = eps;
  a.grad_scale = grad_scale;
#define LAUNCH_WGA(DTV, BPV) \
  hipLaunchKernelGGL((linear_wgrad_adam_kernel<DTV, BPV>), grid, block, 0, st, DYT, XT, O, K, R, scale, kpb, a)
  if (dtype == DSR_BF16) {
    if (Bp == 32) LAUNCH_WGA(DSR_DTYPE_BF16, 32); else LAUNCH_WGA(DSR_DTYPE_BF16, 64);
  } else {
    if (Bp == 32) LAUNCH_WGA(DSR_DTYPE_F16, 32); else LAUNCH_WGA(DSR_DTYPE_F16, 64);
  }
#undef LAUNCH_WGA
  return dsr_launch_status("dsr_linear_wgrad_adam");
}

extern "C" int dsr_dense2_fwd(const float* h, const float* w2, const float* b2, int B, int K1, float* out,
                              dsr_stream_t st) {
  DSR_REQUIRE(h && w2 && out && B > 0 && K1 > 0, "dense2_fwd: null pointer or bad shape");
  hipLaunchKernelGGL(dense2_fwd_kernel, dim3(B), dim3(256), 0, st, h, w2, b2, K1, out);
  return dsr_launch_status("dsr_dense2_fwd");
}

extern "C" int dsr_dense2_bwd(int dtype, const float* dout, const float* out, const float* h, const float* w2, int B,
                              int K1, int Bp, float slope, float* dw2, float* db2, float* db1, void* dy16, void* dyT16,
                              dsr_stream_t st) {
  DSR_REQUIRE(dout && out && h && w2 && dw2 && db2 && db1 && dy16 && dyT16 && DSR_DTYPE_OK(dtype) && B > 0 && K1 > 0 && Bp >= B, "dense2_bwd: null pointer or bad shape");
  dim3 grid((K1 + 127) / 128), block(128);
  if (dtype == DSR_BF16)
    hipLaunchKernelGGL((dense2_bwd_kernel<DSR_DTYPE_BF16>), grid, block, 0, st, dout, out, h, w2, B, K1, Bp, slope, dw2,
                       db2, db1, (unsigned short*)dy16, (unsigned short*)dyT16);
  else
    hipLaunchKernelGGL((dense2_bwd_kernel<DSR_DTYPE_F16>), grid, block, 0, st, dout, out, h, w2, B, K1, Bp, slope, dw2,
                       db2, db1, (unsigned short*)dy16, (unsigned short*)dyT16);
  return dsr_launch_status("dsr_dense2_bwd");
}
