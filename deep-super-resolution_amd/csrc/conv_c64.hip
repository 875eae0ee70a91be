// 3x3 / stride 1 / zero-pad convolution with 64 input and 64 output channels -- the SRGAN trunk
// (models/GAN/generator.py:7,11,52: 33 forward + 33 input-grad launches per generator pass) and every other
// 64->64 layer (VGG conv1_2).  The generic gather kernel spends this shape's short K loop (9 steps) mostly in its
// prologue/epilogue and re-reads weights and input per K-step; here
//   * the block is persistent and its waves keep ALL weight fragments of their 32-channel half in registers
//     (9 taps x 2 k-halves x 2 n-tiles = 36 fragments = 144 VGPRs) for the whole launch,
//   * the 2-row x 32-column pixel tile's input HALO (4 x 34 pixels) is staged once in LDS and every tap is an
//     address offset into it (A fragments: one conflict-free ds_read_b128 per 2 MFMAs),
//   * the NEXT tile's halo travels HBM -> LDS by LDS-DMA (no VGPRs) under the current tile's 72 MFMAs per wave,
// so the MFMA phase touches LDS only.  Same epilogue contract as conv_gemm (bias, activation, BatchNorm
// sum / sum-of-squares rows -- one row per spatial TILE here --, 16-byte NHWC stores).  dgrad = the same kernel on the
// [tap][ci][co] weight image with mirrored tap offsets.
#include "dsr_common.h"
#include "dsr_kernels.h"

template <int DT, bool FOLD>     // FOLD: inference epilogue (eval-mode BatchNorm scale/shift, residual); a separate
                                 // instantiation because the training one has no registers to spare (249 of 256)
__global__ __launch_bounds__(256, 2) void conv_c64_kernel(const C64Args a) {
  constexpr int HC = 40;                       // halo row pitch in pixels (34 used; multiple of 8 keeps the swizzle row-free)
  constexpr int TR = 2;                        // tile rows (one per wave pair)
  constexpr int HRW = TR + 2;                  // halo rows
  constexpr int X_BYTES = HRW * HC * 128;      // 20,480
  constexpr int C_STRIDE = 64 * 2 + 16;
  // one LDS object (a second one beside an LDS-DMA target makes hipcc drain the DMA before every LDS read):
  // two halo stages | C tile | statistics
  constexpr int C_BYTES = TR * 32 * C_STRIDE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * X_BYTES + C_BYTES + 2 * 2 * 64 * 4];
  unsigned char* sC = smem + 2 * X_BYTES;
  float (*sStat)[2][64] = reinterpret_cast<float (*)[2][64]>(smem + 2 * X_BYTES + C_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int wp = wave >> 1, wc = wave & 1;     // tile row, channel half (32 co)
  const int c0 = blockIdx.y * 64;              // this block's slice of the output channels (Cout = 64 * gridDim.y)
  const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(a.w);
  unsigned short* __restrict__ Y = reinterpret_cast<unsigned short*>(a.y);

  // ---- weights: registers, once (B operand: rows = output channels wc*32 + nt*16 + r16)
  U4 fw[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        fw[t][kk][nt] = *reinterpret_cast<const U4*>(W + ((size_t)(t * a.CoutP + c0 + wc * 32 + nt * 16 + r16)) * 64 + kk * 32 + g * 8);

  // A-fragment LDS offsets: pixel column (tx + r16) -> (tx + r16)*128 + swizzled chunk; + row*HC*128 + hx*2048
  int lds_off[3][2];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) lds_off[tx][kk] = (tx + r16) * 128 + (((4 * kk + g) ^ ((tx + r16) & 7)) << 4);

  // loader: LDS-DMA (buffer_load ... lds): wave-instruction u of wave w fills halo slots 32u + 8w .. +7 (8 pixels x
  // 128 B, lane-linear), slot position (lane & 7) of pixel q holding channel chunk (lane & 7) ^ (q & 7); an
  // out-of-range offset writes zeros (image border, pad columns 34..39).  The tile never passes through VGPRs.
  const int c = (tid & 7) ^ ((tid >> 3) & 7), pb = tid >> 3;
  constexpr int NV = (HRW * HC + 31) / 32;     // 5
  const int hr0 = pb / HC, hc0 = pb - hr0 * HC;
  const int per_img = a.tiles_y * a.tiles_x;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto fetch = [&](int t, int buf) {
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * TR - 1, ox0 = (rem % a.tiles_x) * 32 - 1;
    int hr = hr0, hc = hc0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int iy = oy0 + hr, ix = ox0 + hc;
      const bool ok = hr < HRW && hc < 34 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)(smem + buf * X_BYTES + (32 * u + 8 * wave) * 128), 16,
                                               ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * 64 + c * 8) * 2) : OOB, 0, 0, 0);
      hc += 32;
      if (hc >= HC) {
        hc -= HC;
        ++hr;
      }
    }
  };

  const float slope = (a.flags & DSR_F_PRELU_PTR) ? a.prelu[0] : a.slope;
  const bool do_stats = (a.flags & DSR_F_STATS) != 0;
  float bias_v[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bias_v[nt] = (a.flags & DSR_F_BIAS) ? a.bias[c0 + wc * 32 + nt * 16 + r16] : 0.f;

  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tstep = gridDim.x;
  if (t < a.ntiles) fetch(t, 0);
  int buf = 0;
  bool first = true;
  for (; t < a.ntiles; t += tstep, buf ^= 1) {
    // my DMA of this tile is done; after the barrier everyone's is, and every wave is past the previous tile.
    // The previous tile's two output stores per thread (always issued: range-checked buffer stores) are younger than
    // this tile's DMA and stay in flight through the MFMA phase: vmcnt retires in order, so "at most 2 outstanding"
    // already means the DMA has landed.  A vmcnt(0) here exposed the full store latency once per 72-MFMA tile.
    if (first)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    first = false;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + tstep < a.ntiles) fetch(t + tstep, buf ^ 1);
    const unsigned char* sX = smem + buf * X_BYTES;

    f32x4 acc[2][2];                           // m-tile = half hx of the wave's row ; n-tile nt
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ty = a.tap_y[tp], tx = a.tap_x[tp];   // uniform, 0..2 (halo-relative)
      const unsigned char* rowp = sX + (wp + ty) * HC * 128;
      const int o0 = tx == 0 ? lds_off[0][0] : (tx == 1 ? lds_off[1][0] : lds_off[2][0]);
      const int o1 = tx == 0 ? lds_off[0][1] : (tx == 1 ? lds_off[1][1] : lds_off[2][1]);
      U4 fa[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[0][i] = *reinterpret_cast<const U4*>(rowp + i * 2048 + o0);
        fa[1][i] = *reinterpret_cast<const U4*>(rowp + i * 2048 + o1);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[i][nt] = mfma16<DT>(fa[kk][i], fw[tp][kk][nt], acc[i][nt]);
    }

    // ---- epilogue
    const int n = t / per_img;
    const int rem = t - n * per_img;
    const int oy0 = (rem / a.tiles_x) * TR, ox0 = (rem % a.tiles_x) * 32;
    // inference: eval-mode BatchNorm folded in.  The per-column scale / shift are fetched per tile (L1-resident) rather
    // than held for the life of the block: the training path has no registers to spare (249 of 256 VGPRs)
    [[maybe_unused]] float sc_v[2] = {1.f, 1.f}, sh_v[2] = {0.f, 0.f};
    if constexpr (FOLD) {
      if (a.flags & DSR_F_AFFINE) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          sc_v[nt] = a.scale[c0 + wc * 32 + nt * 16 + r16];
          sh_v[nt] = a.shift[c0 + wc * 32 + nt * 16 + r16];
        }
      }
    }
    auto epilogue = [&](auto actf) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = wc * 32 + nt * 16 + r16;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int oy = oy0 + wp;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int px = i * 16 + 4 * g + r;
            const float val = FOLD ? (acc[i][nt][r] + bias_v[nt]) * sc_v[nt] + sh_v[nt] : acc[i][nt][r] + bias_v[nt];
            const float vm = (oy < a.H && ox0 + px < a.W) ? val : 0.f;
            s1 += vm;
            s2 += vm * vm;
            const int prow = wp * 32 + px;                       // pixel index inside the tile
            *reinterpret_cast<unsigned short*>(sC + prow * C_STRIDE + col * 2) = f2h<DT>(actf(val));
          }
        }
        if (do_stats) {
          s1 += __shfl_xor(s1, 16, 64);
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 16, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (g == 0) {
            sStat[wp][0][col] = s1;
            sStat[wp][1][col] = s2;
          }
        }
      }
    };
    if (a.act == DSR_ACT_NONE)
      epilogue([](float x) { return x; });
    else if (a.act == DSR_ACT_RELU)
      epilogue([](float x) { return x > 0.f ? x : 0.f; });
    else if (a.act == DSR_ACT_LEAKY || a.act == DSR_ACT_PRELU)
      epilogue([slope](float x) { return x >= 0.f ? x : x * slope; });
    else
      epilogue([&](float x) { return act_apply(a.act, x, slope); });
    // raw barrier: a __syncthreads() here would also drain the next tile's DMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (do_stats && tid < 128) {
      const int which = tid >> 6, col = tid & 63;
      a.stats[((size_t)t * 2 + which) * a.CoutP + c0 + col] = sStat[0][which][col] + sStat[1][which][col];
    }
    if (!(a.flags & DSR_F_PIXSHUF)) {
#pragma unroll
      for (int it = 0; it < TR * 32 * 8 / 256; ++it) {          // exactly 2 stores per thread (see the wait above)
        const int idx = tid + it * 256;
        const int prow = idx >> 3, ch = idx & 7;
        const int oy = oy0 + (prow >> 5), ox = ox0 + (prow & 31);
        const bool ok = oy < a.H && ox < a.W;
        const size_t off = ok ? ((size_t)(n * a.H + oy) * a.W + ox) * a.CoutP + c0 + ch * 8 : 0;
        U4 v = *reinterpret_cast<const U4*>(sC + prow * C_STRIDE + ch * 16);
        if (FOLD && (a.flags & DSR_F_RESIDUAL)) {   // skip connection (generator.py:24,74): added after the activation
          float f[8], rr[8];
          unpack8<DT>(v, f);
          unpack8<DT>(*reinterpret_cast<const U4*>(reinterpret_cast<const unsigned short*>(a.res) + off), rr);
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] += rr[q];
          v = pack8<DT>(f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrsrc, ok ? (unsigned)(off * 2) : OOB, 0, 0);
      }
    } else {
      // PixelShuffle(2): conv channel 4c + 2i + j of pixel (h, w) -> channel c of pixel (2h+i, 2w+j); this slice's 64
      // conv channels are 16 output channels (two 8-channel vectors) of each of the 4 sub-pixels
      const int OCp = a.CoutP / 4;
#pragma unroll
      for (int it = 0; it < TR * 32 * 8 / 256; ++it) {          // exactly 2 stores per thread
        const int idx = tid + it * 256;
        const int prow = idx >> 3, sub = (idx >> 1) & 3, cq = idx & 1;
        const int oy = oy0 + (prow >> 5), ox = ox0 + (prow & 31);
        const bool ok = oy < a.H && ox < a.W;
        const unsigned short* src = reinterpret_cast<const unsigned short*>(sC + prow * C_STRIDE);
        unsigned short v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[4 * (cq * 8 + q) + sub];
        U4 o;
        o.x = v[0] | ((unsigned)v[1] << 16);
        o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16);
        o.w = v[6] | ((unsigned)v[7] << 16);
        const int py = 2 * oy + (sub >> 1), px = 2 * ox + (sub & 1);
        const size_t off = ((size_t)(n * 2 * a.H + py) * (2 * a.W) + px) * OCp + c0 / 4 + cq * 8;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, ok ? (unsigned)(off * 2) : OOB, 0, 0);
      }
    }
  }
}

// statistics rows written by one launch (= spatial tiles)
int dsr_c64_tiles(int N, int H, int W) {
  return N * ((H + 1) / 2) * ((W + 31) / 32);
}

void dsr_launch_conv_c64(C64Args& a, int N, int dtype, hipStream_t st) {
  a.tiles_y = (a.H + 1) / 2;
  a.tiles_x = (a.W + 31) / 32;
  a.ntiles = N * a.tiles_y * a.tiles_x;
  a.x_bytes = (unsigned)((size_t)N * a.H * a.W * 128);
  a.y_bytes = (unsigned)((size_t)N * a.H * a.W * a.CoutP * 2);   // (PixelShuffle: [N][2H][2W][CoutP/4] is the same size)
  const int slices = a.CoutP / 64;                            // blockIdx.y: 64-channel slice of the output
  const int per_slice = 512 / slices;                         // 2 resident blocks per CU over all slices
  dim3 grid(a.ntiles < per_slice ? a.ntiles : per_slice, slices), block(256);
  const bool fold = (a.flags & (DSR_F_AFFINE | DSR_F_RESIDUAL)) != 0;
  if (dtype == DSR_DTYPE_BF16) {
    if (fold) hipLaunchKernelGGL((conv_c64_kernel<DSR_DTYPE_BF16, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_c64_kernel<DSR_DTYPE_BF16, false>), grid, block, 0, st, a);
  } else {
    if (fold) hipLaunchKernelGGL((conv_c64_kernel<DSR_DTYPE_F16, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_c64_kernel<DSR_DTYPE_F16, false>), grid, block, 0, st, a);
  }
}
