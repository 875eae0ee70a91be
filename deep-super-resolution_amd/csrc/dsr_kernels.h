// Internal launch interfaces shared by the .hip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define DSR_MAX_TAPS 96
#define DSR_WGRAD_SCRATCH_SLABS 16

#define DSR_F_BIAS 1
#define DSR_F_STATS 2
#define DSR_F_PIXSHUF 4
#define DSR_F_OUT_NCHW_F32 8
#define DSR_F_PRELU_PTR 16
#define DSR_F_AFFINE 32      // (acc + bias) * scale[c] + shift[c] before the activation (eval-mode BatchNorm folded in)
#define DSR_F_RESIDUAL 64    // + residual tile after the activation
#define DSR_F_MASK 128       // (conv_c64, with DSR_F_RESIDUAL): the `res` tile is an activation output o: y = conv * act'(o) instead of conv + res

// q = floor(m / d) for 0 <= m < 2^31 via multiply-high (no integer division in kernels)
struct FastDiv {
  unsigned magic;
  unsigned shift;
};
static inline FastDiv fd_make(unsigned d) {
  FastDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
  return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int fd_div(const FastDiv f, int m) {
  return (int)((__umulhi((unsigned)m, f.magic) + (unsigned)m) >> f.shift);
}
#endif

struct ConvGemmArgs {
  const void* x;       // [N][IH][IW][CinP] 16-bit
  const void* w;       // [slices][NB][CinP] 16-bit
  void* y;             // [N][OH][OW][CoutP] 16-bit
  float* out_f32;      // optional NCHW fp32 output [N][cout][OH][OW]
  const float* bias;   // [cout]
  const float* prelu;  // 1 float
  float* stats;        // [tiles_m][2][stats_stride]
  unsigned x_bytes, w_bytes;   // buffer sizes for the hardware range check of the operand loads
  unsigned y_bytes;            // size of y (range-checked buffer stores of the persistent kernel)
  int M, GH, GW;
  int IH, IW, CinP;
  int OH, OW, CoutP;
  int NB;     // weight rows per tap slice (>= cout, multiple of 8)
  int cout;   // valid output columns
  int stats_stride;
  int isy, isx, osy, osx, ooy, oox;
  int ntaps, CU, U, ksteps;
  int pad_mode, act;
  float slope;
  int flags;
  int tiles_m, tiles_n;
  FastDiv fd_ghw, fd_gw, fd_cu, fd_cu8;
  // dsr_conv_dgrad_masked: the stored output is multiplied by act'(mask_x) (mask_x: the activation OUTPUT this gradient is
  // taken with respect to, same shape as y); dense-output launches of the one-tile-per-block kernel only
  const void* mask_x;
  int mask_act;
  float mask_slope;
  int taps[DSR_MAX_TAPS];   // (dy & 0xff) | (dx & 0xff) << 8 | widx << 16
};

void dsr_launch_conv_gemm(const ConvGemmArgs& a, int dtype, hipStream_t st);
// 256x256-tile variant of the gather kernel (conv_gemm.hip): LDS-DMA fast path, N a multiple of 256 and at least one
// tile per CU.  Shared by the dispatcher and by dsr_conv_kernel_name().
// DSR_CONV_BIG: 0 = 128x128 tiles only, 1 = the 256x128 three-stage experiment, 2 (default) = 256x256 where it applies.
static inline int dsr_conv_big_mode(void) {
  const char* e = getenv("DSR_CONV_BIG");            // read per call: tests switch it inside one process
  return e ? atoi(e) : 2;
}
static inline bool dsr_conv_gemm_use_256(long long M, int NB, bool fast, bool stats) {
  const long long tiles = ((M + 255) / 256) * ((NB + 255) / 256);
  (void)stats;   // both epilogues exist on the 256x256 tile (statistics from the channel-major accumulators)
  const char* t = getenv("DSR_CONV_BIG_TILES");      // tuning switch: fewest 256x256 tiles worth taking (default 150: a 196-tile launch on 256 CUs still beats four times as many 128x128 tiles, measured on VGG 512->512 at 28x28)
  return dsr_conv_big_mode() == 2 && fast && NB % 256 == 0 && tiles >= (t ? atoi(t) : 150);
}

// 224x256 tile (the same 8-wave kernel with 7 m-tiles per wave): a launch whose 256-row tiles leave most of the chip idle in
// their last round -- VGG19 at batch 32: 196 tiles (512 channels at 28x28) or 392 tiles (256 channels at 56x56) on the 256
// CUs of an MI355X, i.e. 77 % of the slots of one / two rounds -- runs 224 / 448 tiles of 7/8 the work each instead: the same
// number of rounds, every round 12.5 % shorter.  Taken when it lowers rounds x rows per tile, for launches without BatchNorm
// statistics (one statistics row per 128 tile rows has no place in a 224-row tile) and without PixelShuffle stores.
// DSR_CONV_BM224: 0 = never, 1 (default) = by that cost, 2 = wherever the 256x256 tile would be taken (tests).
static inline bool dsr_conv_gemm_use_224(long long M, int NB, bool fast, int flags) {
  const char* e = getenv("DSR_CONV_BM224");          // read per call: tests switch it inside one process
  const int mode = e ? atoi(e) : 1;
  if (mode == 0 || (flags & (DSR_F_STATS | DSR_F_PIXSHUF | DSR_F_OUT_NCHW_F32))) return false;
  if (!dsr_conv_gemm_use_256(M, NB, fast, false)) return false;
  if (mode == 2) return true;
  const long long cus = 256, nt = NB / 256;
  const long long r256 = (((M + 255) / 256) * nt + cus - 1) / cus, r224 = (((M + 223) / 224) * nt + cus - 1) / cus;
  return r224 * 7 < r256 * 8;
}

// 64x128 tile (4 waves of 64x32): launches whose 128x128 tiles do not even give every CU one block (VGG conv5_x at batch 32:
// M = 6,272 -> 49 x 4 = 196 blocks).  Without statistics only (a statistics row covers 128 tile rows).
// DSR_CONV_BM64: 0 = never, 1 (default) = fewer than 256 tiles of 128x128, 2 = every launch the 128x128 tile would take (tests).
static inline bool dsr_conv_gemm_use_64(long long M, int NB, bool fast, int flags) {
  const char* e = getenv("DSR_CONV_BM64");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0 || !fast || NB % 128 != 0 || (flags & (DSR_F_STATS | DSR_F_PIXSHUF | DSR_F_OUT_NCHW_F32))) return false;
  if (mode == 2) return true;
  return ((M + 127) / 128) * (NB / 128) < 256;
}

bool dsr_launch_conv_gemm_persist(const ConvGemmArgs& a, int dtype, hipStream_t st);   // conv_gemm_persist.hip

struct WgradArgs {
  const void* x;    // [N][IH][IW][CinP]
  const void* dy;   // [N][OH][OW][CoutP]
  float* partial;   // [splits][ntaps][CoutP][CinP] fp32
  int M, OH, OW, IH, IW, CinP, CoutP;
  int stride, pad, pad_mode, KH, KW;
  int tiles_co, tiles_ci, splits, chunk;   // chunk = pixels per split (multiple of 128)
  FastDiv fd_ohw, fd_ow;
};
void dsr_launch_wgrad(const WgradArgs& a, int dtype, hipStream_t st);

// tile-resident weight gradient (3x3 stride 1|2 and 1x1): all taps from one staged halo tile
struct WgradTileArgs {
  const void* x;
  const void* dy;
  float* partial;   // [ychunks][ntaps][CoutP][CinP]
  int N, OH, OW, IH, IW, CinP, CoutP;
  int pad, pad_mode;
  int tiles_co, tiles_ci;
  int tiles_y, tiles_x, ntiles, tiles_per_block;
  unsigned x_bytes, dy_bytes;
};
// grouped launch of the 3x3 / stride-1 kernel: problems by value in the kernel-argument segment (4 KB limit)
#define DSR_WGRAD_BATCH_MAX 36
struct WgradBatchArgs {
  WgradTileArgs p[DSR_WGRAD_BATCH_MAX];
  int first_block[DSR_WGRAD_BATCH_MAX + 1];
  int count;
};
static_assert(sizeof(WgradBatchArgs) <= 4096, "kernel arguments are limited to 4 KB");
void dsr_launch_wgrad_dma_batch(const WgradBatchArgs& b, int dtype, hipStream_t st);
// one reduction launch for a batch: entry e sums `splits` consecutive slabs into dw (PyTorch [co][ci][kh][kw] layout)
struct WgradReduceBatchArgs {
  struct Entry {
    const float* partial;
    float* dw;
    int splits, Cout, Cin, CoutP, CinP, first_block;
  } e[DSR_WGRAD_BATCH_MAX];
  int count, total_blocks;
};
void dsr_launch_wgrad_reduce_batch(const WgradReduceBatchArgs& r, hipStream_t st);
// returns the number of partial slabs (ychunks) the launch will write, 0 if the shape is not handled
int dsr_wgrad_tile_plan(int KH, int KW, int stride, int N, int OH, int OW, int CinP, int CoutP, WgradTileArgs* a);
void dsr_launch_wgrad_tile(const WgradTileArgs& a, int KH, int stride, int ychunks, int dtype, hipStream_t st);

// 9x9 stride-1 "same" convolution with <= 3 output channels and 64 input channels (conv_smalln.hip)
int dsr_wgrad_toeplitz_plan(int N, int H, int W, WgradTileArgs* a);
void dsr_launch_wgrad_toeplitz(const WgradTileArgs& a, int dtype, hipStream_t st);
void dsr_launch_dgrad_toeplitz(const void* dy, const void* w_dgrad, void* dx, int N, int H, int W, int dtype, hipStream_t st,
                               const void* act_out = nullptr, const float* prelu = nullptr, void* dyu = nullptr,
                               float* ps_partial = nullptr);      // act_out != nullptr: the PixelShuffle-PReLU backward in the epilogue
int dsr_dgrad_toeplitz_ps_blocks(int N, int H, int W);
int dsr_wgrad_taps_plan(int KH, int KW, int stride, int N, int OH, int OW, int CinP, int CoutP, WgradTileArgs* a);
void dsr_launch_wgrad_taps(const WgradTileArgs& a, int KH, int ychunks, int dtype, hipStream_t st);

// forward conv with <= 16 output channels from an LDS-resident halo tile (conv_smalln.hip)
struct SmallNArgs {
  const void* x;
  const void* w;      // [taps][NB][CinP]
  void* y;            // NHWC 16-bit [N][OH][OW][CoutP] (or null)
  float* out_f32;     // NCHW fp32 (or null)
  const float* bias;
  const float* prelu;
  int IH, IW, CinP, OH, OW, CoutP, NB, cout, KH, KW, pad, act;
  float slope;
  int tiles_y, tiles_x, ntiles;
  int flip;           // read weight slice ntaps-1-t for tap t (input gradient = correlation with the mirrored kernel)
  unsigned x_bytes;
};
int dsr_launch_conv_smalln(SmallNArgs& a, int N, int dtype, hipStream_t st);   // returns 0 if the halo does not fit

// 64 -> 64 channel 3x3 stride-1 zero-pad conv, weights register-resident, persistent (conv_c64.hip)
struct C64Args {
  const void* x;       // [N][H][W][64]
  const void* w;       // [9][64 rows][64]  (rows = output channels of this GEMM)
  void* y;             // [N][H][W][CoutP]  (PixelShuffle: [N][2H][2W][CoutP/4])
  const float* bias;
  const float* prelu;
  float* stats;        // [ntiles][2][CoutP]
  const float* scale;  // DSR_F_AFFINE: [CoutP]
  const float* shift;
  const void* res;     // DSR_F_RESIDUAL: [N][H][W][CoutP]
  int H, W;
  int CoutP;           // 64 * slices (one 64-channel slice per blockIdx.y; weight image [9][CoutP][64])
  int act;
  float slope;
  int flags;
  int mask_act;        // DSR_F_MASK: activation whose derivative (from its output `res`) multiplies the result
  float mask_slope;
  int tiles_y, tiles_x, ntiles;
  int tap_y[9], tap_x[9];   // halo-relative row / column offset (0..2) of weight slice t
  unsigned x_bytes, y_bytes;
};
int dsr_c64_tiles(int N, int H, int W, int tile_rows);
int dsr_c64_stat_rows(int N, int H, int W, int CoutP);
void dsr_launch_conv_c64(C64Args& a, int N, int dtype, hipStream_t st);

struct Cin8Args {
  const void* x;       // [N][H][W][8]
  const void* w;       // forward weight image [9][64][8]
  void* y;             // [N][H][W][64]
  const float* bias;
  const float* prelu;
  int H, W;
  int act;
  float slope;
  int tiles_y, tiles_x, ntiles;
  unsigned x_bytes;
  int nt_store;        // output beyond the Infinity Cache: nontemporal stores (set by the launcher)
};
void dsr_launch_conv_cin8(Cin8Args& a, int N, int dtype, hipStream_t st);
// the generator's 9x9 RGB -> 64 head, one kernel row per MFMA k-step (conv_rgb9.hip); same arguments
int dsr_conv_rgb9_supported(int KH, int KW, int stride, int pad, int pad_mode, int Cin, int Cout);
void dsr_launch_conv_rgb9(Cin8Args& a, int N, int dtype, hipStream_t st);
struct Rgb9WgradArgs {
  const void* x;       // [N][H][W][8]
  const void* dy;      // [N][H][W][64]
  float* partial;      // dsr_wgrad_rgb9_blocks() slabs of dsr_wgrad_rgb9_slab_floats() floats
  int H, W;
  int tiles_y, tiles_x, ntiles;      // filled by the launcher
  unsigned x_bytes, dy_bytes;
};
int dsr_wgrad_rgb9_blocks(int N, int H, int W);
size_t dsr_wgrad_rgb9_slab_floats();
void dsr_launch_wgrad_rgb9(Rgb9WgradArgs& a, int N, int Cin, float* dw, int dtype, hipStream_t st);   // kernel + slab reduction -> dw [64][Cin][9][9]

// input gradient of a 3x3 stride-2 pad-1 convolution in one launch (conv_dgrad_s2.hip)
struct DgradS2Args {
  const void* dy;     // [N][H/2][W/2][CoutP]
  const void* w;      // dgrad weight image [9][CinP][CoutP]
  void* dx;           // [N][H][W][CinP]
  int H, W, CinP, CoutP;
  int OH, OW, M, ci_blocks, kblocks, tiles_m;      // filled by the launcher
  unsigned dy_bytes, w_bytes, dx_bytes;
  FastDiv fd_ghw, fd_gw;
  // the image layer's backward in the epilogue (dsr_conv_dgrad_first_bwd; img == nullptr: plain input gradient)
  const void* img;        // [N][H][W][8] the image layer's input
  const float* w0;        // its weights [64][Cin0][3][3], fp32
  const float* b0;        // its bias [64] or nullptr
  float* fb_partial;      // [2 blocks][64][32]
  unsigned img_bytes;
  int Cin0, act0;
  float slope0;
  // BatchNorm-backward sums in the epilogue (dsr_conv_dgrad_bn; bn_y == nullptr: none)
  const void* bn_y;       // [N][H][W][CinP] raw conv output of the layer whose BatchNorm + activation output dx is the gradient of
  const float* bn_scale;  // [CinP] its affine map (gamma * rstd, beta - mean * gamma * rstd)
  const float* bn_shift;
  float* bn_partial;      // [blocks][3][CinP]: sum g, sum g*y, 0
  int bn_act;
  float bn_slope;
};
int dsr_dgrad_s2_blocks(int N, int H, int W, int CinP);
bool dsr_dgrad_s2_bn_supported(int KH, int KW, int stride, int pad, int pad_mode, int H, int W, int CinP, int CoutP, int N);
bool dsr_dgrad_s2_supported(int KH, int KW, int stride, int pad, int pad_mode, int H, int W, int CinP, int CoutP, int N);
void dsr_launch_dgrad_s2(DgradS2Args& a, int N, int dtype, hipStream_t st);

// 3x3 stride-1 pad-1 conv with 64 output channels and CinP = 32 k input channels, halo-staged per 32-channel K-block
// (conv_halo64.hip)
struct Halo64Args {
  const void* x;      // [N][H][W][CinP]
  const void* w;      // [9][64 rows = output channels][CinP]
  void* y;            // [N][H][W][64]
  const float* bias;  // [64] or null
  int H, W, CinP;
  int mirror;         // 1: weight slice t sits at the mirrored halo offset (input gradient on the [tap][ci][co] image)
  int act;
  float slope;
  int flags;          // DSR_F_BIAS
  const void* mask_x; // optional: y *= mask_act'(mask_x), mask_x an activation output of y's shape (dsr_conv_dgrad_masked)
  int mask_act;
  float mask_slope;
  int cout_full;      // output channels of the layer: 64, or 128 = two slices of 64 (0 is taken as 64)
  int kblocks, tiles_y, tiles_x, ntiles;      // filled by the launcher
  unsigned w_bytes, y_bytes;
};
bool dsr_halo64_supported(int KH, int KW, int stride, int pad, int pad_mode, int H, int W, int CinP, int CoutP);
void dsr_launch_conv_halo64(Halo64Args& a, int N, int dtype, hipStream_t st);

// the discriminator's first two convolutions in one forward kernel (conv_first2.hip)
struct First2Args {
  const void* x;      // [N][H][W][8]
  const void* w0;     // [9][64][8]   first layer, forward image
  const float* b0;
  float slope0;
  const void* w1;     // [9][64][64]  second layer, forward image
  const float* b1;
  void* a0;           // [N][H][W][64] or null
  void* y1;           // [N][OH][OW][64]
  float* stats;       // [blocks][2][64] or null
  int H, W, OH, OW, tiles_y, tiles_x, ntiles;
  unsigned a0_bytes, y1_bytes;
};

// fused backward of a first layer (conv_first_bwd.hip)
struct FirstBwdArgs {
  const void* x;       // [N][H][W][8]
  const void* dout;    // [N][H][W][64] gradient w.r.t. the activation output
  const void* y;       // [N][H][W][64] activation output
  float* partial;      // [blocks][64][32]
  int N, H, W, act;
  float slope;
  int tiles_y, tiles_x, ntiles;
  unsigned x_bytes, y_bytes;
  const float* w0;     // recompute form: the layer's fp32 OIHW weights [64][Cin][3][3] and bias (nullable); y is not read
  const float* b0;
  int Cin;
};

void dsr_launch_wgrad_reduce(const float* partial, float* dw, int splits, int ntaps, int Cout, int Cin, int CoutP,
                             int CinP, hipStream_t st);

// Opt-in for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize): a property of (kernel, DEVICE), so
// the "already done" flag is one bit per device ordinal, and the launchers may be entered from the main and the autograd
// thread at once, so it is atomic.  One static LdsOptIn per kernel instantiation; setting the attribute twice is harmless.
#include <atomic>
struct LdsOptIn {
  std::atomic<unsigned long long> done{0};
  void ensure(const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      done.fetch_or(bit, std::memory_order_release);
    }
  }
};

// records hipGetLastError() under `what`; returns 0 or a negative code (see dsr_last_error()).
int dsr_launch_status(const char* what);
int dsr_fail(int code, const char* fmt, ...);

// Argument validation of the C-ABI entry points: a bad argument returns DSR_E_ARG (see dsr_last_error()) before
// anything is launched -- a null device pointer would otherwise become a GPU memory fault, which the runtime turns
// into an abort of the whole process.
#define DSR_REQUIRE(cond, ...)                          \
  do {                                                  \
    if (!(cond)) return dsr_fail(DSR_E_ARG, __VA_ARGS__); \
  } while (0)
#define DSR_DTYPE_OK(dt) ((dt) == DSR_DTYPE_BF16 || (dt) == DSR_DTYPE_F16)
// channel-chunked pointwise kernels: one thread per 8-channel chunk, 256 / (Cp / 8) pixel rows per block iteration
#define DSR_CP_OK(Cp) ((Cp) >= 8 && (Cp) % 8 == 0 && (Cp) <= 2048)
