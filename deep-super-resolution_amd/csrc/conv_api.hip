// C-ABI entry points for convolution forward / input-grad / weight-grad (include/dsr_hip.h).
// Host code only: turns an nn.Conv2d-style descriptor into the tap table + grid mapping the
// gather-GEMM kernel (conv_gemm.hip) consumes.  No allocation, no synchronisation: capturable.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include "../../include/dsr_hip.h"
#include "dsr_common.h"
#include "dsr_kernels.h"

static thread_local char g_err[512] = "";

int dsr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int dsr_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dsr_fail(DSR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DSR_OK;
}
extern "C" const char* dsr_last_error(void) { return g_err; }
extern "C" int dsr_abi_version(void) { return 7; }

static inline int r8(int c) { return (c + 7) & ~7; }

static int check_desc(const dsr_conv_desc* d) {
  if (!d) return dsr_fail(DSR_E_ARG, "conv: null descriptor");
  if (d->dtype != DSR_BF16 && d->dtype != DSR_F16) return dsr_fail(DSR_E_ARG, "conv: dtype %d", d->dtype);
  if (d->N < 1 || d->H < 1 || d->W < 1 || d->Cin < 1 || d->Cout < 1) return dsr_fail(DSR_E_ARG, "conv: empty shape");
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->pad < 0) return dsr_fail(DSR_E_ARG, "conv: bad kernel/stride/pad");
  if (d->KH * d->KW > DSR_MAX_TAPS) return dsr_fail(DSR_E_UNSUPPORTED, "conv: %dx%d taps > %d", d->KH, d->KW, DSR_MAX_TAPS);
  if (d->pad_mode < 0 || d->pad_mode > 2) return dsr_fail(DSR_E_ARG, "conv: pad_mode %d", d->pad_mode);
  if (d->pad_mode == DSR_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W))
    return dsr_fail(DSR_E_ARG, "conv: reflect pad %d >= input size", d->pad);   // torch raises the same
  if (d->KH > 127 || d->KW > 127) return dsr_fail(DSR_E_UNSUPPORTED, "conv: kernel too large");
  int OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1, OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  if (OH < 1 || OW < 1) return dsr_fail(DSR_E_ARG, "conv: output would be empty");
  {
    // operands are addressed with 32-bit byte offsets (buffer loads): every tensor must stay below 2 GiB
    const long long in_b = (long long)d->N * d->H * d->W * r8(d->Cin) * 2, out_b = (long long)d->N * OH * OW * r8(d->Cout) * 2;
    if (in_b >= (1ll << 31) || out_b >= (1ll << 31)) return dsr_fail(DSR_E_UNSUPPORTED, "conv: tensor of 2 GiB or more");
  }
  return DSR_OK;
}

extern "C" int dsr_conv_out_size(const dsr_conv_desc* d, int* OH, int* OW) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!OH || !OW) return dsr_fail(DSR_E_ARG, "conv_out_size: null output pointer");
  *OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  *OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  return DSR_OK;
}

static bool is_cin8(const dsr_conv_desc* d, const dsr_epilogue* e) {   // first layer: RGB (padded to 8) -> 64, 3x3 s1 p1
  return d->Cin <= 8 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == DSR_PAD_ZERO && !(e && (e->stats_partial || e->pixel_shuffle || e->out_nchw_f32));
}
static bool is_rgb9(const dsr_conv_desc* d, const dsr_epilogue* e) {   // the generator's head: RGB -> 64, 9x9 s1 p4 (generator.py:48)
  return dsr_conv_rgb9_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->Cin, d->Cout) &&
         !(e && (e->stats_partial || e->pixel_shuffle || e->out_nchw_f32 || e->bn_scale || e->residual));
}
// weight gradient of the generator's 9x9 RGB head on conv_rgb9_wgrad_kernel (DSR_CONV_RGB9=0: the tap-per-MFMA kernel)
static bool is_rgb9_wgrad(const dsr_conv_desc* d) {
  return dsr_conv_rgb9_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->Cin, d->Cout) != 0;
}

static bool is_tail9(const dsr_conv_desc* d) {
  return d->KH == 9 && d->KW == 9 && d->stride == 1 && d->pad == 4 && d->pad_mode == DSR_PAD_ZERO && d->Cout <= 3 &&
         r8(d->Cin) == 64;
}
// Cin = 64 forward convs with Cout a multiple of 64 (PixelShuffle convs 64->256, D's 64->128, VGG 64->128): their K
// loop is 9 steps, where the gather kernel's per-tile prologue/epilogue costs as much as the loop; the weights-in-
// registers kernel runs one 64-channel output slice per block row instead
static bool is_c64_wide(const dsr_conv_desc* d) {
  static const bool on = [] { const char* e = getenv("DSR_C64_WIDE"); return !(e && e[0] == '0'); }();
  return on && d->Cin == 64 && d->Cout > 64 && d->Cout % 64 == 0 && d->Cout <= 256 && d->KH == 3 && d->KW == 3 &&
         d->stride == 1 && d->pad == 1 && d->pad_mode == DSR_PAD_ZERO;
}
static bool is_smalln_dgrad(const dsr_conv_desc* d) {
  return d->Cin <= 16 && r8(d->Cout) == 64 && d->stride == 1 && d->pad_mode == DSR_PAD_ZERO && d->KH == d->KW &&
         (d->KW == 3 || d->KW == 9) && 2 * d->pad == d->KH - 1;
}
static bool is_c64(const dsr_conv_desc* d) {
  return d->Cin == 64 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == DSR_PAD_ZERO;
}

extern "C" int dsr_conv_fwd_affine_supported(const dsr_conv_desc* d) {
  return d && !check_desc(d) && (is_c64(d) || is_c64_wide(d));
}

extern "C" int dsr_conv_stats_rows(const dsr_conv_desc* d) {
  int OH, OW;
  if (dsr_conv_out_size(d, &OH, &OW)) return -1;
  if (is_c64(d) || is_c64_wide(d)) return dsr_c64_stat_rows(d->N, OH, OW, r8(d->Cout));   // one statistics row per persistent block
  long long M = (long long)d->N * OH * OW;
  return (int)((M + 127) / 128);
}

extern "C" size_t dsr_conv_packed_elems(const dsr_conv_desc* d, int dgrad) {
  if (check_desc(d)) return 0;
  size_t T = (size_t)d->KH * d->KW;
  return dgrad ? T * r8(d->Cin) * r8(d->Cout) : T * r8(d->Cout) * r8(d->Cin);
}

extern "C" int dsr_conv_pack_weight(const dsr_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad,
                                    dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!w || !w_fwd) return dsr_fail(DSR_E_ARG, "pack: null pointer");
  return dsr_pw_pack_weight(d->dtype, w, w_fwd, w_dgrad, d->Cout, d->Cin, d->KH * d->KW, r8(d->Cout), r8(d->Cin),
                            r8(d->Cin), r8(d->Cout), s);
}

static inline int pack_tap(int dy, int dx, int widx) { return (dy & 0xff) | ((dx & 0xff) << 8) | (widx << 16); }

static void finish_args(ConvGemmArgs& a, int N, int wslices) {
  a.x_bytes = (unsigned)((size_t)N * a.IH * a.IW * a.CinP * 2);
  a.w_bytes = (unsigned)((size_t)wslices * a.NB * a.CinP * 2);
  a.y_bytes = (unsigned)((size_t)N * a.OH * a.OW * a.CoutP * 2);
  a.CU = a.CinP / 8;
  a.U = a.ntaps * a.CU;
  a.ksteps = (a.U + 7) / 8;
  a.fd_ghw = fd_make((unsigned)(a.GH * a.GW));
  a.fd_gw = fd_make((unsigned)a.GW);
  a.fd_cu = fd_make((unsigned)a.CU);
  a.fd_cu8 = fd_make((unsigned)((a.CU >> 3) > 0 ? (a.CU >> 3) : 1));
}

extern "C" int dsr_conv_fwd(const dsr_conv_desc* d, const void* x, const void* w_fwd, const dsr_epilogue* e, void* y,
                            dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !w_fwd || !e) return dsr_fail(DSR_E_ARG, "conv_fwd: null pointer");
  if (!y && !e->out_nchw_f32) return dsr_fail(DSR_E_ARG, "conv_fwd: no output");
  int OH, OW;
  dsr_conv_out_size(d, &OH, &OW);
  ConvGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x;
  a.w = w_fwd;
  a.y = y;
  a.out_f32 = e->out_nchw_f32;
  a.bias = e->bias;
  a.prelu = e->prelu;
  a.stats = e->stats_partial;
  a.GH = OH;
  a.GW = OW;
  a.M = d->N * OH * OW;
  a.IH = d->H;
  a.IW = d->W;
  a.CinP = r8(d->Cin);
  a.NB = r8(d->Cout);
  a.cout = d->Cout;
  a.stats_stride = r8(d->Cout);
  a.isy = a.isx = d->stride;
  a.osy = a.osx = 1;
  a.pad_mode = d->pad_mode;
  a.act = e->act;
  a.slope = e->slope;
  a.flags = (e->bias ? DSR_F_BIAS : 0) | (e->stats_partial ? DSR_F_STATS : 0) |
            (e->out_nchw_f32 ? DSR_F_OUT_NCHW_F32 : 0) | ((e->act == DSR_ACT_PRELU && e->prelu) ? DSR_F_PRELU_PTR : 0);
  if (e->act == DSR_ACT_PRELU && !e->prelu) return dsr_fail(DSR_E_ARG, "conv_fwd: PReLU needs its weight pointer");
  if (e->pixel_shuffle) {
    if (d->Cout % 4 || d->Cout < 32 || e->out_nchw_f32 || e->stats_partial)
      return dsr_fail(DSR_E_UNSUPPORTED, "conv_fwd: pixel-shuffle epilogue needs Cout %% 4 == 0, Cout >= 32, 16-bit output");
    a.flags |= DSR_F_PIXSHUF;
    a.OH = 2 * OH;
    a.OW = 2 * OW;
    a.CoutP = r8(d->Cout / 4);
  } else {
    a.OH = OH;
    a.OW = OW;
    a.CoutP = r8(d->Cout);
  }
  const bool fold = e->bn_scale || e->bn_shift || e->residual;
  if (fold && (!dsr_conv_fwd_affine_supported(d) || e->pixel_shuffle || e->out_nchw_f32 || e->stats_partial ||
               (!e->bn_scale) != (!e->bn_shift)))
    return dsr_fail(DSR_E_UNSUPPORTED, "conv_fwd: folded BatchNorm / residual epilogue not available for this layer");
  if (!fold && !e->pixel_shuffle && !e->out_nchw_f32 && !e->stats_partial && d->Cout % 64 == 0 && d->Cout >= 128 &&
      (e->act == DSR_ACT_NONE || e->act == DSR_ACT_RELU || e->act == DSR_ACT_LEAKY) &&
      dsr_halo64_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cin), d->Cout)) {
    // 128 outputs over >= 128 input channels without BatchNorm statistics (VGG conv2_2, utils/GAN.py:26): two 64-channel
    // slices per spatial tile on the halo-staged kernel (conv_halo64.hip) instead of the gather kernel's 128x128 tile
    Halo64Args q;
    memset(&q, 0, sizeof(q));
    q.x = x;
    q.w = w_fwd;
    q.y = y;
    q.bias = e->bias;
    q.H = d->H;
    q.W = d->W;
    q.CinP = r8(d->Cin);
    q.cout_full = d->Cout;
    q.mirror = 0;
    q.act = e->act;
    q.slope = e->slope;
    q.flags = e->bias ? DSR_F_BIAS : 0;
    dsr_launch_conv_halo64(q, d->N, d->dtype, s);
    return dsr_launch_status("dsr_conv_fwd(halo64)");
  }
  if (((is_c64(d) && !e->pixel_shuffle) || is_c64_wide(d)) && !e->out_nchw_f32) {
    C64Args c;
    memset(&c, 0, sizeof(c));
    c.CoutP = d->Cout;
    c.scale = e->bn_scale;
    c.shift = e->bn_shift;
    c.res = e->residual;
    c.x = x;
    c.w = w_fwd;
    c.y = y;
    c.bias = e->bias;
    c.prelu = e->prelu;
    c.stats = e->stats_partial;
    c.H = d->H;
    c.W = d->W;
    c.act = e->act;
    c.slope = e->slope;
    c.flags = a.flags | (e->bn_scale ? DSR_F_AFFINE : 0) | (e->residual ? DSR_F_RESIDUAL : 0);
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        c.tap_y[kh * 3 + kw] = kh;
        c.tap_x[kh * 3 + kw] = kw;
      }
    dsr_launch_conv_c64(c, d->N, d->dtype, s);
    return dsr_launch_status("dsr_conv_fwd(c64)");
  }
  if (is_cin8(d, e) || is_rgb9(d, e)) {
    Cin8Args c;
    memset(&c, 0, sizeof(c));
    c.x = x;
    c.w = w_fwd;
    c.y = y;
    c.bias = e->bias;
    c.prelu = (e->act == DSR_ACT_PRELU) ? e->prelu : nullptr;
    c.H = d->H;
    c.W = d->W;
    c.act = e->act;
    c.slope = e->slope;
    if (is_rgb9(d, e)) {
      dsr_launch_conv_rgb9(c, d->N, d->dtype, s);
      return dsr_launch_status("dsr_conv_fwd(rgb9)");
    }
    dsr_launch_conv_cin8(c, d->N, d->dtype, s);
    return dsr_launch_status("dsr_conv_fwd(cin8)");
  }
  if (d->Cout <= 16 && d->stride == 1 && d->pad_mode == DSR_PAD_ZERO && d->KH * d->KW >= 9 && !e->stats_partial &&
      !e->pixel_shuffle) {
    // few output channels + many taps: stage the input halo once instead of gathering it once per tap
    SmallNArgs sn;
    memset(&sn, 0, sizeof(sn));
    sn.x = x;
    sn.w = w_fwd;
    sn.y = y;
    sn.out_f32 = e->out_nchw_f32;
    sn.bias = e->bias;
    sn.prelu = (e->act == DSR_ACT_PRELU) ? e->prelu : nullptr;
    sn.IH = d->H;
    sn.IW = d->W;
    sn.CinP = r8(d->Cin);
    sn.OH = OH;
    sn.OW = OW;
    sn.CoutP = r8(d->Cout);
    sn.NB = r8(d->Cout);
    sn.cout = d->Cout;
    sn.KH = d->KH;
    sn.KW = d->KW;
    sn.pad = d->pad;
    sn.act = e->act;
    sn.slope = e->slope;
    if (dsr_launch_conv_smalln(sn, d->N, d->dtype, s)) return dsr_launch_status("dsr_conv_fwd(small-n)");
  }
  a.ntaps = d->KH * d->KW;
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) a.taps[kh * d->KW + kw] = pack_tap(kh - d->pad, kw - d->pad, kh * d->KW + kw);
  finish_args(a, d->N, d->KH * d->KW);
  dsr_launch_conv_gemm(a, d->dtype, s);
  return dsr_launch_status("dsr_conv_fwd");
}

// ---- reflect-padding adjoint: dx[i][j] = sum of dxp over the padded coordinates that mirror onto (i,j)
template <int DT>
__global__ void reflect_fold_kernel(const unsigned short* __restrict__ dxp, unsigned short* __restrict__ dx, int N,
                                    int H, int W, int Cp, int p) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = Cp / 8;
  size_t total = (size_t)N * H * W * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t pix = idx / cpr;
  int j = (int)(pix % W);
  int i = (int)((pix / W) % H);
  int n = (int)(pix / ((size_t)W * H));
  const int HP = H + 2 * p, WP = W + 2 * p;
  int ys[3], xs[3], ny = 0, nx = 0;
  ys[ny++] = i + p;
  if (i >= 1 && i <= p) ys[ny++] = p - i;
  if (i <= H - 2 && i >= H - 1 - p) ys[ny++] = 2 * (H - 1) - i + p;
  xs[nx++] = j + p;
  if (j >= 1 && j <= p) xs[nx++] = p - j;
  if (j <= W - 2 && j >= W - 1 - p) xs[nx++] = 2 * (W - 1) - j + p;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b) {
      float f[8];
      unpack8<DT>(*reinterpret_cast<const U4*>(dxp + ((size_t)(n * HP + ys[a]) * WP + xs[b]) * Cp + ch * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
  *reinterpret_cast<U4*>(dx + pix * Cp + ch * 8) = pack8<DT>(acc);
}

static void launch_c64_dgrad(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, void* dx,
                             dsr_stream_t s, int mask_act = DSR_ACT_NONE, float mask_slope = 0.f) {
  C64Args c;
  memset(&c, 0, sizeof(c));
  c.CoutP = 64;
  c.x = dy;
  c.w = w_dgrad;
  c.y = dx;
  c.res = addend;                       // dx = dgrad(dy) + addend: the skip path's gradient rides in the epilogue
  c.flags = addend ? DSR_F_RESIDUAL : 0;
  if (addend && mask_act != DSR_ACT_NONE) {   // ... or dx = dgrad(dy) * act'(addend): the tile is an activation output
    c.flags |= DSR_F_MASK;
    c.mask_act = mask_act;
    c.mask_slope = mask_slope;
  }
  c.H = d->H;
  c.W = d->W;
  c.act = DSR_ACT_NONE;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      c.tap_y[kh * 3 + kw] = 2 - kh;
      c.tap_x[kh * 3 + kw] = 2 - kw;
    }
  dsr_launch_conv_c64(c, d->N, d->dtype, s);
}

extern "C" size_t dsr_conv_dgrad_workspace(const dsr_conv_desc* d) {
  if (check_desc(d)) return 0;
  if (d->pad_mode == DSR_PAD_ZERO || d->pad == 0) return 0;
  return (size_t)d->N * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * r8(d->Cin) * 2;
}

static bool dgrad_mask_supported(const dsr_conv_desc* d);

// mask_x != null: dx = dgrad(dy) * act'(mask_x) (dsr_conv_dgrad_masked; the caller has checked dgrad_mask_supported)
static int conv_dgrad_impl(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, void* workspace,
                           size_t ws_bytes, const void* mask_x, int mask_act, float mask_slope, dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !w_dgrad || !dx) return dsr_fail(DSR_E_ARG, "conv_dgrad: null pointer");
  if (d->pad_mode == DSR_PAD_REPLICATE && d->pad > 0)
    return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad: replicate padding has no consumer on the hot path");
  int OH, OW;
  dsr_conv_out_size(d, &OH, &OW);
  const bool folded = d->pad_mode == DSR_PAD_REFLECT && d->pad > 0;
  // with reflect padding the gradient is first taken w.r.t. the PADDED input (pad = 0 problem), then folded
  const int H = folded ? d->H + 2 * d->pad : d->H;
  const int W = folded ? d->W + 2 * d->pad : d->W;
  const int pad = folded ? 0 : d->pad;
  void* target = dx;
  if (folded) {
    size_t need = dsr_conv_dgrad_workspace(d);
    if (!workspace || ws_bytes < need) return dsr_fail(DSR_E_WORKSPACE, "conv_dgrad: workspace %zu < %zu", ws_bytes, need);
    target = workspace;
  }
  if (is_c64(d)) {   // mirrored taps on the [tap][ci][co] weight image
    launch_c64_dgrad(d, dy, w_dgrad, mask_x, dx, s, mask_act, mask_slope);
    return dsr_launch_status("dsr_conv_dgrad(c64)");
  }
  if (is_tail9(d)) {   // the generator's 9x9 64->3 tail: Toeplitz K = (kw, co) mapping (conv_smalln.hip)
    dsr_launch_dgrad_toeplitz(dy, w_dgrad, dx, d->N, d->H, d->W, d->dtype, s);
    return dsr_launch_status("dsr_conv_dgrad(toeplitz)");
  }
  if (is_smalln_dgrad(d)) {
    // few input channels (the RGB first layers, discriminator.py:22): dx = dy correlated with the mirrored kernel,
    // a stride-1 "forward" problem with Cin output channels -> the halo-staged small-N kernel
    SmallNArgs sn;
    memset(&sn, 0, sizeof(sn));
    sn.x = dy;
    sn.w = w_dgrad;
    sn.y = dx;
    sn.IH = OH;
    sn.IW = OW;
    sn.CinP = r8(d->Cout);
    sn.OH = d->H;
    sn.OW = d->W;
    sn.CoutP = r8(d->Cin);
    sn.NB = r8(d->Cin);
    sn.cout = d->Cin;
    sn.KH = d->KH;
    sn.KW = d->KW;
    sn.pad = d->KH - 1 - d->pad;
    sn.act = DSR_ACT_NONE;
    sn.flip = 1;
    if (dsr_launch_conv_smalln(sn, d->N, d->dtype, s)) return dsr_launch_status("dsr_conv_dgrad(small-n)");
  }
  if (!folded && dsr_halo64_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cout), r8(d->Cin))) {
    // 64 input channels of a layer with 128 / 256 outputs (PixelShuffle convs, D's 64 -> 128, VGG conv2_1): the gradient is a
    // 64-output convolution over many channels -- halo staged per 32-channel K-block (conv_halo64.hip)
    Halo64Args q;
    memset(&q, 0, sizeof(q));
    q.x = dy;
    q.w = w_dgrad;
    q.y = dx;
    q.H = d->H;
    q.W = d->W;
    q.CinP = r8(d->Cout);
    q.cout_full = r8(d->Cin);
    q.mirror = 1;
    q.act = DSR_ACT_NONE;
    q.mask_x = mask_x;
    q.mask_act = mask_act;
    q.mask_slope = mask_slope;
    dsr_launch_conv_halo64(q, d->N, d->dtype, s);
    return dsr_launch_status("dsr_conv_dgrad(halo64)");
  }
  if (!folded && dsr_dgrad_s2_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cin), r8(d->Cout), d->N)) {
    // 3x3 stride 2 (discriminator.py:29-35): all four output-parity classes from one staged dY tile, one launch
    DgradS2Args q;
    memset(&q, 0, sizeof(q));
    q.dy = dy;
    q.w = w_dgrad;
    q.dx = dx;
    q.H = d->H;
    q.W = d->W;
    q.CinP = r8(d->Cin);
    q.CoutP = r8(d->Cout);
    dsr_launch_dgrad_s2(q, d->N, d->dtype, s);
    return dsr_launch_status("dsr_conv_dgrad(s2)");
  }
  const int st = d->stride;
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      int GH = (H - ph + st - 1) / st, GW = (W - pw + st - 1) / st;
      if (GH <= 0 || GW <= 0) continue;
      ConvGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.x = dy;
      a.w = w_dgrad;
      a.y = target;
      a.GH = GH;
      a.GW = GW;
      a.M = d->N * GH * GW;
      a.IH = OH;
      a.IW = OW;
      a.CinP = r8(d->Cout);
      a.NB = r8(d->Cin);
      a.cout = d->Cin;
      a.CoutP = r8(d->Cin);
      a.OH = H;
      a.OW = W;
      a.isy = a.isx = 1;
      a.osy = a.osx = st;
      a.ooy = ph;
      a.oox = pw;
      a.pad_mode = DSR_PAD_ZERO;
      a.act = DSR_ACT_NONE;
      a.mask_x = mask_x;
      a.mask_act = mask_act;
      a.mask_slope = mask_slope;
      int nt = 0;
      for (int kh = 0; kh < d->KH; ++kh) {
        if ((ph + pad - kh) % st != 0) continue;
        for (int kw = 0; kw < d->KW; ++kw) {
          if ((pw + pad - kw) % st != 0) continue;
          // C '/' truncates toward zero, but (ph+pad-kh) is an exact multiple of st here
          a.taps[nt++] = pack_tap((ph + pad - kh) / st, (pw + pad - kw) / st, kh * d->KW + kw);
        }
      }
      a.ntaps = nt;
      finish_args(a, d->N, d->KH * d->KW);
      dsr_launch_conv_gemm(a, d->dtype, s);
    }
  if (folded) {
    size_t total = (size_t)d->N * d->H * d->W * (r8(d->Cin) / 8);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (d->dtype == DSR_BF16)
      hipLaunchKernelGGL((reflect_fold_kernel<DSR_DTYPE_BF16>), grid, block, 0, s, (const unsigned short*)workspace,
                         (unsigned short*)dx, d->N, d->H, d->W, r8(d->Cin), d->pad);
    else
      hipLaunchKernelGGL((reflect_fold_kernel<DSR_DTYPE_F16>), grid, block, 0, s, (const unsigned short*)workspace,
                         (unsigned short*)dx, d->N, d->H, d->W, r8(d->Cin), d->pad);
  }
  return dsr_launch_status("dsr_conv_dgrad");
}

extern "C" int dsr_conv_dgrad(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, void* workspace,
                              size_t ws_bytes, dsr_stream_t s) {
  return conv_dgrad_impl(d, dy, w_dgrad, dx, workspace, ws_bytes, nullptr, DSR_ACT_NONE, 0.f, s);
}

// dx = dgrad(dy) * act'(x_act): the backward of the activation that produced this conv's input, folded into the store loop
// of the input-gradient kernel (x_act = the conv's own input = that activation's output; ReLU, or LeakyReLU with slope > 0).
// Replaces the separate dsr_pw_act_bwd pass of the producing layer (read g, read o, write g') by one extra tile read --
// the VGG19 trunk of the perceptual loss (utils/GAN.py:19-57) is a chain of such pairs.  Stride-1 zero-pad layers on the
// 64 -> 64 kernel or the one-tile-per-block gather kernel; bit-identical to dsr_conv_dgrad followed by dsr_pw_act_bwd.
static bool dgrad_mask_supported(const dsr_conv_desc* d) {
  if (d->stride != 1 || d->pad_mode != DSR_PAD_ZERO || is_tail9(d) || is_smalln_dgrad(d)) return false;
  if (is_c64(d)) return true;
  return r8(d->Cout) % 64 == 0 && r8(d->Cin) >= 32;      // the gather kernel's LDS-DMA fast path with 16-byte output vectors
}
extern "C" int dsr_conv_dgrad_masked_supported(const dsr_conv_desc* d) { return d && !check_desc(d) && dgrad_mask_supported(d) ? 1 : 0; }
extern "C" int dsr_conv_dgrad_masked(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* x_act, int act,
                                     float slope, void* dx, dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !w_dgrad || !x_act || !dx) return dsr_fail(DSR_E_ARG, "conv_dgrad_masked: null pointer");
  if (act != DSR_ACT_RELU && !(act == DSR_ACT_LEAKY && slope > 0.f))
    return dsr_fail(DSR_E_ARG, "conv_dgrad_masked: ReLU, or LeakyReLU with a positive slope (the branch is read off the output)");
  if (!dgrad_mask_supported(d)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad_masked: layer shape not taken (stride 1, zero padding, >= 32 input channels)");
  return conv_dgrad_impl(d, dy, w_dgrad, dx, nullptr, 0, x_act, act, slope, s);
}

static void wgrad_plan(const dsr_conv_desc* d, WgradArgs& a) {
  int OH, OW;
  dsr_conv_out_size(d, &OH, &OW);
  a.M = d->N * OH * OW;
  a.OH = OH;
  a.OW = OW;
  a.IH = d->H;
  a.IW = d->W;
  a.CinP = r8(d->Cin);
  a.CoutP = r8(d->Cout);
  a.stride = d->stride;
  a.pad = d->pad;
  a.pad_mode = d->pad_mode;
  a.KH = d->KH;
  a.KW = d->KW;
  a.tiles_co = (a.CoutP + 63) / 64;
  a.tiles_ci = (a.CinP + 63) / 64;
  long long base = (long long)a.tiles_co * a.tiles_ci * d->KH * d->KW;
  long long want = (1536 + base - 1) / base;   // aim at ~6 blocks per CU in flight
  long long max_splits = (a.M + 127) / 128;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  long long chunk = (a.M + want - 1) / want;
  chunk = (chunk + 127) / 128 * 128;
  a.chunk = (int)chunk;
  a.splits = (int)((a.M + chunk - 1) / chunk);
  a.fd_ohw = fd_make((unsigned)(OH * OW));
  a.fd_ow = fd_make((unsigned)OW);
}

static int tile_plan(const dsr_conv_desc* d, WgradTileArgs& t, bool* taps_kernel = nullptr, bool* toeplitz = nullptr) {
  int OH, OW;
  dsr_conv_out_size(d, &OH, &OW);
  memset(&t, 0, sizeof(t));
  int ych = dsr_wgrad_tile_plan(d->KH, d->KW, d->stride, d->N, OH, OW, r8(d->Cin), r8(d->Cout), &t);
  if (taps_kernel) *taps_kernel = false;
  if (toeplitz) *toeplitz = false;
  if (ych == 0 && d->pad_mode == DSR_PAD_ZERO) {
    if (is_tail9(d)) {
      ych = dsr_wgrad_toeplitz_plan(d->N, d->H, d->W, &t);   // the generator's tail (generator.py:62)
      if (toeplitz) *toeplitz = true;
    } else {
      ych = dsr_wgrad_taps_plan(d->KH, d->KW, d->stride, d->N, OH, OW, r8(d->Cin), r8(d->Cout), &t);
      if (taps_kernel) *taps_kernel = ych > 0;
    }
  }
  t.N = d->N;
  t.OH = OH;
  t.OW = OW;
  t.IH = d->H;
  t.IW = d->W;
  t.CinP = r8(d->Cin);
  t.CoutP = r8(d->Cout);
  t.pad = d->pad;
  t.pad_mode = d->pad_mode;
  t.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * r8(d->Cin) * 2);       // check_desc keeps both below 2 GiB
  t.dy_bytes = (unsigned)((size_t)d->N * OH * OW * r8(d->Cout) * 2);
  return ych;
}

extern "C" size_t dsr_conv_wgrad_workspace(const dsr_conv_desc* d) {
  if (check_desc(d)) return 0;
  if (is_rgb9_wgrad(d)) return (size_t)dsr_wgrad_rgb9_blocks(d->N, d->H, d->W) * dsr_wgrad_rgb9_slab_floats() * sizeof(float);
  WgradTileArgs t;
  int ych = tile_plan(d, t);
  if (ych > 0) return (size_t)(ych + DSR_WGRAD_SCRATCH_SLABS) * d->KH * d->KW * t.CoutP * t.CinP * sizeof(float);
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  wgrad_plan(d, a);
  return (size_t)(a.splits + DSR_WGRAD_SCRATCH_SLABS) * d->KH * d->KW * a.CoutP * a.CinP * sizeof(float);
}

extern "C" int dsr_conv_wgrad(const dsr_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                              size_t ws_bytes, dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !dy || !dw) return dsr_fail(DSR_E_ARG, "conv_wgrad: null pointer");
  size_t need = dsr_conv_wgrad_workspace(d);
  if (!workspace || ws_bytes < need) return dsr_fail(DSR_E_WORKSPACE, "conv_wgrad: workspace %zu < %zu", ws_bytes, need);
  if (is_rgb9_wgrad(d)) {
    Rgb9WgradArgs q;
    memset(&q, 0, sizeof(q));
    q.x = x;
    q.dy = dy;
    q.partial = (float*)workspace;
    q.H = d->H;
    q.W = d->W;
    dsr_launch_wgrad_rgb9(q, d->N, d->Cin, dw, d->dtype, s);
    return dsr_launch_status("dsr_conv_wgrad(rgb9)");
  }
  WgradTileArgs t;
  bool taps_kernel = false, toeplitz = false;
  int ych = tile_plan(d, t, &taps_kernel, &toeplitz);
  if (ych > 0) {   // 3x3 (stride 1|2), 1x1, and 9x9 with few channels: taps derived from one staged halo tile
    t.x = x;
    t.dy = dy;
    t.partial = (float*)workspace;
    if (toeplitz)
      dsr_launch_wgrad_toeplitz(t, d->dtype, s);
    else if (taps_kernel)
      dsr_launch_wgrad_taps(t, d->KH, ych, d->dtype, s);
    else
      dsr_launch_wgrad_tile(t, d->KH, d->stride, ych, d->dtype, s);
    dsr_launch_wgrad_reduce(t.partial, dw, ych, d->KH * d->KW, d->Cout, d->Cin, t.CoutP, t.CinP, s);
    return dsr_launch_status("dsr_conv_wgrad");
  }
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  wgrad_plan(d, a);
  a.x = x;
  a.dy = dy;
  a.partial = (float*)workspace;
  dsr_launch_wgrad(a, d->dtype, s);
  dsr_launch_wgrad_reduce(a.partial, dw, a.splits, d->KH * d->KW, d->Cout, d->Cin, a.CoutP, a.CinP, s);
  return dsr_launch_status("dsr_conv_wgrad");
}

// dx = dgrad(dy) + addend in one launch: a residual block's input receives the gradient of its conv path AND of its skip
// path (generator.py:24: `return x + z`); the sum otherwise costs an elementwise pass over three tensors per block.  Taken
// by the 64 -> 64 3x3 kernel (its folded epilogue adds a residual tile: round(bf16(acc) + addend), the same two roundings
// as a separate add); other shapes: DSR_E_UNSUPPORTED, the caller adds itself.
extern "C" int dsr_conv_dgrad_add_supported(const dsr_conv_desc* d) { return d && !check_desc(d) && is_c64(d) ? 1 : 0; }
extern "C" int dsr_conv_dgrad_add(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, void* dx,
                                  dsr_stream_t s) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !w_dgrad || !addend || !dx) return dsr_fail(DSR_E_ARG, "conv_dgrad_add: null pointer");
  if (!is_c64(d)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad_add: 3x3 stride-1 zero-pad 64 -> 64 layers only");
  launch_c64_dgrad(d, dy, w_dgrad, addend, dx, s);
  return dsr_launch_status("dsr_conv_dgrad_add");
}

// ---- batched weight gradients: every 3x3 / stride-1 layer of a backward pass in one contraction launch + one reduction
// launch (conv_wgrad_tile.hip: conv_wgrad_dma_batch_kernel).  Entries with the same `dws[i]` (a weight used twice in the
// graph: the discriminator on the real and on the generated batch, train_GAN.py:44-47) must be adjacent; their
// contributions are summed -- what autograd's accumulation would do with two separate gradients.
extern "C" int dsr_conv_wgrad_batchable(const dsr_conv_desc* d) {
  if (!d || check_desc(d)) return 0;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 ? 1 : 0;
}

namespace {
struct BatchChunk {
  int first, count;        // problems [first, first + count)
  size_t ws_floats;
};
// Chunks of <= DSR_WGRAD_BATCH_MAX problems that do not split a group of equal dw; per chunk the pixel range of problem i
// is cut so that all blocks of the launch carry about the same number of tile steps and there are ~`target` of them.
int batch_chunks(int count, const dsr_conv_desc* descs, float* const* dws, WgradTileArgs* t, int* ych, BatchChunk* chunks,
                 int max_chunks) {
  const char* e = getenv("DSR_WGRAD_BATCH_BLOCKS");      // tuning switch: blocks per batched launch (default 1024 = two rounds of 2 per CU)
  const long long target = e ? atoi(e) : 1024;
  int nch = 0, i = 0;
  while (i < count) {
    if (nch == max_chunks) return -1;
    int j = i;
    while (j < count) {                                   // extend by whole groups
      int k = j + 1;
      while (k < count && dws && dws[k] == dws[j]) ++k;
      if (k - i > DSR_WGRAD_BATCH_MAX) break;
      j = k;
    }
    if (j == i) return -2;                                // one group larger than a launch can hold
    long long cost = 0;
    for (int q = i; q < j; ++q) {
      tile_plan(&descs[q], t[q]);
      cost += (long long)t[q].ntiles * t[q].tiles_co * t[q].tiles_ci;
    }
    long long per = (cost + target - 1) / (target > 0 ? target : 1);
    if (per < 1) per = 1;
    size_t wsf = 0;
    for (int q = i; q < j; ++q) {
      long long tpb = per < t[q].ntiles ? per : t[q].ntiles;
      t[q].tiles_per_block = (int)tpb;
      ych[q] = (int)((t[q].ntiles + tpb - 1) / tpb);
      wsf += (size_t)ych[q] * 9 * t[q].CoutP * t[q].CinP;
    }
    chunks[nch].first = i;
    chunks[nch].count = j - i;
    chunks[nch].ws_floats = wsf;
    ++nch;
    i = j;
  }
  return nch;
}
constexpr int kMaxBatchProblems = 256, kMaxBatchChunks = 64;
}  // namespace

extern "C" size_t dsr_conv_wgrad_batched_workspace(int count, const dsr_conv_desc* descs, float* const* dws) {
  if (count <= 0 || count > kMaxBatchProblems || !descs) return 0;
  for (int i = 0; i < count; ++i)
    if (!dsr_conv_wgrad_batchable(&descs[i])) return 0;
  static thread_local WgradTileArgs t[kMaxBatchProblems];
  static thread_local int ych[kMaxBatchProblems];
  BatchChunk ch[kMaxBatchChunks];
  const int n = batch_chunks(count, descs, dws, t, ych, ch, kMaxBatchChunks);
  size_t m = 0;
  for (int c = 0; c < n; ++c) m = ch[c].ws_floats > m ? ch[c].ws_floats : m;
  return m * sizeof(float);
}

extern "C" int dsr_conv_wgrad_batched(int count, const dsr_conv_desc* descs, const void* const* xs, const void* const* dys,
                                      float* const* dws, void* workspace, size_t ws_bytes, dsr_stream_t s) {
  if (count <= 0 || count > kMaxBatchProblems) return dsr_fail(DSR_E_ARG, "conv_wgrad_batched: count %d outside 1..%d", count, kMaxBatchProblems);
  if (!descs || !xs || !dys || !dws || !workspace) return dsr_fail(DSR_E_ARG, "conv_wgrad_batched: null table");
  for (int i = 0; i < count; ++i) {
    int rc = check_desc(&descs[i]);
    if (rc) return rc;
    if (!dsr_conv_wgrad_batchable(&descs[i])) return dsr_fail(DSR_E_UNSUPPORTED, "conv_wgrad_batched: entry %d is not a 3x3 stride-1 pad-1 convolution", i);
    if (descs[i].dtype != descs[0].dtype) return dsr_fail(DSR_E_UNSUPPORTED, "conv_wgrad_batched: mixed dtypes");
    if (!xs[i] || !dys[i] || !dws[i]) return dsr_fail(DSR_E_ARG, "conv_wgrad_batched: null pointer in entry %d", i);
    if (i && dws[i] == dws[i - 1] && (descs[i].Cin != descs[i - 1].Cin || descs[i].Cout != descs[i - 1].Cout))
      return dsr_fail(DSR_E_ARG, "conv_wgrad_batched: entries %d and %d share dw but not its shape", i - 1, i);
    for (int k = 0; k + 1 < i; ++k)
      if (dws[k] == dws[i] && dws[i - 1] != dws[i]) return dsr_fail(DSR_E_ARG, "conv_wgrad_batched: entries that share dw must be adjacent");
  }
  static thread_local WgradTileArgs t[kMaxBatchProblems];
  static thread_local int ych[kMaxBatchProblems];
  BatchChunk ch[kMaxBatchChunks];
  const int n = batch_chunks(count, descs, dws, t, ych, ch, kMaxBatchChunks);
  if (n < 0) return dsr_fail(DSR_E_UNSUPPORTED, "conv_wgrad_batched: more than %d entries share one dw", DSR_WGRAD_BATCH_MAX);
  for (int c = 0; c < n; ++c) {
    if (ws_bytes < ch[c].ws_floats * sizeof(float))
      return dsr_fail(DSR_E_WORKSPACE, "conv_wgrad_batched: workspace %zu < %zu", ws_bytes, ch[c].ws_floats * sizeof(float));
    WgradBatchArgs b;
    WgradReduceBatchArgs r;
    memset(&b, 0, sizeof(b));
    memset(&r, 0, sizeof(r));
    float* ws = (float*)workspace;
    int blocks = 0, rblocks = 0;
    for (int q = 0; q < ch[c].count; ++q) {
      const int i = ch[c].first + q;
      t[i].x = xs[i];
      t[i].dy = dys[i];
      t[i].partial = ws;
      b.p[q] = t[i];
      b.first_block[q] = blocks;
      blocks += ych[i] * t[i].tiles_co * t[i].tiles_ci;
      const size_t slab = (size_t)9 * t[i].CoutP * t[i].CinP;
      if (q && dws[i] == dws[i - 1]) {
        r.e[r.count - 1].splits += ych[i];                 // its slabs follow the previous entry's: one longer sum
      } else {
        WgradReduceBatchArgs::Entry& e = r.e[r.count++];
        e.partial = ws;
        e.dw = dws[i];
        e.splits = ych[i];
        e.Cout = descs[i].Cout;
        e.Cin = descs[i].Cin;
        e.CoutP = t[i].CoutP;
        e.CinP = t[i].CinP;
        e.first_block = rblocks;
        rblocks += (int)((slab + 255) / 256);
      }
      ws += (size_t)ych[i] * slab;
    }
    b.first_block[ch[c].count] = blocks;
    b.count = ch[c].count;
    r.total_blocks = rblocks;
    dsr_launch_wgrad_dma_batch(b, descs[0].dtype, s);
    dsr_launch_wgrad_reduce_batch(r, s);
  }
  return dsr_launch_status("dsr_conv_wgrad_batched");
}

// ---- measurement aid: the kernel family the dispatch above selects (kept next to it so the two cannot drift far)
static const char* gemm_name(int nb, long long M = 0, bool fast = false, bool stats = true, int flags = 0) {
  if (nb > 64 && dsr_conv_gemm_use_224(M, nb, fast, flags | (stats ? DSR_F_STATS : 0))) return "conv_gemm_kernel<224x256>";
  if (nb > 64 && dsr_conv_gemm_use_256(M, nb, fast, stats)) return "conv_gemm_kernel<256x256>";
  if (nb > 64 && dsr_conv_gemm_use_64(M, nb, fast, flags | (stats ? DSR_F_STATS : 0))) return "conv_gemm_kernel<64x128>";
  return nb > 64 ? "conv_gemm_kernel<128x128>" : (nb > 16 ? "conv_gemm_kernel<128x64>" : "conv_gemm_kernel<128x16>");
}
extern "C" const char* dsr_conv_kernel_name(const dsr_conv_desc* d, int op, const dsr_epilogue* e) {
  if (!d || check_desc(d)) return "invalid";
  const bool ps = e && e->pixel_shuffle, nchw = e && e->out_nchw_f32, stats = e && e->stats_partial;
  if (op == 0) {
    // (named like the symbols of a rocprof summary: <mode 0> statistics epilogue, <1> plain, <2> folded inference epilogue)
    if (((is_c64(d) && !ps) || is_c64_wide(d)) && !nchw)
      return (e && (e->bn_scale || e->residual)) ? "conv_c64_kernel<2>" : (stats ? "conv_c64_kernel<0>" : "conv_c64_kernel<1>");
    if (!ps && !nchw && !stats && !(e && (e->bn_scale || e->residual)) && d->Cout % 64 == 0 && d->Cout >= 128 &&
        (!e || e->act == DSR_ACT_NONE || e->act == DSR_ACT_RELU || e->act == DSR_ACT_LEAKY) &&
        dsr_halo64_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cin), d->Cout))
      return "conv_halo64_kernel";
    if (is_cin8(d, e)) return "conv_cin8_kernel";
    if (is_rgb9(d, e)) return "conv_rgb9_kernel";
    if (d->Cout <= 16 && d->stride == 1 && d->pad_mode == DSR_PAD_ZERO && d->KH * d->KW >= 9 && !stats && !ps &&
        d->KW == 9 && d->KH <= 9 && r8(d->Cin) == 64)
      return "conv_smalln_kernel";
    int OH, OW;
    dsr_conv_out_size(d, &OH, &OW);
    return gemm_name(r8(d->Cout), (long long)d->N * OH * OW, d->pad_mode == DSR_PAD_ZERO && r8(d->Cin) % 64 == 0, stats,
                     (ps ? DSR_F_PIXSHUF : 0) | (nchw ? DSR_F_OUT_NCHW_F32 : 0));
  }
  if (op == 1) {
    if (is_c64(d)) return "conv_c64_kernel<1>";
    if (is_tail9(d)) return "conv_dgrad_toeplitz9_kernel";
    if (is_smalln_dgrad(d)) return "conv_smalln_kernel";
    if (dsr_halo64_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cout), r8(d->Cin)))
      return "conv_halo64_kernel";
    if (dsr_dgrad_s2_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cin), r8(d->Cout), d->N))
      return "conv_dgrad_s2_kernel";
    // input gradient on the gather kernel: grid = the input pixels (stride 1) or one output-parity class of them (stride 2)
    const long long Mg = (long long)d->N * ((d->H + d->stride - 1) / d->stride) * ((d->W + d->stride - 1) / d->stride);
    return gemm_name(r8(d->Cin), Mg, r8(d->Cout) % 64 == 0 && (d->pad_mode == DSR_PAD_ZERO || d->pad == 0), false);
  }
  if (is_rgb9_wgrad(d)) return "conv_rgb9_wgrad_kernel";
  WgradTileArgs t;
  bool taps = false;
  bool toep = false;
  int ych = tile_plan(d, t, &taps, &toep);
  if (ych > 0) {
    if (toep) return "conv_wgrad_toeplitz9_kernel";
    if (taps) return "conv_wgrad_taps_kernel";
    // the three tile-resident kernels are different symbols in a rocprof summary: name them as it does
    if (d->KH == 3 && d->stride == 1) return "conv_wgrad_dma_kernel";
    if (d->KH == 3) return "conv_wgrad_dma_s2_kernel";
    return "conv_wgrad_tile_kernel<1x1>";
  }
  return "conv_wgrad_kernel";
}

// ---- input gradient of a 3x3 stride-2 layer + the BatchNorm-backward sums of the layer in front (conv_dgrad_s2_kernel<BN>)
static bool dgrad_bn_ok(const dsr_conv_desc* d) {
  return d && (d->dtype == DSR_BF16 || d->dtype == DSR_F16) && d->N > 0 && d->Cin % 8 == 0 &&
         dsr_dgrad_s2_bn_supported(d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->H, d->W, r8(d->Cin), r8(d->Cout), d->N);
}
extern "C" int dsr_conv_dgrad_bn_supported(const dsr_conv_desc* d) { return dgrad_bn_ok(d) ? 1 : 0; }
extern "C" int dsr_conv_dgrad_bn_rows(const dsr_conv_desc* d) {
  return d ? dsr_dgrad_s2_blocks(d->N, d->H, d->W, r8(d->Cin)) : 0;
}
extern "C" int dsr_conv_dgrad_bn(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* bn_y,
                                 const float* bn_scale, const float* bn_shift, int act, float slope, float* partial,
                                 dsr_stream_t s) {
  if (!dgrad_bn_ok(d)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad_bn: unsupported layer");
  if (!dy || !w_dgrad || !dx || !bn_y || !bn_scale || !bn_shift || !partial) return dsr_fail(DSR_E_ARG, "conv_dgrad_bn: null pointer");
  if (act != DSR_ACT_NONE && act != DSR_ACT_LEAKY) return dsr_fail(DSR_E_ARG, "conv_dgrad_bn: activation %d (LeakyReLU or none)", act);
  DgradS2Args q;
  memset(&q, 0, sizeof(q));
  q.dy = dy;
  q.w = w_dgrad;
  q.dx = dx;
  q.H = d->H;
  q.W = d->W;
  q.CinP = r8(d->Cin);
  q.CoutP = r8(d->Cout);
  q.bn_y = bn_y;
  q.bn_scale = bn_scale;
  q.bn_shift = bn_shift;
  q.bn_partial = partial;
  q.bn_act = act;
  q.bn_slope = slope;
  dsr_launch_dgrad_s2(q, d->N, d->dtype, s);
  return dsr_launch_status("dsr_conv_dgrad_bn");
}

// ---- input gradient of the 9x9 64 -> 3 tail (generator.py:78) with the backward of the PixelShuffle + PReLU in front of it
// (generator.py:37-39) in its epilogue (conv_dgrad_toeplitz9_kernel<PS>)
static bool dgrad_ps_ok(const dsr_conv_desc* d) {
  const char* e = getenv("DSR_DGRAD_PS");            // tuning switch, read per call: 0 = never
  if (e && e[0] == '0') return false;
  return d && is_tail9(d) && !(d->H & 1) && !(d->W & 1) && d->H >= 2 && d->W >= 2;
}
extern "C" int dsr_conv_dgrad_ps_supported(const dsr_conv_desc* d) { return dgrad_ps_ok(d) ? 1 : 0; }
extern "C" int dsr_conv_dgrad_ps_rows(const dsr_conv_desc* d) { return d ? dsr_dgrad_toeplitz_ps_blocks(d->N, d->H, d->W) : 0; }
extern "C" int dsr_conv_dgrad_ps(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* act_out,
                                 const float* prelu, void* dyu, float* partial, dsr_stream_t s) {
  if (!dgrad_ps_ok(d)) return dsr_fail(DSR_E_UNSUPPORTED, "conv_dgrad_ps: unsupported layer");
  if (!dy || !w_dgrad || !act_out || !prelu || !dyu || !partial) return dsr_fail(DSR_E_ARG, "conv_dgrad_ps: null pointer");
  dsr_launch_dgrad_toeplitz(dy, w_dgrad, nullptr, d->N, d->H, d->W, d->dtype, s, act_out, prelu, dyu, partial);
  return dsr_launch_status("dsr_conv_dgrad_ps");
}
