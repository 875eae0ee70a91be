/* dsr_hip.h -- C ABI of libdsr_hip.so, the MI355X (gfx950) kernel library behind the
 * nn.Module surface of LewisClifton/Deep-Super-Resolution.
 *
 * The reference has no FFI of its own: every device op is reached through
 * torch.nn modules.  This ABI is what those modules' forward/backward bind to in this build
 * (deep-super-resolution_amd/_lib.py, ctypes).  Conventions:
 *   - plain C: raw DEVICE pointers borrowed for the duration of the call, sizes as ints; no torch
 *     types.  The caller allocates every output and workspace.
 *   - every function only ENQUEUES work on `stream` and never synchronises, allocates or copies
 *     from the host: all of them are HIP-graph capturable.
 *   - return 0 on success, a negative code otherwise; dsr_last_error() gives the message.  The
 *     Python side turns that into RuntimeError (the reference raises on bad configs:
 *     utils/downsampler.py:12,38; models/DIP/utils.py:74,92).
 *   - activation tensors: NHWC, 16-bit (dtype 0 = bf16, 1 = f16), channel count padded up to a
 *     multiple of 8 ("Cp"); parameters, statistics, gradients of parameters: fp32 in the
 *     reference's own layouts (OIHW conv weights etc.).
 * Each entry cites the reference code whose device work it replaces.
 */
#ifndef DSR_HIP_H
#define DSR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dsr_stream_t; /* == hipStream_t */

#define DSR_BF16 0
#define DSR_F16 1

enum { DSR_OK = 0, DSR_E_ARG = -1, DSR_E_LAUNCH = -2, DSR_E_WORKSPACE = -3, DSR_E_UNSUPPORTED = -4 };

const char* dsr_last_error(void);
int dsr_abi_version(void);

/* ------------------------------------------------------------------ convolution
 * One descriptor for nn.Conv2d as the reference uses it:
 *   generator.py:7,11,52 (3x3 s1 p1), :30 (64->256), :47,62 (9x9 p4); discriminator.py:7,25 (3x3 s1|s2 p1);
 *   models/DIP/utils.py:83-105 (ReflectionPad2d + Conv2d k in {1,3}, stride 1|2, padding 0);
 *   utils/GAN.py:69-72 (VGG19 3x3 p1).
 * pad_mode: 0 zero, 1 reflect (the DIP padder folded into the tile loader), 2 replicate. */
typedef struct dsr_conv_desc {
  int dtype;
  int N, H, W;      /* input spatial size */
  int Cin, Cout;    /* real channel counts; tensors hold round_up(C, 8) channels */
  int KH, KW, stride, pad, pad_mode;
} dsr_conv_desc;

/* fused epilogue of the forward conv */
typedef struct dsr_epilogue {
  int act;                 /* 0 none, 1 leaky(slope), 2 prelu(*prelu), 3 relu, 4 tanh, 5 sigmoid, 6 elu(alpha=1) */
  float slope;
  const float* prelu;      /* 1-element device tensor (nn.PReLU(), generator.py:9,34,48) or NULL */
  const float* bias;       /* [Cout] or NULL */
  float* stats_partial;    /* [dsr_conv_stats_rows()][2][round_up(Cout,8)] sum / sum-of-squares rows for
                              a following train-mode BatchNorm (pre-activation values), or NULL */
  int pixel_shuffle;       /* 1: store through nn.PixelShuffle(2) (generator.py:32,38):
                              y is [N][2*OH][2*OW][round_up(Cout/4, 8)] */
  float* out_nchw_f32;     /* non-NULL: write fp32 NCHW [N][Cout][OH][OW] here instead of y (last layers) */
  /* inference-time folding of an eval-mode BatchNorm2d and a skip connection (generator.py:14-25 under gan_G.eval()):
   * y = act((conv + bias) * bn_scale[c] + bn_shift[c]) + residual.  All NULL for the plain epilogue.  Only layers for
   * which dsr_conv_fwd_affine_supported() is non-zero accept them; others return DSR_E_UNSUPPORTED. */
  const float* bn_scale;   /* [round_up(Cout,8)] */
  const float* bn_shift;   /* [round_up(Cout,8)] */
  const void* residual;    /* same layout as y */
} dsr_epilogue;

int dsr_conv_fwd_affine_supported(const dsr_conv_desc* d);

int dsr_conv_out_size(const dsr_conv_desc* d, int* OH, int* OW);
/* rows of the BatchNorm statistics partial buffer written by dsr_conv_fwd */
int dsr_conv_stats_rows(const dsr_conv_desc* d);
/* element counts (16-bit) of the two packed weight images */
size_t dsr_conv_packed_elems(const dsr_conv_desc* d, int dgrad);
/* w [Cout][Cin][KH][KW] fp32 -> w_fwd [KH*KW][round_up(Cout,8)][round_up(Cin,8)],
 *                               w_dgrad [KH*KW][round_up(Cin,8)][round_up(Cout,8)] (may be NULL) */
int dsr_conv_pack_weight(const dsr_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad, dsr_stream_t s);
/* the same packing for `count` weights in one launch per 48 (HOST arrays of device pointers and of Cout / Cin / KH*KW,
 * read before the call returns): an optimiser refreshes the images of everything it just updated */
int dsr_conv_pack_weight_multi(int dtype, int count, const float* const* w, void* const* w_fwd, void* const* w_dgrad,
                               const int* cout, const int* cin, const int* taps, dsr_stream_t s);
/* y = epilogue(conv(x, w) ) */
int dsr_conv_fwd(const dsr_conv_desc* d, const void* x, const void* w_fwd, const dsr_epilogue* e, void* y,
                 dsr_stream_t s);
/* dx = conv_transpose(dy, w) : autograd of nn.Conv2d w.r.t. its input.
 * workspace: dsr_conv_dgrad_workspace(d) bytes (non-zero only for reflect padding). */
size_t dsr_conv_dgrad_workspace(const dsr_conv_desc* d);
int dsr_conv_dgrad(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, void* workspace,
                   size_t ws_bytes, dsr_stream_t s);
/* dw [Cout][Cin][KH][KW] fp32 (overwritten) = autograd of nn.Conv2d w.r.t. its weight */
size_t dsr_conv_wgrad_workspace(const dsr_conv_desc* d);
int dsr_conv_wgrad(const dsr_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace, size_t ws_bytes,
                   dsr_stream_t s);

/* Fused backward of a first layer whose input needs no gradient -- Conv2d(<=3 -> 64, 3x3, s1, p1) + LeakyReLU | ReLU |
 * nothing (discriminator.py:22,25-27): dw (OIHW fp32) and db (nullable) from x, the gradient dout w.r.t. the activation
 * output and that output y, in one pass (g = dout * act'(y) is never written).  _supported() says whether a descriptor
 * qualifies; callers otherwise use dsr_pw_act_bwd + dsr_conv_wgrad. */
int dsr_conv_first_bwd_supported(const dsr_conv_desc* d, int act);
size_t dsr_conv_first_bwd_workspace(const dsr_conv_desc* d);
int dsr_conv_first_bwd(const dsr_conv_desc* d, const void* x, const void* dout, const void* y, int act, float slope,
                       float* dw, float* db, void* workspace, size_t ws_bytes, dsr_stream_t s);
/* The same without y: the sign of the pre-activation is recomputed from x and the layer's own fp32 weights w (OIHW) and
 * bias (nullable) -- the 1.07 GB activation of discriminator.py:25 at 512x512, batch 32, is not read by this pass. */
int dsr_conv_first_bwd_recompute(const dsr_conv_desc* d, const void* x, const void* dout, const float* w, const float* bias,
                                 int act, float slope, float* dw, float* db, void* workspace, size_t ws_bytes, dsr_stream_t s);

/* dsr_conv_dgrad of the 9x9 64 -> 3 tail (generator.py:78) whose input is PReLU(PixelShuffle(2)(conv)) (generator.py:37-39), with
 * that activation's whole backward in the same launch: the gradient w.r.t. the tail's input is never written; what comes out is
 * dyu [N][H/2][W/2][256], the gradient of the shuffle conv's output (channel 4c + 2i + j <- pixel (2h+i, 2w+j), masked by the
 * PReLU derivative taken from the sign of act_out = the tail's own input), and dsr_conv_dgrad_ps_rows(d) partial rows of [2][256]:
 * the column sums of dyu (that conv's bias gradient) and the PReLU-weight gradient terms (column 0 of the second slice) --
 * exactly what dsr_pw_act_bwd(pixshuf = 1) produces from dx and act_out.  prelu: the (positive) PReLU weight on the device. */
int dsr_conv_dgrad_ps_supported(const dsr_conv_desc* d);
int dsr_conv_dgrad_ps_rows(const dsr_conv_desc* d);
int dsr_conv_dgrad_ps(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* act_out, const float* prelu,
                      void* dyu, float* partial, dsr_stream_t s);

/* dsr_conv_dgrad of a 3x3 stride-2 layer (discriminator.py:31,33,35) whose input is the output of BatchNorm + LeakyReLU
 * (:14-19), with the two per-channel sums the BatchNorm backward of that layer needs formed in the same launch: partial gets
 * dsr_conv_dgrad_bn_rows(d) rows of [3][r8(Cin)] = (sum g, sum g*y, 0) with g = dx * act'(scale*y + shift) -- what
 * dsr_pw_bn_act_bwd_reduce would write after reading dx and y (bn_y: that layer's raw conv output, same shape as dx) once more;
 * feed them to dsr_pw_bn_bwd_finalize.  act: LeakyReLU or none.  _supported(): H, W even, tiles of 256 gradient pixels that
 * are whole rows of one image, one 64-channel slice per block. */
int dsr_conv_dgrad_bn_supported(const dsr_conv_desc* d);
int dsr_conv_dgrad_bn_rows(const dsr_conv_desc* d);
int dsr_conv_dgrad_bn(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* bn_y,
                      const float* bn_scale, const float* bn_shift, int act, float slope, float* partial, dsr_stream_t s);

/* The backward counterpart of dsr_conv_first2_fwd for a step that needs no image gradient (the discriminator's own update,
 * train_GAN.py:47-56): the input gradient of the stride-2 layer d1 (discriminator.py:29) and the whole backward of the image
 * layer d0 under it (:25-27: activation mask, bias gradient, weight gradient) in ONE launch.  The gradient of the 64-channel
 * activation between them (1.07 GB at 512x512, batch 32) is formed per tile in LDS and never written.  dy: gradient of d1's
 * conv output [N][H/2][W/2][64]; w1_dgrad: d1's packed input-gradient weights (dsr_conv_pack_weight); img: NHWC 16-bit image
 * with 8 channels; w0 (OIHW fp32) / b0 (nullable): d0's parameters as the forward used them; dw0 (OIHW fp32), db0 (nullable).
 * _supported(): 64 -> 64 channels, even H and W with (W/2) % 256 == 0, d0 as dsr_conv_first_bwd_supported. */
int dsr_conv_dgrad_first_bwd_supported(const dsr_conv_desc* d0, const dsr_conv_desc* d1, int act0);
size_t dsr_conv_dgrad_first_bwd_workspace(const dsr_conv_desc* d1);
int dsr_conv_dgrad_first_bwd(const dsr_conv_desc* d0, const dsr_conv_desc* d1, const void* dy, const void* w1_dgrad,
                             const void* img, const float* w0, const float* b0, int act0, float slope0, float* dw0, float* db0,
                             void* workspace, size_t ws_bytes, dsr_stream_t s);

/* The discriminator's first two convolutions (discriminator.py:25 Conv2d(3,64,3,1,1) + LeakyReLU; :29 Conv2d(64,64,3,2,1) in
 * front of its BatchNorm) as ONE forward launch: the first layer's 64-channel activation (1.07 GB at 512x512, batch 32) is
 * recomputed per tile in LDS and goes to HBM only if `a0` is given (training: the second layer's weight gradient reads it).
 * d0 / d1: the two layers' descriptors (same N, H, W); x: NHWC 16-bit image with 8 channels; w0 / w1: packed forward images;
 * y1: raw second-layer output [N][OH][OW][64]; stats (nullable): dsr_conv_first2_stats_rows(d0) rows of [2][64] partial sums. */
int dsr_conv_first2_supported(const dsr_conv_desc* d0, const dsr_conv_desc* d1);
int dsr_conv_first2_stats_rows(const dsr_conv_desc* d0);
int dsr_conv_first2_fwd(const dsr_conv_desc* d0, const dsr_conv_desc* d1, const void* x, const void* w0, const float* bias0,
                        float slope0, const void* w1, const float* bias1, void* a0, void* y1, float* stats, dsr_stream_t s);

/* Which kernel the dispatcher launches for this descriptor (measurement aid: bench.py labels its HIP-event timings
 * with it so they can be matched against rocprofv3's kernel names).  op: 0 forward, 1 dgrad, 2 wgrad.  `e` may be
 * NULL (no statistics, no pixel shuffle, NHWC output).  Returns a static string; never NULL. */
/* dx = dgrad(dy) * act'(x_act) in one launch: x_act = this conv's own input = the OUTPUT of the activation in front of it
 * (ReLU, or LeakyReLU with slope > 0); replaces the producing layer's separate activation-backward pass (the VGG19 trunk of
 * utils/GAN.py:19-57 is a chain of conv + ReLU pairs).  Bit-identical to dsr_conv_dgrad followed by dsr_pw_act_bwd. */
int dsr_conv_dgrad_masked_supported(const dsr_conv_desc* d);
int dsr_conv_dgrad_masked(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* x_act, int act, float slope,
                          void* dx, dsr_stream_t s);
/* dx = dgrad(dy) + addend in one launch (the gradient of a residual block's input: conv path + skip path,
 * generator.py:24); dsr_conv_dgrad_add_supported tells whether the shape is taken (64 -> 64 3x3 stride 1 zero pad). */
int dsr_conv_dgrad_add_supported(const dsr_conv_desc* d);
int dsr_conv_dgrad_add(const dsr_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, void* dx,
                       dsr_stream_t s);
/* Every 3x3 / stride-1 / pad-1 weight gradient of a backward pass in one contraction launch + one reduction launch
 * (replaces the per-layer weight-gradient kernels of loss.backward(), train_GAN.py:52,63).  Entries whose dws[i] are equal
 * must be adjacent and are summed into that one gradient (a weight applied to two batches).  All entries share one dtype.
 * dsr_conv_wgrad_batchable: 1 if the shape is taken; the workspace size depends on the whole table. */
int dsr_conv_wgrad_batchable(const dsr_conv_desc* d);
size_t dsr_conv_wgrad_batched_workspace(int count, const dsr_conv_desc* descs, float* const* dws);
int dsr_conv_wgrad_batched(int count, const dsr_conv_desc* descs, const void* const* xs, const void* const* dys,
                           float* const* dws, void* workspace, size_t ws_bytes, dsr_stream_t s);
const char* dsr_conv_kernel_name(const dsr_conv_desc* d, int op, const dsr_epilogue* e);

/* ------------------------------------------------------------------ pointwise / reductions (pointwise.hip)
 * nn.BatchNorm2d / PReLU / LeakyReLU / Tanh / Sigmoid / residual add / PixelShuffle backward / losses / Adam.
 * See the kernel comments in csrc/pointwise.hip for the reference lines each covers. */
int dsr_pw_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, dsr_stream_t s);
int dsr_pw_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, dsr_stream_t s);
int dsr_pw_pack_weight(int dtype, const float* w, void* wf, void* wd, int Cout, int Cin, int T, int NBo, int CinP,
                       int NBi, int CoutP, dsr_stream_t s);
/* Two-stage deterministic reductions: kernels write one partial ROW per block; the finalize entry points first
 * compact many rows in parallel into <= dsr_pw_scratch_rows() rows stored right behind the partial buffer, so EVERY
 * partial buffer handed to dsr_pw_bn_finalize / dsr_pw_bn_bwd_finalize / dsr_pw_sum_rows(compact=1) must have room
 * for `rows + dsr_pw_scratch_rows()` rows. */
int dsr_pw_scratch_rows(void);
/* out[c] (+)= scale * sum_r partial[r*row_stride + col_offset + c], c < C */
int dsr_pw_sum_rows(const float* partial, int rows, int row_stride, int col_offset, int C, float scale, float* out,
                    int accumulate, int compact, dsr_stream_t s);
int dsr_pw_bn_finalize(const float* partial, int tiles, int stride, int C, int Cp, float count, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                       float momentum, float eps, int updates, float* scale, float* shift, float* mean, float* rstd,
                       dsr_stream_t s);
int dsr_pw_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, int C, int Cp, float* scale, float* shift, float* mean, float* rstd,
                          dsr_stream_t s);
/* grid sizing shared by the two-stage reductions: returns #blocks, writes rows per block */
int dsr_pw_reduce_blocks(size_t P, int* rows_per_block);
int dsr_pw_channel_stats(int dtype, const void* x, size_t P, int Cp, int blocks, int rpb, float* partial,
                         dsr_stream_t s);
int dsr_pw_bn_act_fwd(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                      size_t P, int Cp, int act, float slope, const float* prelu, dsr_stream_t s);
int dsr_pw_bn_act_bwd_reduce(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                             const float* mean, const float* rstd, size_t P, int Cp, int blocks, int rpb, int act,
                             float slope, const float* prelu, float* partial, dsr_stream_t s);
int dsr_pw_bn_bwd_finalize(const float* partial, int blocks, int C, int Cp, float count, const float* mean,
                           const float* rstd, float* dgamma, float* dbeta, float* dprelu, float* c1, float* c2,
                           dsr_stream_t s);
int dsr_pw_bn_act_bwd_apply(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                            const float* mean, const float* rstd, const float* c1, const float* c2, void* dy, size_t P,
                            int Cp, int act, float slope, const float* prelu, int train, dsr_stream_t s);
int dsr_pw_act_bwd(int dtype, const void* dout, const void* out, void* dy, int N, int H, int W, int CyP, int CoP,
                   int pixshuf, int act, float slope, const float* prelu, int blocks, int rpb, float* partial,
                   dsr_stream_t s);
int dsr_pw_act_bwd_nchw(int dtype, const float* dout, const float* out, void* dy, int N, int C, int H, int W, int Cp,
                        int act, dsr_stream_t s);
int dsr_pw_colsum(int dtype, const void* x, size_t P, int Cp, int blocks, int rpb, float* partial, dsr_stream_t s);
int dsr_pw_add(int dtype, const void* a, const void* b, void* out, size_t nvec, dsr_stream_t s);
/* out[i] = (a * x[i] + b * y[i]) * g[0] over n fp32 elements; y nullable (-> 0), g nullable device scalar (-> 1).
 * The scalar arithmetic of the step recipes (utils/GAN.py:105,122 loss sums; loss-gradient scaling). */
int dsr_pw_axpby_f32(const float* x, const float* y, float a, float b, const float* g, float* out, size_t n,
                     dsr_stream_t s);
int dsr_pw_diff_loss(const float* pred, const float* tgt, float* grad, size_t n, int mode, float* partial, int blocks,
                     dsr_stream_t s);
int dsr_pw_bce_const(const float* p, int n, float target, float* loss, float* grad, int accumulate, dsr_stream_t s);
/* torch.optim.Adam defaults; g is multiplied by grad_scale first (1/S when a static loss scale S is in use);
 * shadow_bf16 (nullable): also write the bf16 image of the updated parameter (same layout) */
int dsr_pw_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                const int* step, float grad_scale, void* shadow_bf16, dsr_stream_t s);
/* the same update for `count` tensors in as few launches as possible (64 tensors per launch); the four pointer
 * arrays and n[] are HOST arrays of device pointers / element counts, read before the call returns */
int dsr_pw_adam_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                      const size_t* n, float lr, float b1, float b2, float eps, const int* step, float grad_scale,
                      dsr_stream_t s);
int dsr_pw_incr(int* step, dsr_stream_t s);

/* ------------------------------------------------------------------ discriminator dense head (linear.hip)
 * models/GAN/discriminator.py:37-45,65-72: flatten(C,H,W) -> Linear(K,O) -> LeakyReLU(0.2) -> Linear(O,1) -> Sigmoid */
/* 16-bit shadow copy of an fp32 tensor (n % 8 == 0) */
int dsr_cast16(int dtype, const float* src, void* dst, size_t n, dsr_stream_t s);
/* mode 0: flat[b][c*HW+p] = act[b][p][c];  1: flatT[c*HW+p][b] (Bp columns, zero padded);  2: act <- flat */
int dsr_flatten(int dtype, const void* src, void* dst, int B, int HW, int C, int Cp, int Bp, int mode, dsr_stream_t s);
size_t dsr_linear_fwd_workspace(int B, size_t K, int O);
/* out[b][o] (fp32) = act(sum_k x[b][k] w16[o][k] + bias[o]) */
int dsr_linear_fwd(int dtype, const void* x, const void* w16, const float* bias, int act, float slope, float* out,
                   int B, size_t K, int O, void* workspace, size_t ws_bytes, dsr_stream_t s);
/* dx[b][k] (16-bit) = sum_o dy16[b][o] w16[o][k] */
int dsr_linear_dgrad(int dtype, const void* dy16, const void* w16, void* dx, int B, int O, size_t K, dsr_stream_t s);
/* dw[o][k] (fp32, overwritten) = sum_b dyT16[o][b] xT16[k][b];  Bp in {32, 64} */
int dsr_linear_wgrad(int dtype, const void* dyT16, const void* xT16, float* dw, int Bp, int O, size_t K,
                     dsr_stream_t s);
/* data-parallel form: dw = scale * sum over R gathered rank-local factor pairs, dyT16_all [R][O][Bp], xT16_all [R][K][Bp]
 * (all-gather the 67 MB + 128 KB factors instead of all-reducing the 2.1 GB gradient) */
int dsr_linear_wgrad_gathered(int dtype, const void* dyT16_all, const void* xT16_all, float* dw, int Bp, int O, size_t K,
                              int R, float scale, dsr_stream_t s);
/* the same contraction with torch.optim.Adam's update applied to p / m / v (and p's bf16 shadow) in the epilogue: the
 * gradient is never written (R = 1, scale = 1: bit-identical to dsr_linear_wgrad followed by dsr_pw_adam); K % 64 == 0.
 * Replaces loss.backward() writing dense1.weight.grad + optimizer.step() reading it (train_GAN.py:52-53 on
 * discriminator.py:54). */
int dsr_linear_wgrad_adam(int dtype, const void* dyT16_all, const void* xT16_all, int Bp, int O, size_t K, int R, float scale,
                          float* p, float* m, float* v, void* shadow_bf16, const int* step, float lr, float b1, float b2,
                          float eps, float grad_scale, dsr_stream_t s);
/* out[b] = sigmoid(h[b][:] . w2 + b2) */
int dsr_dense2_fwd(const float* h, const float* w2, const float* b2, int B, int K1, float* out, dsr_stream_t s);
/* backward of the fp32 tail; also emits the 16-bit dy / dy^T operands of the two dense1 GEMMs */
int dsr_dense2_bwd(int dtype, const float* dout, const float* out, const float* h, const float* w2, int B, int K1,
                   int Bp, float slope, float* dw2, float* db2, float* db1, void* dy16, void* dyT16, dsr_stream_t s);

/* ------------------------------------------------------------------ resampling / data movement (resample.hip) */
/* nn.MaxPool2d(2,2) of the VGG19 trunk (utils/GAN.py:24,29,38,47); backward routes to the first maximum */
int dsr_maxpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t s);
int dsr_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t s);
/* the same with the backward of the ReLU that produced x folded in (conv + ReLU + MaxPool of the VGG trunk, utils/GAN.py:24-47):
 * dx = routed dy where the window maximum is > 0 */
int dsr_maxpool2_relu_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t s);
/* nn.AvgPool2d(2,2) after a stride-1 conv: downsample_mode='avg' of models/DIP/utils.py:86-94 (floor mode);
 * H, W are the INPUT size of the pool.  downsample_mode='max' uses dsr_maxpool2_* above. */
int dsr_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t s);
int dsr_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t s);
/* nn.Upsample(scale_factor=2, mode='nearest') (models/DIP/skip.py:77, the builder's default); H, W = INPUT size */
int dsr_nearest2x_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t s);
int dsr_nearest2x_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t s);
/* nn.Upsample(scale_factor=2, mode='bilinear') (models/DIP/skip.py:77); H, W are the INPUT size */
int dsr_bilinear2x_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, dsr_stream_t s);
int dsr_bilinear2x_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, dsr_stream_t s);
/* VGG19_Weights.IMAGENET1K_V1.transforms() (utils/GAN.py:82-83): separable antialiased resize + crop + normalise.
 * src fp32 NCHW [N][C<=3][H][W] -> dst NHWC 16-bit [N][OH][OW][8]; tables are host-built (utils/GAN.py mirror):
 * per output index start/count/weights[KT]; the backward takes the transposed tables. */
int dsr_resize_norm_fwd(int dtype, const float* src, void* dst, int N, int C, int H, int W, int OH, int OW,
                        const int* ys, const int* yc, const float* yw, const int* xs, const int* xc, const float* xw,
                        int KT, const float* mean3, const float* std3, dsr_stream_t s);
int dsr_resize_norm_bwd(int dtype, const void* dout, float* dsrc, int N, int C, int H, int W, int OH, int OW,
                        const int* ty_s, const int* ty_c, const float* ty_w, const int* tx_s, const int* tx_c,
                        const float* tx_w, int KT, const float* std3, dsr_stream_t s);
/* Concat with centre crop (models/DIP/utils.py:18-38) and its adjoint: 16-bit NHWC box copy
 * dst[n][dy0+y][dx0+x][cd0+c] = src[n][sy0+y][sx0+x][cs0+c], y<BH, x<BW, c<C */
int dsr_box_copy(const void* src, void* dst, int N, int BH, int BW, int C, int SH, int SW, int SCp, int sy0, int sx0,
                 int cs0, int DH, int DW, int DCp, int dy0, int dx0, int cd0, dsr_stream_t s);
/* Downsampler (utils/downsampler.py:44-71): depthwise k x k kernel, stride f, ReplicationPad2d(p); fp32 NCHW, NC = N*C */
int dsr_downsample_fwd(const float* x, const float* kern, float* y, int NC, int H, int W, int k, int f, int p,
                       dsr_stream_t s);
int dsr_downsample_bwd(const float* dy, const float* kern, float* dx, int NC, int H, int W, int k, int f, int p,
                       dsr_stream_t s);

/* SSIM as the reference's scripts measure it (torchmetrics StructuralSimilarityIndexMeasure at train_GAN.py:31,111,
 * eval_GAN.py:31,48): Gaussian 11x11 sigma 1.5 window, K1 0.01, K2 0.03, per plane (planes = N*C fp32 H x W images),
 * window positions inside the image.  Writes dsr_ssim_blocks(planes, H, W) partial sums; their total divided by
 * planes * (H-10) * (W-10) is the mean SSIM. */
int dsr_ssim_blocks(int planes, int H, int W);
int dsr_ssim_f32(const float* img1, const float* img2, int planes, int H, int W, float data_range, float* partial,
                 dsr_stream_t s);

/* measurement aid: out16[2x] = shader-clock cycle counter (s_memtime) and out16[2x+1] = 100 MHz real-time counter
 * (s_memrealtime) of XCD x (8 pairs; zero-fill before), read when the stream reaches this launch; two samples give the average
 * shader clock of what ran between them (difference pairs of the same XCD only: the cycle counters are per XCD) */
int dsr_clock_sample(unsigned long long* out16, dsr_stream_t s);
/* ---- data-side byte kernels (dataset.py:9-62,121-159; utils/degradation.py:5-20) on device-resident
 * uint8 HWC images.  Integer / byte arithmetic, bit-identical to Pillow / numpy.
 * dsr_resample_u8: ONE pass of Pillow's 8-bit resampler along `axis` (1 = width, 0 = height); `bounds` [out_size][2] and `kk`
 * [out_size][ksize] are the 22-bit fixed-point tables of Pillow's precompute_coeffs + normalize_coeffs_8bpc (device int32,
 * built on the host by utils/degradation.py: resample_tables).  Image.resize(.., BICUBIC) = width pass, then height pass. */
int dsr_resample_u8(const unsigned char* src, unsigned char* dst, int H, int W, int C, int axis, int out_size, const int* bounds,
                    const int* kk, int ksize, dsr_stream_t s);
/* out = uint8(clip(img + noise, 0, 255)) (truncating cast); noise float64 (drawn by numpy, as the reference does) or float32 */
int dsr_noise_gaussian_u8(const unsigned char* img, const void* noise, int noise_is_f64, unsigned char* out, size_t n, dsr_stream_t s);
/* salt -> 255, then pepper -> 0, per pixel over all channels; salt / pepper: [H][W] bytes (non-zero = hit) */
int dsr_salt_pepper_u8(const unsigned char* img, const unsigned char* salt, const unsigned char* pepper, unsigned char* out, int H,
                       int W, int C, dsr_stream_t s);
/* B patches of ph x pw pixels, one from each of B RGB images (HOST tables of device pointers / sizes / corners), converted to an
 * fp32 [B][3][ph][pw] batch: ToTensor (/255) followed by the scaling `mode` selects */
enum { DSR_PATCH_UNIT = 0,      /* [0,1]: ToTensor only */
       DSR_PATCH_LR_REF = 1,    /* dataset.py:152: /255 a second time (the reference's LR scaling as written) */
       DSR_PATCH_HR_REF = 2,    /* dataset.py:155-157: /255 a second time, *2, -1 (as written) */
       DSR_PATCH_HR_UNIT = 3 }; /* *2 - 1: the [-1,1] its comments intend */
/* dataset.py:149-159 in place on an fp32 tensor ToTensor already put in [0,1]: mode DSR_PATCH_LR_REF: x /= 255;
 * DSR_PATCH_HR_REF: x = x / 255 * 2 - 1 (true divisions, as on the host) */
int dsr_scale_images_f32(float* x, size_t n, int mode, dsr_stream_t s);
#define DSR_PATCH_BATCH_MAX 64
int dsr_patch_batch_u8(int count, const unsigned char* const* images, const int* heights, const int* widths, const int* tops,
                       const int* lefts, int ph, int pw, int mode, float* out, dsr_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
