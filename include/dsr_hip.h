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
  int act;                 /* 0 none, 1 leaky(slope), 2 prelu(*prelu), 3 relu, 4 tanh, 5 sigmoid */
  float slope;
  const float* prelu;      /* 1-element device tensor (nn.PReLU(), generator.py:9,34,48) or NULL */
  const float* bias;       /* [Cout] or NULL */
  float* stats_partial;    /* [dsr_conv_stats_rows()][2][round_up(Cout,8)] sum / sum-of-squares rows for
                              a following train-mode BatchNorm (pre-activation values), or NULL */
  int pixel_shuffle;       /* 1: store through nn.PixelShuffle(2) (generator.py:32,38):
                              y is [N][2*OH][2*OW][round_up(Cout/4, 8)] */
  float* out_nchw_f32;     /* non-NULL: write fp32 NCHW [N][Cout][OH][OW] here instead of y (last layers) */
} dsr_epilogue;

int dsr_conv_out_size(const dsr_conv_desc* d, int* OH, int* OW);
/* rows of the BatchNorm statistics partial buffer written by dsr_conv_fwd */
int dsr_conv_stats_rows(const dsr_conv_desc* d);
/* element counts (16-bit) of the two packed weight images */
size_t dsr_conv_packed_elems(const dsr_conv_desc* d, int dgrad);
/* w [Cout][Cin][KH][KW] fp32 -> w_fwd [KH*KW][round_up(Cout,8)][round_up(Cin,8)],
 *                               w_dgrad [KH*KW][round_up(Cin,8)][round_up(Cout,8)] (may be NULL) */
int dsr_conv_pack_weight(const dsr_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad, dsr_stream_t s);
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

/* ------------------------------------------------------------------ pointwise / reductions (pointwise.hip)
 * nn.BatchNorm2d / PReLU / LeakyReLU / Tanh / Sigmoid / residual add / PixelShuffle backward / losses / Adam.
 * See the kernel comments in csrc/pointwise.hip for the reference lines each covers. */
int dsr_pw_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, dsr_stream_t s);
int dsr_pw_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, dsr_stream_t s);
int dsr_pw_pack_weight(int dtype, const float* w, void* wf, void* wd, int Cout, int Cin, int T, int NBo, int CinP,
                       int NBi, int CoutP, dsr_stream_t s);
int dsr_pw_sum_rows(const float* partial, int rows, int row_stride, int C, float scale, float* out, int accumulate,
                    dsr_stream_t s);
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
int dsr_pw_bn_bwd_finalize(const float* partial, int blocks, int C, int Cp, float count, float* dgamma, float* dbeta,
                           float* dprelu, float* c1, float* c2, dsr_stream_t s);
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
int dsr_pw_diff_loss(const float* pred, const float* tgt, float* grad, size_t n, int mode, float* partial, int blocks,
                     dsr_stream_t s);
int dsr_pw_bce_const(const float* p, int n, float target, float* loss, float* grad, int accumulate, dsr_stream_t s);
int dsr_pw_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                const int* step, dsr_stream_t s);
int dsr_pw_incr(int* step, dsr_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
