/*
 * oct_hip.h -- C ABI of liboct_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * U-Net training hot path of ZhangHH233/Retinal_OCT_Image_Segmentation_via_Deep_Learning.
 *
 * The reference has no FFI of its own: its boundary is Python, and every op on
 * the path is a stock torch.nn call.  Each entry point below replaces the torch op(s) at the
 * cited reference call site; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless marked host.
 *   - the library never allocates or frees memory and never synchronises: work is enqueued on the
 *     hipStream_t passed as `void* stream`; hipSetDevice is the caller's job.
 *   - every function returns 0 on success or a negative code (OCT_E_*); it never throws/aborts.
 *     oct_get_last_error() returns the message of the calling thread's last failure.
 *   - re-entrant: no global mutable state besides the thread-local error string (backward is
 *     called from PyTorch's autograd thread).
 *   - activation tensors are NHWC ("channels-last") in the dtype named by `dtype`
 *     (OCT_DT_BF16 production, OCT_DT_F32 parity mode); accumulation is always fp32.
 *     Parameters, gradients, BN statistics and losses are fp32.  The network input
 *     (B,Cin,H,W) and output (B,Ccls,H,W) keep the reference's NCHW layout.
 */
#ifndef OCT_HIP_H
#define OCT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCT_VERSION 220 /* 0.2.2 (round 3): 7x3 on the pipelined kernels, oct_rowdot_* up to 12 outputs, frozen-BatchNorm backward, depth-rolling 3-D kernel; 0.2.1: + oct_bilinear_resize_*, floor-mode max-pooling (MGU-Net); 0.2.0: (kh,kw) kernels, depth taps, partials, ReLayNet / 3-D / per-class metric entry points */

/* dtypes of activation storage */
#define OCT_DT_BF16 0
#define OCT_DT_F32 1

/* error codes */
#define OCT_OK 0
#define OCT_IMG_SHIFT_ALL 2  /* OctWgradDesc.in_img_shift: all depth taps in one launch */
#define OCT_E_INVALID (-22)  /* bad descriptor / unsupported shape */
#define OCT_E_LAUNCH (-5)    /* hipLaunch failure */
#define OCT_E_NODEVICE (-19) /* no HIP device */

/* source transform applied while a conv stages its input (BN-apply + ReLU fused on load) */
#define OCT_XF_NONE 0
#define OCT_XF_AFFINE_RELU 1 /* a = max(x*scale[c] + shift[c], 0) */
#define OCT_XF_AFFINE 2      /* a = x*scale[c] + shift[c]: a deferred bias add (SD_Layer_Net conv_block.init_conv) */

/* input / output addressing of the implicit GEMM */
#define OCT_IN_PLAIN 0
#define OCT_IN_S2D 1  /* input pixel (y,x) gathers the 2x2 block of a 2H x 2W tensor: k=(dy*2+dx)*C+c */
#define OCT_OUT_PLAIN 0
#define OCT_OUT_D2S 1 /* GEMM column n=(dy*2+dx)*C+co is stored at pixel (2y+dy, 2x+dx), channel co */

/* weight packing modes (fp32 torch layout -> MFMA A-fragment order in the activation dtype) */
#define OCT_PACK_CONV_FPROP 0   /* (Cout,Cin,3,3): row=co,  k=(tap,ci)               */
#define OCT_PACK_CONV_DGRAD 1   /* (Cout,Cin,3,3): row=ci,  k=(flipped tap,co)       */
#define OCT_PACK_DECONV_FPROP 2 /* (Cin,Cout,2,2): row=(dydx,co), k=ci               */
#define OCT_PACK_DECONV_DGRAD 3 /* (Cin,Cout,2,2): row=ci,  k=(dydx,co)              */
#define OCT_PACK_1X1_DGRAD 4    /* (Cout,Cin,1,1): row=ci,  k=co                     */
#define OCT_PACK_1X1_FPROP 5    /* (Cout,Cin,1,1): row=co,  k=ci                     */
/* 3-D (oct_pack_weights3d): Conv3d(3x3x3) and ConvTranspose3d(k=2,s=2) of the cfg5 volumetric U-Net */
#define OCT_PACK_CONV3D_FPROP 6   /* (Cout,Cin,3,3,3): row=co, taps=(kh,kw), k=(kd,ci)                        */
#define OCT_PACK_CONV3D_DGRAD 7   /* (Cout,Cin,3,3,3): row=ci, k=(kd,co), kernel point-reflected in d, h, w   */
#define OCT_PACK_DECONV3D_FPROP 8 /* (Cin,Cout,2,2,2), one depth slice kdi: row=(dydx,co), k=ci              */
#define OCT_PACK_DECONV3D_DGRAD 9 /* (Cin,Cout,2,2,2): row=ci, k=(kd,dydx,co)                                */

const char* oct_version_string(void);
int oct_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf */
int oct_get_last_error(char* buf, size_t len);
/* number of HIP devices visible (0 on a machine without a GPU); never fails */
int oct_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (replaces nn.Conv2d(3x3,p=1) YNet_2022.py:578-596 and
 * nn.ConvTranspose2d(k=2,s=2) YNet_2022.py:526-540, forward and data-gradient).
 *   fprop  3x3 : taps=9, IN_PLAIN, OUT_PLAIN, weights packed CONV_FPROP
 *   dgrad  3x3 : taps=9, IN_PLAIN (x0 = dY), OUT_PLAIN (+split for a concat), CONV_DGRAD
 *   deconv fwd : taps=1, IN_PLAIN, OUT_D2S (+bias),  DECONV_FPROP, cout = 4*Cout
 *   deconv dgrd: taps=1, IN_S2D (x0 = dU at 2H x 2W), OUT_PLAIN, DECONV_DGRAD
 * The two sources x0|x1 are a virtual torch.cat((x0,x1),1) (YNet_2022.py:557).
 * ------------------------------------------------------------------------------------------ */
typedef struct OctConvDesc {
  int dtype;         /* OCT_DT_* */
  int n, h, w;       /* batch and GEMM pixel grid (low-res grid for S2D / D2S) */
  int c0, c1;        /* channels of source 0 / 1 (c1 = 0: single source) */
  int cout;          /* GEMM columns (output channels; 4*Cout for D2S) */
  int taps;          /* 9 or 1 */
  int xform0, xform1;
  int in_mode, out_mode;
  int split;         /* OUT_PLAIN: channels [0,split) -> y0, [split,cout) -> y1; 0: all to y0 */
  int want_stats;    /* write per-workgroup partial sum / sum of squares of the fp32 outputs */
  int kh, kw;        /* kernel size; 0, 0 = derive from taps (9 -> 3x3, 1 -> 1x1).  7, 3 with taps = 21 is ReLayNet's
                      * BasicBlock conv (ReLayNet_2017.py:155-160): stride 1, padding ((kh-1)/2, (kw-1)/2), plain in/out */
  int depth;         /* 0: 2-D.  D > 0: the n images are n/D volumes of D slices each (NDHWC) and the GEMM gains depth taps:
                      *   taps 9, plain in : Conv3d 3x3x3, padding 1 -- K = 3*(c0+c1), tap kd reads slice d+kd-1 (zero outside
                      *                      the volume); weights packed OCT_PACK_CONV3D_FPROP / _DGRAD
                      *   taps 1, IN_S2D   : ConvTranspose3d(k2,s2) data gradient -- K = 8*c0, k = (kd,dy,dx,c) gathers from
                      *                      slice 2d+kd of the 2D x 2H x 2W tensor; weights OCT_PACK_DECONV3D_DGRAD        */
  int out_img_mul, out_img_add; /* OUT_D2S only, 0,0 = identity: the output image index is img*mul + add.  ConvTranspose3d
                      * forward = two D2S launches (kd = 0, 1) with mul = 2, add = kd and the OCT_PACK_DECONV3D_FPROP slice */
} OctConvDesc;

typedef struct OctConvArgs {
  const void* x0; const void* x1;
  const float* scale0; const float* shift0; /* per channel of source 0 (xform0 != NONE) */
  const float* scale1; const float* shift1;
  const void* wpacked;                       /* from oct_pack_weights */
  const float* bias;                         /* D2S only, per real output channel; may be NULL */
  void* y0; void* y1;
  float* stat_partials;                      /* [oct_conv_stat_blocks][2][cout] fp32 */
} OctConvArgs;

/* number of partial-statistics rows oct_conv_forward writes (= spatial workgroups) */
int oct_conv_stat_blocks(const OctConvDesc* d);
/* elements (of the activation dtype) of the packed weight buffer for a GEMM with `rows` output
 * rows, `taps` taps and `kch` input channels */
size_t oct_packed_weight_elems(int rows, int taps, int kch);
int oct_pack_weights(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, void* stream);
/* OCT_PACK_CONV_FPROP / OCT_PACK_CONV_DGRAD of a (Cout,Cin,kh,kw) weight with any odd kernel size up to 7x7
 * (buffer: oct_packed_weight_elems(rows, kh*kw, kch)); ReLayNet_2017.py:155-160 uses 7x3.                  */
int oct_pack_weights_kk(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int kh, int kw,
                        void* stream);
/* The OCT_PACK_*3D modes (buffer: oct_packed_weight_elems(rows, taps, kch) with rows/taps/kch = cout/9/3cin, cin/9/3cout,
 * 4cout/1/cin, cin/1/8cout).  kdi: depth slice for OCT_PACK_DECONV3D_FPROP, 0 otherwise.                         */
int oct_pack_weights3d(int mode, int dtype, const float* w, void* wpacked, int cout, int cin, int kdi, void* stream);
/* The same for up to OCT_PACK_BATCH_MAX weights per launch (all re-packings that follow an optimizer
 * step in one go); longer lists are split.  Jobs are plain structs read on the host.              */
#define OCT_PACK_BATCH_MAX 96
typedef struct OctPackJob {
  int mode, cout, cin, reserved;
  const float* w;   /* torch-layout fp32 weight (device) */
  void* wpacked;    /* oct_packed_weight_elems(...) elements of dtype (device) */
} OctPackJob;
int oct_pack_weights_batch(int dtype, int count, const OctPackJob* jobs, void* stream);
int oct_conv_forward(const OctConvDesc* d, const OctConvArgs* a, void* stream);

/* Weight gradient (replaces the autograd of the same two modules).
 *   3x3   : taps=9, dy plain  -> dwp[tap][cout][ktot]
 *   deconv: taps=1, dy_mode=OCT_IN_S2D (dU at 2H x 2W, cout = 4*Cout) -> dwp[0][(dydx,co)][ci]
 * dwp must be zeroed by the caller; partial sums are accumulated with fp32 atomics.           */
typedef struct OctWgradDesc {
  int dtype;
  int n, h, w;
  int c0, c1;
  int cout;
  int taps;
  int xform0, xform1;
  int dy_mode;
  int kh, kw;        /* as in OctConvDesc; dwp is [kh*kw][cout][ktot] */
  int depth;         /* D > 0: volumes of D slices (see OctConvDesc).  One launch computes the weight gradient of ONE depth
                      * tap: the input tile is read from slice d + in_img_shift (zero outside the volume)              */
  int in_img_shift;  /* -1, 0, +1 = kd - 1 (Conv3d); 0 otherwise; OCT_IMG_SHIFT_ALL: every depth tap in one launch, dwp =
                      * [3][taps][cout][cin] (only where oct_conv_wgrad_all_depth_taps_ok says so)                    */
  int dy_img_mul, dy_img_add; /* dy_mode S2D only, 0,0 = identity: dY is gathered from image img*mul + add
                      * (ConvTranspose3d weight gradient: two launches, mul = 2, add = kd)                               */
  int partials;      /* 0: partial sums of the workgroups meet in dwp through fp32 atomics (dwp zeroed by the caller; the
                      * summation order, hence the last bits, vary from run to run).  1: DETERMINISTIC two-stage reduction --
                      * every workgroup writes its partial sums with plain stores into its own slab of
                      * dwp[oct_conv_wgrad_partials(desc)][taps][cout][ktot] (nothing to zero), and the unpack pass
                      * (OctUnpackJob.nparts) adds the slabs in index order.  Same for the bias gradient through
                      * OctWgradArgs.dbias_partials + oct_reduce_bias_partials.                                          */
} OctWgradDesc;
typedef struct OctWgradArgs {
  const void* x0; const void* x1;
  const float* scale0; const float* shift0;
  const float* scale1; const float* shift1;
  const void* dy;
  float* dwp;
  float* dbias; /* optional: += sum over pixels of dy per real output channel (bias gradient); caller zeroes */
  /* optional fused BatchNorm-backward apply (first layer only, OCT_E_INVALID elsewhere): when dy_coef != NULL,
   * `dy` holds dA and the kernel forms dy = coef0*[y*scale+shift>0]*dA + coef1*y + coef2 on the fly */
  const void* dy_y; const float* dy_coef; const float* dy_scale; const float* dy_shift;
  float* dbias_partials; /* partials mode with dbias != NULL: [slabs][cout] fp32 (GEMM rows, i.e. 4*Cout for the deconv) */
} OctWgradArgs;
int oct_conv_wgrad(const OctWgradDesc* d, const OctWgradArgs* a, void* stream);
/* slabs a launch of this descriptor writes when partials = 1 (host query, never fails; >= 1) */
int oct_conv_wgrad_partials(const OctWgradDesc* d);
/* dbias[c] (+)= sum over slabs, in order, of the `rows / channels` GEMM rows that map to channel c (rows = 4*Cout for the
 * transposed convolution, else Cout)                                                                     */
int oct_reduce_bias_partials(const float* part, int nparts, int rows, int channels, float* dbias, int accumulate,
                             void* stream);
/* 1 when oct_conv_wgrad accepts dy_coef (fused BatchNorm-backward apply) for this descriptor, else 0: the
 * caller then materialises dY with oct_bn_bwd_apply first.  Host-only query, never fails.               */
int oct_conv_wgrad_fused_apply_ok(const OctWgradDesc* d);
/* 1 when oct_conv_wgrad accepts in_img_shift = OCT_IMG_SHIFT_ALL for this descriptor (nn.Conv3d(1, F, 3) weight gradient,
 * engine3d: all 27 taps in one pass over dY instead of three), else 0.  Host-only query, never fails.    */
int oct_conv_wgrad_all_depth_taps_ok(const OctWgradDesc* d);
/* dwp -> torch-layout gradient.  mode: OCT_PACK_CONV_FPROP (grad[co][ci][tap]),
 * OCT_PACK_DECONV_FPROP (grad[ci][co][dydx]) or OCT_PACK_1X1_FPROP (grad[co][ci]).
 * accumulate != 0: grad += */
int oct_unpack_wgrad(int mode, const float* dwp, float* grad, int cout, int cin, int accumulate, void* stream);
/* dwp[kh*kw][cout][cin] -> grad (Cout,Cin,kh,kw) for any kernel size */
int oct_unpack_wgrad_kk(const float* dwp, float* grad, int cout, int cin, int kh, int kw, int accumulate, void* stream);
/* 3-D: OCT_PACK_CONV3D_FPROP   dwp[3][9][cout][cin] (three launches, one slab per depth tap) -> grad (Cout,Cin,3,3,3)
 *      OCT_PACK_DECONV3D_FPROP dwp[0][(dydx,co)][ci] of depth slice kdi -> grad (Cin,Cout,2,2,2)[:, :, kdi]          */
int oct_unpack_wgrad3d(int mode, const float* dwp, float* grad, int cout, int cin, int kdi, int accumulate, void* stream);
/* The same for up to OCT_PACK_BATCH_MAX gradients per launch (a whole backward pass).               */
typedef struct OctUnpackJob {
  int mode, cout, cin, accumulate;
  int nparts, reserved; /* slabs to sum in order (partials mode); 0 or 1: dwp is a single accumulated buffer */
  const float* dwp;   /* [taps][rows][kch] fp32 from oct_conv_wgrad (device) */
  float* grad;        /* torch-layout fp32 gradient (device) */
} OctUnpackJob;
int oct_unpack_wgrad_batch(int count, const OctUnpackJob* jobs, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d in training mode (nn.BatchNorm2d, YNet_2022.py:586,598; torch defaults eps=1e-5,
 * momentum=0.1, biased var to normalise, unbiased var into running_var).
 * ------------------------------------------------------------------------------------------ */
/* partials [nblocks][2][c] -> mean, invstd, scale=gamma*invstd, shift=beta-mean*scale, and
 * running_mean/var update (skipped when running_mean == NULL).  count = N*H*W.
 * conv_bias (may be NULL): bias of the conv in front of the BN (BioNet_2020.py:45-53, MGUNet_2021.py).
 * The stored y excludes it -- in training mode it cancels in the BN output and its gradient is
 * exactly zero -- so it only enters the running mean here and the eval-mode shift below.         */
int oct_bn_finalize(const float* partials, int nblocks, int c, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, const float* conv_bias,
                    void* stream);
/* eval mode: scale/shift from the running statistics */
int oct_bn_eval_coeffs(int c, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift,
                       const float* conv_bias, void* stream);

/* a = relu(y*scale+shift); p = maxpool2x2(a)  (nn.MaxPool2d(2,2), YNet_2022.py:516-522) */
int oct_bn_relu_pool_fwd(int dtype, const void* y, const float* scale, const float* shift,
                         void* pooled, int n, int h, int w, int c, void* stream);
/* a = relu(y*scale+shift) materialised (not used on the training path; for tests / export) */
int oct_bn_relu_fwd(int dtype, const void* y, const float* scale, const float* shift,
                    void* out, size_t npix, int c, void* stream);

/* g = (da + route(dpool)) * [y*scale+shift > 0], written to g; partial sums of g and
 * g*xhat (xhat = (y-mean)*invstd) per workgroup into partials[nblocks][2][c].
 * da or dpool may be NULL (not both).  dpool is at (h/2, w/2); the gradient goes to the first
 * maximum of each 2x2 window (ATen max_pool2d tie rule).  g may alias da; without dpool g may be
 * NULL (reduce only: no masked copy is written, oct_bn_bwd_apply re-derives the mask).           */
int oct_dact_bn_reduce(int dtype, const void* da, const void* dpool, const void* y,
                       const float* scale, const float* shift, const float* mean,
                       const float* invstd, void* g, float* partials, int n, int h, int w, int c,
                       void* stream);
int oct_dact_bn_reduce_blocks(int n, int h, int w, int c, int has_pool);
/* Pooled layers (nn.MaxPool2d(2, 2) behind conv + BN + ReLU, YNet_2022.py:516-522) without a stored masked gradient: when
 * oct_bn_bwd_apply_pool_ok(...) == 1, oct_dact_bn_reduce(da, dpool, ..., g = NULL) is the reduce-only pass and
 * oct_bn_bwd_apply_pool re-derives g (pool routing to the first maximum + ReLU mask, rounded to the activation dtype) and
 * writes dy = coef0*g + coef1*y + coef2 -- bit-identical to reduce(g) + oct_bn_bwd_apply(g), 11 instead of 12.5 bytes per
 * element.  da may be NULL (no skip gradient); dy may alias da.                                              */
int oct_bn_bwd_apply_pool_ok(int dtype, int n, int h, int w, int c);
int oct_bn_bwd_apply_pool(int dtype, const void* da, const void* dpool, const void* y, const float* scale,
                          const float* shift, const float* coef, void* dy, int n, int h, int w, int c, void* stream);
/* partials -> dgamma, dbeta and the three coefficients of dy = k[0]*g + k[1]*y + k[2]        */
int oct_bn_bwd_finalize(const float* partials, int nblocks, int c, double count,
                        const float* gamma, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* coef /* [3][c] */, int accumulate,
                        void* stream);
/* dy = coef0*g + coef1*y + coef2, in place over g.  With scale/shift != NULL, g holds dA and the ReLU
 * mask [y*scale+shift > 0] is applied here (pairs with oct_dact_bn_reduce(g = NULL), reduce-only). */
int oct_bn_bwd_apply(int dtype, void* g, const void* y, const float* coef, const float* scale,
                     const float* shift, size_t npix, int c, void* stream);
/* The same, written to `dst` (g is left intact: a gradient that autograd still hands to another consumer -- the
 * residual branch of SD_Layer_Net's conv_block, common.py:22-25 -- needs no copy first). dst == g is allowed.   */
int oct_bn_bwd_apply_to(int dtype, void* dst, const void* g, const void* y, const float* coef, const float* scale,
                        const float* shift, size_t npix, int c, void* stream);
/* ------------------------------------------------------------------------------------------
 * 1x1 convolution with k <= 12 output channels: Attention_block's psi = Conv2d(F_int, 1, 1) (SD_Layer_Net/common.py:79-83)
 * and the heads Conv_1x1 = Conv2d(64, output_ch, 1) (SD_Layer_Net/unet.py:38,113).  Three streaming kernels instead of
 * GEMMs padded k -> 32 (c = 8 * 2^j <= 512 with k <= 4, c <= 128 with k = 5..12 -- ReLayNet's / MGU-Net's class heads: oct_rowdot_ok; other shapes stay on oct_conv_forward / oct_conv_wgrad).
 * x, dx: [npix][c] NHWC; y, dy: [npix][k]; w, dw: [k][c] fp32 (torch (k,c,1,1)).
 *   fwd       : y = x . w^T; stats (may be NULL): [oct_rowdot_blocks][2][k] partial sum / sum of squares (oct_bn_finalize rows)
 *   bwd_data  : dx[pix][c] = sum_k dy[pix][k] * w[k][c]
 *   bwd_weight: dw[k][c] (+)= sum_pix dy[pix][k] * x[pix][c]; partials: scratch [oct_rowdot_blocks][k*c], summed in a fixed order
 * ------------------------------------------------------------------------------------------ */
int oct_rowdot_ok(int c, int k);
int oct_rowdot_blocks(size_t npix, int c);
int oct_rowdot_fwd(int dtype, const void* x, const float* w, void* y, float* stats, size_t npix, int c, int k, void* stream);
int oct_rowdot_bwd_data(int dtype, const void* dy, const float* w, void* dx, size_t npix, int c, int k, void* stream);
int oct_rowdot_bwd_weight(int dtype, const void* dy, const void* x, float* dw, float* partials, size_t npix, int c, int k,
                          int accumulate, void* stream);
/* + the bias gradient dbias[k] (+)= sum_pix dy[pix][k] from the same pass (library 0.2.2); partials: [oct_rowdot_blocks][k*c + k] */
int oct_rowdot_bwd_weight_bias(int dtype, const void* dy, const void* x, float* dw, float* dbias, float* partials, size_t npix, int c,
                               int k, int accumulate, void* stream);

/* per-channel sum over pixels of an NHWC tensor (bias gradient of ConvTranspose2d) */
int oct_channel_sum(int dtype, const void* x, float* out, size_t npix, int c, int accumulate,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Head: 1x1 conv + Softmax2d (YNet_2022.py:543-546,569) fused with the loss head the reference
 * lacks: CE = nll_loss(log p) and soft Dice.
 * ------------------------------------------------------------------------------------------ */
typedef struct OctHeadDesc {
  int dtype;
  int n, h, w;
  int feat;      /* input channels of the 1x1 conv (init_features) */
  int classes;   /* <= OCT_MAX_CLASSES */
} OctHeadDesc;
#define OCT_MAX_CLASSES 16
#define OCT_HEAD_LOSS_SLOTS (2 + 3 * OCT_MAX_CLASSES)
int oct_head_blocks(const OctHeadDesc* d);
/* forward: y (NHWC dtype) -> probs (NCHW fp32, may be NULL), argmax (int64 [n,h,w], may be NULL);
 * when target != NULL also per-workgroup loss partials [blocks][OCT_HEAD_LOSS_SLOTS].         */
int oct_head_forward(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                     const float* w /* [classes][feat] */, const float* b, const int64_t* target,
                     float* probs, int64_t* argmax, float* logits /* NCHW fp32, may be NULL */,
                     double* loss_partials, void* stream);
/* reduces the partials: loss_out[0..2] = total, ce, dice; dice_coef[2][classes] for backward  */
int oct_head_loss_finalize(const OctHeadDesc* d, const double* loss_partials, int nblocks,
                           float w_ce, float w_dice, float dice_eps, float* loss_out,
                           float* dice_coef, void* stream);
/* d(loss)/d(logits) as an NHWC tensor of the activation dtype [n,h,w,classes]: recomputes the
 * logits and the softmax from y, then either dl = w_ce*(p - onehot)/N + p*(dp - <p,dp>) with the
 * Dice term dp_c = A_c*onehot_c + B_c taken from dice_coef (fused loss path; dice_coef may be
 * NULL), or dl = p*(dprobs - <p,dprobs>) for an explicit dprobs (NCHW fp32, autograd path).
 * The rest of the head backward is the generic 1x1 machinery: oct_conv_forward (taps=1, weights
 * packed OCT_PACK_1X1_DGRAD) gives dA, oct_conv_wgrad (taps=1) gives dW, oct_channel_sum db.  */
int oct_head_dlogits(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                     const float* w, const float* b, const int64_t* target, const float* dice_coef,
                     float w_ce, const float* dprobs, void* dlogits, void* stream);

/* Fused head backward (feat == 32 only, else OCT_E_INVALID -> use the pieces above): in one pass over
 * y it writes dlogits (optional, NHWC dtype; needed by oct_conv_wgrad for dW), dA = W^T dlogits
 * (NHWC dtype, [n,h,w,feat], UNMASKED: pair it with oct_bn_bwd_apply(scale, shift)), the BN-backward
 * partial sums of the masked gradient [oct_head_blocks][2][feat], the bias gradient (atomics into
 * dbias[classes]; caller zeroes) and -- when dweight is not NULL (classes <= 8) -- the weight gradient
 * dW[c][f] = sum dlogits[c]*relu(bn(y))[f] (atomics into dweight[classes][feat], torch layout
 * (Cout,Cin,1,1); caller zeroes), in which case dlogits may be NULL and is then never written.
 * loss_partials (needs target; may be NULL): [oct_head_blocks][OCT_HEAD_LOSS_SLOTS] rows like
 * oct_head_forward's with only the cross-entropy slot filled -- a training step whose loss has no
 * Dice term can skip the forward head pass and feed these rows to oct_head_loss_finalize.
 * bf16 with classes <= 8, a target, dweight and no dlogits runs on the matrix pipe (head_mfma.hip);
 * every other combination on the vector kernel (head.hip).  OCT_HEAD_MFMA=0 forces the latter.     */
int oct_head_backward_fused(const OctHeadDesc* d, const void* y, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* w, const float* b,
                            const int64_t* target, const float* dice_coef, float w_ce,
                            const float* dprobs, void* dlogits, void* da, float* partials,
                            float* dbias, float* dweight, double* loss_partials, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout / dtype helpers and the optimizer
 * ------------------------------------------------------------------------------------------ */
/* (B,C,H,W) fp32 -> (B,H,W,C) dtype */
int oct_nchw_to_nhwc(int dtype, const float* x, void* out, int n, int c, int h, int w, void* stream);
/* (B,H,W,C) dtype -> (B,C,H,W) fp32 (tests / export) */
int oct_nhwc_to_nchw(int dtype, const void* x, float* out, int n, int c, int h, int w, void* stream);
/* torch.optim.SGD (momentum, dampening 0, no nesterov): buf = mu*buf + g (buf = g when
 * first != 0); p -= lr*buf.  grad_scale multiplies g first (1/world_size after all-reduce).  */
int oct_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                 float weight_decay, float grad_scale, int first, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks of the reference's other U-Net families; NHWC tensors
 * of `dtype`, materialised activations.
 * ------------------------------------------------------------------------------------------ */
#define OCT_ACT_NONE 0
#define OCT_ACT_RELU 1
#define OCT_ACT_SIGMOID 2
/* out = act(y*scale[c] + shift[c] (+ res)).  Covers BN+ReLU (MGUNet_2021.py:46-55), conv bias
 * without BN (scale=1, shift=bias; MGUNet_2021.py:57-64, common.py:9), the residual sum
 * `conv(x) + init_conv` + act (common.py:21-25), BN+Sigmoid (common.py:77-81).  res may be NULL. */
int oct_affine_act_fwd(int dtype, const void* y, const float* scale, const float* shift,
                       const void* res, int act, void* out, size_t npix, int c, void* stream);
/* The same with a residual whose bias add was deferred (OCT_XF_AFFINE producer): out = act(scale*y + shift + T(res + res_shift[c])),
 * the sum rounded to the storage type first -- bit-identical to a residual that had been materialised.            */
int oct_affine_res_act_fwd(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                           const float* res_shift, int act, void* out, size_t npix, int c, void* stream);
/* dz = dout * act'(z), from the stored output: relu [out>0], sigmoid out*(1-out). dz may alias dout */
int oct_act_bwd(int dtype, const void* dout, const void* out, int act, void* dz, size_t n, void* stream);
/* nn.MaxPool2d(k) on a materialised activation (SD_Layer_Net/unet.py:85, MGUNet_2021.py:211-217);
 * Output (n, h/k, w/k, c), torch's floor mode: trailing rows / columns that no whole window covers are
 * ignored (MGR_Module's 3x3 / 5x5 pools, MGUNet_2021.py:163,168).  Backward writes all of da: dout to the
 * first maximum of each window in row-major order (ATen's tie rule), zero elsewhere.            */
int oct_maxpool_fwd(int dtype, const void* a, void* out, int n, int h, int w, int c, int k, void* stream);
int oct_maxpool_bwd(int dtype, const void* a, const void* dout, void* da, int n, int h, int w, int c,
                    int k, void* stream);
/* nn.Upsample(scale_factor=f, mode="bilinear", align_corners=True) / nn.UpsamplingBilinear2d
 * (common.py:31, MGUNet_2021.py:79,98): x (n,h,w,c) -> out (n,h*f,w*f,c); backward is the exact
 * transpose (dout at h*f x w*f -> dx at h x w), deterministic (gather, no atomics).             */
int oct_bilinear_up_fwd(int dtype, const void* x, void* out, int n, int h, int w, int c, int factor,
                        void* stream);
int oct_bilinear_up_bwd(int dtype, const void* dout, void* dx, int n, int h, int w, int c, int factor,
                        void* stream);
/* F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True) (MGR_Module.forward,
 * MGUNet_2021.py:180,184,188): x (n,h,w,c) -> out (n,ho,wo,c) for any output size; src = dst*(in-1)/(out-1),
 * a one-pixel axis reads pixel 0.  Backward is the exact transpose (gather, deterministic).     */
int oct_bilinear_resize_fwd(int dtype, const void* x, void* out, int n, int h, int w, int c, int ho, int wo,
                            void* stream);
int oct_bilinear_resize_bwd(int dtype, const void* dout, void* dx, int n, int h, int w, int c, int ho, int wo,
                            void* stream);
/* Scatter step of nn.ConvTranspose2d(kernel=s, stride=s) (MGUNet_2021.py:95, s=4): the GEMM output
 * in[n,h,w,(dy*s+dx)*cout+co] goes to out[n,h*s+dy,w*s+dx,co] (+ bias[co], may be NULL);
 * space_to_depth is its inverse, used on dOut before the weight / data gradients.               */
int oct_depth_to_space(int dtype, const void* in, const float* bias, void* out, int n, int h, int w,
                       int cout, int s, void* stream);
int oct_space_to_depth(int dtype, const void* in, void* out, int n, int h, int w, int cout, int s,
                       void* stream);
/* Attention gate product `x * psi` (SD_Layer_Net/common.py:91): out[pix,c] = x[pix,c]*p[pix];
 * backward dx = dout*p, dp[pix] = sum_c dout*x.                                                 */
int oct_gate_fwd(int dtype, const void* x, const void* p, void* out, size_t npix, int c, void* stream);
int oct_gate_bwd(int dtype, const void* dout, const void* x, const void* p, void* dx, void* dp,
                 size_t npix, int c, void* stream);

/* ReLayNet blocks (SOTAS/Lesions_Segment/ReLayNet_2017.py:133-201).
 * BasicBlock :164-168: out = prelu(bn(conv7x3(x))) with nn.PReLU()'s single learnable slope `alpha` (device float[1]):
 *   forward  out = z > 0 ? z : alpha*z,  z = y*scale[c] + shift[c]  (y = raw conv output, scale/shift from oct_bn_finalize)
 *   backward dz = dout * (z > 0 ? 1 : alpha);  dalpha[0] += sum dout*z*[z <= 0]  (caller zeroes dalpha); dz may alias dout */
int oct_affine_prelu_fwd(int dtype, const void* y, const float* scale, const float* shift, const float* alpha,
                         void* out, size_t npix, int c, void* stream);
int oct_affine_prelu_bwd(int dtype, const void* dout, const void* y, const float* scale, const float* shift,
                         const float* alpha, void* dz, float* dalpha, size_t npix, int c, void* stream);
/* The same backward WITHOUT the dz tensor (library 0.2.2): the two BatchNorm-backward passes re-derive dz = dout * (z > 0 ? 1 : alpha)
 * themselves -- oct_dact_bn_reduce_prelu (sums for oct_bn_bwd_finalize, dalpha[0] += ...; partial rows as oct_dact_bn_reduce_blocks
 * (n, h, w, c, 0)) and oct_bn_bwd_apply_prelu_to (dst = k0*dz + k1*y + k2) -- bit-identical to oct_affine_prelu_bwd followed by the
 * plain passes.  oct_prelu_bn_fused_ok: c % 8 == 0 with 256 % (c/8) == 0.                                                  */
int oct_prelu_bn_fused_ok(int dtype, int c);
int oct_dact_bn_reduce_prelu(int dtype, const void* da, const void* y, const float* scale, const float* shift, const float* alpha,
                             const float* mean, const float* invstd, float* partials, float* dalpha, int n, int h, int w, int c,
                             void* stream);
int oct_bn_bwd_apply_prelu_to(int dtype, void* dst, const void* da, const void* y, const float* coef, const float* scale,
                              const float* shift, const float* alpha, size_t npix, int c, void* stream);
/* EncoderBlock :174-179, nn.MaxPool2d(k, k, return_indices=True): out (n,h/k,w/k,c) and idx (same shape, int64) holding
 * torch's index of the winner inside its (n, c) input plane, iy*w + ix; first maximum in row-major window order.      */
int oct_maxpool_idx_fwd(int dtype, const void* a, void* out, int64_t* idx, int n, int h, int w, int c, int k, void* stream);
/* DecoderBlock :185-188, nn.MaxUnpool2d: out[n, idx[n,p,c], c] = v[n,p,c] over a zero-filled (by the caller) out of
 * `plane` = H*W pixels per image (also the backward of the pooling above); the gather is its transpose (unpool
 * backward): v[n,p,c] = x[n, idx[n,p,c], c].  npool = pooled pixels per image.  Out-of-range indices are skipped / 0. */
int oct_index_scatter(int dtype, const void* v, const int64_t* idx, void* out, int n, size_t npool, size_t plane, int c,
                      void* stream);
int oct_index_gather(int dtype, const void* x, const int64_t* idx, void* v, int n, size_t npool, size_t plane, int c,
                     void* stream);
/* The same pair by WINDOW CODE (library 0.2.2; the indices of ReLayNet's encoder -> decoder path never leave the library): code
 * (n,h/k,w/k,c) uint8 = (iy - yo*k)*k + (ix - xo*k) of the winner oct_maxpool_idx_fwd would report.  oct_window_scatter writes the
 * WHOLE (n, hp*k, wp*k, c) output (value at the code, zeros elsewhere: no zero-fill by the caller) = MaxUnpool2d forward and
 * max-pool backward; oct_window_gather = MaxUnpool2d backward.  k <= 15.                                                      */
int oct_maxpool_code_fwd(int dtype, const void* a, void* out, unsigned char* code, int n, int h, int w, int c, int k, void* stream);
int oct_window_scatter(int dtype, const void* v, const unsigned char* code, void* out, int n, int hp, int wp, int c, int k, void* stream);
int oct_window_gather(int dtype, const void* x, const unsigned char* code, void* v, int n, int hp, int wp, int c, int k, void* stream);

/* MaxPool3d(2) of the cfg5 volumetric U-Net = oct_bn_relu_pool_fwd inside every slice, then the pairwise maximum of
 * consecutive slices: p2 (nslab, 2, m) -> out (nslab, m), m contiguous elements per slice ((h/2)*(w/2)*c).  Backward:
 * dout goes to the first slice unless the second is strictly larger (torch's first-maximum rule in (d,h,w) order);
 * the 2-D routing inside the slice is oct_dact_bn_reduce's.                                                   */
int oct_depth_pool_fwd(int dtype, const void* p2, void* out, size_t nslab, size_t m, void* stream);
int oct_depth_pool_bwd(int dtype, const void* p2, const void* dout, void* dp2, size_t nslab, size_t m, void* stream);

/* ------------------------------------------------------------------------------------------
 * Metrics (Metrics/Region_based_metrics.py:3-61, Metrics/ConfusionMatrix_based_metrics.py:4-63)
 * One pass over the two masks; integer inputs are reduced exactly in 64-bit with numpy's
 * same-dtype product semantics, float inputs in fp64.
 *   out_i[6] = { sum(t*p), sum(t), sum(p), sum((1-t)*(1-p)), sum((1-t)*p), sum(t*(1-p)) }
 * elem: 0 u8/bool, 1 i32, 2 i64, 3 f32, 4 f64, 5 i8, 6 i16, 7 u16                                   */
int oct_confusion_counts(const void* y_true, const void* y_pred, int elem, size_t n,
                         int64_t* out_i /* [6] device, zeroed by callee */,
                         double* out_f /* [6] device, zeroed by callee */, void* stream);

/* Per-class variant for CLASS MAPS (integer element types only): out[c][6] holds the same six sums for the
 * one-vs-rest masks (y_true == c), (y_pred == c), c = 0..classes-1 (classes <= 16), all classes in ONE pass --
 * what the reference's functions (Region_based_metrics.py:3-61, ConfusionMatrix_based_metrics.py:4-63) compute
 * when they are called once per class on binarised maps.  Labels outside [0, classes) belong to no class.
 * scratch: 48 x uint64 device words (zeroed by the callee).                                               */
int oct_class_confusion_counts(const void* y_true, const void* y_pred, int elem, size_t n, int classes,
                               int64_t* out /* [classes][6] device */, uint64_t* scratch /* [48] device */,
                               void* stream);

/* Metrics/PixelError_based_metrics.py:3-37: out[0] = sum (double(t) - double(p))^2 (device double)      */
int oct_sqdiff_sum(const void* y_true, const void* y_pred, int elem, size_t n, double* out, void* stream);
/* Metrics/Biomarker_based_metrics.py:3-21: out[0] = sum over columns of |colsum(t) - colsum(p)| for a
 * (rows, cols) view (axis 0 = rows).  unsigned_wrap != 0 reproduces numpy's uint64 wrap-around for
 * unsigned inputs; bool masks pass elem 0 with unsigned_wrap = 0 (numpy sums bool in int64).          */
int oct_column_absdiff_sum(const void* y_true, const void* y_pred, int elem, int unsigned_wrap, size_t rows,
                           size_t cols, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCT_HIP_H */
