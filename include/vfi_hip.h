/*
 * vfi_hip.h -- C ABI of libvfi_hip.so, the MI355X (gfx950) implementation of the
 * per-frame inference hot path of "Fusion Method for Video Frame Interpolation":
 * steerable-pyramid phase decomposition + PhaseNet, AdaCoF deformable sampling and
 * the FusionNet blend.
 *
 * Conventions (every entry point):
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - Returns VFI_OK (0) or a negative vfi_status; never throws, never exits.
 *     vfi_status_string() names a code, vfi_last_error() gives the detail of the last
 *     failure on the calling thread.
 *   - Every data pointer is a DEVICE pointer owned by the caller (the Python host passes
 *     torch-allocated HBM); tensors are dense fp32, NCHW unless stated otherwise.
 *   - `stream` is a hipStream_t passed as void* (the host passes torch's current HIP
 *     stream).  Calls only enqueue work; none synchronises the device, allocates or
 *     frees device memory, so a sequence of calls can be captured into a hipGraph.
 *     Workspace is caller-provided or owned by an explicit plan object.
 *   - Process-wide state is limited to idempotent per-device launch setup (the dynamic-LDS
 *     attribute of three kernels, the CU count), tuning switches read once from the
 *     environment (VFI_CONV_WINOGRAD, VFI_CONV_WINOGRAD4, VFI_CONV_STREAM1X1, VFI_ADACOF_VARIANT, VFI_ADACOF_MARGIN; VFI_CONV_WINOGRAD4M at every call) and the
 *     thread-local last-error string; everything else lives in explicit plan objects,
 *     whose tables are immutable after creation and whose workspace belongs to ONE stream
 *     at a time (frames in flight on different streams use different plans).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef VFI_HIP_H
#define VFI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFI_ABI_VERSION 1

typedef void *vfi_stream_t; /* hipStream_t */

typedef enum vfi_status {
    VFI_OK = 0,
    VFI_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, unsupported value */
    VFI_ERR_SHAPE = -2,       /* shape relation violated (mirrors the reference's asserts) */
    VFI_ERR_LAUNCH = -3,      /* HIP launch / runtime error */
    VFI_ERR_UNSUPPORTED = -4, /* valid request this build has no kernel for */
    VFI_ERR_FFT = -5,         /* reserved (round 1 linked an FFT library; the transforms are hand-written kernels now) */
    VFI_ERR_NOMEM = -6        /* host or device allocation failed (plan creation only) */
} vfi_status;

int vfi_abi_version(void);
const char *vfi_status_string(int status);
const char *vfi_last_error(void);
/* Test aid (no reference counterpart): fills the LDS of every CU with NaNs, so that the kernels enqueued next on `stream`
 * start on NaN-filled LDS -- a kernel whose result depends on LDS it has not written itself then shows deterministically
 * (tests/test_pyramid_gpu.py::test_results_do_not_depend_on_stale_lds). */
int vfi_debug_poison_lds(vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * AdaCoF deformable sampling
 * ---------------------------------------------------------------------------------- */

/* Replaces FunctionAdaCoF.forward + kernel_AdaCoF_updateOutput
 * (src/adacof/cupy_module/adacof.py:313-361 and :6-65).
 *   input    (N, C, Hin, Win)   already padded by the caller, as in the reference
 *   weight, offset_i, offset_j  (N, F*F, H, W)
 *   output   (N, C, H, W)
 * Shape relation of adacof.py:326-327 is enforced: Hin - ((F-1)*dilation + 1) == H - 1
 * (same for W), otherwise VFI_ERR_SHAPE.  Sampling semantics are the reference's:
 * (int) truncation of the offsets, each of the four taps clamped to the input
 * independently, bilinear weights from the un-clamped fractional parts. */
int vfi_adacof_forward(const float *input, const float *weight, const float *offset_i,
                       const float *offset_j, float *output, int N, int C, int Hin, int Win,
                       int H, int W, int F, int dilation, vfi_stream_t stream);

/* One fused pass over both sampling sides of AdaCoFNet.forward
 * (src/fusion_net/fusion_adacofnet.py:195-213; plain variant src/adacof/models/adacofnet.py:195-199):
 *   t1 = AdaCoF(ReplicationPad(frame0), W1, A1, B1);  t2 = AdaCoF(ReplicationPad(frame2), W2, A2, B2)
 *   frame1 = Occ * t1 + (1 - Occ) * t2
 *   mask   = clip(max(sum Var(W1; A1, B1), sum Var(W2; A2, B2)), 0, 20) / 20
 * The replication pad of (F-1)*dilation/2 pixels is folded into the tap clamp, so
 * frame0/frame2 are the UN-padded (N, C, H, W) frames.  w*, a*, b* are (N, F*F, H, W),
 * occ is (N, 1, H, W).  out_t1 / out_t2 / out_mask may be NULL (not produced);
 * out_frame is required.  (F-1)*dilation must be even (as in the reference, where
 * kernel_pad = int((F-1)*dilation/2)). */
int vfi_adacof_fused(const float *frame0, const float *frame2,
                     const float *w1, const float *a1, const float *b1,
                     const float *w2, const float *a2, const float *b2, const float *occ,
                     float *out_t1, float *out_t2, float *out_frame, float *out_mask,
                     int N, int C, int H, int W, int F, int dilation, vfi_stream_t stream);

/* Same computation on PIXEL-INTERLEAVED frames (N, H, W, 4) = (r, g, b, unused), as written by
 * vfi_adacof_prepare(..., rgbx=1): one 16-byte gather per bilinear corner instead of three 4-byte ones
 * (the kernel is bound by gather issue, not by HBM).  Outputs stay planar (N, 3, H, W).
 * weights_are_logits != 0: w1 / w2 are the pre-softmax outputs of Subnet_weight and the channel softmax
 * (fusion_adacofnet.py:56) is folded into the accumulation, so the normalised weights never touch HBM. */
int vfi_adacof_fused_rgbx(const float *frame0_rgbx, const float *frame2_rgbx,
                          const float *w1, const float *a1, const float *b1,
                          const float *w2, const float *a2, const float *b2, const float *occ,
                          float *out_t1, float *out_t2, float *out_frame, float *out_mask,
                          int N, int H, int W, int F, int dilation, int weights_are_logits, vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dense convolution on the fp32 matrix cores (exact fp32: v_mfma_f32_16x16x4_f32 in the Winograd
 * kernels that take the 3x3 layers, v_mfma_f32_32x32x2_f32 in the direct 1x1 / 5x5 kernels)
 * ---------------------------------------------------------------------------------- */

typedef enum vfi_act {
    VFI_ACT_NONE = 0,
    VFI_ACT_RELU = 1,
    VFI_ACT_ELU = 2, /* alpha = 1 */
    VFI_ACT_TANH = 3,
    VFI_ACT_SIGMOID = 4
} vfi_act;

typedef enum vfi_pad { VFI_PAD_ZERO = 0, VFI_PAD_REFLECT = 1 } vfi_pad;

/* Number of floats of the packed weight buffer for an (Cout, Cin, KS, KS) filter bank, -1 on bad arguments.
 * The buffer is opaque to the caller: [round_up(Cin,8)][KS*KS][round_up(Cout,32)] (zero filled) and, for KS = 3,
 * the Winograd F(2x2,3x3) transformed weights G g G^T as [round_up(Cin,8)][4][round_up(Cout,32)][4] behind it. */
long long vfi_conv2d_packed_floats(int Cout, int Cin, int KS);

/* Packs torch-layout (Cout, Cin, KS, KS) weights once at model-load time.  `scale` (Cout) or NULL
 * multiplies each output channel (BatchNorm2d in eval mode folded into the preceding conv:
 * reference src/phase_net/phase_net.py:191-193). */
int vfi_conv2d_pack(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin, int KS,
                    vfi_stream_t stream);

/* y = act(conv2d(x, w) + bias) (+ residual): stride 1, padding (KS-1)/2 in `pad_mode`, KS in {1,3,5}.
 * Replaces every nn.Conv2d (+ following BatchNorm / ReLU / ELU / Tanh / Sigmoid, + the additive U-Net
 * skip) on the path: reference src/phase_net/phase_net.py:190-200, src/fusion_net/fusion_adacofnet.py:18-155,
 * src/fusion_net/fusion_net.py:24-41,56-69.
 * KS = 3 runs Winograd on the fp32 matrix cores -- F(4x4,3x3) (16x64-pixel x 32-channel work items, one workgroup per
 * CU) or F(2x2,3x3) (8x32-pixel items, two per CU, K split for few long items), whichever a cost model of the two kernels
 * (rounds of resident workgroups x time per item, fitted to a per-layer A/B of the 1080p frame) puts ahead;
 * vfi_conv2d_algo tells which kernel a layer gets; same result up to fp32 rounding of the transforms: rms 3e-7 resp.
 * 2e-6 of the output rms, <= 3e-5 resp. 1e-4 at worst on unit-variance data (2.5e-4 on the statistics of trained PhaseNet
 * weights with folded BatchNorm).  Environment (A/B aids): VFI_CONV_WINOGRAD4=0 keeps F(2x2) everywhere, =2 sends every
 * plain 3x3 layer to F(4x4); VFI_CONV_WINOGRAD4M=0 (read at every call) runs F(4x4) on round 3's kernel -- 16 channels per
 * wave, two waves per SIMD; bit-identical outputs -- instead of the 32-channel-per-wave one; VFI_CONV_WINOGRAD=0 selects
 * the direct kernel.  KS = 5: the direct one.  KS = 1: the direct one, except layers with at most 16 output channels on an even
 * number of >= 4096 pixels without a residual (8-byte-aligned operands): a streaming kernel that reads the input once at the HBM
 * rate (vector ALU, fp32 FMA chain over the input channels in order; VFI_CONV_STREAM1X1=0 keeps the direct kernel).
 *   x        (N, Cin, H, W); consecutive samples are x_bstride floats apart, so x may be a channel
 *            slice of a wider tensor (no concat / split copies)
 *   residual NULL or (N, Cout, H, W) with stride res_bstride, added AFTER the activation
 *   y        (N, Cout, H, W) with stride y_bstride
 *   workspace / workspace_floats  optional scratch (NULL / 0 = none).  When the launch would otherwise fill only a
 *            fraction of a round of resident workgroups (deep U-Net levels: few, long workgroups) the channel
 *            loop is split over several workgroups whose partial sums go through the workspace and are reduced
 *            in a fixed order (deterministic).  16 * N*Cout*H*W floats always suffice; results do not depend on
 *            whether a workspace is given beyond fp32 summation order. */
enum { VFI_CONV_ALGO_DIRECT = 0, VFI_CONV_ALGO_WINOGRAD2 = 1, VFI_CONV_ALGO_WINOGRAD4 = 2, VFI_CONV_ALGO_STREAM1X1 = 3 };
/* Which kernel vfi_conv2d / vfi_conv2d_pool2 runs for a layer on the current device (>= 0: VFI_CONV_ALGO_*; < 0: status):
 * direct implicit GEMM, Winograd F(2x2,3x3), Winograd F(4x4,3x3), or the streaming kernel of 1x1 layers with at most 16 output
 * channels (an even number of >= 4096 pixels, no residual; operands assumed 8-byte aligned).  For profiling labels and flop
 * counts: the selection rule lives in the library only. */
int vfi_conv2d_algo(int N, int Cin, int H, int W, int Cout, int KS, int has_residual, int pooled, int act);

int vfi_conv2d(const float *x, long long x_bstride, const float *packed_w, const float *bias,
               const float *residual, long long res_bstride, float *y, long long y_bstride, int N, int Cin,
               int H, int W, int Cout, int KS, int pad_mode, int act, float *workspace,
               long long workspace_floats, vfi_stream_t stream);

/* vfi_conv2d followed by 2x2 / stride-2 pooling of its result (AvgPool2d after every encoder block of the U-Net,
 * src/fusion_net/fusion_adacofnet.py:76-89,116-126; MaxPool2d in FusionNet, src/fusion_net/fusion_net.py:41,59):
 * y as vfi_conv2d (no residual), pooled (N, Cout, H/2, W/2) = pool(y).  For KS = 3 ReLU layers the pooled value is
 * written by the convolution's epilogue (the lane that holds a 2x2 output block averages / maximises it); other
 * layers run the pooling pass afterwards.  Same results as the two calls. */
int vfi_conv2d_pool2(const float *x, long long x_bstride, const float *packed_w, const float *bias, float *y,
                     long long y_bstride, float *pooled, long long pooled_bstride, int is_max, int N, int Cin, int H, int W,
                     int Cout, int KS, int pad_mode, int act, float *workspace, long long workspace_floats,
                     vfi_stream_t stream);

/* conv2d( nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)(x_lowres) ) in one launch: the
 * `Upsample -> Conv2d` pairs of KernelEstimation (src/fusion_net/fusion_adacofnet.py:28-33 and the heads'
 * tails :41-43,53-56,67-70).  x_lowres is (N, Cin, H/2, W/2); H, W are the OUTPUT size (even); the upsampled
 * tensor is never written to HBM (the tile loader interpolates it).  KS = 3, zero padding only. */
int vfi_conv2d_upsample2x(const float *x_lowres, long long x_bstride, const float *packed_w, const float *bias,
                          const float *residual, long long res_bstride, float *y, long long y_bstride, int N,
                          int Cin, int H, int W, int Cout, int KS, int pad_mode, int act, float *workspace,
                          long long workspace_floats, vfi_stream_t stream);

/* conv2d over the virtual concat [ resize(x2) | x[:, prefix_channels:] ]: the first prefix_channels input channels
 * are torch's bilinear resize (align_corners=False) of x2 (N, prefix_channels, Hs, Ws) to (H, W), the remaining
 * Cin - prefix_channels channels are read from x (N, Cin, H, W) -- whose first prefix_channels channels are never
 * touched.  This is PhaseNet's block input `cat(Upsample(feature), phase, amp, Upsample(prediction))`
 * (src/phase_net/phase_net.py:138-141, channels permuted at pack time) without writing the resized maps.
 * KS = 3, Cout % 64 == 0, prefix_channels % 8 == 0. */
int vfi_conv2d_resized_prefix(const float *x, long long x_bstride, const float *x2, long long x2_bstride,
                              int prefix_channels, int Hs, int Ws, const float *packed_w, const float *bias,
                              float *y, long long y_bstride, int N, int Cin, int H, int W, int Cout, int KS,
                              int pad_mode, int act, float *workspace, long long workspace_floats,
                              vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Glue between the convolutions (HBM-bound; batch strides as for vfi_conv2d)
 * ---------------------------------------------------------------------------------- */

/* AdaCoFNet.forward prologue (src/fusion_net/fusion_adacofnet.py:182-193, src/adacof/utility.py:86-87):
 * reflect-pads both (N,3,H,W) frames at the bottom/right to (Hp,Wp) (multiples of 32), writes the padded
 * raw frames pad0/pad2 for the sampler -- planar (N,3,Hp,Wp) when rgbx == 0, pixel-interleaved (N,Hp,Wp,4)
 * when rgbx == 1 (4th float left untouched) -- and x6 = cat(pad0 - mean, pad2 - mean) (N,6,Hp,Wp). */
int vfi_adacof_prepare(const float *frame0, const float *frame2, float *pad0, float *pad2, float *x6, int N,
                       int H, int W, int Hp, int Wp, int rgbx, vfi_stream_t stream);

/* 2x2 / stride-2 pooling, floor output size.  is_max=0: AvgPool2d (fusion_adacofnet.py:76-89);
 * is_max=1: MaxPool2d (src/fusion_net/fusion_net.py:41,59). */
int vfi_pool2(const float *x, long long x_bstride, float *y, long long y_bstride, int N, int C, int H, int W,
              int is_max, vfi_stream_t stream);

/* torch bilinear interpolation to (Hout,Wout): align_corners=1 is nn.Upsample(scale_factor=2,
 * align_corners=True) of fusion_adacofnet.py:30,42,54,68; align_corners=0 is nn.Upsample(size) of
 * src/phase_net/phase_net.py:138-139 and nn.Upsample(scale_factor=2) of fusion_net.py:42.
 * relu_input=1 applies ReLU to the source first and residual (NULL or (N,C,Hout,Wout)) is added
 * to the result: `deconvolution(relu(x)) + s` of fusion_net.py:65-66 in one pass. */
int vfi_resize_bilinear(const float *x, long long x_bstride, const float *residual, long long res_bstride,
                        float *y, long long y_bstride, int N, int C, int Hin, int Win, int Hout, int Wout,
                        int align_corners, int relu_input, vfi_stream_t stream);

/* Tail of `Upsample(x2, bilinear, align_corners=True) -> Conv2d(C, 1, 3, padding=1) -> act` (Subnet_occlusion,
 * fusion_adacofnet.py:68-70) after the channel reduction has been done at LOW resolution: taps_lowres (N,9,Hs,Ws)
 * holds m_t = sum_c w[0][c][t] * x_c (a 1x1 vfi_conv2d with the 3x3 filter's taps as output channels);
 * out (N,1,2Hs,2Ws) = act(bias + sum_t U(m_t) shifted by tap t, zero outside).  Exact by linearity of both maps;
 * replaces a 64->1 conv at full resolution that would waste 31/32 of each matrix-core tile. */
int vfi_upsample2x_tapsum(const float *taps_lowres, float *out, int N, int Hs, int Ws, float bias, int act,
                          vfi_stream_t stream);

/* Softmax over the channel axis of (N,C,HW) (Subnet_weight, fusion_adacofnet.py:56).  In place allowed. */
int vfi_softmax_channels(const float *x, long long x_bstride, float *y, long long y_bstride, int N, int C, int HW,
                         vfi_stream_t stream);

/* dst[n][e] = src[n][e] / div_per_sample[n] * mul for e < count (div_per_sample may be NULL).
 * Slice copies into concat buffers; phase/pi and amplitude/max of PhaseNet.normalize_vals
 * (src/phase_net/phase_net.py:61-70). */
int vfi_affine_slice(const float *src, long long src_bstride, float *dst, long long dst_bstride, int N,
                     long long count, const float *div_per_sample, float mul, vfi_stream_t stream);

/* out_max[n] = max(x[n][0..count)) + eps (phase_net.py:55,69).  workspace_u32: N 32-bit words. */
int vfi_batch_max(const float *x, long long x_bstride, int N, long long count, float eps, float *out_max,
                  void *workspace_u32, vfi_stream_t stream);

/* PhaseNet per-level outputs (phase_net.py:155-168 + reverse_normalize :80-90), pred (N,8,HW) from the
 * block's tanh head, amp_in (N,8,HW) the normalised input amplitudes:
 *   phase_out (N,4,HW) = pred[:,0:4]*pi ; amp_out (N,4,HW) = (b*amp_in[:,4:8] + (1-b)*amp_in[:,0:4])*max_amp[n],
 *   b = (pred[:,4:8]+1)/2. */
int vfi_phasenet_emit(const float *pred, long long pred_bstride, const float *amp_in, long long amp_bstride,
                      const float *max_amp, float *phase_out, float *amp_out, int N, int HW, vfi_stream_t stream);

/* The prediction head of one PhaseNet level in one pass (phase_net.py:149-168, 190-207): pred (N,8,H*W) = tanh(1x1 conv of feat
 * (N,Cin,H*W) with packed_w / bias: the weights of vfi_conv2d_pack for Cout = 8, KS = 1), written out (the next level resizes
 * it), and vfi_phasenet_emit's outputs from the same registers.  Same results as vfi_conv2d(act = tanh) followed by
 * vfi_phasenet_emit, which is what small or odd-sized levels still run. */
int vfi_phasenet_predict(const float *feat, long long feat_bstride, const float *packed_w, const float *bias,
                         const float *amp_in, long long amp_bstride, const float *max_amp, float *pred, long long pred_bstride,
                         float *phase_out, float *amp_out, int N, int Cin, int H, int W, vfi_stream_t stream);

/* phase_net.py:113-116 + :96-98: low_out (N,HW) = (a*low_in[:,0] + (1-a)*low_in[:,1])*max_low[n], a = (pred+1)/2. */
int vfi_phasenet_emit_low(const float *pred, long long pred_bstride, const float *low_in, long long low_bstride,
                          const float *max_low, float *low_out, int N, int HW, vfi_stream_t stream);

/* FusionNet tail (fusion_net.py:70-77): y = clamp(base + tanh(x), 0, 1) over `count` floats. */
int vfi_tanh_residual_clamp(const float *x, const float *base, float *y, long long count, vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Complex steerable pyramid (frequency domain), scale_factor-generalised
 * ---------------------------------------------------------------------------------- */

/* Opaque plan: level geometry, mask tables, the tables of the hand-written FFT kernels (twiddles, Bluestein chirps and
 * filters; no FFT library is linked) and workspace for one (H, W, height, nbands, scale_factor).  Replaces the
 * per-frame construction of `SCFpyr_PyTorch(height, nbands, scale_factor, device)` (reference
 * src/train/pyramid.py:28-33, rebuilt for every frame at src/fusion_net/interpolate_twoframe.py:124-129).  Creation
 * allocates device memory and synchronises; everything else only enqueues.  A plan's workspace belongs to one stream
 * at a time: frames in flight use one plan each. */
typedef struct vfi_pyr_plan vfi_pyr_plan;

enum {
    VFI_PYR_BAND_MAJOR = 1,    /* default plane of image d, band b is d*nbands+b; with this flag b*N+d */
    VFI_PYR_COMPLEX_COEFF = 2  /* bands are interleaved (re, im) coefficients instead of (phase, amplitude) */
};

/* Every level size must be transformable: any length with prime factors 2, 3, 5 up to 8704, any other length up to 4096
 * (frames up to 3840x2160 qualify; VFI_ERR_UNSUPPORTED otherwise).  Lengths the wave-private register engine has a
 * configuration for (the level sizes of 256x256 ... 1920x1080 frames among them) run on it, the others on the generic
 * LDS engine; results agree to fp32 rounding. */
int vfi_pyr_plan_create(int H, int W, int height, int nbands, double scale_factor, int max_images,
                        vfi_pyr_plan **out);
int vfi_pyr_plan_destroy(vfi_pyr_plan *plan);
/* Size of band level `level` (0 = finest .. height-3) or of the low residual (level == height-2):
 * ceil(H / scale_factor^level) x ceil(W / scale_factor^level). */
int vfi_pyr_plan_level_size(const vfi_pyr_plan *plan, int level, int *h, int *w);

/* Analysis followed by synthesis of an UNMODIFIED subset of the pyramid is a real, radially symmetric linear
 * filter (the four orientation masks form a partition of unity):
 *   inv_filter(keep(filter(x))) = real(ifft2(fft2(x) * G)),
 *   G = [keep_high] hi0^2 + lo0^2 * ( sum_{k in level_mask} (prod_{j<k} lomask_j^2) himask_k^2 + [keep_low] prod_j lomask_j^2 ).
 * vfi_pyr_plan_prepare_filter builds G once (allocates; returns an id), vfi_pyr_apply_filter applies it to
 * N images (N,H,W) with one R2C, one multiply and one C2R.  Replaces
 * `pyr.inv_filter(get_last_value_levels(vals, 1))` of src/fusion_net/interpolate_twoframe.py:205-209
 * (24 band FFTs forward and back per call) wherever the kept values are not modified in between. */
int vfi_pyr_plan_prepare_filter(vfi_pyr_plan *plan, unsigned long long level_mask, int keep_high, int keep_low,
                                int *filter_id);
int vfi_pyr_apply_filter(vfi_pyr_plan *plan, int filter_id, const float *img, int N, float *out, vfi_stream_t stream);
/* out = real(ifft2(fft2(img_a) * G_a + fft2(img_b) * G_b)): the sum of two such filters applied to two image sets of N
 * images each (2N <= the plan's max_images), one C2R.  Replaces the "baseline" mix of
 * src/fusion_net/interpolate_twoframe.py:288-322 -- filter(lab_ada), filter(lab_phase), a DecompValues assembled from the
 * fine levels + low residual of one and the coarse levels + high residual of the other, inv_filter -- which only moves
 * UNMODIFIED values: two full analyses and one full synthesis become two R2C, one multiply-add and one C2R. */
int vfi_pyr_apply_filter_pair(vfi_pyr_plan *plan, int filter_a, const float *img_a, int filter_b, const float *img_b, int N,
                              float *out, vfi_stream_t stream);

/* Pyramid.filter = SCFpyr_PyTorch.build + coeff_to_values (src/train/pyramid.py:35-39,48-78).
 *   img    (N, H, W)
 *   high   (N, H, W) or NULL;  low (N, hL, wL) or NULL
 *   phase[k], amp[k] (k < height-2; host arrays of device pointers): band level k, finest first.
 *          The plane (h_k*w_k floats) of image d, band b starts at plane index
 *          plane_index[k*N + d] + b*band_stride from phase[k] / amp[k]; plane_index == NULL gives the
 *          reference's per-image layout (N*nbands, 1, h, w) with index d*nbands+b.  A custom table lets
 *          the bands land directly in PhaseNet's (colour, frame*nbands+band) block-input buffers
 *          (separate_vals + get_concat_layers_inf, src/train/utils.py:47-127, without the copies).
 *   phase_scale multiplies the phase (1, or 1/pi for PhaseNet.normalize_vals, src/phase_net/phase_net.py:64)
 *   level_mask  bit k set = produce band level k (clear bits skip that level's inverse FFTs)
 *   flags       VFI_PYR_* */
int vfi_pyr_analyze(vfi_pyr_plan *plan, const float *img, int N, float *high, float *const *phase,
                    float *const *amp, const int *plane_index, float *low, float phase_scale,
                    unsigned long long level_mask, int flags, vfi_stream_t stream);
/* The same, plus the maxima PhaseNet.normalize_vals needs (src/phase_net/phase_net.py:55: max over the 8 channels
 * x pixels of each colour, per level), reduced inside the row kernel that writes the amplitudes (wave shuffles, one
 * atomic per wave): amp_max[k * groups + g] = max amplitude of level k over the images d with d % groups == g,
 * + eps.  amp_max must hold (height-2) * groups floats; levels outside level_mask get eps. */
int vfi_pyr_analyze_max(vfi_pyr_plan *plan, const float *img, int N, float *high, float *const *phase,
                        float *const *amp, const int *plane_index, float *low, float phase_scale,
                        unsigned long long level_mask, int flags, float *amp_max, int groups, float eps,
                        vfi_stream_t stream);

/* Pyramid.inv_filter = values_to_coeff + SCFpyr_PyTorch.reconstruct (src/train/pyramid.py:41-46,85-112).
 * high / low may be NULL (treated as zeros, e.g. PhaseNet's high_level, src/phase_net/phase_net.py:127-128);
 * band levels whose level_mask bit is clear are treated as zeros (get_last_value_levels /
 * get_first_value_levels, src/train/utils.py:242-320, without materialising the zero tensors). */
int vfi_pyr_synthesize(vfi_pyr_plan *plan, const float *high, const float *const *phase,
                       const float *const *amp, const int *plane_index, const float *low,
                       unsigned long long level_mask, int flags, float *img, int N, vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Image-space stages the reference runs on the host CPU (skimage / scipy round trips)
 * ---------------------------------------------------------------------------------- */

/* rgb2lab_single / lab2rgb_single (src/train/transform.py:17-25,40-49): skimage D65/2deg conversion
 * with the reference's scaling L/100, (a+128)/255, (b+128)/255.  (N,3,HW) planar. */
int vfi_rgb2lab(const float *rgb, float *lab, int N, int HW, vfi_stream_t stream);
int vfi_lab2rgb(const float *lab, float *rgb, int N, int HW, vfi_stream_t stream);

/* out (N,HW) = mean_c a  (b == NULL)   or   |mean_c a - mean_c b|,  times scale; clamp01 is a flag word:
 * bit 0 clamps to [0,1], bit 1 keeps the SIGNED difference (no abs)
 * (src/fusion_net/interpolate_twoframe.py:207-211,219-220: `.mean(1)`, abs, `*100`/`*30`, clamp). */
int vfi_channel_mean_diff(const float *a, const float *b, float *out, int N, int C, int HW, float scale,
                          int clamp01, vfi_stream_t stream);

/* out = |x - y| * scale, or |x| * scale when y is NULL (clamped to [0,1] if clamp01): subtract_values (src/train/utils.py:322-346) and
 * `abs(freq_diff - median) * 5` clamp (interpolate_twoframe.py:223-224). */
int vfi_absdiff(const float *x, const float *y, float *out, long long count, float scale, int clamp01,
                vfi_stream_t stream);

/* scipy.ndimage.gaussian_filter(x, sigma, truncate=truncate) per (H,W) image, mode='reflect'
 * (interpolate_twoframe.py:212-213 uses sigma=5, default truncate=4 -> 41 taps).  tmp: N*H*W floats. */
int vfi_gaussian_filter(const float *x, float *tmp, float *y, int N, int H, int W, float sigma, float truncate,
                        vfi_stream_t stream);

/* scipy.ndimage.median_filter(x, size=size) per (H,W) image: size x size window covering
 * [i - size/2, i - size/2 + size - 1], mode='reflect', rank size*size/2 (interpolate_twoframe.py:221-222,
 * size=50).  Exact selection (returns an element of the window). */
int vfi_median_filter(const float *x, float *y, int N, int H, int W, int size, vfi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Quality scoring on device (src/evaluation/evaluate_image.py:7-30)
 * ---------------------------------------------------------------------------------- */

/* out2[0] = sum(a - b), out2[1] = sum((a - b)^2) over `count` floats, accumulated in double in a fixed order
 * (deterministic).  PSNR / SSD / L1 / MSE / variance of evaluate_image.py:22-28 follow from the two sums.
 * workspace: 2 * 1024 doubles. */
int vfi_diff_sums(const float *a, const float *b, long long count, double *out2, void *workspace, vfi_stream_t stream);

/* Sum of the SSIM map (piq.ssim formula) over the interior [border, H-border) x [border, W-border) of `planes`
 * planes, from the Gaussian-filtered moments mu_x, mu_y, E[xx], E[yy], E[xy]; out2[0] = sum.  workspace as above. */
int vfi_ssim_sum(const float *mu_x, const float *mu_y, const float *e_xx, const float *e_yy, const float *e_xy,
                 int planes, int H, int W, int border, float c1, float c2, double *out2, void *workspace,
                 vfi_stream_t stream);

/* out = a * b elementwise. */
int vfi_mul(const float *a, const float *b, float *out, long long count, vfi_stream_t stream);

/* F.avg_pool2d(kernel_size=f) on `planes` dense (H,W) planes -> (H/f, W/f) (floor): the down-sampling step of piq.ssim
 * (called at src/evaluation/evaluate_image.py:21) for images whose short side exceeds 384 px. */
int vfi_avg_pool(const float *x, float *y, int planes, int H, int W, int f, vfi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VFI_HIP_H */
