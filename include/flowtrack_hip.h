/*
 * flowtrack_hip.h — C ABI of libflowtrack_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the two dense-CNN hot paths of FlowTrack
 * (pose: ResNet + 3x deconv head; flow: FlowNet2-family).  It replaces the
 * reference's operator FFI — the cffi/THC entry points that the reference
 * binds through torch.utils.ffi — and the stock torch.nn layers the reference
 * runs through cuDNN.  Conventions (differences from the reference FFI are
 * deliberate):
 *
 *   - plain device pointers + explicit sizes + a hipStream_t (as void*); no
 *     torch / THC types, no global THCState;
 *   - the caller owns every buffer, including outputs and workspaces; nothing
 *     is resized, zero-filled or freed behind the caller's back
 *     (reference: correlation_cuda.c:36-42,83-84 resizes + frees scratch);
 *   - every entry point returns an int status (FT_OK == 0) and never aborts
 *     (reference: THError("aborting") correlation_cuda.c:87-89; exit(-1)
 *     roi_pooling_kernel.cu:94-98);
 *   - all work is enqueued asynchronously on the given stream; entry points
 *     are re-entrant and hold no global state (one process per GPU).
 *
 * Reference paths are relative to /root/reference.
 */
#ifndef FLOWTRACK_HIP_H_
#define FLOWTRACK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
enum {
  FT_OK = 0,
  FT_ERR_INVALID_ARG = 1,   /* bad size / alignment / enum */
  FT_ERR_UNSUPPORTED = 2,   /* valid but not implemented for this combination */
  FT_ERR_HIP = 3,           /* a HIP runtime call failed; see ft_last_hip_error */
  FT_ERR_NO_DEVICE = 4
};

/* ---- element types / layouts --------------------------------------------- */
enum { FT_F16 = 0, FT_F32 = 1 };
enum { FT_ACT_NONE = 0, FT_ACT_RELU = 1, FT_ACT_LEAKY = 2 };
enum { FT_LAYOUT_NHWC = 0, FT_LAYOUT_NCHW_F32 = 1 };

typedef void* ft_stream_t; /* hipStream_t */

/* ---- runtime ------------------------------------------------------------- */
int ft_version(void);
const char* ft_status_string(int status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char* ft_last_hip_error(void);
/* name (<= name_len bytes incl. NUL), CU count, HBM bytes of `device` */
int ft_device_info(int device, char* name, int name_len, int* cu_count,
                   uint64_t* hbm_bytes);

/* HIP-graph capture of a launch sequence issued on `stream` through this ABI
 * (replaces per-layer host launches in the hot loop). */
int ft_graph_begin_capture(ft_stream_t stream);
int ft_graph_end_capture(ft_stream_t stream, void** graph_exec_out);
int ft_graph_launch(void* graph_exec, ft_stream_t stream);
int ft_graph_destroy(void* graph_exec);

/* hipEvent helpers so a host in any language can time work on `stream`
 * (torch.cuda.Event only sees torch's current stream). */
int ft_event_create(void** event_out);
int ft_event_record(void* event, ft_stream_t stream);
/* later work on `stream` waits for `event` (fork / join of two streams; inside a stream capture this makes the second
 * stream a parallel branch of the graph) */
int ft_stream_wait_event(ft_stream_t stream, void* event);
int ft_event_synchronize(void* event);
int ft_event_elapsed_ms(void* start, void* stop, float* ms_out);
int ft_event_destroy(void* event);
int ft_stream_synchronize(ft_stream_t stream);
/* hipMemcpyAsync(hipMemcpyDefault) on `stream`: a host loop that drives launches through this ABI (the per-frame pose runner of
 * the tracking glue, lib/tracking/net_utils.py:36-71) brings its few kB of results back without a framework call in between;
 * a pinned host buffer makes the copy asynchronous. */
int ft_memcpy_async(void* dst, const void* src, size_t bytes, ft_stream_t stream);

/* ---- P1-P7 / F1-F3: fused conv / transposed-conv (implicit GEMM on MFMA) ---
 *
 * Replaces, on the reference path, the stock layers it runs via torch/cuDNN:
 *   nn.Conv2d (+BatchNorm2d eval, +ReLU/LeakyReLU, +residual add)
 *     lib/pose/models/resnet.py:19-23, blocks.py:89-119,
 *     lib/flownet/networks/submodules.py:7-32
 *   nn.ConvTranspose2d(k=4, s=2, p=1) (+BN, +ReLU / +bias, +LeakyReLU)
 *     lib/pose/models/pose_deconv.py:19-28, submodules.py:34-38,
 *     FlowNetS.py:42-45
 *
 *   y[n, oy, ox, co] = act( scale[co] * sum_{tap,ci} x[n, iy, ix, ci] * w[co,tap,ci]
 *                           + shift[co] (+ residual[n, oy, ox, co]) )
 *
 * Activations are NHWC with an explicit channel stride/offset so producers can
 * write straight into a channel slice of a concat buffer (torch.cat in
 * FlowNetS.py:73-88 disappears).  Cin is rounded up when reading: channels
 * [Cin, cin_pad) of every pixel of x (cin_pad from ft_conv_pack_geometry, at
 * least roundup8(Cin)) must exist inside x_cstride and be zero.
 */
typedef struct ft_conv_desc {
  int dtype;       /* FT_F16 (fp16 storage, fp32 accumulate) or FT_F32 */
  int N, Hi, Wi;   /* input batch / spatial size */
  int Cin;         /* logical input channels */
  int x_cstride;   /* elements between consecutive input pixels */
  int x_coff;      /* first channel inside the pixel (multiple of 8) */
  int Cout;        /* logical output channels */
  int kh, kw;      /* kernel (transposed: must be 4,4) */
  int stride, pad; /* conv: any; transposed: must be 2,1 */
  int transposed;  /* 0 = Conv2d, 1 = ConvTranspose2d(4,2,1) */
  int Ho, Wo;      /* output spatial size (validated against the formula) */
  int y_cstride;   /* NHWC: elements between output pixels; NCHW: ignored */
  int y_coff;      /* NHWC: first output channel inside the pixel (mult of 4) */
  int out_layout;  /* FT_LAYOUT_NHWC (dtype) or FT_LAYOUT_NCHW_F32 */
  int has_residual;
  int res_cstride, res_coff; /* residual is NHWC, output geometry */
  int act;         /* FT_ACT_* */
  float slope;     /* LeakyReLU negative slope */
  /* Row-packed input (small-Cin stem layers, e.g. the 7x7/s2 convs on 3/6/12 channels): when
   * x_wpitch > 0 the input buffer is [N, Hi, x_wpitch, x_cstride] with x_lpad >= pad zero columns
   * physically present left of pixel 0 and zero columns on the right up to x_wpitch.  A whole kernel
   * ROW (kw taps x x_cstride channels, contiguous in memory) then is one K-run, so the direct-to-LDS
   * kernel applies although Cin is tiny.  0 = plain NHWC. */
  int x_lpad, x_wpitch;
  /* 0 = let the library choose the workgroup tile; else one of the values ft_conv_tile_candidates()
   * returned for this descriptor (the pick of a plan-build-time benchmark, the counterpart of the
   * reference's `cudnn.benchmark = True`, tools/pose/main.py:59).  Only the speed depends on it, never
   * the packed-weight layout; an inapplicable value falls back to the heuristic. */
  int tile_hint;
  /* Optional SECOND input summed into the same accumulator ("K-concat"):
   *   y = act(scale .* (W1 . x + W2 . x2[n, q*x2_stride]) + shift),   W = [W1 | W2] packed along K (K-run 1 = Cin
   * channels of x, K-run 2 = x2_cin channels of x2, each padded as ft_conv_pack_geometry reports).
   * It fuses the projection shortcut of a bottleneck block (`downsample` conv + bn, lib/pose/models/blocks.py:
   * 104-119) into its conv3: the caller folds both BatchNorms into the weights (scale = NULL) and sums the shifts.
   * Requirements: 1x1 / stride 1 / pad 0 main conv, has_residual = 0 — x2 is passed in ft_conv2d_fwd's `residual`
   * argument — x2 is NHWC `dtype` [N, x2_hi, x2_wi, x2_cstride] with Ho = (x2_hi - 1) / x2_stride + 1 (same for Wo).
   * x2_cin = 0: no second input. */
  int x2_cin, x2_hi, x2_wi, x2_cstride, x2_coff, x2_stride;
  /* Optional fused TAIL 1x1 conv: y = Wt . act(scale .* conv(x) + shift) + bt with tail_cout <= 32 outputs.  The
   * intermediate (all Cout channels of a pixel tile) stays in LDS and is never written: it fuses the pose head's
   * last layers, `heatmap(deconv(...))` (lib/pose/models/pose_deconv.py:43-45), and saves the round trip of the
   * largest tensor of the head.  Requirements: FT_F16, Cout in {64, 128, 256}, has_residual = 0, x2_cin = 0; the
   * tail pack — fp16 [32][Cout] weights `hi` (rows >= tail_cout zero), fp32 [32] bias, fp16 [32][Cout] `lo` with
   * w = hi + lo (the tail weights keep ~22 bits: the heatmap conv decides the arg-max), and, REQUIRED when Cout = 256
   * and `tile_hint` selects the 8-phase tile (bits 28-29 == 3; never chosen without a hint): the same weights once more in
   * that kernel's operand order, fp16 [2][8][2][64][8] = [channel half wc][step st][hi, lo][lane = lhi * 32 + tail output][e]
   * holding channel wc * 128 + st * 16 + 4 * lhi + e % 4 + 8 * (e / 4) (32 KiB; total pack 64 KiB + 128 B) — is passed in
   * ft_conv2d_fwd's `residual` argument; `y` / out_layout / y_cstride / y_coff describe the TAIL output
   * (Ho x Wo x tail_cout).  tail_cout = 0: none. */
  int tail_cout;
  /* 1: fuse the 3x3 / stride 2 / pad 1 max-pool of the ResNet stem behind the activation (resnet.py:19-23: conv1 -> bn1
   * -> relu -> maxpool).  Ho / Wo stay the CONV's output size (even); `y` is the pooled NHWC map [N, Ho/2, Wo/2, y_cstride].
   * Supported for the row-packed 7x7 / stride 2 fp16 stem with relu, 64 outputs and x_lpad >= pad + 2 (the pooled patch
   * starts one stem column further left); FT_ERR_UNSUPPORTED otherwise.  Bit-identical to conv + ft_maxpool3x3s2_fwd. */
  int pool;
  /* 0: `shift` is one vector [Cout] shared by the batch.  > 0: `shift` is [N][shift_nstride] floats, sample n takes
   * shift + n * shift_nstride — the fold of FlowNet2*'s per-sample rgb mean into conv1 (ft_flow_mean_fold below).
   * Supported by the persistent row-packed 7x7 / stride 2 fp16 stem only; FT_ERR_UNSUPPORTED elsewhere. */
  int shift_nstride;
  /* 1: `x` is the reference's NCHW fp32 network input [N, Cin, Hi, Wi] itself (Cin <= 4), not a packed copy: the stem kernel
   * gathers its input patch from the planes, casts to fp16 and lays it out in LDS exactly as ft_pack_nchw_to_nhwc would have
   * written it, so the pack launch and its [N, Hi, x_wpitch, 4] buffer disappear and the result is bit-identical.  x_cstride (4),
   * x_lpad and x_wpitch then describe that VIRTUAL row-packed view.  Supported with pool = 1 (the fp16 ResNet stem,
   * lib/pose/models/resnet.py:19-23) only; FT_ERR_UNSUPPORTED elsewhere. */
  int x_nchw_f32;
} ft_conv_desc;

/* Packed-weight geometry (ft_conv_pack_geometry): w_packed is [nphases][cout_pad][kpad] of d->dtype,
 * zeros in all padding, and the element for (tap, sub, ci) sits at k = tap*cin_pad + sub*run_cpad + ci.
 * Plain layers: run_taps = 1 (sub = 0), tap = kernel element.  Row-packed layers: tap = kernel row,
 * sub = kernel column (run_taps = kw, run_cpad = x_cstride). ft_conv_tap_source maps (phase,tap,sub)
 * to the reference kernel element (ky,kx). */
typedef struct ft_conv_geometry {
  int nphases;   /* 1 conv, 4 transposed conv (one 2x2 conv per output parity) */
  int ntaps;     /* K-runs per phase */
  int cin_pad;   /* elements per K-run (channels padded; row-packed: roundup(kw * x_cstride)) */
  int cout_pad, kpad;
  int run_taps, run_cpad;
  int cin2_pad;  /* elements of the second input's K-run (x2_cin padded), 0 without one; kpad = ntaps * cin_pad + cin2_pad */
} ft_conv_geometry;

/* The layout depends only on (dtype, Cin, Cout, kernel, transposed, x_cstride - x_coff, row-packing):
 * when the input view has room for roundup(Cin, 32 fp16 / 16 fp32) channels per pixel (padding
 * channels zero) - or is row-packed - the direct-to-LDS kernel is used and every K-run is padded to
 * that multiple; otherwise cin_pad = roundup(Cin, 8) and the generic kernel runs. */
int ft_conv_pack_geometry(const ft_conv_desc* d, ft_conv_geometry* out);
/* Which original kernel element (ky, kx) element `sub` of K-run `tap` of phase `phase` reads. */
int ft_conv_tap_source(const ft_conv_desc* d, int phase, int tap, int sub, int* ky, int* kx);
/* scale/shift: float[Cout_pad] or NULL (=> 1 / 0). residual may be NULL. */
int ft_conv2d_fwd(const ft_conv_desc* d, const void* x, const void* w_packed,
                  const float* scale, const float* shift, const void* residual,
                  void* y, ft_stream_t stream);
/* Same as ft_conv2d_fwd with a scratch buffer (>= ft_conv_workspace_bytes(d), 16-byte aligned, contents don't care):
 * unlocks the tile variants that split K across workgroups (fp32 partial tiles in the workspace + a reduce launch) for
 * layers with few pixels and a long K (layer4, FlowNet conv5..6, every deep layer at small batch).  Without a
 * workspace those hints fall back to the unsplit variant of the same tile. */
int ft_conv2d_fwd_ws(const ft_conv_desc* d, const void* x, const void* w_packed,
                     const float* scale, const float* shift, const void* residual,
                     void* y, void* workspace, size_t workspace_bytes, ft_stream_t stream);
size_t ft_conv_workspace_bytes(const ft_conv_desc* d);
/* Writes up to `max` valid `tile_hint` values for `d` (pixel tile x channel tile x split-K variants of
 * the implicit-GEMM kernel) into hints[] and returns their count (0: the layer runs on a kernel without
 * tile variants); negative = error status. */
int ft_conv_tile_candidates(const ft_conv_desc* d, int* hints, int max);
/* algorithmic FLOPs (2*MACs, unpadded) of one ft_conv2d_fwd call */
double ft_conv_flops(const ft_conv_desc* d);

/* ---- 1x1 convolutions of the deep, small-map stages: weight-streaming GEMM, weights straight to registers ----------
 * Same math as ft_conv2d_fwd for a 1x1 / stride 1 / pad 0 fp16 NHWC layer (optionally with the K-concat second input
 * x2_* or an identity residual), for layers with few pixels and long K (ResNet layer3/4 at batch 64, blocks.py:89,95,
 * 98-103): a workgroup owns 96 pixels, the pixel operand streams through an LDS ring, each wave loads its weight
 * fragments from a fragment-ordered stream directly into registers (csrc/conv_direct.hip).
 * The same five entry points also carry the other weight-streaming / weight-stationary forms (each with its own stream layout,
 * told apart by ft_conv_direct_stream_id): 3x3 / stride 1 and 2 on whole small maps or row strips (512 / 1024 channels: ResNet
 * layer4's conv2, blocks.py:92-95; FlowNet conv5 .. conv6_1, FlowNetS.py:27-32), the 3x3 gather form, the persistent K = 256
 * form, and — round 4 — the REGISTER-STATIONARY 5x5 / stride 2 / pad 2 form on 64 input channels with Cout % 64 == 0 (FlowNet's
 * conv2, FlowNetS.py:21: csrc/conv_wstat.hip; plain NHWC fp16 input and output, no residual / second input).
 *   ft_conv_direct_supported     FT_OK when the shape qualifies (1x1: Cin, x2_cin multiples of 64, Cout multiple of 256 — or
 *                                Cin, x2_cin multiples of 256 and Cout of 64 for the K-split form)
 *   ft_conv_direct_weight_bytes  size of the weight stream
 *   ft_conv_direct_pack          builds it from the ft_conv_pack_geometry layout w_packed [cout_pad][kpad] (once per weight set)
 *   ft_conv_direct_fwd           arguments as ft_conv2d_fwd (residual = identity residual or the second input x2)
 *   ft_conv_direct_stream_id     identifies the stream LAYOUT `d` needs (>= 0; -1 when unsupported).  The kernel form — and
 *                                with it the fragment order of the stream — depends on the pixel count N * Ho * Wo, not only on
 *                                the weights: a stream packed for one descriptor may be passed to ft_conv_direct_fwd with
 *                                another descriptor only when both report the same id (all layouts of a layer have the same
 *                                byte count, so the size does not tell them apart). */
int ft_conv_direct_supported(const ft_conv_desc* d);
int ft_conv_direct_stream_id(const ft_conv_desc* d);
long long ft_conv_direct_weight_bytes(const ft_conv_desc* d);
int ft_conv_direct_pack(const ft_conv_desc* d, const void* w_packed, int kpad, int cout_pad, void* wstream, ft_stream_t stream);
int ft_conv_direct_fwd(const ft_conv_desc* d, const void* x, const void* wstream, const float* scale, const float* shift,
                       const void* residual, void* y, ft_stream_t stream);

/* ---- whole-bottleneck fusion (fp16) --------------------------------------------------------------
 * One launch for an identity-shortcut Bottleneck (reference: Bottleneck.forward, lib/pose/models/blocks.py:105-120,
 * the blocks whose `downsample` is empty, resnet.py:29-36):
 *   y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(relu(bn1(conv1_1x1(x)))))))) + x)
 * t1 / t2 live in LDS only; x is read once (+ halo / residual re-reads from L2), y written once.
 * Supported: dtype FT_F16, C = 256, P = 64, stride 1 (ResNet layer1.1+); y must not alias x.
 * w1 / w2 / w3 are the ft_conv_pack_geometry layouts of the three convs on channel-aligned views
 * ([64][256], [64][9*64] with k = (ky*3+kx)*64 + ci, [256][64]); scale_shift = the three folded BatchNorms as one
 * float[2P + 2P + 2C] table: scale1[P] shift1[P] scale2[P] shift2[P] scale3[C] shift3[C]. */
typedef struct ft_bottleneck_desc {
  int dtype;
  int N, H, W;            /* block input = output size */
  int C, P;               /* block width (in = out channels) and planes */
  int x_cstride, x_coff;  /* NHWC views, element units, multiples of 8 */
  int y_cstride, y_coff;
  int head_only;          /* 1: conv1 + conv2 only, y = t2 [N,H,W,P] (C = 64: the stage's entry block, whose conv3 is
                             K-concatenated with its projection shortcut by ft_conv2d_fwd); w3 = NULL,
                             scale_shift = float[4P] */
  int projection;         /* 1: the stage's entry block WHOLE (blocks.py:104-119 with `downsample`): C = 64 input channels,
                             y = relu(bn3(conv3(t2)) + bn_d(conv_d(x))) [N,H,W,4P]; w3 = [4P][P | C] fp16, the two 1x1
                             convs K-concatenated with their BatchNorm scales folded in (the layout ft_conv2d_fwd's x2_*
                             path takes), scale_shift = float[4P + 8P]: {s1 b1 s2 b2} then scale3 = 1, shift3 = b3 + b_d */
  int stride;             /* stride of conv2 (0 / 1 = 1).  2 only with head_only in the streamed-weights form
                             (ft_bottleneck_stream_*: P = 256, C = 512, even H and W: layer3.0 of the ResNets, resnet.py:44-49):
                             y = t2 [N, H/2, W/2, P]; weight stream = w1 [P][C] then w2 [P][9P]; tables = float[6][2P] of
                             which {scale1, shift1} {scale2, shift2} are read.  (Appended in round 3: zero-initialised
                             descriptors of older callers keep their meaning.) */
  int folded;             /* ft_bottleneck_stream_fwd only (round 6; see ft_bottleneck_stream_folds).  1: the weight stream was packed
                             from weights with the BatchNorm SCALE already folded in (fp16(w * scale[co]), one rounding from the fp32
                             weights) and `tables` is NOT float[6][2P] but uint32 [P + P + C]: shift1, shift2, shift3 as (hi, lo) fp16
                             pairs, hi = fp16(shift) in bits 0-15, lo = fp16(shift - hi) in bits 16-31.  The kernels add the shift and
                             the identity residual as extra MFMA k-steps and their epilogues are fp16(relu(acc)).  0 = the scale /
                             shift tables of round 2. */
} ft_bottleneck_desc;
int ft_bottleneck_supported(const ft_bottleneck_desc* d);   /* FT_OK or FT_ERR_UNSUPPORTED / FT_ERR_INVALID_ARG */
int ft_bottleneck_fwd(const ft_bottleneck_desc* d, const void* x,
                      const void* w1, const void* w2, const void* w3,
                      const float* scale_shift, void* y, ft_stream_t stream);
/* algorithmic FLOPs of the three convs (2*MACs, no halo recompute) */
double ft_bottleneck_flops(const ft_bottleneck_desc* d);

/* Streamed-weights form of the same fusion for the 128- and 256-plane stages (fp16, C = 4P, P = 128 or 256, stride 1,
 * head_only = 0; ResNet layer2.1+ / layer3.1+): a workgroup owns a full-width strip of output rows of one image, keeps
 * t1 / t2 in LDS and streams the block's weights once through an LDS ring (csrc/bottleneck_stream.hip).
 *   ft_bottleneck_stream_weight_bytes  size of the packed weight stream (0 when unsupported)
 *   ft_bottleneck_stream_pack          builds it from the three convs' ft_conv_pack_geometry layouts
 *                                      (w1 [P][C], w2 [P][9P] with k = (ky*3+kx)*P + ci, w3 [C][P]); once per weight set
 *   ft_bottleneck_stream_fwd           tables = float[6][2P]: {scale1, shift1} {scale2, shift2} then {scale3, shift3} of
 *                                      output channels [qP, qP+P) for q = 0..3 (d->folded = 0), or the shift pairs described at
 *                                      ft_bottleneck_desc.folded; y must not alias x. */
int ft_bottleneck_stream_supported(const ft_bottleneck_desc* d);
/* 1 when ft_bottleneck_stream_fwd takes the FOLDED operands for this block (d->folded = 1; every stride-1 identity block), 0 when only
 * the table form exists (the stride-2 head); the value of d->folded itself is ignored. */
int ft_bottleneck_stream_folds(const ft_bottleneck_desc* d);
long long ft_bottleneck_stream_weight_bytes(const ft_bottleneck_desc* d);
int ft_bottleneck_stream_pack(const ft_bottleneck_desc* d, const void* w1, const void* w2, const void* w3, void* wstream,
                              ft_stream_t stream);
int ft_bottleneck_stream_fwd(const ft_bottleneck_desc* d, const void* x, const void* wstream, const float* tables, void* y,
                             ft_stream_t stream);

/* ---- layout / pooling helpers -------------------------------------------- */
/* NCHW fp32 [N,C,H,W] -> NHWC `dtype` [N,H,wpitch,cpad]: pixel x lands in column lpad + x, channels
 * >= C and all other columns are zeroed (wpitch = W, lpad = 0: plain NHWC; cpad multiple of 4).
 * First step of DeconvResnet.forward (pose_deconv.py:32-33). */
int ft_pack_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W,
                         int cpad, int lpad, int wpitch, int dtype,
                         ft_stream_t stream);
/* NHWC `dtype` (channel stride/offset) -> NCHW fp32 */
int ft_unpack_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W,
                           int x_cstride, int x_coff, int dtype,
                           ft_stream_t stream);
/* nn.MaxPool2d(3, 2, 1) on NHWC (resnet.py:23); C multiple of 8 */
int ft_maxpool3x3s2_fwd(const void* x, void* y, int N, int Hi, int Wi, int C,
                        int dtype, ft_stream_t stream);

/* ---- P4: BatchNorm2d batch statistics -----------------------------------------
 * Per-channel mean and BIASED variance over N*H*W of an NHWC view (C and x_cstride multiples of 8) — the
 * reduction nn.BatchNorm2d performs in training mode (tools/pose/main.py:207,342; running-stat
 * recalibration).  The inference path does not need it (eval-mode BN is folded into ft_conv2d_fwd).
 * workspace: float[2*C] scratch (zeroed by the call); mean, var: float[C]. */
int ft_bn_batch_stats(const void* x, int N, int H, int W, int C, int x_cstride, int dtype,
                      float* workspace, float* mean, float* var, ft_stream_t stream);

/* ---- P8/P9: heatmap -> keypoints ------------------------------------------
 * max_preds + the adjust_coords nudge of final_preds
 * (lib/pose/utils/evaluation.py:11-35): per (n,k) map the arg-max over H*W
 * (first occurrence on ties), x = idx % W, y = idx / W (integer floor, torch
 * 0.4 semantics), coords zeroed where score <= 0, optional +/-0.25 px nudge.
 * heatmaps: NCHW fp32 [N,K,H,W]; idx int32 [N*K]; score fp32 [N*K];
 * coords fp32 [N*K*2] (x,y in heatmap pixels, before transform_preds). */
int ft_heatmap_max_preds(const float* heatmaps, int N, int K, int H, int W,
                         int adjust_coords, int32_t* idx, float* score,
                         float* coords, ft_stream_t stream);
/* The same launch with the key points written as rows (x, y, score): rows fp32
 * [N*K*3] — the layout the tracking glue and the multi-GPU gather consume
 * (lib/tracking/net_utils.py: `preds` and `maxvals` concatenated), so no
 * concat launch follows the network. */
int ft_heatmap_keypoint_rows(const float* heatmaps, int N, int K, int H, int W,
                             int adjust_coords, int32_t* idx, float* rows,
                             ft_stream_t stream);

/* Arg-max margin screen of the fast (fp16) mode: min_margin[n] = min over the K maps of crop n of (largest - second largest
 * value, at different pixels; 0 for a tie).  A crop whose min_margin exceeds twice the heat-map error bound of the fp16
 * arithmetic has the same arg-max in every map as the fp32 parity mode (max_preds, evaluation.py:11-20); the others are the
 * ones a caller re-runs in fp32 (DeconvResnet.forward_keypoint_rows_exact).  heatmaps NCHW fp32 [N,K,H,W], min_margin fp32 [N]. */
int ft_heatmap_min_margin(const float* heatmaps, int N, int K, int H, int W, float* min_margin, ft_stream_t stream);

/* The same screen with a RELATIVE bound and the two cases the margin alone misses.  Per crop n: R = largest - smallest value
 * over its K maps, E = rel_bound * R (the fp16 mode's heat-map error scales with the maps' range); flags[n] = 1 when some map
 * has (top1 - top2) < 2 E (two values an error of E per element can reorder), or |top1| < E (max_preds zeroes the coordinates
 * where score <= 0, lib/pose/utils/evaluation.py:17-19: that mask can flip), or any non-finite value (NaN compares false: the
 * tests are written negated).  stats fp32 [N,4] = (smallest margin, R, smallest |top1|, E); flags int32 [N]; K <= 256. */
int ft_heatmap_argmax_screen(const float* heatmaps, int N, int K, int H, int W, float rel_bound, int32_t* flags, float* stats,
                             ft_stream_t stream);

/* The re-run set of that screen, built ON THE DEVICE (round 5; DeconvResnet.exact_submit / exact_finish): header[0] = number
 * of set flags, header[1 + j] = index of the j-th flagged row, and row idx[j] of `src` ([N] rows of row_bytes, a multiple of
 * 16) is copied to row j of `dst`.  No host read sits between the screen and the gather: the host looks at `header` one step
 * later, behind an event, and only then decides whether a re-run is needed.  N <= 1024; header: int32 [1 + N]. */
int ft_gather_flagged_rows(const int32_t* flags, int N, const void* src, long long row_bytes, void* dst, int32_t* header,
                           ft_stream_t stream);

/* ---- F1: FlowNet2* input normalisation ------------------------------------
 * rgb_mean over (pair,H,W) per (b,colour) then (x - mean) / rgb_max
 * (lib/flownet/model/models.py:255-257).  inputs: fp32 [B,3,2,H,W].
 * partial: float workspace [B*3*FT_RGB_MEAN_SPLITS]; mean: float [B*3]. */
#define FT_RGB_MEAN_SPLITS 64
int ft_flow_rgb_mean(const float* inputs, int B, int H, int W, float* partial,
                     float* mean, ft_stream_t stream);
/* mode 0: y = NHWC [B,H,wpitch,8]   channels (r0,g0,b0,r1,g1,b1,0,0)   (FlowNet2S)
 * mode 1: y = NHWC [2B,H,wpitch,4]  images 0..B-1 = frame0, B..2B-1 = frame1,
 *         channels (r,g,b,0)                           (FlowNetC siamese trunk)
 * pixel x lands in column lpad + x; the other columns are zeroed (wpitch = W, lpad = 0: plain). */
int ft_flow_pack_pair(const float* inputs, const float* mean, float rgb_max,
                      void* y, int B, int H, int W, int mode, int lpad,
                      int wpitch, int dtype, ft_stream_t stream);

/* The two calls above as ONE launch, writing the 6-channel (mode 0) view y [B,H,wpitch,8] and / or the siamese (mode 1) view y3
 * [2B,H,wpitch3,4] (either may be NULL): every workgroup keeps its rows of the sample in registers
 * across the mean reduction (partial sums exchanged as tagged 8-byte words in `state`), so the frame pair is read once
 * (lib/flownet/model/models.py:255-257).  `state`: ft_flow_mean_pack_pair_state_words(B, H, W) 8-byte words (0 = the shape is not
 * covered: W % 4 != 0, more than 42 x 6144 pixels per frame, or more than 512 workgroups = B x ceil(H x W / 6144) so that not
 * all of them would be resident while they wait for each other -> use the two calls), zeroed ONCE by the caller at allocation and
 * private to this (B, H, W) from then on; its last word becomes non-zero if a workgroup ever timed out waiting for its sample's
 * sums (means are NaN then).  mean fp32 [B*3] is written as well.  Same arithmetic per element as ft_flow_pack_pair; the mean's
 * summation order differs from ft_flow_rgb_mean's (last-bit differences). */
/* The mean folded into conv1 (FlowNet2S; lib/flownet/model/models.py:255-257 + FlowNetS.py:20 as ONE pass over the frames):
 * conv_zero-padded((x - mean) / rgb_max) = conv_mean-padded(x / rgb_max) - (mean / rgb_max) . sum of the kernel, so
 *  (1) ft_flow_pack_pair_sums writes the UN-centred pair y = fp16 [B, H + 2 pad, wpitch, 8] (pixel (yy, xx) at row pad + yy,
 *      column pad + xx, channels (r0,g0,b0,r1,g1,b1,0,0) = x / rgb_max; padding pixels untouched) and per-chunk colour sums
 *      partial fp32 [B*3][ft_flow_pack_pair_sums_chunks(H)] in the same pass (no dependency on the mean: the mean launch and
 *      its second read of the frames disappear);
 *  (2) ft_flow_mean_fold finishes the mean (fixed summation order), fills y's padding pixels with m16 = fp16(mean / rgb_max)
 *      per colour (the reference's zero padding of the centred input) and writes the per-sample shift
 *      shift_n[b][co] = shift[co] - scale[co] * sum_c m16[c % 3] * wsum[co][c]   (scale NULL = 1)
 *      with wsum fp32 [Cout][8] = the fp16-rounded conv1 weights summed over the kernel window per input channel;
 *  (3) conv1 runs as a pad-0 conv on the (H + 2 pad) x (W + 2 pad) row-packed view with ft_conv_desc.shift_nstride = Cout.
 * Differences from the two-launch path: the input is rounded to fp16 before the mean is removed (|x / rgb_max| <= 1 instead
 * of <= 0.5: one more bit of input rounding), and mean / rgb_max enters as its fp16 value.  W % 4 == 0, FT_F16 only. */
#define FT_PACK_SUMS_ROWS 4
int ft_conv_shift_nstride_supported(const ft_conv_desc* d);   /* FT_OK when ft_conv2d_fwd takes `d` with its shift_nstride */
long long ft_flow_pack_pair_sums_chunks(int H);
int ft_flow_pack_pair_sums(const float* inputs, float rgb_max, void* y, int B, int H, int W, int pad, int wpitch, int dtype,
                           float* partial, ft_stream_t stream);
int ft_flow_mean_fold(const float* partial, float rgb_max, void* y, int B, int H, int W, int pad, int wpitch, int dtype,
                      const float* wsum, const float* scale, const float* shift, int Cout, float* shift_n, float* mean,
                      ft_stream_t stream);

long long ft_flow_mean_pack_pair_state_words(int B, int H, int W);
int ft_flow_mean_pack_pair(const float* inputs, float rgb_max, void* y, int lpad, int wpitch, void* y3, int lpad3, int wpitch3,
                           int B, int H, int W, int dtype, unsigned long long* state, float* mean, ft_stream_t stream);

/* ---- F7: nn.Upsample(scale_factor=4, mode='bilinear') * mul ----------------
 * (FlowNetS.py:58, models.py:292; align_corners=False).  NCHW fp32 in/out. */
int ft_upsample_bilinear4x(const float* x, float* y, int N, int C, int h, int w,
                           float mul, ft_stream_t stream);

/* ---- F4: Correlation forward ---------------------------------------------
 * Replaces Correlation_forward_cuda (correlation_package/src/correlation_cuda.c:11-93,
 * kernels correlation_cuda_kernel.cu:10-106).  Same parameters and output
 * geometry; no rInput scratch tensors (padding is implicit), NCHW fp32 in/out:
 *   out[n, tj*D+ti, y, x] = 1/(k*k*C) * sum_{j,i,c} in1[n,c,y1+j,x1+i] * in2[n,c,y2+j,x2+i]
 * corr_type_multiply must be 1 (the only mode the reference implements). */
int ft_correlation_out_shape(int C, int H, int W, int pad_size, int kernel_size,
                             int max_displacement, int stride1, int stride2,
                             int* out_c, int* out_h, int* out_w);
int ft_correlation_fwd(const float* in1, const float* in2, float* out, int B,
                       int C, int H, int W, int pad_size, int kernel_size,
                       int max_displacement, int stride1, int stride2,
                       int corr_type_multiply, ft_stream_t stream);
/* Fused in-network form (FlowNetC.py:86-92): NHWC `dtype` features in, the
 * LeakyReLU'd cost volume written into channels [y_coff, y_coff+D*D) of an
 * NHWC concat buffer.  kernel_size=1, stride1=1, pad=max_displacement. */
int ft_correlation_nhwc_fwd(const void* f1, const void* f2, void* y, int B,
                            int C, int H, int W, int max_displacement,
                            int stride2, int f_cstride, int y_cstride,
                            int y_coff, int act, float slope, int dtype,
                            ft_stream_t stream);

/* ---- F5: Resample2d forward (flow warp) ------------------------------------
 * Replaces Resample2d_cuda_forward (resample2d_package/src/Resample2d_cuda.c,
 * kernel Resample2d_kernel.cu:20-66); kernel_size is fixed at 1 as in
 * modules/resample2d.py:8.  in1 [B,C,H,W], flow [B,2,H,W], out [B,C,H,W]. */
int ft_resample2d_fwd(const float* in1, const float* flow, float* out, int B,
                      int C, int H, int W, ft_stream_t stream);

/* ---- F6: ChannelNorm forward ------------------------------------------------
 * Replaces ChannelNorm_cuda_forward (channelnorm_package/src/ChannelNorm_cuda.c,
 * kernel ChannelNorm_kernel.cu:19-51): out[b,0,y,x] = sqrt(sum_c in^2). */
int ft_channelnorm_fwd(const float* in1, float* out, int B, int C, int H, int W,
                       ft_stream_t stream);

/* ---- F5+F6 fused stage between stacked FlowNets (models.py:396-403) -------
 * From x6 = NHWC `dtype` [B,H,x_wpitch,8] (normalised img0|img1, pixel x in column x_lpad + x) and
 * flow NCHW fp32 [B,2,H,W] (already multiplied by div_flow) builds the 12-channel input of the
 * next FlowNetS: (img0, img1, warp(img1,flow), flow/div_flow, |img0-warp|)
 * as NHWC `dtype` [B,H,y_wpitch,16] (channels 12..15 and the padding columns zero). */
int ft_flow_warp_concat(const void* x6, const float* flow, float div_flow,
                        void* y, int B, int H, int W, int x_lpad, int x_wpitch,
                        int y_lpad, int y_wpitch, int dtype, ft_stream_t stream);

/* ---- N4: full FlowNet2 stack helpers -------------------------------------------
 * nn.Upsample(scale_factor=4, mode='nearest') of (x * mul) (lib/flownet/model/models.py:59-60,448). */
int ft_upsample_nearest4x(const float* x, float* y, int N, int C, int h, int w, float mul,
                          ft_stream_t stream);
/* Input of FlowNetFusion (models.py:140-168) in one pass: from x6 (as ft_flow_warp_concat) and two flow fields
 * NCHW fp32 [B,2,H,W] builds NHWC `dtype` [B,H,y_wpitch,16] (pixel x at column y_lpad + x; padding columns untouched) = (img0(3), flow_sd(2), flow_s2(2), |flow_sd|, |flow_s2|,
 * |img0 - warp(img1, flow_sd)|, |img0 - warp(img1, flow_s2)|, 0...) — two Resample2d + four ChannelNorm + cat. */
int ft_flow_fusion_concat(const void* x6, const float* flow_sd, const float* flow_s2, void* y, int B, int H,
                          int W, int x_lpad, int x_wpitch, int y_lpad, int y_wpitch, int dtype,
                          ft_stream_t stream);

/* ---- N2: person crops for the pose net --------------------------------------
 * Replaces the per-box host loop `transform_image` = cv2.warpAffine(img, t[:2], (res_w, res_h))
 * (lib/pose/utils/transforms.py:231-240, geometry :173-184; caller lib/tracking/net_utils.py:49-57).
 * img: HWC uint8 frame on the device (C <= 4); boxes: float[nb*3] = (center_x, center_y, scale) per crop;
 * out: NCHW fp32 [nb, C, rh, rw] = (bilinear(img) * pre_scale - mean[c]) * inv_std[c] (mean / inv_std may be
 * NULL), constant-0 border, fp32 interpolation weights (cv2's 1/32-px weight quantisation is not reproduced). */
int ft_crop_affine_fwd(const uint8_t* img, int H, int W, int C, const float* boxes, int nb, int rh, int rw,
                       const float* mean, const float* inv_std, float pre_scale, float* out,
                       ft_stream_t stream);

/* The same crop in the fixed-point arithmetic of cv2.warpAffine for a uint8 frame (round 5) — bit-exact to the RESTATED classic OpenCV
 * path (oracle/tracking_ref.py::warp_affine_cv2_ref; no cv2 in this image, so parity with a real cv2 build is UNPINNED; OpenCV >= 4.11,
 * IPP and HAL builds may differ in the last bit): OpenCV's fixed-point INTER_LINEAR
 * (imgwarp.cpp cv::warpAffine / WarpAffineInvoker / remapBilinear<FixedPtCast<int, uchar, 15>>; the third-party
 * dependency behind lib/pose/utils/transforms.py:238).  minv: double[nb*6], per box the dst -> src 2x3 map that
 * cv::warpAffine derives from the matrix it is handed (the caller inverts in double, cv2's operation order:
 * tracking.net_utils.cv2_crop_matrices).  out_u8: NHWC uint8 [nb, rh, rw, C] = cv2's return value (may be NULL);
 * out: NCHW fp32 [nb, C, rh, rw] = (float(u8) * pre_scale - mean[c]) * inv_std[c] (may be NULL; not both). */
int ft_crop_affine_cv2_fwd(const uint8_t* img, int H, int W, int C, const double* minv, int nb, int rh, int rw,
                           const float* mean, const float* inv_std, float pre_scale, uint8_t* out_u8, float* out,
                           ft_stream_t stream);

/* ---- experimental entry points (NOT part of the drop-in boundary) -------------------------------------------
 * Measured alternatives of the fused-block kernels that the default plans do not record (profiles/README.md has their
 * numbers).  They are exported by the library so that the tests and tools/dev can reach them, but they are declared only
 * when the includer defines FT_EXPERIMENTAL; signatures may change between rounds. */
#ifdef FT_EXPERIMENTAL
/* Register-stationary strip form of the same identity block (csrc/bottleneck_rstat.hip; fp16, C = 256, P = 64, stride 1,
 * 3 <= W <= 62): one persistent 8-wave workgroup per strip of full-width rows keeps all weights in registers.  The folded
 * BatchNorms travel INSIDE the weight buffer, which the caller builds once per weight set (fp16, from the fp32 weights):
 *   [64][272]  w1[co][ci] * scale1[co], then 16 columns {hi(shift1[co]), lo(shift1[co]), 0 x 14}
 *   [64][592]  w2[co][(ky*3+kx)*64 + ci] * scale2[co], then the 16 shift columns
 *   [256][80]  w3[co][ci] * scale3[co], then the 16 shift columns          (hi = fp16(shift), lo = fp16(shift - hi))
 * (the shift is added by one extra MFMA k-step against a vector of ones, the residual by two k-steps against an identity).
 * _supported also applies the cost rule (at least eight 64-pixel steps per strip) unless FT_BNK_RSTAT=2; FT_BNK_RSTAT=0 -> unsupported. */
int ft_bottleneck_rstat_supported(const ft_bottleneck_desc* d);
long long ft_bottleneck_rstat_weight_bytes(void);
int ft_bottleneck_rstat_fwd(const ft_bottleneck_desc* d, const void* x, const void* wpack, void* y, ft_stream_t stream);

/* CLUSTER form of the same block for the 256-plane stage on maps of <= 192 pixels (fp16, P = 256, C = 1024, stride 1;
 * layer3.1+ of the ResNets at 256 x 192; csrc/bottleneck_cluster.hip, round 5): the work of one image is shared by four
 * workgroups that exchange t1 / t2 through `workspace` INSIDE the launch (agent-scope hand-off, bounded spins), so a CU
 * streams about half the weight bytes of the strip form and no MFMA tile is padded.  Same wstream / tables as
 * ft_bottleneck_stream_fwd.  workspace: ft_bottleneck_cluster_workspace_bytes(d) bytes, ZEROED ONCE by the caller and
 * then left alone (it holds monotonic per-cluster arrival counters across calls); one workspace must not be used by two
 * launches that may run concurrently.  The 32-bit word at ft_bottleneck_cluster_status_offset(d) becomes non-zero if a
 * hand-off ever timed out (a member was not resident in time: the output of that call is wrong). */
int ft_bottleneck_cluster_supported(const ft_bottleneck_desc* d);
long long ft_bottleneck_cluster_workspace_bytes(const ft_bottleneck_desc* d);
long long ft_bottleneck_cluster_status_offset(const ft_bottleneck_desc* d);
int ft_bottleneck_cluster_fwd(const ft_bottleneck_desc* d, const void* x, const void* wstream, const float* tables, void* y,
                              void* workspace, ft_stream_t stream);
#endif /* FT_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* FLOWTRACK_HIP_H_ */
