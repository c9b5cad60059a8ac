/* c2s_hip.h -- C ABI of libc2s_hip.so: MI355X (gfx950) kernels for the Crop2Seg backbone hot path
 * (U-TAE / TimeUNet_v1 / W-TAE forward + backward).
 *
 * The reference (Many98/Crop2Seg) has no FFI / plugin layer: its hot path is stock ATen ops called
 * from nn.Module.forward.  Each entry point below therefore cites the reference
 * *call site* (file:line under the reference root) whose ATen op(s) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) unless named host_*; tensors are dense fp32,
 *     NCHW with the (batch, time) axes folded into N = B*T "frames" where the reference folds them
 *     (src/backbones/temp_shared_block.py:28);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - entry points never allocate, never synchronise and never throw: they validate their arguments,
 *     enqueue kernels on `stream` and return C2S_OK or a negative C2S_E* code, so a sequence of calls
 *     is hipGraph-capturable PROVIDED c2s_init() ran on the device beforehand (it raises the dynamic-LDS
 *     limits of the kernels and caches the CU count; an entry point that finds its device not set up does
 *     that set-up itself -- thread-safe, once per device -- which is not legal inside a stream capture);
 *     scratch memory is caller-provided (query with the *_workspace_floats calls);
 *   - `valid` is an int32[N] per-frame flag array (1 = real frame, 0 = temporal padding) or NULL when
 *     every frame is real; padded frames are skipped (reference: temp_shared_block.py:31-40).
 */
#ifndef C2S_HIP_H
#define C2S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2S_OK 0
#define C2S_EINVAL (-1)   /* bad shape / unsupported configuration */
#define C2S_ELAUNCH (-2)  /* hipLaunch error (see c2s_last_error) */
#define C2S_ENOSPACE (-3) /* workspace too small */

#define C2S_PAD_ZEROS 0
#define C2S_PAD_REFLECT 1

#define C2S_NORM_GROUP 0 /* statistics over (frame, channel group) : nn.GroupNorm */
#define C2S_NORM_BATCH 1 /* statistics over (all frames, channel)  : nn.BatchNorm2d / BatchNorm1d (train) */

int c2s_abi_version(void);
const char* c2s_last_error(void);
/* One-time set-up of the CURRENT device (pass its index, or -1 for "whichever is current"): function attributes
 * (160 KB dynamic LDS) and the cached CU count.  Idempotent and thread-safe; call it once per device before capturing
 * a hipGraph.  Build-defined (the reference has no counterpart: torch initialises its kernels lazily). */
int c2s_init(int device);
/* number of compute units of the current device (for split-K sizing); <=0 on error */
int c2s_device_cus(void);

/* ------------------------------------------------------------------------------------------------
 * Weight packing.  The implicit-GEMM kernels read weights as wpk[tap][cin][coutP] (coutP = cout rounded
 * up to 32, zero filled).  A "tap table" selects and orders the (ky,kx) taps of the source kernel, so one
 * pack kernel serves: Conv2d forward (conv.py:70-80), its data gradient (flipped taps, cin<->cout
 * swapped), ConvTranspose2d (conv.py:384-390) as four 2x2 parity sub-kernels, and their data gradients.
 *   src element (o, c, t) is read at  src[o*stride_o + c*stride_c + tap_off[t]].
 * ------------------------------------------------------------------------------------------------ */
int c2s_pack_weights(const float* src, float* wpk, int cin, int cout, int coutP, int ntaps,
                     long stride_o, long stride_c, const int* host_tap_off, void* stream);
/* All weight packs of a step in one launch.  The caller builds a table of job records once (host memory, record size
 * c2s_pack_job_bytes(), filled by c2s_pack_job_fill: winograd = 0 plain tap-major pack as c2s_pack_weights, 1 the Winograd
 * transform of c2s_pack_weights_winograd, 2 the same in the layout of c2s_pack_weights_winograd16, 3 the parity sub-filters
 * of c2s_pack_weights_s2wino, 4 those of c2s_pack_weights_s2dgrad with cin = gy channels, cout = input channels), uploads it, and calls c2s_pack_batch(device_table, njobs, total_blocks) every step;
 * block_start of job i = sum of c2s_pack_job_blocks over the jobs before it. */
size_t c2s_pack_job_bytes(void);
int c2s_pack_job_fill(void* host_record, const float* src, float* dst, int cin, int cout, int coutP, int ntaps,
                      long stride_o, long stride_c, int winograd, const int* host_tap_off, int block_start);
int c2s_pack_job_blocks(int cin, int coutP, int ntaps, int winograd);
int c2s_pack_batch(const void* device_table, int njobs, int total_blocks, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on f32 MFMA (v_mfma_f32_32x32x2_f32), forward and data-gradient.
 *   out[n, o, oy*osy+ooy, ox*osx+oox] (+)= bias[o] + sum_{c,ky,kx} in[n, c, map(oy*S+ky-pad_y), map(ox*S+kx-pad_x)]
 *                                                               * wpk[ky*KW+kx][c][o]
 * `in` is the channel concatenation of src0 (C0 channels) and src1 (C1 channels; may be NULL/0): the
 * torch.cat of conv.py:408 is never materialised.  map() = zero padding or reflection (conv.py:72-79).
 * Replaces: nn.Conv2d 3x3 / 4x4-stride-2 / 1x1 (conv.py:70-80,263-271,378-382), nn.ConvTranspose2d
 * (conv.py:384-390; four launches, one per output parity) and every convolution_backward-input.
 * Supported (KH,KW,S): (3,3,1) (1,1,1) (2,2,1) (4,4,2).
 * ------------------------------------------------------------------------------------------------ */
typedef struct c2s_conv_desc {
    int N, C0, C1;        /* frames; channels of src0 / src1 */
    int Hin, Win;         /* input plane */
    int Cout, CoutP;      /* real / packed output channels */
    int Hout, Wout;       /* logical output grid computed by this launch */
    int OutH, OutW;       /* physical output plane (>= grid*stride+offset) */
    int KH, KW, S;        /* kernel, input stride */
    int pad_y, pad_x;
    int pad_mode;         /* C2S_PAD_* */
    int osy, osx, ooy, oox; /* output placement stride / offset */
    int accumulate;       /* 0: out = r ; 1: out += r */
    int reflect_adjoint;  /* 1: this launch is the (zero-padded) data gradient of a reflect-padded 3x3 conv, or one
                             2x2 parity sub-kernel of the data gradient of a reflect-padded 4x4-stride-2 conv: the
                             adjoint of the reflection (halo gradient folded back onto rows/cols 1 and H-2/W-2;
                             reflection_pad2d_backward of conv.py:72-79) is applied inside the kernel */
} c2s_conv_desc;

int c2s_conv_igemm(const c2s_conv_desc* d, const float* src0, const float* src1, const float* wpk,
                   const float* bias, float* out, const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3x3 stride-1 pad-1 convolution of a few input channels (<= 10: the Sentinel-2 bands of the first layer), same
 * arithmetic and wpk layout as c2s_conv_igemm, specialised: the 9*Cin x 64 weights of a wave stay in registers, the
 * padded input patch of an 8x32 tile is staged once, no channel chunking.  c2s_conv3x3_smallcin_supported tells
 * whether a descriptor qualifies (one source, Hin % 8 == 0, Win % 32 == 0, Cout % 64 == 0, plain placement).
 * Replaces the first nn.Conv2d of the in_conv block (conv.py:70-80 via utae.py:64-71, timeunet_v1.py, wtae.py).
 * ------------------------------------------------------------------------------------------------ */
int c2s_conv3x3_smallcin_supported(const c2s_conv_desc* d);
int c2s_conv3x3_smallcin(const c2s_conv_desc* d, const float* src, const float* wpk, const float* bias, float* out,
                         const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Transposed 4x4 stride-2 pad-1 convolution, one output-row parity per launch, both column parities fused:
 *   out[n, o, 2i+py, 2j+px] (+)= bias[o] + sum_{c,ty,tx} in[n, c, i-(1-py)+ty, j-(1-px)+tx] * wpk[px][ty*2+tx][c][o]
 * with py = 1 - d->pad_y.  Descriptor: KH = KW = 2, S = 1, pad_mode = ZEROS, C1 = 0, Hout = Hin, Wout = Win,
 * OutH = 2*Hout, OutW = 2*Wout, osy = osx = 2, ooy = py (pad_x / oox unused); wpk = c2s_pack_weights with the 8 taps
 * (px-major).  Each lane stores the two column parities as one float2 (full-line stores; the single-parity
 * launches of c2s_conv_igemm wrote every other float).  reflect_adjoint = 1: data gradient of a reflect-padded
 * 4x4-stride-2 convolution (the halo gradient is folded back inside the kernel).
 * Replaces: nn.ConvTranspose2d(4,2,1) forward (conv.py:384-390) and the convolution_backward-input of the strided
 * down convolutions (conv.py:263-271): two launches (py = 0, 1) each.
 * ------------------------------------------------------------------------------------------------ */
int c2s_conv_xpair(const c2s_conv_desc* d, const float* src, const float* wpk, const float* bias, float* out,
                   const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3x3 stride-1 pad-1 convolution as Winograd F(2x2,3x3) on the f32 MFMA (2.25x fewer multiplies than
 * c2s_conv_igemm; all arithmetic fp32, error a few 1e-7 of sum|a*b|).  Same descriptor as c2s_conv_igemm with
 * KH = KW = 3, S = 1, pad = 1, dense same-size output, even planes with W >= 8, CoutP a multiple of 64;
 * upk = c2s_pack_weights_winograd (G g Gt per filter, c2s_winograd_packed_floats(Cin, CoutP) floats laid out
 * [cout block of 64][chunk of 8 input channels][16][8][64]; tap table as for c2s_pack_weights, so the flipped table
 * gives the data-gradient filters).  reflect_adjoint = 1 (zero-padded launch): the adjoint of the
 * reflection is applied to the raw patches of the border blocks inside the kernel.
 * Replaces: nn.Conv2d 3x3 forward (conv.py:70-80,378-382) and its convolution_backward-input for layers with
 * >= 32 input and >= 64 output channels (the engine keeps c2s_conv_igemm for the rest).
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_winograd_packed_floats(int cin, int coutP);
int c2s_pack_weights_winograd(const float* src, float* upk, int cin, int cout, int coutP, long stride_o, long stride_c,
                              const int* host_tap_off, void* stream);
int c2s_conv3x3_winograd(const c2s_conv_desc* d, const float* src0, const float* src1, const float* upk,
                         const float* bias, float* out, const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The same Winograd F(2x2,3x3) convolution on v_mfma_f32_16x16x4_f32 with the output transform in registers (8-wave
 * workgroups, 8 x 32 pixel tiles; conv_winograd16.hip) for planes at least 32 wide and 8 high.  Same descriptor and
 * semantics as c2s_conv3x3_winograd; upk = c2s_pack_weights_winograd16 (c2s_winograd16_packed_floats(Cin, CoutP) floats,
 * [cout block of 64][chunk of 8 input channels][8 c][4 xi][64 o][4 nu]).  c2s_conv3x3_winograd16_supported = the host-side
 * predicate.
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_winograd16_packed_floats(int cin, int coutP);
int c2s_pack_weights_winograd16(const float* src, float* upk, int cin, int cout, int coutP, long stride_o, long stride_c,
                                const int* host_tap_off, void* stream);
int c2s_conv3x3_winograd16_supported(const c2s_conv_desc* d);
int c2s_conv3x3_winograd16(const c2s_conv_desc* d, const float* src0, const float* src1, const float* upk,
                           const float* bias, float* out, const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 4x4 stride-2 pad-1 convolution (forward) as Winograd F(2x2,2x2) over the four input parities (conv_s2wino.hip): 36 instead
 * of 64 multiplies per (cin, cout) pair and 2x2 output block, integer transform matrices.  Descriptor as c2s_conv_igemm with
 * KH = KW = 4, S = 2, pad = 1, one source with an even number (>= 8) of channels, Hin = 2 Hout, Win = 2 Wout, even output
 * planes at least 32 wide and 8 high, dense output, CoutP a multiple of 64 (c2s_conv4x4s2_winograd_supported).
 * upk = c2s_pack_weights_s2wino (c2s_s2wino_packed_floats(Cin, CoutP) floats, [cout block][chunk of 2 channels][2][4 parities]
 * [64][12]; 16-entry tap table as for c2s_pack_weights).
 * Replaces: nn.Conv2d(4, stride 2, padding 1) forward of DownConvBlock (conv.py:263-271).
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_s2wino_packed_floats(int cin, int coutP);
int c2s_pack_weights_s2wino(const float* src, float* upk, int cin, int cout, int coutP, long stride_o, long stride_c,
                            const int* host_tap_off, void* stream);
int c2s_conv4x4s2_winograd_supported(const c2s_conv_desc* d);
int c2s_conv4x4s2_winograd(const c2s_conv_desc* d, const float* src, const float* upk, const float* bias, float* out,
                           const int* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data gradient of the same 4x4 stride-2 pad-1 convolution as Winograd F(2x2,2x2) per output parity (conv_s2dgrad.hip):
 * gx[2m+e] is a 2-tap filter over gy per parity e; one workgroup per row parity, both column parities per lane (16-byte
 * stores).  The descriptor describes the gradient launch: C0 = gy channels (a multiple of 8, >= 24), Hin x Win = the gy plane
 * (even, at least 32 wide and 8 high), Cout / CoutP (multiple of 64) = the input channels receiving the gradient, Hout x Wout
 * = 2 Hin x 2 Win, dense output; reflect_adjoint = 1: the forward pass was reflect-padded -- the halo gradient is folded in
 * by a variant 3-multiply algorithm in the border blocks (no second pass); accumulate as for c2s_conv_igemm.
 * upk = c2s_pack_weights_s2dgrad (c2s_s2dgrad_packed_floats(C0, CoutP) floats; element (k = gy channel, c = input channel,
 * tap) read at src[c * stride_c + k * stride_k + tap_off[ky * 4 + kx]]).
 * Replaces: convolution_backward-input of nn.Conv2d(4, stride 2, padding 1) in DownConvBlock (conv.py:263-271).
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_s2dgrad_packed_floats(int kc, int csP);
int c2s_pack_weights_s2dgrad(const float* src, float* upk, int kc, int cs, int csP, long stride_c, long stride_k,
                             const int* host_tap_off, void* stream);
int c2s_conv4x4s2_dgrad_winograd_supported(const c2s_conv_desc* d);
int c2s_conv4x4s2_dgrad_winograd(const c2s_conv_desc* d, const float* gy, const float* upk, float* gx, const int* valid,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Opt-in split-precision variant of the 3x3 stride-1 pad-1 convolution (forward and data gradient, same
 * descriptor as c2s_conv_igemm): every fp32 operand is split into two bf16 halves (hi + lo, 16 significant bits)
 * and each product evaluated with three v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi) accumulating in fp32
 * ("bf16x3"): ~1e-5 relative accuracy at ~5x the rate of the exact fp32 MFMA.  Weights are pre-split by
 * c2s_pack_weights_bf16x3 into two bf16 arrays of c2s_bf16x3_packed_elems(cin, coutP) elements each
 * ([cin/8][10 taps][coutP][8]).  Channel counts must be multiples of 8.
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_bf16x3_packed_elems(int cin, int coutP);
int c2s_pack_weights_bf16x3(const float* src, void* whi, void* wlo, int cin, int cout, int coutP, int ntaps,
                            long stride_o, long stride_c, const int* host_tap_off, void* stream);
int c2s_conv3x3_bf16x3(const c2s_conv_desc* d, const float* src0, const float* src1, const void* whi,
                       const void* wlo, const float* bias, float* out, const int* valid, void* stream);
/* Measurement switch: != 0 drops the lo parts (plain bf16 x bf16 products, fp32 accumulation) in the two calls above, to report
 * the deviation of bf16 inputs from the fp32 path (tools/bf16_deviation.py).  Process-wide; off by default. */
void c2s_bf16x3_set_single_product(int on);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient (convolution_backward-weight of the same call sites), split-K over output tiles:
 *   slab[s][t][c][o] = sum over the tiles of slice s of  in[n,c,map(oy*S+ky-pad),..] * gout[n,o,oy,ox]
 * followed by c2s_wgrad_reduce which sums the slices (fixed order: bitwise reproducible) and scatters to
 * the destination layout  dst[o*stride_o + c*stride_c + tap_off[t]] (+)= .
 * ------------------------------------------------------------------------------------------------ */
typedef struct c2s_wgrad_desc {
    int N, C0, C1, Hin, Win;
    int Cout;
    int Hout, Wout;       /* gout plane */
    int KH, KW, S, pad_y, pad_x, pad_mode;
    int nslices;
} c2s_wgrad_desc;

size_t c2s_wgrad_workspace_floats(const c2s_wgrad_desc* d);
int c2s_conv_wgrad(const c2s_wgrad_desc* d, const float* src0, const float* src1, const float* gout,
                   float* slabs, size_t slab_floats, const int* valid, void* stream);
/* Which algorithm c2s_conv_wgrad picks where both apply (process-wide; both forms write the same slabs): winograd_3x3 -- the
 * Winograd F(2x2,3x3) kernel for wide 3x3 stride-1 layers; winograd_4x4s2 -- the F(2x2,2x2) kernel over the four input parities
 * for the 4x4 stride-2 layers of DownConvBlock (conv.py:263-271).  1 = on, 0 = the direct split-K kernel, -1 = the default
 * (on unless C2S_WINOGRAD=0 / C2S_S2WINO=0 in the environment).  For A/B tests of two exact-fp32 evaluations. */
int c2s_wgrad_algorithms(int winograd_3x3, int winograd_4x4s2);
int c2s_wgrad_reduce(const c2s_wgrad_desc* d, const float* slabs, float* dst, long stride_o, long stride_c,
                     const int* host_tap_off, int accumulate, void* stream);
/* All slice sums of a backward pass in one launch: one slab buffer per layer, a table of job records (record size
 * c2s_wgrad_reduce_job_bytes(), filled by c2s_wgrad_reduce_job_fill with the arguments of c2s_wgrad_reduce; block_start of
 * job i = sum of c2s_wgrad_reduce_job_blocks over the jobs before it) built once on the host, uploaded, and passed to
 * c2s_wgrad_reduce_batch(device_table, njobs, total_blocks) every step.  Same order of additions as c2s_wgrad_reduce. */
size_t c2s_wgrad_reduce_job_bytes(void);
int c2s_wgrad_reduce_job_blocks(const c2s_wgrad_desc* d);
int c2s_wgrad_reduce_job_fill(void* host_record, const c2s_wgrad_desc* d, const float* slabs, float* dst, long stride_o,
                              long stride_c, const int* host_tap_off, int accumulate, int block_start);
int c2s_wgrad_reduce_batch(const void* device_table, int njobs, int total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise convolution (groups = C, no bias) forward / data gradient / weight gradient.
 * Replaces DepthwiseSeparableConv2D.depthwise (conv.py:18-20,24), used by W-TAE (wtae.py:148-162).
 * ------------------------------------------------------------------------------------------------ */
int c2s_dwconv_fwd(const float* in, const float* w, float* out, const int* valid, int N, int C, int Hin, int Win,
                   int K, int S, int pad, int pad_mode, void* stream);
/* accumulate != 0: gin += the data gradient (the residual branch of the block already left its gradient there) */
int c2s_dwconv_dgrad(const float* gout, const float* w, float* gin, const int* valid, int N, int C, int Hin, int Win,
                     int K, int S, int pad, int pad_mode, int accumulate, void* stream);
/* gw[C,K,K] = sum_n ...; `partial` is scratch of N*C*K*K floats */
int c2s_dwconv_wgrad(const float* in, const float* gout, float* partial, float* gw, const int* valid, int N, int C,
                     int Hin, int Win, int K, int S, int pad, int pad_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation + ReLU (+ residual), GroupNorm and train/eval BatchNorm share one set of kernels:
 *   rows = (frame n, channel c), each row = HW contiguous floats.
 *   c2s_norm_fwd     : two passes over x: (1) per-(row,segment) mean / M2 (wave per segment); (2) every wave combines
 *                      the partials of its own group (Chan, double) and writes y = relu((x-mean)*a+beta) (+ residual);
 *                      rows of padded frames are filled with pad_value.  Left for the backward: per-row
 *                      (a, beta, mean) in row_ab[n,c,3] and (mean, rstd) per group in group_stats.
 *   c2s_norm_bwd     : given g = dL/dy, two passes over (x, g): (1) per-(row,segment) sums of g' = g*[y>0] and
 *                      g'*xhat, (2) every wave derives its row's coefficients of dx = k1*g' + k2*(x-mean) + k3 from the
 *                      sums of its group and writes dx; then one small launch for dgamma, dbeta and dbias (gradient
 *                      of the bias of the producing convolution).
 * Replaces nn.GroupNorm(4) + ReLU (conv.py:56-60,85-88), nn.BatchNorm2d + ReLU (conv.py:52-53,380,388),
 * the residual add of conv.py:292,410 and their backward ops.
 * stats layout: group stats [G2][2] = (mean, rstd), G2 = N*groups (GROUP) or C (BATCH).
 * ------------------------------------------------------------------------------------------------ */
typedef struct c2s_norm_desc {
    int N, C, HW;
    int kind;             /* C2S_NORM_* */
    int groups;           /* GROUP: channel groups per frame */
    int training;         /* BATCH: 1 = batch statistics (+ running update), 0 = running statistics */
    float eps, momentum;
} c2s_norm_desc;

size_t c2s_norm_workspace_floats(const c2s_norm_desc* d);
/* running_mean/var: BATCH only (may be NULL for GROUP); updated in place when training, and so is the int64 counter
 * num_batches_tracked (may be NULL) */
int c2s_norm_fwd(const c2s_norm_desc* d, const float* x, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, long long* num_batches_tracked, float* group_stats, float* row_ab,
                 const float* residual, float* y, int relu, float* workspace, size_t ws_floats, const int* valid,
                 float pad_value, void* stream);
/* gx may alias g.  dgamma, dbeta, dbias may be NULL (all NULL: the parameter-gradient launch is skipped).  g_residual_out: if non-NULL receives a copy of g (residual branch) */
int c2s_norm_bwd(const c2s_norm_desc* d, const float* x, const float* g, const float* gamma,
                 const float* group_stats, const float* row_ab, int relu, float* gx, float* dgamma,
                 float* dbeta, float* dbias, float* workspace, size_t ws_floats, const int* valid, void* stream);
/* The parameter gradients alone, from the workspace a c2s_norm_bwd call (with dgamma = dbeta = dbias = NULL) left behind:
 * lets the caller run this small launch on another stream, next to the data-gradient chain. */
int c2s_norm_bwd_params(const c2s_norm_desc* d, const float* workspace, float* dgamma, float* dbeta, float* dbias,
                        const int* valid, void* stream);

/* One-pass forms of the two calls above (same results and reference call sites; same arithmetic in the same order): the
 * activation is read ONCE per direction.  A wave keeps its (row, 2048-float segment) in registers, publishes its partial
 * sums as tagged 8-byte words in `sync`, and one wave per workgroup sweeps the words of its normalisation group before
 * the workgroup applies from registers (5 tensor passes per layer and step instead of 8).  `sync` is caller-provided
 * device memory of c2s_norm_onepass_sync_bytes() bytes, ALL ZERO before its first use and never cleared afterwards
 * (tags carry a launch epoch kept in the area); one area serves every launch of ONE stream, of any shape (launches that
 * may run concurrently need areas of their own).  The grid is persistent and sized to what the device holds at once.
 * Word 3 of the area is an error flag: non-zero if a sweep gave up (it never hangs).
 * c2s_norm_onepass_sync_bytes returns 0 for shapes the one-pass form does not take (HW not a whole number of full
 * segments of 256 / 512 / 1024 / 2048 floats; groups of more than 512 segments or, above one segment, not a multiple
 * of 4 segments; BatchNorm in eval mode or with frame flags): use c2s_norm_fwd / c2s_norm_bwd there.  has_valid: whether
 * the call will pass frame flags. */
size_t c2s_norm_onepass_sync_bytes(const c2s_norm_desc* d, int has_valid);
int c2s_norm_fwd_onepass(const c2s_norm_desc* d, const float* x, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked, float* group_stats,
                         float* row_ab, const float* residual, float* y, int relu, const int* valid, float pad_value,
                         void* sync, size_t sync_bytes, void* stream);
int c2s_norm_bwd_onepass(const c2s_norm_desc* d, const float* x, const float* g, const float* gamma,
                         const float* group_stats, const float* row_ab, int relu, float* gx, float* dgamma, float* dbeta,
                         float* dbias, float* workspace, size_t ws_floats, const int* valid, void* sync,
                         size_t sync_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze-and-excitation (squeeze_and_excitation.py:7-30; conv.py:90-91,286-294; constructor flag add_squeeze_excit):
 *   y[n,c,:] = x[n,c,:] * sigmoid(W2 relu(W1 mean_hw(x[n])))[c],  W1 [C/16, C], W2 [C, C/16], no biases.
 * c2s_se_fwd leaves pooled [N,C], hidden [N,C/16] (post-ReLU) and scale [N,C] for the backward; rows of padded frames are
 * filled with pad_value.  c2s_se_bwd: gx (may alias g), gW1 / gW2 (acc_* != 0: accumulate).  16 <= C <= 1024.
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_se_workspace_floats(int N, int C, int HW);
/* x[n,c,:] += bias[c] on real frames: the bias of MBConv's depthwise convolution (mbconv.py:71-79), applied after
 * c2s_dwconv_fwd; its gradient is the per-channel sum of dx that c2s_norm_bwd_params delivers as dbias. */
int c2s_channel_bias_add(float* x, const float* bias, const int* valid, int N, int C, int HW, void* stream);
int c2s_se_fwd(const float* x, const float* W1, const float* W2, float* pooled, float* hidden, float* scale, float* y,
               const int* valid, int N, int C, int HW, float pad_value, float* workspace, size_t ws_floats, void* stream);
int c2s_se_bwd(const float* x, const float* g, const float* W1, const float* W2, const float* pooled, const float* hidden,
               const float* scale, float* gx, float* gW1, float* gW2, int acc_w1, int acc_w2, const int* valid, int N, int C,
               int HW, float* workspace, size_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Frame flags: valid[n] = any(x[n] != pad_value)   (utae.py:201-203, temp_shared_block.py:31)
 * ------------------------------------------------------------------------------------------------ */
int c2s_frame_flags(const float* x, int* valid, int N, long frame_elems, float pad_value, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L-TAE attention (tae.py:451-481 + 738-847), fused and re-associated:
 *   per-pixel GroupNorm(16 groups over C/16 channels x T, padded frames included, tae.py:461),
 *   folded key projection  score[h,t] = U[h,:].xhat_t + s0[b,t,h]   (U = q_h^T Wk_h Wc / sqrt(d_k) [16,C];
 *   s0 [B,T,16] from the biases + positional table; both folded by the host from the parameters),
 *   -1e6 masking of padded frames (tae.py:831), softmax over T, optional dropout (counter-based RNG or an
 *   explicit keep mask [16,P,T]; the attention output is post-dropout, tae.py:836-847),
 *   z[h,:] = sum_t attn[h,t] xhat_t  and  emb[16h+j] = Wc[16h+j,:].z[h,:] + (sum_t attn) bc + sum_t attn pe_t.
 * Everything stays NCHW (pixel index fastest): x [B,T,C,hw]; attn, attn_pre [16,B,T,hw]; emb [B,256,hw]
 * (NULL for W-TAE's attention-only variant, tae.py:619); stats [P,16,2] = (mean, rstd); pe [B,T,16].
 * ------------------------------------------------------------------------------------------------ */
typedef struct c2s_ltae_desc {
    int B, T, C, HW;      /* HW = h*w pixels per frame at the L-TAE resolution */
    int n_head, d_model;  /* built for 16 / 256 */
    float eps;
    float dropout_p;      /* 0 => no dropout */
    uint64_t seed;        /* dropout RNG key */
    const float* keep;    /* optional explicit keep mask [16,P,T] (tests); NULL => RNG when dropout_p>0 */
    const uint64_t* seed_dev; /* optional DEVICE counter mixed into the seed (advanced by the caller between
                                 hipGraph replays so that every step draws a fresh mask); NULL => seed only */
    uint64_t* keep_bits;  /* optional [B*hw][16] words, bit t of word (pixel, head) = "time step t was kept by the attention
                             dropout": written by c2s_ltae_attn_fwd_ws on the register-resident path, read by c2s_ltae_attn_bwd
                             when attn == NULL (c2s_ltae_attn_optional): 1 bit per element instead of a second
                             [16,B,T,hw] float tensor.  NULL otherwise. */
} c2s_ltae_desc;

int c2s_ltae_attn_fwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                      const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                      const int* valid, float* attn, float* attn_pre, float* emb, float* stats, void* stream);
/* Same operation with a caller-provided workspace (c2s_ltae_fwd_workspace_floats).  When the pixel count fills the chip
 * (TimeUNet: L-TAE at full resolution) and C == 64 the full-resolution kernels are used: the register-resident kernel on
 * 16-pixel tiles (x read once; h*w a multiple of 16, >= 4 tiles per CU) or, for other shapes, the streaming kernel on
 * 64-pixel tiles (lane = pixel, 256-byte row segments, weights through the scalar cache); otherwise, or with
 * workspace == NULL, the 16-pixel LDS kernel of c2s_ltae_attn_fwd.  All of them compute the same function to fp32
 * rounding; the choice is made from the descriptor alone. */
size_t c2s_ltae_fwd_workspace_floats(const c2s_ltae_desc* d);
int c2s_ltae_attn_fwd_ws(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                         const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                         const int* valid, float* attn, float* attn_pre, float* emb, float* stats,
                         float* workspace, size_t ws_floats, void* stream);
/* 1 when this shape runs the full-resolution (register-resident or streaming) kernels on the current device (forward: with
 * a workspace and attn_pre; backward: additionally needs g_emb), 0 for the 16-/8-pixel LDS kernels.  Lets tests assert
 * which path they cover. */
int c2s_ltae_uses_streaming(const c2s_ltae_desc* d);
/* Which forward kernel c2s_ltae_attn_fwd_ws launches for this shape: 0 = the 16-pixel LDS kernel (ltae_fwd_kernel: small maps),
 * 1 = the three-pass streaming kernels (ltae_prep + ltae_stream_fwd), 2 = the register-resident kernel (ltae_reg_fwd_kernel);
 * -1 = bad descriptor.  (Measurement harnesses label their numbers with it.) */
int c2s_ltae_fwd_path(const c2s_ltae_desc* d);
/* 1 when a caller that never reads the post-dropout attention weights (TimeUNet_v1.forward without return_att,
 * timeunet.py:176-178,204-205) may pass attn == NULL to c2s_ltae_attn_fwd_ws AND to c2s_ltae_attn_bwd: both then take the
 * register-resident kernels, the forward stores attn_pre and the keep flags as bits (d->keep_bits: 16*B*T*hw floats less to
 * write) and the backward reads those (as many floats less to read).  Needs d->keep_bits on both calls, the RNG mask
 * (d->keep == NULL) and the embedding output. */
int c2s_ltae_attn_optional(const c2s_ltae_desc* d);
size_t c2s_ltae_bwd_workspace_floats(const c2s_ltae_desc* d);
/* g_emb [B,256,hw] or NULL; g_attn [16,B,T,hw] or NULL.  Outputs (all overwritten): gx [B,T,C,hw],
 * gU [16,C], gs0 [B,T,16], gWc [256,C] (embedding path only), gbc [256], ggamma [C], gbeta [C]. */
int c2s_ltae_attn_bwd(const c2s_ltae_desc* d, const float* x, const float* gamma, const float* beta,
                      const float* U, const float* s0, const float* Wc, const float* bc, const float* pe,
                      const int* valid, const float* attn, const float* attn_pre, const float* stats,
                      const float* g_emb, const float* g_attn, float* gx, float* gU, float* gs0, float* gWc,
                      float* gbc, float* ggamma, float* gbeta, float* workspace, size_t ws_floats, void* stream);

/* Parameter side of the L-TAE re-association, n_head = 16, d_model = 256, d_k = 4:
 *   c2s_positional_table : pe[b,t,j] = sin / cos(dates[b,t] / period^(2 floor(j/2)/16))   (positional_encoding.py:16-33)
 *   c2s_ltae_fold_fwd    : U [16,C], s0 [BT,16] from Q [16,1,4], fc1_k (Wk [64,256], bk [64]), inconv (Wc [256,C], bc [256])
 *                          and pe [BT,16]; qwk [16,256] is saved for the adjoint
 *   c2s_ltae_fold_bwd    : final gradients of Q, Wk, bk, Wc, bc from gU, gs0 (+ the attention kernel's direct d Wc / d bc,
 *                          may be NULL); acc_mask bit i set = accumulate into output i (order gQ, gWk, gbk, gWc, gbc) */
int c2s_positional_table(const long long* dates, float* pe, long n, float period, void* stream);
int c2s_ltae_fold_fwd(const float* Q, const float* Wk, const float* bk, const float* Wc, const float* bc, const float* pe,
                      float* U, float* s0, float* qwk, int BT, int C, void* stream);
int c2s_ltae_fold_bwd(const float* Q, const float* Wk, const float* bk, const float* Wc, const float* bc, const float* pe,
                      const float* qwk, const float* gU, const float* gs0, const float* gWc_attn, const float* gbc_attn,
                      float* gQ, float* gWk, float* gbk, float* gWc, float* gbc, int BT, int C, int acc_mask,
                      float* workspace, size_t ws_floats, void* stream);
size_t c2s_ltae_fold_bwd_workspace_floats(void);

/* Learnable positional encoders of the L-TAE (constructor flags use_doy / use_abs_rel_enc / add_linear: tae.py:404-430,
 * 467-479; positional_encoding.py:7-73).  The attention kernels above keep running with a ZERO table pe[BT,16]; the general
 * table pe256 [B,T,256] enters next to them (ltae_pe.hip):
 *   c2s_ltae_pe_table : mode 1 (use_doy): pe256[bt,16h+j] = W[j,dates0] + b[j], W [16,365];  mode 2 (use_abs_rel_enc):
 *                       sinusoid(dates0)[j] + W[j,dates1] + b[j];  mode 3 (add_linear): W [256,256] applied to the tiled
 *                       sinusoid, which is saved in sin256 [BT,256] for the adjoint.  *bad_days counts days outside [0,365)
 *                       (the reference's one_hot raises on them; here they are clamped).
 *   c2s_ltae_pe_fwd   : phase 0, before c2s_ltae_attn_fwd: s0[bt,h] += qwk[h,:] . pe256[bt,:];  phase 1, after it:
 *                       emb[b,16h+j,p] += sum_t attn[h,b,t,p] pe256[b,t,16h+j]   (emb may be NULL: W-TAE)
 *   c2s_ltae_pe_gattn : g_attn_out = g_attn_in (or 0) + <g_emb_h, pe256_h>: the upstream gradient c2s_ltae_attn_bwd takes
 *   c2s_ltae_pe_bwd   : after c2s_ltae_attn_bwd and c2s_ltae_fold_bwd: g_pe [BT,256] (scratch output), the pe part of
 *                       d fc1_k.weight / d Q ACCUMULATED into gWk / gQ, and the encoder's parameter gradients gW, gb.
 *   c2s_ltae_pe_abs_add / _abs_bwd : use_abs_rel_enc TOGETHER with use_doy / add_linear (tae.py:407-423,473): the first
 *                       encoder (mode 1 or 3, on dates0) fills pe256, the second -- AbsolutePositionalEncoder on dates1 -- is
 *                       added to it: pe256[bt,16h+j] += W2[j,dates1] + b2[j]; its parameter gradients from the same g_pe. */
int c2s_ltae_pe_table(int mode, const long long* dates0, const long long* dates1, float period, const float* W, const float* b,
                      float* pe256, float* sin256, int* bad_days, int BT, void* stream);
int c2s_ltae_pe_abs_add(const long long* dates1, const float* W2, const float* b2, float* pe256, int* bad_days, int BT,
                        void* stream);
int c2s_ltae_pe_abs_bwd(const long long* dates1, const float* g_pe, float* gW2, float* gb2, int BT, void* stream);
int c2s_ltae_pe_fwd(const float* qwk, const float* pe256, const float* attn, float* s0, float* emb, int B, int T, int HW,
                    int phase, void* stream);
int c2s_ltae_pe_gattn(const float* g_emb, const float* pe256, const float* g_attn_in, float* g_attn_out, int B, int T, int HW,
                      void* stream);
int c2s_ltae_pe_bwd(int mode, const long long* dates0, const long long* dates1, const float* Q, const float* Wk,
                    const float* qwk, const float* pe256, const float* sin256, const float* attn, const float* g_emb,
                    const float* gs0, float* g_pe, float* gWk, float* gQ, float* gW, float* gb, int B, int T, int HW,
                    void* stream);

/* L-TAE tail (tae.py:442-449,486-488).  Linear(256,C') is a 1x1 convolution on the NCHW embedding
 * (c2s_conv_igemm), BatchNorm1d over P is c2s_norm_* with kind BATCH; the two pieces below are the rest:
 * Dropout(0.2) with the keep mask indexed pixel-major [P,C] like the reference's activations, and the
 * per-pixel GroupNorm(16) over channel groups (out_norm). */
int c2s_dropout_nchw(const float* x, float* y, int B, int C, int HW, float p, uint64_t seed, const uint64_t* seed_dev,
                     const float* keep, void* stream);
int c2s_pixel_gn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int B, int C,
                     int HW, int groups, float eps, void* stream);
size_t c2s_pixel_gn_bwd_workspace_floats(int B, int C, int HW);
int c2s_pixel_gn_bwd(const float* x, const float* gy, const float* gamma, const float* stats, float* gx,
                     float* dgamma, float* dbeta, int B, int C, int HW, int groups, float* workspace,
                     size_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Temporal aggregation of skip features (temporal_aggregator.py:14-45,58-70, mode att_group):
 *   out[b,c,Y,X] = sum_t valid[b,t] * bilinear_up(attn[g(c),b,t])(Y,X) * x[b,t,c,Y,X]
 * bilinear align_corners=False (temporal_aggregator.py:17-19); g(c) = c / (C/n_head).
 * Backward: gx[b,t,c] = up(attn)*gout ; gattn (low resolution, adjoint of the upsample) is ACCUMULATED
 * into gattn (caller zeroes it once; U-TAE sums three resolutions into it).
 * ------------------------------------------------------------------------------------------------ */
typedef struct c2s_agg_desc {
    int B, T, C, H, W;    /* feature maps x [B,T,C,H,W] */
    int n_head, h, w;     /* attention maps [n_head,B,T,h,w] */
} c2s_agg_desc;

/* agg_mode "att_mean" / "mean" (temporal_aggregator.py:46-56,71-77) run the same kernels on a derived weight tensor:
 *   c2s_attn_head_mean     : v[g][i] = mean_h attn[h][i] for every g < n_head (n = B*T*h*w entries per head)
 *   c2s_attn_head_mean_bwd : gattn[h][i] (+)= (1/n_head) sum_g gv[g][i]
 *   c2s_frame_mean_weights : v[g][b][t] = valid[b,t] / #valid frames of b   (a 1x1 "attention map" per frame; valid may be NULL) */
int c2s_attn_head_mean(const float* attn, float* v, int n_head, long n, void* stream);
int c2s_attn_head_mean_bwd(const float* gv, float* gattn, int n_head, long n, int accumulate, void* stream);
int c2s_frame_mean_weights(const int* valid, float* v, int n_head, int B, int T, void* stream);
int c2s_temporal_aggregate_fwd(const c2s_agg_desc* d, const float* x, const float* attn, const int* valid,
                               float* out, void* stream);
size_t c2s_temporal_aggregate_bwd_workspace_floats(const c2s_agg_desc* d);
int c2s_temporal_aggregate_bwd(const c2s_agg_desc* d, const float* x, const float* attn, const int* valid,
                               const float* gout, float* gx, int gx_accumulate, float* gattn,
                               float* workspace, size_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss + optimiser of the train step (train.py:454,463-468; src/learning/utils.py:314-328).
 *   c2s_cross_entropy: nn.CrossEntropyLoss(weight=w, label_smoothing=eps) (train.py:463-468), reduction "mean":
 *   loss = sum_i [(1-eps) w[y_i] nll_i + (eps/K) sum_k w[k] (-log p_ik)] / sum_i w[y_i] over the pixels whose target is not
 *   ignore_index (torch's default -100; targets outside [0,K) are skipped the same way -- torch raises instead -- and COUNTED: the
 *   last three floats of the workspace hold (sum, weight sum, number of such targets) after the call); writes
 *   loss (1 float) and, if glogits != NULL, dL/dlogits.  workspace: see the query.
 *   c2s_adam_flat: torch.optim.Adam defaults on a flat parameter buffer.
 * ------------------------------------------------------------------------------------------------ */
size_t c2s_cross_entropy_workspace_floats(int B, int HW);
int c2s_cross_entropy(const float* logits, const int64_t* target, const float* class_w, float* loss,
                      float* glogits, int B, int K, int HW, float label_smoothing, long long ignore_index,
                      float* workspace, size_t ws_floats, void* stream);
/* step_dev: optional DEVICE int holding the 1-based step count (takes precedence over `step`; lets a captured
 * hipGraph advance the bias correction on replay) */
int c2s_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                  int step, const int* step_dev, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Metrics tail of iterate() (src/learning/utils.py:332-336,377-380; src/learning/miou.py:55-117).
 *   c2s_metrics_update: one pass over logits [B,K,HW] and target int64 [B,HW]:
 *       pred      = argmax_k logits (first maximum, = torch.argmax)
 *       second    = the largest of the remaining classes, lowest index on ties (torch.topk leaves tie order unspecified)
 *       pred_top2 = (target == second) ? second : pred                                   (utils.py:377)
 *       conf[t*K + pred] += 1 ; conf_top2[t*K + pred_top2] += 1   for targets in [0,K)   (miou.py:98-114, rows = target)
 *     conf / conf_top2 are int64 [K,K] DEVICE accumulators (never reset by the call); conf_top2, pred, pred_top2 may be
 *     NULL.  Integer atomics only: bit-exact and order independent.  K <= 32.
 *   c2s_loss_meter_add: acc[0] += *loss, acc[1] += 1 (double[2], device): tnt AverageValueMeter.add(loss.item())
 *     (utils.py:380) without the per-step host synchronisation.
 * Boundary-loss pieces (N4; utils.py:198-222,283-285; focal_loss.py:7-44):
 *   c2s_boundary_target: y_b = (get_dilated(y, K, connectivity 4).sum(1) > 1): 1 where a 4-neighbour holds another class.
 *   c2s_focal_ce: FocalCELoss(gamma, size_average=True, weight=None): mean over targets != ignore_index of
 *     -(1-pt)^gamma log pt; accumulate_loss != 0 adds it to *loss (utils.py:324 loss = loss + loss_b).
 *   c2s_focal_ce_ex: the module's other constructor arguments (focal_loss.py:12-45): size_average == 0: the sum; class_w[K]
 *     non-NULL: the reference multiplies a [N,1] column of gathered weights by the [N] row of focal terms (an N x N outer
 *     product, focal_loss.py:36-39), i.e. its value is mean_i(w[t_i]) * mean_j(f_j) (sum_i * sum_j without size_average) --
 *     computed as that product, without the N^2 tensor.
 * ------------------------------------------------------------------------------------------------ */
int c2s_metrics_update(const float* logits, const long long* target, long long* conf, long long* conf_top2,
                       long long* pred, long long* pred_top2, int B, int K, int HW, void* stream);
/* ConfusionMatrix.add for class-index predictions [n] (miou.py:98-114); pairs outside [0,K) are skipped */
int c2s_confusion_add(const long long* pred, const long long* target, long long* conf, long n, int K, void* stream);
int c2s_loss_meter_add(const float* loss, double* acc, void* stream);
int c2s_boundary_target(const long long* y, long long* y_b, int B, int H, int W, void* stream);
/* test_region of iterate() (utils.py:362-373): keep_boundary != 0: 'boundary' (interior pixels -> ignore_label), else
 * 'interior' (boundary pixels -> ignore_label); boundary as in c2s_boundary_target.  Out of place (y_out != y). */
int c2s_region_relabel(const long long* y, long long* y_out, int B, int H, int W, int keep_boundary, long long ignore_label,
                       void* stream);
size_t c2s_focal_ce_workspace_floats(void);
int c2s_focal_ce(const float* logits, const long long* target, float* loss, float* glogits, int B, int K, int HW,
                 float gamma, long long ignore_index, int accumulate_loss, float* workspace, size_t ws_floats,
                 void* stream);
int c2s_focal_ce_ex(const float* logits, const long long* target, const float* class_w, float* loss, float* glogits, int B,
                    int K, int HW, float gamma, long long ignore_index, int size_average, int accumulate_loss,
                    float* workspace, size_t ws_floats, void* stream);
/* SmoothCrossEntropy2D (smooth_loss.py:18-84): soft targets from the 4-neighbourhood dilation of the label map (classes
 * present at the pixel or a 4-neighbour share 1 - eps*(K - n) evenly, the others get eps = label_smoothing / K; pixels
 * labelled bg_index take bg_distrib[K] instead when it is non-NULL: background_treatment), then CrossEntropyLoss with
 * probability targets and optional class weights: loss = mean over all B*H*W pixels of -sum_k w_k t_k log_softmax(z)_k.
 * glogits (may be NULL) receives dL/dlogits.  workspace[2*512 + 1] holds the number of labels outside [0, K) the pass
 * met (the reference raises on them; here such a pixel contributes nothing and is counted). */
size_t c2s_smooth_ce_workspace_floats(void);
int c2s_smooth_ce(const float* logits, const long long* target, const float* class_w, const float* bg_distrib, float* loss,
                  float* glogits, int B, int K, int H, int W, float label_smoothing, long long bg_index, int accumulate_loss,
                  float* workspace, size_t ws_floats, void* stream);
/* the same with CrossEntropyLoss's other reductions: reduction 0 'mean', 1 'sum', 2 'none' -- pixel_loss [B,H,W] (required
 * for 2, optional otherwise) receives the per-pixel terms; with 2, *loss is their sum and glogits the gradient of that sum */
int c2s_smooth_ce_ex(const float* logits, const long long* target, const float* class_w, const float* bg_distrib, float* loss,
                     float* glogits, float* pixel_loss, int B, int K, int H, int W, float label_smoothing, long long bg_index,
                     int reduction, int accumulate_loss, float* workspace, size_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batch movers on either side of the path.
 *   c2s_collate_series: dataset tail + pad_collate in one pass (s2_ts_cz_crop.py:366-374,393-398; src/utils.py:14-32;
 *     train.py:291).  src = the B series of a batch back to back, [sum_b T_b][Cs][HW] in the storage type of the .npy
 *     files (host-pinned or device memory); offsets int64 [B+1] and src_dates int64 [sum_b T_b] must be device-accessible.
 *     x[b,t,c] = (float(src[b][t][order[c]]) - mean[c]) / std[c] for t < T_b (IEEE fp32 subtract and divide, as torch),
 *     pad_value frames for t >= T_b; dates[b,t] = src_dates or 0; valid[b*T+t] = (t < T_b).  mean / std (host arrays, in
 *     OUTPUT channel order like the reference's norm_values) may both be NULL: no normalisation.
 *   c2s_softmax_stitch: prediction.py:310-333: Softmax(dim=1), top-1 class (first maximum of the probabilities), tiling
 *     '(h w) c h1 w1 -> c (h h1) (w w1)' with grid_w patches per row and the crop to out_h x out_w, for the `npatch`
 *     patches starting at index first_patch: proba [K,out_h,out_w] f32, top1 [out_h,out_w] int64 (may be NULL).
 * ------------------------------------------------------------------------------------------------ */
#define C2S_SRC_F32 0
#define C2S_SRC_I16 1
#define C2S_SRC_U16 2
int c2s_collate_series(const void* src, int src_dtype, const long long* offsets, const long long* src_dates, float* x,
                       long long* dates, int* valid, int B, int T, int C, int Cs, int HW,
                       const int* host_channel_order, const float* host_mean, const float* host_std, float pad_value,
                       void* stream);
/* The same with the dataset's add_ndvi option (s2_ts_cz_crop.py:376-391,401-402): the LAST of the C output channels is
 * (a - b) / (a + b) of the RAW source channels ndvi_a (NIR) and ndvi_b (red) -- 0 where a + b == 0 or where the quotient leaves
 * [-1, 1] -- not normalised; channel_order / mean / std then describe the first C - 1 channels.  ndvi_a < 0: no NDVI channel. */
int c2s_collate_series_ndvi(const void* src, int src_dtype, const long long* offsets, const long long* src_dates, float* x,
                            long long* dates, int* valid, int B, int T, int C, int Cs, int HW, const int* host_channel_order,
                            const float* host_mean, const float* host_std, float pad_value, int ndvi_a, int ndvi_b,
                            void* stream);
int c2s_softmax_stitch(const float* logits, float* proba, long long* top1, int first_patch, int npatch, int K, int ph,
                       int pw, int grid_w, int out_h, int out_w, void* stream);

/* elementwise helpers */
int c2s_fill(float* p, long n, float v, void* stream);
int c2s_add_inplace(float* dst, const float* src, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* C2S_HIP_H */
