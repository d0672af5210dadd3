/* libi2v_hip.so -- C ABI of the MI355X-native (gfx950) cINN-sampling + VAE-decoder hot path.
 *
 * The reference (CompVis/image2video-synthesis-using-cINNs) is pure PyTorch and has no FFI of
 * its own: the boundary it offers is the Python class surface.  This header is
 * the NEW native boundary underneath that surface; each entry point names the reference
 * method (file:line relative to the reference repo) whose arithmetic it replaces.  The Python
 * mirror of the reference classes (image2video-synthesis-using-cinns_amd/...) calls these via
 * ctypes, see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.
 *   - "host tensors" (i2v_tensor) are read during *_load only and are not retained.
 *   - all pointers passed to the compute calls are DEVICE pointers owned by the caller (PyTorch
 *     allocates inputs, outputs and the workspace); the library owns only its packed weights.
 *   - compute calls only ENQUEUE on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and never synchronise.  A handle is bound to the device current at create time,
 *     is not re-entrant, and must be used from one stream at a time.
 *   - return value: 0 = ok, negative = error (I2V_E_*); i2v_last_error() gives the message of
 *     the last failure on the calling thread.
 */
#ifndef I2V_HIP_H
#define I2V_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2V_OK 0
#define I2V_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define I2V_E_MISSING (-2)   /* a state_dict key is missing or has the wrong size */
#define I2V_E_HIP (-3)       /* a HIP runtime call failed */
#define I2V_E_WORKSPACE (-4) /* workspace too small */
#define I2V_E_STATE (-5)     /* weights not loaded */
#define I2V_E_RANGE (-6)     /* an activation left the fp16 range of the split-fp16 operand format (see i2v_dec_status) */

#define I2V_F32 0
#define I2V_I64 1
#define I2V_U8 2

/* One named host tensor of a PyTorch state_dict (contiguous). */
typedef struct {
    const char* name;  /* state_dict key, e.g. "sub_layers.3.coupling.s.0.main.2.weight" */
    const void* data;  /* HOST pointer */
    int64_t numel;
    int32_t dtype;     /* I2V_F32 / I2V_I64 / I2V_U8 */
} i2v_tensor;

const char* i2v_last_error(void);
int i2v_version(void);
/* Number of HIP devices visible; <0 on error.  Used by the Python side to fail loudly. */
int i2v_device_count(void);

/* ------------------------------------------------------------------------------------------
 * cINN flow: ConditionalFlow (stage2_cINN/modules/flow_blocks.py:8-60)
 * ---------------------------------------------------------------------------------------- */
typedef struct i2v_flow i2v_flow;

typedef struct {
    int32_t in_channels;   /* 64 (flow_blocks.py:13); the kernels map channel <-> wavefront lane */
    int32_t embedding_dim; /* E (+30 with control), flow_blocks.py:14 */
    int32_t hidden_dim;    /* 512 = z_dim * flow_mid_channels_factor, get_model.py:34 */
    int32_t hidden_depth;  /* 2, flow_blocks.py:16 */
    int32_t n_flows;       /* 20 */
    int32_t control;       /* 1: blocks fl%4 != 0 run in mode 'cond' (flow_blocks.py:24); 2: every block does */
    int32_t activation;    /* 1 = InvLeakyRelu(0.9) (default), 0 = IgnoreLeakyRelu */
    int32_t skip_actnorm;  /* 1: no ActNorm  (used to expose the bare coupling block, :63-105) */
    int32_t skip_shuffle;  /* 1: no Shuffle */
    int32_t use_graph;     /* 1: replay the launch chain from a captured hipGraph */
    int32_t linear_f16;    /* 1: fp16-operand mode of the s- / t-net Linear layers (BASELINE configs[4]: "fp16 MFMA conditioning
                            * GEMM"): weights rounded to fp16 once at load, activations per layer, v_mfma_f32_16x16x16_f16 with fp32
                            * accumulation; bias, LeakyReLU, coupling and log-det stay fp32.  NOT within the 1e-4 fp32 gate (z rel-L2
                            * ~1e-3); 0 (default): exact fp32 matrix cores.  Needs the tile-chain geometry. */
} i2v_flow_cfg;

int i2v_flow_create(const i2v_flow_cfg* cfg, i2v_flow** out);
void i2v_flow_destroy(i2v_flow* f);
/* Packs the 80 MLPs, ActNorm and Shuffle parameters of ConditionalFlow.state_dict() into the
 * streaming layout and uploads them.  Keys: sub_layers.{i}.{norm_layer.{loc,scale},
 * coupling.{s,t}.{0,1}.main.{0,2,4,6}.{weight,bias}, shuffle.{forward,backward}_shuffle_idx}. */
int i2v_flow_load(i2v_flow* f, const i2v_tensor* tensors, int32_t n_tensors);
size_t i2v_flow_workspace_bytes(const i2v_flow* f, int32_t batch);
/* Bytes of parameters streamed per pass (the algorithmic HBM traffic of SURVEY §8d). */
size_t i2v_flow_param_bytes(const i2v_flow* f);
/* ConditionalFlow.forward(x, embedding, reverse=False), flow_blocks.py:42-51.
 * x [B,64], embed [B,E] -> zt [B,64], logdet [B]. */
int i2v_flow_forward(i2v_flow* f, const float* x, const float* embed, float* zt, float* logdet,
                     void* workspace, size_t workspace_bytes, int32_t batch, void* stream);
/* ConditionalFlow.forward(x, embedding, reverse=True), flow_blocks.py:53-57. */
int i2v_flow_inverse(i2v_flow* f, const float* residual, const float* embed, float* z,
                     void* workspace, size_t workspace_bytes, int32_t batch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Leaf modules of the flow, individually callable (sub-module classes stay usable)
 * ---------------------------------------------------------------------------------------- */
/* BasicFullyConnectedNet (stage2_cINN/modules/modules.py:9-30): Linear(dim,hidden) -> LeakyReLU(0.01) ->
 * depth x [Linear(hidden,hidden) -> LeakyReLU(0.01)] -> Linear(hidden,out_dim).  Keys main.{0,2,...}.{weight,bias}. */
typedef struct i2v_mlp i2v_mlp;
int i2v_mlp_create(int32_t dim, int32_t hidden_dim, int32_t depth, int32_t out_dim, i2v_mlp** out);
void i2v_mlp_destroy(i2v_mlp* m);
int i2v_mlp_load(i2v_mlp* m, const i2v_tensor* tensors, int32_t n_tensors);
size_t i2v_mlp_workspace_bytes(const i2v_mlp* m, int32_t batch);
/* x [B,dim] -> y [B,out_dim] */
int i2v_mlp_forward(i2v_mlp* m, const float* x, float* y, void* workspace, size_t workspace_bytes, int32_t batch,
                    void* stream);

/* Per-channel elementwise ops on a contiguous [B][C][inner] tensor. */
#define I2V_OP_ACTNORM_FWD 0  /* out = p1[c] * (x + p0[c])        ActNorm.forward, modules.py:80 (p0 = loc, p1 = scale) */
#define I2V_OP_ACTNORM_REV 1  /* out = x / p1[c] - p0[c]          ActNorm.reverse, modules.py:100 */
#define I2V_OP_INVLRELU_FWD 2 /* out = x * (x >= 0 ? 1 : alpha)   InvLeakyRelu.forward, flow_blocks.py:180-181 */
#define I2V_OP_INVLRELU_REV 3 /* out = x / (x >= 0 ? 1 : alpha)   InvLeakyRelu.reverse, flow_blocks.py:185-186 */
#define I2V_OP_GATHER 4       /* out[b,c] = x[b, idx[c]]          Shuffle, flow_blocks.py:152-154 (idx: device int64) */
int i2v_channel_op(int32_t op, const float* x, float* out, int32_t batch, int32_t channels, int32_t inner,
                   const float* p0, const float* p1, const int64_t* idx, float alpha, void* stream);
/* Per-row mean and unbiased std of x [rows][n] (ActNorm.initialize, modules.py:43-63). */
int i2v_row_mean_std(const float* x, int32_t rows, int32_t n, float* mean, float* std, void* stream);
/* out[b] = hw * sum_c log|scale[c]| for b < batch (ActNorm log-det, modules.py:86-88). */
int i2v_actnorm_logdet(const float* scale, int32_t channels, float hw, float* out, int32_t batch, void* stream);

/* Measurement helper (no reference counterpart; used by bench.py only): enqueues an MFMA-only loop -- `workgroups` x 512
 * threads, `iters` k-steps of 12 v_mfma_f32_32x32x16_f16 per wavefront on live pseudo-random register operands, no
 * memory traffic -- and returns the fp16 MFMA FLOPs it executes in *flops.  Timed by the caller with events on `stream`,
 * it gives the matrix-core rate the chip SUSTAINS under power management, next to the data-sheet peak.
 * scratch: device buffer of workgroups * 512 floats. */
int i2v_probe_mfma_f16(int32_t workgroups, int32_t iters, float* scratch, double* flops, void* stream);


/* ------------------------------------------------------------------------------------------
 * Stage-1 decoder: Generator (stage1_VAE/modules/decoder.py:55-120)
 * ---------------------------------------------------------------------------------------- */
typedef struct i2v_dec i2v_dec;

typedef struct {
    int32_t channel_factor; /* nf, decoder.py:59 (multiple of 8) */
    int32_t z_dim;          /* 64 */
    int32_t upsample_s[2];  /* decoder.py:66 */
    int32_t upsample_t[2];  /* decoder.py:67 */
    int32_t spectral_norm;  /* decoder.py:64 */
    int32_t mma;            /* 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = split-fp16 3-term MFMA; 2 = auto: split-fp16 with a
                             * per-layer fallback to the exact-fp32 kernels behind the range guard -- both weight sets are packed, every
                             * forward synchronises its stream, looks at the operand maxima the writers published, switches the 3x3x3 convs
                             * whose operand left the window the split format holds 1e-4 in (for the life of the handle) and runs the call
                             * again; a checkpoint inside the window runs exactly the launches of mma = 1 (i2v_dec_fallback_layers) */
} i2v_dec_cfg;

int i2v_dec_create(const i2v_dec_cfg* cfg, i2v_dec** out);
void i2v_dec_destroy(i2v_dec* d);
/* Folds W/sigma (signed sigma = u.(W_mat v), torch spectral_norm eval semantics, hook at
 * decoder.py:20-25), re-lays every conv for the implicit-GEMM kernels and uploads.  Keys as in
 * Generator.state_dict(): fc.*, {head_0,g_0..g_4}.{conv_0,conv_1,conv_s}.{weight_orig,weight_u,
 * weight_v,bias} (or .weight when spectral_norm = 0), .norm_0.{conv,conv_gamma,conv_beta}.*,
 * .norm_1.linear.*, .norm_s.bn.*, conv_img.*. */
int i2v_dec_load(i2v_dec* d, const i2v_tensor* tensors, int32_t n_tensors);
/* Output geometry [T, H, W] of one decoder pass (T = 16 for every shipped config). */
int i2v_dec_out_shape(const i2v_dec* d, int32_t* t, int32_t* h, int32_t* w);
size_t i2v_dec_workspace_bytes(const i2v_dec* d, int32_t batch, int32_t img_h, int32_t img_w);
double i2v_dec_flops_per_sample(const i2v_dec* d, int32_t img_h, int32_t img_w);
/* Generator.forward(img, motion), decoder.py:97-120.
 * img [B,3,img_h,img_w] (NCHW, [-1,1]), motion [B,z_dim] -> out [B,T,3,H,W] contiguous. */
int i2v_dec_forward(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, const float* motion,
                    float* out, void* workspace, size_t workspace_bytes, int32_t batch, void* stream);
/* The same with explicit SAMPLE strides (in floats; 0 = dense) for the start frames and the output: img sample b starts at
 * img + b * img_bstride (its [3,img_h,img_w] planes stay contiguous), out sample b at out + b * out_bstride (its [T,3,H,W]
 * block stays contiguous).  This is what the autoregressive loop of Model.forward (get_model.py:68-73) needs to run without
 * torch.cat / .contiguous() copies: pass k decodes straight into frames [16k, 16k+16) of ONE pre-allocated
 * [B, vid_length, 3, H, W] buffer (out = buf + 16k * 3*H*W, out_bstride = vid_length * 3*H*W) and pass k+1 reads its start
 * frames seq[:, -1] from the same buffer (img = buf + (16k + 15) * 3*H*W, img_bstride = vid_length * 3*H*W). */
int i2v_dec_forward_strided(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, int64_t img_bstride,
                            const float* motion, float* out, int64_t out_bstride, void* workspace, size_t workspace_bytes,
                            int32_t batch, void* stream);
/* Optional first half of Generator.forward: the SPADE conditioning branches of all six blocks (normalization_layer.py:20-23:
 * F.interpolate(start frame) -> Conv2d(3,128) + lrelu -> conv_gamma | conv_beta).  They depend on the start frame only, not on
 * the motion latent, so a caller can enqueue them on a SIDE stream while the cINN pass that produces the latent runs
 * (get_model.py:59-66), and then call i2v_dec_forward with the SAME img pointer, size, batch and workspace (after making its
 * stream wait for the side stream): that forward skips the branches and reads the prepared gamma | beta maps from the
 * workspace.  One prepare serves AT MOST the next forward call on the handle: every i2v_dec_forward* entry -- matching or not,
 * successful or not (I2V_E_RANGE of the previous call, bad arguments, workspace too small) -- consumes or discards it before
 * anything else.  The identity test is by address: the CONTENTS of img must not change between the prepare and its forward;
 * a caller that refills the buffer in place calls i2v_dec_prepare_cancel (the Python binding does, keyed on the tensor's
 * version counter).  Same kernels, same bits.
 * Since round 5 the branches are enqueued on a side stream the HANDLE owns, ordered behind everything already on `stream` (an event),
 * and the consuming forward waits per level (events): the caller's stream stays free, e.g. for the cINN pass, and needs no stream
 * of its own for this.  (While `stream` captures a graph, with I2V_DEC_OVERLAP=0 or the debug tap on they run inline on `stream`.)
 * A forward WITHOUT prepared maps forks its own branches the same way, underneath its first levels.
 * LIFETIME CONTRACT of a prepare that ran on the side stream: `workspace` and `img` must stay allocated and unmodified until the
 * NEXT i2v_dec_forward* / i2v_dec_prepare / i2v_dec_join call on the handle has been enqueued -- each of them either consumes the
 * prepared maps (waiting per level) or makes its stream wait for the whole side stream before it does anything else, so from then
 * on synchronising THAT stream bounds the lifetime of both buffers again.  A caller that wants to release or reuse them without
 * another forward calls i2v_dec_join(d, stream) and orders the release behind `stream`.  i2v_dec_destroy synchronises the side
 * stream.  Graph capture: a forward captured on `stream` runs everything inline (no side stream, no cross-stream events); a prepare
 * that was forked BEFORE the capture began is dropped by it (join it with i2v_dec_join before capturing). */
int i2v_dec_prepare(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, void* workspace, size_t workspace_bytes,
                    int32_t batch, void* stream);
/* Drops a pending prepare (no-op without one). */
int i2v_dec_prepare_cancel(i2v_dec* d);
/* Makes `stream` wait for everything the handle has enqueued on its own side stream (a forked i2v_dec_prepare that no forward has
 * consumed) and drops the pending prepare: behind this call `stream` bounds the lifetime of the workspace / start frames again.
 * No reference counterpart (the reference has one stream: generate_samples.py:47-54). */
int i2v_dec_join(i2v_dec* d, void* stream);
/* The handle's side work (SPADE conditioning branches, learned shortcuts, i2v_dec_prepare) runs on `side_stream` instead of a stream
 * the handle creates -- typically the stream the caller's cINN prefetch already runs on, so that a job uses main + ONE side stream
 * (+ its collation stream) next to RCCL's: HIP multiplexes streams onto four hardware queues, and streams that share a queue
 * serialise.  `side_stream` stays the caller's (it must outlive the handle or be reset with NULL: the handle then creates its own
 * again).  Same kernels, same events, same bits.  No reference counterpart (the reference has one stream). */
int i2v_dec_set_side_stream(i2v_dec* d, void* side_stream);
/* mma = 2 (auto): which 3x3x3 convs the range guard has switched to the exact-fp32 kernels so far: bit 2 * block + (0: conv_0, 1: conv_1),
 * blocks head_0, g_0 .. g_4; bit 30: the whole handle (an overflow outside the conv operands: SPADE's activation, a shortcut, conv_img).
 * reruns (optional): forwards that were run a second time because a layer had to be switched.  Always 0 / 0 for mma = 0, 1.
 * No reference counterpart (the reference computes in fp32 throughout: decoder.py:99-120). */
int i2v_dec_fallback_layers(i2v_dec* d, int32_t* mask, int32_t* reruns);
/* Roofline instrumentation.  With profiling on, every 3x3x3 Conv3d launch (the dominant kernel) is bracketed by HIP
 * events recorded on the launch stream -- no synchronisation is added to the forward.  After the caller has
 * synchronised, i2v_dec_get_profile resolves the pending pairs and returns the totals since set_profile(d, 1):
 * summed kernel time [ms], summed algorithmic FLOPs (2*M*N*K of the reference's conv per launch), summed matrix-core
 * FLOPs actually issued (3 per product in split-fp16 mode, 18 of 27 taps in temporal-duplication mode) and launches. */
int i2v_dec_set_profile(i2v_dec* d, int32_t on);
int i2v_dec_get_profile(i2v_dec* d, double* conv3_ms, double* conv3_flops, double* conv3_mfma_flops, int64_t* conv3_launches);
/* The same totals per layer: layer = 2 * block + {0: conv_0, 1: conv_1}, block 0..5 = head_0, g_0 .. g_4 (decoder.py:74-79).
 * name receives "<block>.conv_<i>"; kernel: 0 = fp32 MFMA implicit GEMM, 1 = split-fp16 direct, 2 = split-fp16 Winograd. */
int i2v_dec_get_layer_profile(i2v_dec* d, int32_t layer, char* name, int32_t name_len, double* ms, double* flops,
                              double* mfma_flops, int64_t* launches, int32_t* kernel);
/* Test hook: during the next forwards copy up to max_floats of one channels-last intermediate of GeneratorBlock
 * `block` (0 = head_0 .. 5 = g_4) into dst (device).  which: 0 = SPADE (1+gamma | beta) [B,H,W,2C], 1 = lrelu(Spade(x)),
 * 2 = conv_0 output, 3 = lrelu(ADAIN(.)), 4 = shortcut (low resolution), 5 = block output.  dst = NULL disables. */
/* Range guard of the split-fp16 ("hl16") operand format (mma = 1): the kernels that produce conv operands raise a sticky
 * device flag when a value is non-finite or exceeds the fp16 range (|x| > 65504 -- a regime no synthetic-weight parity
 * test reaches, but a released checkpoint with a large SPADE (1 + gamma) might).  Nothing synchronises on the fast path:
 * every i2v_dec_forward ends with an async copy of the flag to pinned host memory, and the NEXT i2v_dec_forward (or
 * i2v_gblock_forward) on the handle returns I2V_E_RANGE when it finds it set.  i2v_dec_status synchronises `stream`, reads
 * the flag (bit 0 = overflow seen) and optionally clears it; use mma = 0 (exact fp32 MFMA) for such checkpoints.
 * Bit 1 (value 2) = UNDERFLOW warning: the format has an absolute error floor of ~2^-25 (the lo part is an fp16 subnormal
 * below |x| = 2^-3), so a conv whose whole operand tensor lies below 2^-10 no longer holds the 1e-4 gate (measured table:
 * INTEGRATION.md §3).  The two operand writers of every block (the inputs of conv_0 and conv_1: 12 tensors per forward) publish
 * the largest |activation| they wrote; a non-zero tensor whose maximum is below 2^-10 sets bit 1.  Coverage, as built: the
 * maximum is per operand TENSOR over the whole batch (one normal sample hides an underflowing one in the same call), and SPADE's
 * internal 128-channel operand, the EPI_HL16 conv epilogue and conv_img's input are not watched (they carry bit 0 only).  It is reported by i2v_dec_status only (sticky until reset) and does NOT make the next call fail:
 * the output is finite and merely less precise; mma = 0 is exact there too. */
int i2v_dec_status(i2v_dec* d, int32_t* flags, int32_t reset, void* stream);

int i2v_dec_debug_tap(i2v_dec* d, int32_t block, int32_t which, float* dst, size_t max_floats);

/* ------------------------------------------------------------------------------------------
 * Decoder sub-modules, individually callable with the reference's [B][C][T][H][W] tensors:
 * GeneratorBlock (decoder.py:7-52), Spade / Norm3D / ADAIN (normalization_layer.py:5-51).
 * T, H, W must be powers of two (>= 1) so the convolutions tile into bricks.
 * ---------------------------------------------------------------------------------------- */
typedef struct i2v_gblock i2v_gblock;
int i2v_gblock_create(int32_t n_in, int32_t n_out, int32_t z_dim, int32_t spectral_norm, int32_t mma, i2v_gblock** out);
void i2v_gblock_destroy(i2v_gblock* g);
/* Keys relative to the block: conv_{0,1,s}.*, norm_0.{conv,conv_gamma,conv_beta}.*, norm_1.linear.*, norm_s.bn.*.
 * Groups of keys that are absent are skipped, so a handle can carry a lone Spade / ADAIN / Norm3D. */
int i2v_gblock_load(i2v_gblock* g, const i2v_tensor* tensors, int32_t n_tensors);
size_t i2v_gblock_workspace_bytes(const i2v_gblock* g, int32_t batch, int32_t t, int32_t h, int32_t w);
/* GeneratorBlock.forward(x, cond1 = z, cond2 = img): x [B,n_in,T,H,W] -> out [B,n_out,T,H,W]. */
int i2v_gblock_forward(i2v_gblock* g, const float* x, const float* z, const float* img, int32_t img_h, int32_t img_w,
                       float* out, void* workspace, size_t workspace_bytes, int32_t batch, int32_t t, int32_t h, int32_t w,
                       void* stream);
/* The block's own range guard (see i2v_dec_status): synchronises `stream`, returns the sticky flag word of the split-fp16
 * operand writers of this handle (bit 0 = an operand left the fp16 range) and optionally clears it. */
int i2v_gblock_status(i2v_gblock* g, int32_t* flags, int32_t reset, void* stream);
/* part 0: Spade.forward(x, cond = img [B,3,img_h,img_w]); part 1: ADAIN.forward(x [B,n_mid,...], cond = z [B,z_dim]);
 * part 2: Norm3D.forward(x).  Output has the shape of x. */
int i2v_gblock_norm(i2v_gblock* g, int32_t part, const float* x, const float* cond, int32_t img_h, int32_t img_w, float* out,
                    void* workspace, size_t workspace_bytes, int32_t batch, int32_t t, int32_t h, int32_t w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conditioning embedder (row N1): ResnetEncoder.encode(x).mode() -- stage2_cINN/AE/modules/AE.py:91-166,
 * distributions.py:41-42.  torchvision-0.8.1 ResNet-50 with InstanceNorm2d (use_batchnorm = 0) or eval-mode
 * BatchNorm2d (1), fc = Conv2d(2048, 2E, 1); returns the posterior mean [B, E].
 * Keys: model.conv1.weight, model.layer{1..4}.{i}.conv{1,2,3}.weight, model.layer{k}.0.downsample.0.weight,
 * (BatchNorm) model.bn1.*, ...bn{1,2,3}.*, ...downsample.1.* {weight,bias,running_mean,running_var},
 * model.fc.sub_layers.0.{weight,bias}.  Image side must be a power of two >= 64.
 * ---------------------------------------------------------------------------------------- */
typedef struct i2v_embedder i2v_embedder;
int i2v_embedder_create(int32_t z_dim, int32_t use_batchnorm, i2v_embedder** out);
void i2v_embedder_destroy(i2v_embedder* e);
int i2v_embedder_load(i2v_embedder* e, const i2v_tensor* tensors, int32_t n_tensors);
size_t i2v_embedder_workspace_bytes(const i2v_embedder* e, int32_t batch, int32_t h, int32_t w);
/* img [B,3,h,w] in [-1,1] (NCHW) -> embed [B, z_dim] */
int i2v_embedder_forward(i2v_embedder* e, const float* img, int32_t h, int32_t w, float* embed, void* workspace,
                         size_t workspace_bytes, int32_t batch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Motion encoder of the transfer path (row N3): Encoder.forward -- stage1_VAE/modules/resnet3D.py:138-219
 * (3D ResNet-18, GroupNorm(16), conv_mu / conv_var).  Model.transfer (get_model.py:87) uses mu.
 * Keys: conv1.weight, norm1.*, layer.{L}.{i}.{conv1,conv2}.weight, .bn{1,2}.*, .downsample.{0.weight,1.*},
 * conv_mu.*, conv_var.*.  Frames must be powers of two >= 64 and reduce to a [1,4,4] map.
 * ---------------------------------------------------------------------------------------- */
typedef struct i2v_encoder3d i2v_encoder3d;
typedef struct {
    int32_t z_dim;        /* 64 */
    int32_t channels[5];  /* e.g. [64,128,256,512,512] */
    int32_t stride_s[4];  /* e.g. [1,2,2,2] */
    int32_t stride_t[4];  /* e.g. [1,2,2,2] */
    int32_t use_max_pool; /* must be 0 (every shipped config) */
} i2v_encoder3d_cfg;
int i2v_encoder3d_create(const i2v_encoder3d_cfg* cfg, i2v_encoder3d** out);
void i2v_encoder3d_destroy(i2v_encoder3d* e);
int i2v_encoder3d_load(i2v_encoder3d* e, const i2v_tensor* tensors, int32_t n_tensors);
size_t i2v_encoder3d_workspace_bytes(const i2v_encoder3d* e, int32_t batch, int32_t t, int32_t h, int32_t w);
/* x [B,3,t,h,w] -> mu [B,z], logvar [B,z]; sample = eps * exp(0.5 logvar) + mu when sample != NULL (eps [B,z]). */
int i2v_encoder3d_forward(i2v_encoder3d* e, const float* x, int32_t t, int32_t h, int32_t w, const float* eps, float* sample,
                          float* mu, float* logvar, void* workspace, size_t workspace_bytes, int32_t batch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I2V_HIP_H */
