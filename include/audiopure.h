/*
 * audiopure.h — C-ABI of the MI355X-native diffusion-purification hot path.
 *
 * The reference (cychomatica/AudioPure) has no FFI layer: its boundary is Python
 * class identity.  This header is the boundary a maintainer
 * would bind from the reference's Python classes (ctypes stub: INTEGRATION.md);
 * each entry point names the reference code it replaces (file:line relative to
 * the reference checkout).
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns 0 on success or a
 *    negative errno-style code and never throws; ap_last_error() gives the text.
 *  - every `*_dev` / float* tensor argument is a DEVICE pointer owned by the
 *    caller (e.g. the PyTorch caching allocator); the library allocates device
 *    memory only inside ap_ctx_load_wavenet / ap_ctx_prepare_backward / ap_m5_create.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *    All work is enqueued asynchronously on it; nothing synchronises the host
 *    (ap_ctx_load_wavenet / ap_ctx_prepare_backward / ap_m5_create excepted), so every
 *    launch function is hipGraph-capturable.
 *  - tensors are fp32, layout [B][C][L] row-major with the sample axis contiguous,
 *    exactly the reference's `[B,1,16000]` / `[B,C,L]` tensors.
 *  - noise: `z` arguments may be NULL; then N(0,1) samples come from the library's
 *    counter-based Philox4x32-10 keyed on (seed, draw, GLOBAL utterance index
 *    = utt_offset + b, sample index), so results do not depend on how a batch
 *    is sharded over GPUs.  Non-NULL z is used as given (parity tests inject it;
 *    the reference draws torch.normal on the global RNG, diffwave_ddpm.py:66,100).
 */
#ifndef AUDIOPURE_H
#define AUDIOPURE_H

#include <stddef.h>
#include <stdint.h>

#define AP_CONV_SPLIT 0x100
#define AP_CONV_1D 0x200
#define AP_CONV_SPLIT_F16 0x400
#define AP_CONV_DILATION(d) ((d) << 16)

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ap_ctx ap_ctx;
typedef struct ap_m5 ap_m5;

/* arithmetic mode of the two residual-block GEMMs */
enum {
  AP_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate   */
  AP_PREC_BF16 = 1,  /* bf16 operands (RNE), fp32 accumulate; activations stay fp32 in HBM.  Built for the shipped
                        configuration only: res_channels == skip_channels == 256 (any dilation of a power-of-two cycle,
                        any clip length); other shapes return -EINVAL at the first launch (ap_last_error says which) */
  AP_PREC_F32_SPLIT = 2, /* fp32 operands split exactly into three bf16 parts, the six partial products >= 2^-16 of
                            each product on v_mfma_f32_32x32x16_bf16, fp32 accumulate: fp32-class results (dropped
                            terms < 2^-23 of a product) at 6/16 of the fp32 matrix instruction's time; C = 256.  Held to the
                            fp32 kernels' tolerances, and on adversarial operands (cancellation, a 2^40 dynamic range)
                            to <= 2 x the direct fp32 kernel's error against fp64
                            (tests/test_gpu_parity.py::test_fp32_class_modes_bound_their_error_on_adversarial_operands) */
  /* (value 3, AP_PREC_F32_SPLIT_F16 -- two fp16 parts per operand -- was removed in round 5: fp16's exponent range loses
     channels a 2^20 dynamic range apart outright, profiles/r5_fp32_class_adversarial_error.txt) */
  AP_PREC_BF16_STORE = 4  /* AP_PREC_BF16's arithmetic with the residual stream kept in HBM as bf16 (third precision
                             row): each layer hands the next one u = bf16(h' + part_t of that layer) -- the dilated conv's operand
                             and, by the reference's in-place alias (WaveNet.py:77,84), the value the residual carries -- as an
                             image [B][C/32][L][32] (ap_resblock_fwd_u); skip stays fp32 (deferred-skip form).  One more rounding
                             per layer than AP_PREC_BF16 (the residual sees bf16(u) instead of fp32 u).  Input gradients through
                             ap_resblock_fwd_u_save + ap_resblock_bwd_bf16_saved (the bf16 backward kernels).  C = 256. */
};

/* configs/config.json "wavenet_config" + "diffusion_config" (reference: configs/config.json:2-17) */
typedef struct ap_config {
  int32_t res_channels;     /* C  (256) */
  int32_t skip_channels;    /* S  (256); this build requires S == C, C % 64 == 0, C <= 256 */
  int32_t num_res_layers;   /* 36 */
  int32_t dilation_cycle;   /* 12 -> dilation 2^(n mod 12) */
  int32_t embed_dim_in;     /* 128 */
  int32_t embed_dim_mid;    /* 512 */
  int32_t embed_dim_out;    /* 512 */
  int32_t T;                /* 200 */
  float beta_0;             /* 1e-4 */
  float beta_T;             /* 0.02 */
  int32_t precision;        /* AP_PREC_* */
} ap_config;

const char *ap_last_error(void);
int ap_version(void);

/* ---- context -------------------------------------------------------------------------------
 * replaces: create_diffwave_model (diffusion_models/diffwave_ddpm.py:395-411) +
 *           calc_diffusion_hyperparams (DiffWave_Unconditional/util.py:96-123).
 * Host-only (no device allocation, usable without a GPU).  The default schedule tables are the
 * closed form evaluated in double and rounded to fp32; a caller that holds the reference's own
 * fp32 tables (the `diffusion_hyperparams` dict every reference caller passes to DiffWave,
 * diffwave_ddpm.py:18-32) installs them with ap_ctx_set_schedule so both sides use identical
 * coefficients. */
int ap_ctx_create(const ap_config *cfg, ap_ctx **out);
int ap_ctx_destroy(ap_ctx *ctx);

/* Install T-entry host tables Beta, Alpha, Alpha_bar, Sigma (util.py:111-122). */
int ap_ctx_set_schedule(ap_ctx *ctx, const float *beta, const float *alpha, const float *alpha_bar,
                        const float *sigma, int T);
/* Install the VP-SDE tables discrete_betas, alphas_cumprod (diffwave_sde.py:56-58). */
int ap_ctx_set_sde_schedule(ap_ctx *ctx, const float *discrete_betas, const float *alphas_cumprod, int T);

/* Number of fp32 elements of the weight blob for this config. */
size_t ap_wavenet_blob_elems(const ap_config *cfg);

/* Load the epsilon-network weights.  `blob_dev` is the concatenation (fp32, device) of the
 * reference checkpoint's `model_state_dict` tensors IN STATE-DICT ORDER, un-folded
 * (`weight_g`/`weight_v` as stored by nn.utils.weight_norm; WaveNet.py:23-34,138-172):
 *   init_conv.0.conv.{bias,weight_g,weight_v}, residual_layer.fc_t1.{weight,bias},
 *   residual_layer.fc_t2.{weight,bias}, then per block n: fc_t.{weight,bias},
 *   dilated_conv_layer.conv.{bias,weight_g,weight_v}, res_conv.{bias,weight_g,weight_v},
 *   skip_conv.{bias,weight_g,weight_v}; final_conv.0.conv.{bias,weight_g,weight_v},
 *   final_conv.2.conv.{weight,bias}.
 * `embed_freq_dev`: the embed_dim_in/2 frequencies exp(-j ln(1e4)/(half-1)) as the host
 * computes them (util.py:86-88).  Allocates the device weight slab, folds W = g*v/||v|| on device (replaces the 110
 * _weight_norm_interface calls per forward) and packs the MFMA
 * operand images.  Synchronises `stream` before returning. */
int ap_ctx_load_wavenet(ap_ctx *ctx, const float *blob_dev, size_t n_elems,
                        const float *embed_freq_dev, void *stream);

/* Copy the folded (un-packed) weight of one conv back out, for fold-parity tests.
 * which: 0 = dilated conv of `layer` [2C][C][3], 1 = res_conv [C][C], 2 = skip_conv [S][C],
 *        3 = final_conv.0 [S][S], 4 = init_conv [C]. */
int ap_ctx_get_folded(ap_ctx *ctx, int which, int layer, float *out_dev, size_t n_elems, void *stream);

/* Host copies of the schedule (T floats each): which 0=Beta 1=Alpha 2=Alpha_bar 3=Sigma. */
int ap_ctx_get_schedule(ap_ctx *ctx, int which, float *out_host, int n);

/* Measurement hook (bench.py roofline leg): while enabled, every residual-block launch is bracketed by a
 * pair of HIP events on the launch stream; ap_profile_read waits for them, returns the summed kernel time
 * and the number of launches since ap_profile_enable, and resets the counter.  Not graph-capturable while on. */
int ap_profile_enable(ap_ctx *ctx, int enable);
int ap_profile_read(ap_ctx *ctx, double *total_ms, int64_t *launches);
int ap_profile_is_enabled(ap_ctx *ctx);   /* 1 while the hook records events (the launches are then not graph-capturable) */
/* The same reading split by kernel: [0] residual-block launches, [1] skip-GEMM launches of the deferred-skip form (whose time
 * ap_profile_read folds into total_ms while counting only the block launches, so "ms per layer" stays comparable). */
int ap_profile_read_split(ap_ctx *ctx, double *ms_by_kind, int64_t *launches_by_kind);

/* The same hook for the conv-as-GEMM family (BASELINE configs[4]; replaces nothing in the reference -- it times the kernels
 * that stand in for nn.Conv2d / nn.Conv1d / nn.Linear of improved_diffusion/unet.py:462-491 and models/resnext.py:67-142):
 * while enabled every ap_conv2d_fwd launch is bracketed by HIP events on its stream.  ap_conv_profile_read sums kernel
 * time, algorithmic flops (2 N M K) and launches per kernel class -- 0 conv2d_f32_big2<128,128>, 1 big2<64,128>,
 * 2 big2<128,64>, 3 split-operand kernels, 4 conv2d_f32_big, 5 generic, 6 conv2d_w3 (3 x 3 layers in F(2,3) form along W: its flops are
 * the DIRECT form's 2 N M K, of which it executes two thirds) -- into caller arrays of n_classes >= 7, and resets.
 * A measurement aid with process-global state: enable it from ONE host thread, for launches of ONE device on ONE stream at a time
 * (bench.py's use); it is off by default and the compute entry points never depend on it. */
/* Caller-owned device buffer for the split-K partial sums of low-resolution conv layers (K sliced over workgroups when a layer
 * has too few output tiles for the chip; slices are summed in order by a second kernel: deterministic).  NULL / 0 disables
 * split-K; a layer whose partial sums do not fit runs un-split.  The library allocates nothing itself.
 * The buffer is registered for the HIP device that is CURRENT at the call (it must be that device's memory, else -EINVAL) and a
 * launch only ever uses the buffer of the device it is issued on; one buffer serves one stream at a time (a split-K layer
 * and its reduce own it between them). */
int ap_conv2d_set_workspace(float *ws, size_t bytes);
int ap_conv_profile_enable(int enable);
int ap_conv_profile_read(double *ms_by_class, double *flop_by_class, int64_t *launches_by_class, int n_classes);
/* Launch i of the current recording (call before ap_conv_profile_read): time, flops, [B Cin H W Cout kh kw stride groups class];
 * returns 1 past the last launch. */
int ap_conv_profile_launch(int i, double *ms, double *flop, int *shape10);

/* Workspace the eps/purify entry points need for a batch of B clips of L samples. */
size_t ap_workspace_bytes(const ap_ctx *ctx, int B, int L);

/* ---- per-layer / per-step pieces -------------------------------------------------------------
 * ap_embed: replaces calc_diffusion_step_embedding + fc_t1/fc_t2 swish MLP + every block's fc_t
 *   (util.py:68-93, WaveNet.py:82-83,124-126).  `step` is the (shared) diffusion step as the
 *   reference passes it: float(t) (diffwave_ddpm.py:157).  part_t_dev: num_res_layers*C + embed_dim_out
 *   floats: [num_res_layers][C] FiLM vectors, then the shared 512-d embedding (scratch). */
int ap_embed(ap_ctx *ctx, float step, float *part_t_dev, void *stream);

/* ap_init_conv: ReLU(Conv1x1 1->C) (WaveNet.py:147,168).  x [B][1][L] -> h [B][C][L]. */
int ap_init_conv(ap_ctx *ctx, const float *x, float *h, int B, int L, void *stream);

/* ap_resblock_fwd: one Residual_block.forward (WaveNet.py:75-97), fused:
 *   u = h_in + part_t[c]; y = DilConv_{k=3,d}(u) + b; g = tanh(y[:C]) * sigmoid(y[C:]);
 *   h_out = (u + W_res g + b_res) * sqrt(.5); skip (+)= W_skip g + b_skip.
 * part_t_layer: [C] for this layer.  accumulate_skip = 0 writes skip (layer 0), else adds.
 * h_out must not alias h_in (taps at t +- d read other tiles). */
int ap_resblock_fwd(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer,
                    float *h_out, float *skip, int accumulate_skip, int B, int L, void *stream);

/* Deferred-skip form of the block (AP_PREC_BF16, res = skip = 256 channels; round 4).  WaveNet.py:90-97,131-135: skip_conv only
 * ever sees bf16(g) in this mode, so the block can hand g on as that bf16 image and the sum over layers `skip += skip_n`
 * (:131-133) can be taken inside one K-concatenated GEMM per group of layers instead of one read-modify-write of the fp32
 * skip tensor per layer.
 *   ap_resblock_fwd_gate: the block of ap_resblock_fwd without skip_conv: writes h_out (bit-identical to ap_resblock_fwd's) and
 *                         g_image [B][L][C] bf16 (512 bytes per sample, channels contiguous); `skip` is not touched.
 *                         h_out may be NULL: res_conv and the h' store are left out (the net's last layer, whose h' nobody reads).
 *   ap_skip_gemm:         skip (+)= sum_{n = layer0 .. layer0 + n_layers - 1} (W_skip,n g_n + b_skip,n) with
 *                         g_images [n_layers][B][L][C] bf16 (slot n - layer0); accumulate_skip = 0 writes skip.
 *   ap_ctx_set_skip_group(G): G > 0 makes ap_eps_fwd / ap_purify_* run this form with groups of G layers (the workspace grows
 *                         by G x B x L x C x 2 bytes: ap_workspace_bytes follows); 0 (default) = the fused block per layer.
 * Results: h identical bit for bit; skip differs from the per-layer form by fp32 summation order only (the bf16 products are
 * the same; a group's K = G x 256 is accumulated in the matrix pipe's fp32 accumulators).
 * ap_resblock_fwd_gate launches of at most one 128-sample tile per CU (one or two 1 s clips) run on 64-sample tiles
 * (ap_resblock_bf16s.hip) with bit-identical results: a clip's h' and g image do not depend on the batch it travels in. */
int ap_resblock_fwd_gate(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out,
                         void *g_image, int B, int L, void *stream);
int ap_skip_gemm(ap_ctx *ctx, int layer0, int n_layers, const void *g_images, float *skip, int accumulate_skip,
                 int B, int L, void *stream);
int ap_ctx_set_skip_group(ap_ctx *ctx, int layers_per_group);

/* AP_PREC_BF16_STORE block interface (ap_resblock_bf16u.hip).  The u image of a layer: [B][C / 32][L][32] bf16 -- a sample's 32
 * channels of one 32-channel chunk are one 64-byte row, and inside a row position p holds the chunk's channel (p with bits 2 and 3
 * swapped: the register order of a 32 x 32 MFMA accumulator tile, so the epilogue stores straight from its accumulators).
 *   ap_init_conv_u:    u_0 = bf16(ReLU(W0 x + b0) + part_t of layer 0)        (WaveNet.py:147,168 then :82-84)
 *   ap_resblock_fwd_u: y = DilConv(u_in) + b;  g = tanh . sigmoid;  h' = (u_in + W_res bf16(g) + b_res) sqrt(1/2)   (WaveNet.py:87-97);
 *                      u_out = bf16(h' + part_t_next) (part_t_next: [C] of layer + 1), g_image [B][L][C] bf16 for ap_skip_gemm.
 *                      u_out NULL (the net's last layer, whose h' nobody reads: WaveNet.py:131-135): res_conv and the store are left out.
 * ap_resblock_fwd / _gate / _save return -22 in this mode (their h tensors are fp32; ap_resblock_bwd_bf16 recomputes from an fp32
 * layer input this mode never forms: use ap_resblock_fwd_u_save + ap_resblock_bwd_bf16_saved).
 * ap_eps_fwd / ap_purify_* run the whole sweep. */
int ap_init_conv_u(ap_ctx *ctx, const float *x, const float *part_t_layer0, void *u_out, int B, int L, void *stream);
int ap_resblock_fwd_u(ap_ctx *ctx, int layer, const void *u_in, const float *part_t_next, void *u_out, void *g_image, int B, int L,
                      void *stream);
/* ... and the form that also keeps the gate's derivative factors for ap_resblock_bwd_bf16_saved (see ap_resblock_fwd_gate_save): the
 * gradient of the bf16-storage forward is the bf16 backward's with the rounding of u passed straight through. */
int ap_resblock_fwd_u_save(ap_ctx *ctx, int layer, const void *u_in, const float *part_t_next, void *u_out, void *g_image,
                           void *gate_factors, int B, int L, void *stream);

/* AP_PREC_F32 arithmetic form of the dilated conv (WaveNet.py:87).  1 (default where built: res = skip = 256 channels): the
 * F(2,3) minimal-filtering form over the dilation pair -- outputs t and t + d share their four taps, so the pair costs four
 * [2C x C] products instead of six (block: 12.58 GFLOP per clip instead of 16.78) on the exact-fp32 matrix instruction;
 * transformed weights are computed in double at load, results differ from the direct form by fp32 rounding only and meet the
 * same tolerances against the reference's vectors.  0: the direct form (every other shape always runs it).
 * AP_PREC_F32_SPLIT contexts take the same switch (round 6) but default to 0: form 1 runs the block as two launches -- GEMM1 in
 * F(2,3) form over half of the gate channels per workgroup + the gate, g handed on as fp32 through h_out, then [res_conv; skip_conv] g
 * with the block's epilogues in place (ap_resblock_f32s2.hip) -- 3/4 of the direct split kernel's matrix work at the same 5e-6
 * block tolerance, but only 3-4 % faster (both forms sit at the board's power cap; the second launch's traffic and the doubled
 * staging eat the saved matrix energy) and, like the fp32 F(2,3) form, up to 3 x the direct fp32 kernel's error on a 2^40 dynamic
 * range (the direct split form stays within 2 x): opt-in, not the default.
 * ap_ctx_get_f32_form returns the form the block launches of this context will use. */
int ap_ctx_set_f32_form(ap_ctx *ctx, int form);
int ap_ctx_get_f32_form(ap_ctx *ctx);

/* ap_resblock_fwd_save: ap_resblock_fwd that also writes the pre-gate activations y = DilConv(u) + b (WaveNet.py:87) to
 * pre_gate [B][2C][L] (rows 0..C-1 the tanh half, C..2C-1 the sigmoid half): what the backward of :90 needs, kept by the
 * differentiable purifier (the reference's autograd keeps it too) so that the adjoint does not recompute the dilated conv.
 * AP_PREC_F32 contexts only (-22 otherwise). */
int ap_resblock_fwd_save(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out, float *skip,
                         float *pre_gate, int accumulate_skip, int B, int L, void *stream);

/* ap_final_affine: final_conv (WaveNet.py:160-162,170) on skip*sqrt(1/N) (WaveNet.py:135) fused
 * with an affine update of the clip:  eps = W_f2 ReLU(W_f1 (skip*sqrt(1/N)) + b_f1) + b_f2;
 *   out = ca * x + cb * eps + cs * z.
 * eps_out and/or out may be NULL.  z: see header comment (NULL + cs != 0 -> Philox draw `draw`). */
/* ap_resblock_bwd: input gradient of one Residual_block.forward (WaveNet.py:75-97; the reference's autograd does this for
 * robustness_eval/white_box_attack.py:392,437-439), two fused launches (ap_resblock_bwd.hip):
 *   dy = gate'(pre_gate) . ([W_res sqrt(1/2); W_skip]^T [dh_out; dskip])       -> dy_scratch [B][2C][L]
 *   dh_in = sqrt(1/2) dh_out + DilConv^T(dy)                                    (F(2,3) form of the transposed dilated conv)
 * dh_out = d loss / d h' [B][C][L] (zeros for the net's last layer, whose h' is unused), dskip = d loss / d skip_n [B][S][L] (the
 * same tensor for every layer: skip is their sum), pre_gate: what ap_resblock_fwd_save kept for this layer.  Parameters are
 * frozen (no weight gradients).  AP_PREC_F32, res = skip = 256 channels; ap_resblock_bwd_available says whether a shape is served. */
int ap_resblock_bwd(ap_ctx *ctx, int layer, const float *dh_out, const float *dskip, const float *pre_gate, float *dy_scratch,
                    float *dh_in, int B, int L, void *stream);
/* The backward kernels read their own weight images (fp32: 94 MB, bf16: 47 MB at the shipped shape).  ap_ctx_prepare_backward
 * allocates and packs them for the context's precision and synchronises the host -- like ap_ctx_load_wavenet it is NOT a launch
 * function (call it outside stream capture, once after every ap_ctx_load_wavenet; a second call is a no-op).  ap_resblock_bwd /
 * ap_resblock_bwd_bf16 return -22 until it has been called: they allocate nothing and stay hipGraph-capturable. */
int ap_ctx_prepare_backward(ap_ctx *ctx, void *stream);
int ap_resblock_bwd_available(ap_ctx *ctx, int B, int L);
/* The same gradient in AP_PREC_BF16 (bf16 MFMA operands, fp32 accumulate), from the layer INPUT instead of kept pre-gate activations --
 * on the bf16 matrix pipe the dilated conv is cheaper to recompute than a [B][2C][L] fp32 store per layer is to write:
 *   y = DilConv(bf16(h_in + part_t)) + b;  dy = gate'(y) . ([W_res sqrt(1/2); W_skip]^T [dh_out; dskip])  -> dy_scratch: a bf16 image
 *   [B][L][2C] (B L 1024 bytes);   dh_in = sqrt(1/2) dh_out + DilConv^T(dy).
 * AP_PREC_BF16 contexts, res = skip = 256 channels. */
int ap_resblock_bwd_bf16(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, const float *dh_out, const float *dskip,
                         void *dy_scratch, float *dh_in, int B, int L, void *stream);
int ap_resblock_bwd_bf16_available(ap_ctx *ctx, int B, int L);
/* The gradient pass can skip the recomputation: ap_resblock_fwd_gate_save is ap_resblock_fwd_gate (bit-identical h_out and g_image)
 * that also keeps the gate's two derivative factors sg (1 - th^2), th sg (1 - sg) as an fp16 pair per (channel, sample) in
 * gate_factors -- an opaque image of ap_gate_factor_bytes(B, L) bytes (128 KB per clip and 128-sample tile, in the block kernel's own
 * accumulator order) -- and ap_resblock_bwd_bf16_saved is ap_resblock_bwd_bf16 that reads them instead of h_in / part_t:
 *   dy = factors . ([W_res sqrt(1/2); W_skip]^T [dh_out; dskip]);   dh_in = sqrt(1/2) dh_out + DilConv^T(dy).
 * (|factor| <= 1 at 11 significant bits; dy is rounded to bf16 behind it either way.)  The reference's autograd keeps its
 * activations too (white_box_attack.py:392,437-439). */
size_t ap_gate_factor_bytes(int B, int L);
int ap_resblock_fwd_gate_save(ap_ctx *ctx, int layer, const float *h_in, const float *part_t_layer, float *h_out, void *g_image,
                              void *gate_factors, int B, int L, void *stream);
int ap_resblock_bwd_bf16_saved(ap_ctx *ctx, int layer, const void *gate_factors, const float *dh_out, const void *dskip,
                               int dskip_is_image, void *dy_scratch, float *dh_in, int B, int L, void *stream);
/* dskip is the same tensor for every layer of an evaluation (skip is their sum): with dskip_is_image != 0 it is handed over once as
 * the bf16 image [B][L][S] ap_bwd_bf16_rows_image makes of the fp32 rows [B][S][L] (C = 256) -- what the kernel's staging would round
 * it to anyway, at half the bytes per layer and without the convert. */
int ap_bwd_bf16_rows_image(const float *rows, void *image, int B, int C, int L, void *stream);

int ap_final_affine(ap_ctx *ctx, const float *skip, const float *x, float *eps_out, float *out,
                    float ca, float cb, float cs, const float *z, uint64_t seed, uint32_t draw,
                    uint64_t utt_offset, int B, int L, void *stream);

/* out = ca * x + cs * z  (q-sample, diffwave_ddpm.py:66-67; also fills z-only buffers with ca = 0). */
int ap_affine_noise(const float *x, float *out, float ca, float cs, const float *z, uint64_t seed,
                    uint32_t draw, uint64_t utt_offset, int B, int L, void *stream);

/* ---- whole network / whole purification -----------------------------------------------------
 * ap_eps_fwd: WaveNet_Speech_Commands.forward((x, step*ones)) (WaveNet.py:164-172;
 *   DiffWave.compute_eps_t, diffwave_ddpm.py:166-172).  x, eps_out: [B][1][L]. */
int ap_eps_fwd(ap_ctx *ctx, const float *x, float step, float *eps_out, int B, int L,
               void *workspace, size_t ws_bytes, void *stream);

/* ap_eps_affine: one eps-evaluation that also returns out = ca*x + cb*eps in the same launch sequence
 * (DiffWave.compute_coefficients -> (eps, mu, sigma), diffwave_ddpm.py:143-164).  eps_out or out may be NULL. */
int ap_eps_affine(ap_ctx *ctx, const float *x, float step, float ca, float cb, float *eps_out, float *out,
                  int B, int L, void *workspace, size_t ws_bytes, void *stream);

/* One link of a sampling chain: eps = net(x, step); x <- ca*x + cb*eps + cs*z_draw.
 * DDPM, the SDE Euler scheme, one-shot denoising and respaced samplers (DiffWave.fast_reverse,
 * diffwave_ddpm.py:106-141) are all chains of this link with different coefficient tables. */
typedef struct ap_step {
  float step;      /* value fed to the step embedding (float(t), diffwave_ddpm.py:157) */
  float ca, cb, cs;
  int32_t draw;    /* index into z_all / Philox draw id; ignored when cs == 0 */
} ap_step;

/* ap_purify_chain: optional q-sample (x <- qa*x0 + qs*z_0; skipped when qa == 1 and qs == 0) followed by
 * n_steps links.  z_all: NULL (Philox) or [n_draws][B][L] device tensors indexed by `draw`. */
int ap_purify_chain(ap_ctx *ctx, const float *x0, float qa, float qs, const ap_step *steps, int n_steps,
                    const float *z_all, uint64_t seed, uint64_t utt_offset, float *x_out, int B, int L,
                    void *workspace, size_t ws_bytes, void *stream);

/* ap_purify_ddpm: DiffWave.forward = _diffusion + _reverse (diffwave_ddpm.py:36-104,143-164):
 *   x <- sqrt(ab[t*-1]) x0 + sqrt(1-ab[t*-1]) z0; for t = t*-1..0: eps = net(x,t);
 *   mu = (x - (1-a_t)/sqrt(1-ab_t) eps)/sqrt(a_t); x <- mu + Sigma[t] z_t (t>0) | mu.
 * z_all: NULL (Philox) or t* device tensors [t*][B][L] (z_all[0] = q-sample draw).
 * do_diffuse = 0 skips the q-sample (DiffWave._reverse alone). */
int ap_purify_ddpm(ap_ctx *ctx, const float *x0, int t_star, int do_diffuse, const float *z_all,
                   uint64_t seed, uint64_t utt_offset, float *x_out, int B, int L,
                   void *workspace, size_t ws_bytes, void *stream);

/* ap_purify_sde: RevDiffWave.audio_editing_sample, sample_step = 1 (diffwave_sde.py:167-212) with
 * torchsde's fixed-step Euler-Maruyama restated: per k = t*-1..0
 *   x <- x (1 + b_k/2) - b_k eps(x,k)/sqrt(1-ac_k) + sqrt(b_k) sqrt((1-ac_{k-1})/(1-ac_k)) z  (0 at k=0)
 * with ac = cumprod(1-b) (diffwave_sde.py:58).  Exactly t* eps-evaluations. */
int ap_purify_sde(ap_ctx *ctx, const float *x0, int t_star, const float *z_all, uint64_t seed,
                  uint64_t utt_offset, float *x_out, int B, int L,
                  void *workspace, size_t ws_bytes, void *stream);

/* ap_one_shot_denoise: DiffWave.one_shot_denoise (diffwave_ddpm.py:174-205), t = t*-1. */
int ap_one_shot_denoise(ap_ctx *ctx, const float *x_t, int t_star, float *x0_hat, int B, int L,
                        void *workspace, size_t ws_bytes, void *stream);

/* ---- classifier front-end --------------------------------------------------------------------
 * ap_m5_*: M5.forward (audio_models/M5/M5Net.py:21-38), BatchNorm in eval mode folded at create.
 * `blob_dev`: fp32 concat in state-dict order per stage i=1..4:
 *   conv{i}.weight, conv{i}.bias, bn{i}.weight, bn{i}.bias, bn{i}.running_mean, bn{i}.running_var;
 *   then fc1.weight, fc1.bias.  x [B][1][L] -> log-probabilities [B][n_output]. */
int ap_m5_create(int n_output, int n_channel, int first_kernel, int stride, float bn_eps,
                 const float *blob_dev, size_t n_elems, void *stream, ap_m5 **out);
int ap_m5_destroy(ap_m5 *m);
size_t ap_m5_blob_elems(int n_output, int n_channel, int first_kernel);
int ap_m5_fwd(ap_m5 *m, const float *x, float *logprobs, int B, int L, void *stream);

/* ap_melspec_db: the eval scripts' torchaudio front-end (adaptive_attack_eval.py:83-85):
 * MelSpectrogram(n_fft=2048, hop=512, n_mels, norm='slaney', mel_scale='slaney', pad_mode='constant')
 * + AmplitudeToDB('power').  x [B][1][L] -> [B][1][n_mels][1 + L/512].
 * mode 0: absolute dB (torchaudio); mode 1: librosa power_to_db(ref=max, top_db=80) as
 * transforms/transforms_stft.py:101-114 (ToSTFT + ToMelSpectrogramFromSTFT). */
int ap_melspec_db(const float *x, float *out, int n_mels, int mode, int B, int L, void *stream);

/* ---- 2-D ConvNet classifiers on the mel front-end (audio_models/ConvNets_SpeechCommands/models/ : vgg.py, resnet.py,
 * wideresnet.py, resnext.py:67-142, dpn.py, densenet.py).  The host lowers an eval-mode network
 * to these NCHW fp32 primitives (audiopure_amd/convnet.py); BatchNorm is folded into the conv or applied as a
 * per-channel affine.  `*_cstride` / `*_coff`: the operand is channels [coff, coff+C) of a tensor with cstride channels
 * (torch.cat results and x[:, :d] slices are consumed in place).
 *
 * ap_conv2d_pack: w [Cout][Cin/g][kh][kw] (times scale[Cout] if non-NULL, the folded BatchNorm gamma/sqrt(var+eps))
 *   -> wT [groups][ (Cin/g) kh kw ][Cout/g], the A-operand image of the implicit GEMM, followed (layers with
 *   Cin/g % 16 == 0 and Cout/g >= 64) by the same weights as MFMA A fragments in tap-major K order, which the
 *   streamed-weight kernel reads straight from L2; allocate ap_conv2d_packed_elems(...) floats.
 * ap_conv2d_fwd: out = [relu]( conv2d(x, w, stride, pad, groups) + bias + res ), nn.Conv2d semantics (cross-correlation,
 *   zero padding); bias / res may be NULL; res has the shape of out.  conv-as-GEMM on v_mfma_f32_32x32x2_f32.
 *   `relu` is a flag word: bit 0 = fused ReLU, bit 8 (AP_CONV_SPLIT) = run eligible layers (Cin/g % 16 == 0,
 *   Cout/g >= 64) on the bf16 MFMA with exactly 3-way-split fp32 operands (AP_PREC_F32_SPLIT's arithmetic); bit 9
 *   (AP_CONV_1D) = padding and dilation apply to W only (nn.Conv1d over [B][C][1][L]); bit 10 (AP_CONV_SPLIT_F16) = the
 *   same layers with operands as two fp16 parts, three partial products on the fp16 MFMA (valid for |w|, |x| < 3750 and a
 *   narrow dynamic range only -- the normalised activations of the UNet; not an fp32-class mode); bits 16-31 = dilation (0 = 1).
 *   nn.Linear is the kh = kw = H = W = 1 case. */
size_t ap_conv2d_packed_elems(int Cout, int Cin_g, int kh, int kw, int groups);   /* floats ap_conv2d_pack writes */
int ap_conv2d_pack(const float *w, const float *scale, float *wT, int Cout, int Cin_g, int kh, int kw, int groups,
                   void *stream);
int ap_conv2d_fwd(const float *x, const float *wT, const float *bias, const float *res, float *out, int B, int Cin,
                  int H, int W, int Cout, int kh, int kw, int stride, int pad, int groups, int relu, int x_cstride,
                  int x_coff, void *stream);
/* ap_conv2d_fwd with `out` a channel slice [out_coff, out_coff + Cout) of a tensor of out_cstride channels -- the destination
 * of the torch.cat that follows the layer in improved_diffusion/unet.py:490-491 (h = th.cat([h, hs.pop()], dim=1)): the layer
 * that produces h writes it in place.  `res` keeps the shape of the convolution's own output.  Layers of the streamed-weight
 * fp32 kernel only (Cin/g % 16 == 0, Cout/g >= 64, no split-operand flag); -22 otherwise. */
int ap_conv2d_fwd_slice(const float *x, const float *wT, const float *bias, const float *res, float *out, int B, int Cin,
                        int H, int W, int Cout, int kh, int kw, int stride, int pad, int groups, int relu, int x_cstride,
                        int x_coff, int out_cstride, int out_coff, void *stream);
/* y[B][C][HW] = [relu](x * scale[c] + shift[c]); scale == NULL: plain (optionally ReLU'd) copy of the slice. */
int ap_affine_nchw(const float *x, const float *scale, const float *shift, float *y, int B, int C, int HW,
                   int x_cstride, int x_coff, int relu, void *stream);
/* y = [relu](a + b) on channel slices. */
int ap_add_nchw(const float *a, const float *b, float *y, int B, int C, int HW, int a_cstride, int a_coff,
                int b_cstride, int b_coff, int relu, void *stream);
/* dst[:, d_coff : d_coff + C] = src[:, s_coff : s_coff + C]  (torch.cat). */
int ap_copy_channels(const float *src, float *dst, int B, int C, int HW, int s_cstride, int s_coff, int d_cstride,
                     int d_coff, void *stream);
/* nn.MaxPool2d / F.avg_pool2d(k, stride, pad) on [BC][H][W]. */
int ap_pool2d(const float *x, float *y, int BC, int H, int W, int k, int stride, int pad, int is_max, void *stream);

/* ---- Improved-Diffusion UNet pieces (Improved_Diffusion_Unconditional/improved_diffusion/unet.py, nn.py; section 8 a15).
 * ap_groupnorm_nchw: GroupNorm32 (nn.py:17-19,95-102) on [B][C][HW], optionally followed by the scale-shift FiLM of
 *   use_scale_shift_norm, y = gn(x) * (1 + scale[b][c]) + shift[b][c] with scale_shift = [B][2C] (unet.py:184-190),
 *   and an activation (act: 0 none, 1 ReLU, 2 SiLU).
 * ap_timestep_embedding: [cos(t f_j), sin(t f_j)] (nn.py:103-121); freqs computed by the host like the reference.
 * ap_silu, ap_upsample_nearest2x (unet.py:60-72).
 * ap_attention_qkv: QKVAttention (unet.py:239-252) on qkv [B][heads][3*ch][T] -> [B][heads*ch][T], fp32 softmax. */
int ap_groupnorm_nchw(const float *x, const float *gamma, const float *beta, const float *scale_shift, float *y, int B,
                      int C, int HW, int groups, float eps, int act, void *stream);
int ap_timestep_embedding(const float *t_dev, const float *freqs_dev, float *out, int B, int dim, void *stream);
int ap_silu(const float *x, float *y, size_t n, void *stream);
int ap_upsample_nearest2x(const float *x, float *y, int BC, int H, int W, void *stream);
int ap_attention_qkv(const float *qkv, float *out, int B, int C, int T, int heads, void *stream);
/* Input gradients of the three above (white-box attack through the DiffSpec purifier, adaptive_attack_eval.py:102-104 +
 * white_box_attack.py:437-439; the convolutions' input gradients are ap_conv2d_fwd on flipped / transposed weights).
 * ap_groupnorm_bwd: x and the forward's parameters + dy -> dx.  ap_attention_qkv_bwd: qkv, the forward's `out`, dout ->
 * dqkv (same layout as qkv); `stats` holds B * heads * T * 3 floats.  ap_upsample_nearest2x_bwd: dy [BC][2H][2W] -> dx. */
int ap_groupnorm_bwd(const float *x, const float *gamma, const float *beta, const float *scale_shift, const float *dy,
                     float *dx, int B, int C, int HW, int groups, float eps, int act, void *stream);
int ap_attention_qkv_bwd(const float *qkv, const float *out, const float *dout, float *dqkv, float *stats, int B, int C,
                         int T, int heads, void *stream);
int ap_upsample_nearest2x_bwd(const float *dy, float *dx, int BC, int H, int W, void *stream);
/* out = a*x + b*y + c elementwise (y may be NULL): melspec_standardize / inv (sc09_spectrogram_dataset.py:65-81) and the
 * Euler links of the spectrogram SDE (improved_diffusion_sde.py:173-221). */
int ap_axpbyc(const float *x, const float *y, float *out, float a, float b, float c, size_t n, void *stream);
/* GaussianDiffusion.p_sample, epsilon prediction + fixed variance (gaussian_diffusion.py:232-387): pred_x0 =
 * clamp(r1 x - r2 eps, -1, 1) (clip != 0), mean = c1 pred_x0 + c2 x, out = mean + sigma z (z NULL at t = 0). */
int ap_psample_update(const float *x, const float *eps, const float *z, float *out, float r1, float r2, float c1, float c2,
                      float sigma, int clip, size_t n, void *stream);

/* Fill out[B][L] with the library's Philox N(0,1) stream (same values the fused paths use). */
int ap_philox_normal(float *out, uint64_t seed, uint32_t draw, uint64_t utt_offset, int B, int L,
                     void *stream);

/* ---- keyword-spotting route: clips of any length ----
 * ap_kws_*: KWSModel.forward (audio_models/RCNN_KWS/model.py:66-114): depthwise Conv1d(k=5, s=2) + grouped pointwise
 * Conv1d(s=8), 2-layer bidirectional GRU, additive attention, linear, log-softmax.  `blob_dev`: the state dict
 * concatenated in state-dict order.  mel [B][n_mels][T] -> log-probabilities [B][num_classes].
 * ap_melspec_db_htk: the front-end that script builds (kws_adaptive_attack_eval.py:65-67):
 * torchaudio MelSpectrogram(sample_rate=16000, n_mels) with default n_fft = 400, hop = 200, reflect padding, HTK scale,
 * + AmplitudeToDB('power').  x [B][1][L] -> [B][1][n_mels][1 + L/200]. */
typedef struct ap_kws ap_kws;
size_t ap_kws_blob_elems(int n_mels, int hidden, int num_classes);
int ap_kws_create(int n_mels, int hidden, int num_classes, const float *blob_dev, size_t n_elems, void *stream, ap_kws **out);
int ap_kws_destroy(ap_kws *k);
int ap_kws_fwd(ap_kws *k, const float *mel, float *logprobs, int B, int T, void *stream);
int ap_melspec_db_htk(const float *x, float *out, int n_mels, int B, int L, void *stream);
/* Input gradients of the two above (the script's default attack is PGD through front-end + classifier,
 * kws_adaptive_attack_eval.py:132-143).  ap_kws_bwd: dlogprobs [B][K] -> dmel [B][n_mels][T]; `scratch` holds
 * ap_kws_bwd_scratch_elems floats (the recomputed GRU gates).  ap_melspec_db_htk_bwd: dout [B][n_mels][1 + L/200] ->
 * dx [B][L]; `scratch` holds B * (1 + L/200) * 400 floats. */
size_t ap_kws_bwd_scratch_elems(const ap_kws *k, int B, int T);
int ap_kws_bwd(ap_kws *k, const float *mel, const float *dlogprobs, float *dmel, float *scratch, int B, int T, void *stream);
int ap_melspec_db_htk_bwd(const float *x, const float *dout, float *dx, float *scratch, int n_mels, int B, int L, void *stream);

/* ---- input gradient of the eps-network (reference: the white-box attack back-propagates through
 * the defender, robustness_eval/white_box_attack.py:392,437-439; diffwave_sde.py:200-204 sdeint_adjoint).  The GEMM-shaped
 * terms run on ap_conv2d_fwd (AP_CONV_1D + dilation); these are the element-wise pieces between them. ---- */
/* WaveNet.py:90 backward: a = [a_t; a_s] [B][2C][L], dg [B][C][L] -> da [B][2C][L] */
int ap_gate_bwd(const float *a, const float *dg, float *da, int B, int C, int L, void *stream);
/* WaveNet.py:160-162 backward through the ReLU: dr[b][c][t] = r > 0 ? w2[c] deps[b][t] : 0 */
int ap_relu_outer_bwd(const float *r, const float *w2, const float *deps, float *dr, int B, int S, int L, void *stream);
/* NES queries of the black-box attack (robustness_eval/_NES.py:14-55) with counter-based noise: ap_nes_perturb writes,
 * per audio a, [lead unperturbed copy,] S/2 copies x + sigma z_p and S/2 copies x - sigma z_p (z_p = Philox(seed, draw,
 * a S/2 + p)) into out [A][S + lead][L]; ap_nes_grad forms grad[a] (+)= (1/S) sum_p (loss[a][p] - loss[a][p + S/2]) z_p
 * from the per-copy losses [A][S], regenerating z_p -- the reference's [A][S][L] noise tensor never exists. */
int ap_nes_perturb(const float *x, float *out, float sigma, uint64_t seed, uint32_t draw, int A, int S, int lead, int L,
                   void *stream);
int ap_nes_grad(const float *loss, float *grad, uint64_t seed, uint32_t draw, int A, int S, int L, int accumulate,
                void *stream);
/* counts[argmax_k scores[b][k]] += 1 for b < B (int64 device histogram of K + 1 slots; certified_robust.py:58-65).
 * Rows holding a NaN or an infinity are not votes: they are counted in slot K. */
int ap_argmax_hist(const float *scores, long long *counts, int B, int K, void *stream);
/* pieces of the input gradient of the lowered 2-D classifiers (audiopure_amd/convnet.py): slice accumulate, backward
 * through a fused ReLU, zero insertion for the transposed conv of a strided conv (the conv itself is ap_conv2d_fwd on
 * the flipped, transposed weights), pooling backward. */
int ap_acc_channels(const float *src, float *dst, int B, int C, int HW, int s_cstride, int s_coff, int d_cstride, int d_coff,
                    void *stream);
int ap_relu_mask(const float *dy, const float *y, float *out, size_t n, void *stream);
int ap_zero_insert2d(const float *dy, float *out, int BC, int Ho, int Wo, int Hz, int Wz, int stride, void *stream);
int ap_pool2d_bwd(const float *x, const float *dy, float *dx, int BC, int H, int W, int k, int stride, int pad, int is_max,
                  void *stream);
/* mode-0 ap_melspec_db backward with respect to the waveform: dout [B][n_mels][F] -> dx [B][1][L]; scratch: B F 2048 floats */
int ap_melspec_db_bwd(const float *x, const float *dout, float *dx, float *scratch, int n_mels, int B, int L, void *stream);
/* M5.forward (M5Net.py:20-38) backward with respect to the waveform: dlogprobs [B][n_output] -> dx [B][1][L] */
int ap_m5_bwd(ap_m5 *m, const float *x, const float *dlogprobs, float *dx, int B, int L, void *stream);
/* WaveNet.py:147,168 backward: dx[b][t] = sum_c [h0 > 0] w0[c] dh0[b][c][t] */
int ap_init_conv_bwd(const float *h0, const float *w0, const float *dh0, float *dx, int B, int C, int L, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOPURE_H */
