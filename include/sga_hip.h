/*
 * sga_hip.h -- C ABI of the MI355X-native SGA hot path (libsga_hip.so, gfx950 only).
 *
 * The reference (mandt-lab/improving-inference-for-neural-image-compression) has no
 * plugin/FFI layer: sga.py is a monolithic TF1 script and the hot path sits behind four
 * tf.Session interactions.  Each entry point below replaces one of them:
 *
 *   sga_create / sga_destroy   <- tf.train.Saver().restore(sess, latest)          sga.py:180-182
 *   sga_encode                 <- sess.run([y_init, z_init], {x})                 sga.py:207
 *   sga_step_grads + sga_adam  <- sess.run([rd_gradients, rd_loss, train_mse,
 *                                  train_bpp, psnr], {y,z,x,T}) + Adam.update     sga.py:212-215
 *   sga_run                    <- the whole per-batch loop                        sga.py:207-247
 *   sga_eval                   <- sess.run(eval_tensors, {y_tilde:..,z_tilde:..}) sga.py:219-225,244-245
 *   sga_op_*                   <- the per-layer operators nn_models.py composes   nn_models.py:14-163
 *                                 and the entropy-model / sampler ops             sga.py:86-136
 *
 * Conventions
 *   - every `float*` / `const float*` data argument is DEVICE memory (owned by the caller,
 *     e.g. a PyTorch-ROCm tensor), float32, NHWC, contiguous; `sga_weights` members are HOST
 *     pointers (read once by sga_create).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous w.r.t. the host
 *     unless stated otherwise; the caller synchronises.
 *   - return 0 on success, a negative sga_status on error; no C++ exception crosses the ABI.
 *   - no allocation after sga_create (the workspace is sized from sga_config), so the step
 *     sequence is hipGraph-capturable; sga_run captures and replays it itself.
 *   - a handle is NOT thread-safe: one handle per (GPU, stream), one process per GPU.
 */
#ifndef SGA_HIP_H_
#define SGA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGA_ABI_VERSION 6

typedef enum sga_status {
  SGA_OK = 0,
  SGA_ERR_BAD_ARG = -1,       /* null pointer, non-positive size */
  SGA_ERR_BAD_SHAPE = -2,     /* B/H/W exceed the handle's workspace */
  SGA_ERR_UNSUPPORTED = -3,   /* num_filters not a multiple of 64, unknown op */
  SGA_ERR_HIP = -4,           /* a HIP call / launch failed: see sga_last_error */
  SGA_ERR_NO_DEVICE = -5,     /* no gfx950 device visible */
  SGA_ERR_NOMEM = -6
} sga_status;

typedef struct sga_handle sga_handle;

typedef struct sga_config {
  int32_t num_filters;   /* C: 192 or 256 in the reference (README.md:58-60); any multiple of 64 */
  int32_t max_batch;     /* workspace is sized for [max_batch, max_height, max_width, 3] */
  int32_t max_height;
  int32_t max_width;
  int32_t bits_back;     /* 0: mbt2018 (sga.py); 1: mbt2018_bb, h_a emits 2C (bb_sga.py:69) */
  int32_t precision;     /* sga_precision; 0 = default (SGA_PRECISION env: "f32" | "bf16x3" | "bf16x2", else f32) */
  float scale_bound;     /* lower bound on the conditional's sigma = exp(sigma_raw), with the lower_bound gradient
                          * rule (math_ops.py:63-76); 0 = none.  See SGA_SCALE_BOUND_* below.  (ABI v3; was reserved[0]) */
  int32_t reserved;
} sga_config;

/* Which bound mirrors the reference?  tfc 1.3's GaussianConditional applies
 * `scale = lower_bound(scale, scale_table[0])` in SymmetricConditional.build(), and only a __call__ of the
 * layer runs build().
 *   SGA_SCALE_BOUND_NONE  (0)     sga.py:130-133, bb_sga.py:121-124, danneal.py:127-130, unoise.py:90-93, ste.py:105-108,
 *                                 map.py:93-96: the layer is constructed and `_likelihood` is called directly -- never
 *                                 built -- so these scripts evaluate the RAW sigma.  The default of a zeroed config.
 *   SGA_SCALE_BOUND_BUILT (0.11)  mbt2018.py:77-80 calls the layer (`conditional_bottleneck(y, training=...)`): sigma is
 *                                 bounded below by scale_table[0] = SCALES_MIN = 0.11 (mbt2018.py:30-32,77).  Set this
 *                                 for sga_base_compress.
 * PROVISIONAL: stated from tfc 1.3's source as remembered, not from a run (TF 1.15 / tfc 1.3 are not installable in the
 * build container); INTEGRATION.md section 3c has the two-line check for a TF box.  Both modes are parity-tested. */
#define SGA_SCALE_BOUND_NONE 0.0f
#define SGA_SCALE_BOUND_BUILT 0.11f

/* Arithmetic of the convolution contractions.  Both modes take f32 operands and accumulate in f32:
 *  F32_MFMA : v_mfma_f32_32x32x2_f32, a bitwise f32 fmaf chain.
 *  BF16X3   : every f32 operand is split EXACTLY into three bf16 values (3 x 8 = 24 mantissa bits);
 *             the 6 partial products of order >= 2^-16 go through v_mfma_f32_32x32x16_bf16 (each
 *             bf16 x bf16 product is exact in f32).  Measured error vs f64 is that of an f32 GEMM
 *             (2.4e-7 vs 4.4e-7 max rel. at K = 4800); the parity suite passes unchanged.  2.67x the
 *             f32 matrix rate.
 * BF16X2    : the same with the two upper planes only (operands rounded to 16 mantissa bits, 3 plane products per MAC,
 *             f32 accumulation): relative error <= 2^-16 per product -- 32x finer than TF32's 2^-11 -- at half the MFMA
 *             work of BF16X3.  NOT f32-grade: a separate, explicitly requested mode with its own tolerances in the
 *             parity suite.  Measured: single layers 1.1e-5 (forward) / 5.4e-6 (data-gradient) of the output scale;
 *             one complete step at 8 x 256x256, C = 192: gy 3.5e-4, gz 3.8e-3 of their maxima (f32 path: 1.1e-5 /
 *             2.3e-5 -- gz passes through 1 / sigma); the 2000-step acceptance sets end within the north-star
 *             tolerance (mean dBPP 1.1e-4 +- 1.3e-4 and -1e-5 +- 9e-5).  Convolution K loops and the IGDN post-phase of
 *             those launches; the stand-alone GDN tile kernel and the prologue-transform instances stay BF16X3. */
typedef enum sga_precision { SGA_PRECISION_DEFAULT = 0, SGA_PRECISION_F32_MFMA = 1, SGA_PRECISION_BF16X3 = 2,
                             SGA_PRECISION_BF16X2 = 3 } sga_precision;

/* Effective (post-reparameterisation) parameters, HOST float32.  Kernels are HWIO
 * (kh,kw,C_in,C_out) as tfc.SignalConv2D stores them; gamma is [C_in(j)][C_out(i)];
 * the factorized-prior tensors are softplus(matrix_k) / bias_k / tanh(factor_k) with the
 * shapes of learned_prior.py:43-66. */
typedef struct sga_weights {
  const float* ga_kernel[4];  const float* ga_bias[4];   /* AnalysisTransform        nn_models.py:14-29  */
  const float* ga_beta[3];    const float* ga_gamma[3];
  const float* gs_kernel[4];  const float* gs_bias[4];   /* SynthesisTransform       nn_models.py:48-63  */
  const float* gs_beta[3];    const float* gs_gamma[3];
  const float* ha_kernel[3];  const float* ha_bias[3];   /* HyperAnalysis (bias[2]=NULL) nn_models.py:85-96 */
  const float* hs_kernel[3];  const float* hs_bias[3];   /* MBT2018HyperSynthesis    nn_models.py:152-163 */
  const float* eb_matrix[4];  const float* eb_bias[4];   /* factorized prior         learned_prior.py:43-66 */
  const float* eb_factor[3];
} sga_weights;

/* ---- lifetime ------------------------------------------------------------------------- */
int sga_abi_version(void);
int sga_create(sga_handle** out, const sga_config* cfg, const sga_weights* w);
int sga_destroy(sga_handle* h);
/* hipError_t of the last failing HIP call on this handle (0 if none); msg may be NULL */
int sga_last_error(const sga_handle* h, char* msg, int msg_len);
/* latent geometry for an HxW image: y is [h,w,C], z is [hz,wz,C] (sga.py:77-78) */
int sga_latent_shape(const sga_handle* h, int H, int W, int* yh, int* yw, int* zh, int* zw);
/* change sga_config.scale_bound of a live handle (e.g. SGA_SCALE_BOUND_BUILT around sga_base_compress on a handle
 * that otherwise runs SGA).  Host-side only since ABI 4: the bound is part of the key of the handle's graph cache, so the
 * graphs captured under the other value stay cached and are selected again when it comes back. */
int sga_set_scale_bound(sga_handle* h, float scale_bound);

/* ---- sharding: which images of the reference batch does this handle hold? ----
 * The reference draws the Gumbel noise of a whole batch from one op (sga.py:95-97,118-120), so an
 * image's noise depends on its position in the batch.  ids[b] = position of the handle's b-th image
 * in its reference batch (HOST pointer, n <= max_batch; n = 0 restores 0,1,2,...).  The device RNG
 * keys element e of image b on ids[b] * elements_per_image + e, which makes a shard's results equal
 * to the same images' results in the un-sharded batch bit for bit, for any world size or chunking.
 * Synchronises the device. */
int sga_set_image_ids(sga_handle* h, const int32_t* ids, int n);
/* Pooling: a rank whose share of every reference batch is a single image (Tecnick: batches of 7 on 8 GPUs,
 * configs.py:5-9) can run images of CONSECUTIVE reference batches in one launch when their loss_scale agree.  Each
 * reference batch has its own noise op, i.e. its own seed: seeds[b] = the seed of the b-th image's batch (HOST pointer,
 * n <= max_batch); the device RNG then keys image b on seeds[b] instead of the `seed` argument of the run.
 * n = 0 restores "every image uses the run's seed".  Synchronises the device. */
int sga_set_image_seeds(sga_handle* h, const uint64_t* seeds, int n);

/* ---- sga.py:207  y_init, z_init = sess.run([y_init, z_init], {x}) ---------------------- */
int sga_encode(sga_handle* h, const float* x, int B, int H, int W,
               float* y, float* z, void* stream);

/* ---- sga.py:212-214  one evaluation of rd_gradients and the logged scalars --------------
 * Noise: if u_y/u_z are non-NULL they hold the uniforms in (0,1), shape latent+(2,) with
 * [...,0] the DOWN(floor) and [...,1] the UP(ceil) draw; if NULL the device Philox4x32-10
 * stream keyed by (seed, it) is used (bit-identical to oracle/philox.py).
 * loss_scale = 1/B_ref (the reference's batch means, sga.py:147,150).
 * Outputs: gy,gz = d rd_loss / d y,z; scalars[3] = {rd_loss, train_mse, train_bpp};
 * psnr[B] (sga.py:174 on the relaxed sample).  Any output pointer may be NULL. */
int sga_step_grads(sga_handle* h, const float* x, int B, int H, int W,
                   const float* y, const float* z, float T, float lambda, float loss_scale,
                   uint64_t seed, uint32_t it, const float* u_y, const float* u_z,
                   float* gy, float* gz, float* scalars, float* psnr, void* stream);

/* ---- adam.py:20-59  one update of one array, t = iterations+1 (float32 arithmetic) ------ */
/* lr/beta/eps are double: adam.py:40-45 evaluates lr_t and (1 - beta) in double before the cast */
int sga_adam(sga_handle* h, float* p, const float* g, float* m, float* v, int64_t n, int t,
             double lr, double beta1, double beta2, double eps, void* stream);

/* ---- sga.py:207-247  encode + `its` x (sample, fwd, bwd, Adam) + round + eval -----------
 * y_hat/z_hat: rounded latents (np.round, half-to-even); metrics[B][7] in the order of
 * sga.py:183 {mse, psnr, msssim, msssim_db, est_bpp, est_y_bpp, est_z_bpp};
 * trace[its][4] = {rd_loss, train_mse, train_bpp, mean psnr} per step, or NULL.
 * y0/z0: optional initial latents (NULL -> sga_encode(x)).
 * lr / annealing_rate / T_ub are double: the schedule (utils.py:166-180) and lr_t (adam.py:40-42)
 * are evaluated in double on the host and cast to float32 per step, as the reference does. */
int sga_run(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
            int its, double lr, double annealing_rate, int t0, double T_ub, uint64_t seed,
            const float* y0, const float* z0,
            float* y_hat, float* z_hat, float* metrics, float* trace, void* stream);

/* ---- the same loop in pieces, for the scripts that decide on the host while it runs --------
 * map.py:167-196 and ste.py:177-194 stop early: every 10 iterations they look at the objective
 * and either continue or go back to the latents of the previous check.  sga_run_begin = everything
 * sga_run does before its loop (encode or y0/z0, fresh Adam state, schedules for `its` iterations);
 * sga_run_steps = the next n iterations (same graph replay as sga_run); sga_run_state copies the
 * current continuous latents and the trace so far out (set = 0) or overwrites the latents (set = 1).
 * Other entry points (sga_step_grads, sga_eval, sga_quantize_centered) may be called in between. */
int sga_run_begin(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
                  int its, double lr, double annealing_rate, int t0, double T_ub, uint64_t seed,
                  const float* y0, const float* z0, void* stream);
int sga_run_steps(sga_handle* h, int n, void* stream);
int sga_run_state(sga_handle* h, int set, float* y, float* z, float* trace, void* stream);

/* ---- tfc `_quantize(., 'dequantize')` with centring (map.py:83,101; mbt2018.py:69,80):
 * z_hat = round(z - median) + median, y_hat = round(y - mu) + mu with (mu, .) = h_s(z), z as given.
 * medians: [num_filters] device array or NULL (= 0). */
int sga_quantize_centered(sga_handle* h, const float* y, const float* z, int B, int H, int W,
                          const float* medians, float* y_hat, float* z_hat, void* stream);

/* ---- sga.py:219-225,244-245  eval with the latents fed directly (no sampler) ------------ */
int sga_eval(sga_handle* h, const float* x, int B, int H, int W,
             const float* y_hat, const float* z_hat, float* metrics, float* x_hat, void* stream);

/* ---- mbt2018.py:64-81,167-180 (cfg 1, estimated-rate path): one-shot encode -------------
 * mbt2018.py:80 calls the conditional layer, so the reference bounds sigma here: use a handle with
 * scale_bound = SGA_SCALE_BOUND_BUILT (or sga_set_scale_bound around the call) to mirror it. */
int sga_base_compress(sga_handle* h, const float* x, int B, int H, int W, const float* medians,
                      float* y_hat, float* z_hat, float* metrics, void* stream);
/* The same with the sigma bound as an argument of THIS call (SGA_SCALE_BOUND_BUILT mirrors mbt2018.py:80): the handle's
 * bound and its cached step graphs are left alone (the call launches eagerly; the bound reaches the device by value).  Both
 * forms encode into scratch, not into the latents of a run opened by sga_run_begin, so they may be called between two
 * sga_run_steps calls; like every entry point they use the handle's activation workspace and must be issued on the run's
 * stream.  What a handle that otherwise runs SGA should use instead of sga_set_scale_bound around sga_base_compress. */
int sga_base_compress_bound(sga_handle* h, const float* x, int B, int H, int W, const float* medians, float scale_bound,
                            float* y_hat, float* z_hat, float* metrics, void* stream);

/* ---- sibling relaxations on the same step ------------------------------------
 * The ablation scripts differ from sga.py only in the op that maps (y,z) -> (y_tilde,z_tilde) and
 * in the temperature schedule; sga_step_grads / sga_run honour the handle's current setting. */
typedef enum sga_relaxation {
  SGA_RELAX_SGA = 0,      /* sga.py:86-98: Gumbel-softmax over {floor, ceil}                         */
  SGA_RELAX_DANNEAL = 1,  /* danneal.py:74-100: softmax(logits), no Gumbel noise (logits carry 1/T)  */
  SGA_RELAX_UNOISE = 2,   /* unoise.py:76,81: v + U(-.5,.5) (first uniform of the element)           */
  SGA_RELAX_STE = 3,      /* ste.py:78-88, utils.py:130-134: round with identity backward           */
  SGA_RELAX_NONE = 4      /* map.py: v_tilde = v                                                     */
} sga_relaxation;
typedef enum sga_schedule {
  SGA_SCHED_EXP0 = 0,     /* utils.py:170-173: T_ub*exp(-r*(t-t0)) clipped to [1e-8, T_ub]            */
  SGA_SCHED_EXP = 1       /* danneal.py:188-193: exp(-r*t) clipped to [1e-8, T_ub]                    */
} sga_schedule;
int sga_set_relaxation(sga_handle* h, int relaxation, int schedule);

/* ---- bb_sga.py (cfg 5): SGA + bits-back; handle created with bits_back = 1 ------------------
 * zml = concat(z_mean, z_logvar) on the channel axis, [B,zh,zw,2C] (bb_sga.py:93-95).
 * eps: optional injected N(0,1) draws [B,zh,zw,C]; NULL -> device Philox Box-Muller. */
/* bb_sga.py:93-94,203-204,247: (z_mean | z_logvar) = h_a(y_tilde) */
int sga_bb_init_z(sga_handle* h, const float* y_tilde, int B, int H, int W, float* zml, void* stream);
/* bb_sga.py:211-214 (rate_only = 0: y is relaxed, grads of rd_loss w.r.t. [y, z_mean, z_logvar])
 * and bb_sga.py:252-254 (rate_only = 1: y is the fixed y_tilde, grads of train_bpp w.r.t. zml).
 * scalars[3] = {rd_loss, train_mse, train_bpp}. */
int sga_bb_step_grads(sga_handle* h, const float* x, int B, int H, int W, const float* y,
                      const float* zml, float T, float lambda, float loss_scale, uint64_t seed,
                      uint32_t it, const float* u_y, const float* eps, int rate_only, float* gy,
                      float* gzml, float* scalars, float* psnr, void* stream);
/* bb_sga.py:199-276: stage 1 (its x Adam(lr) on [y, zml]), round y, stage 2 (r_its x Adam(r_lr)
 * on zml, rate only), eval.  metrics[B][8] = sga.py:183 fields + est_bpp_back (bb_sga.py:181).
 * trace1[its][4] as sga_run; trace2[r_its][4] (column 2 = train_bpp).  Any output may be NULL. */
int sga_bb_run(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
               int its, int r_its, double lr, double r_lr, double annealing_rate, int t0,
               double T_ub, uint64_t seed, float* y_hat, float* zml_out, float* metrics,
               float* trace1, float* trace2, void* stream);
/* bb_sga.py:238-261 alone: from y_hat (device, [B,yh,yw,C]) to the refined posterior parameters zml -- exactly the
 * stage 2 that sga_bb_run performs, as a pure function of (y_hat, seed, r_its, r_lr, loss_scale).  A receiver that has
 * decoded y_hat obtains the sender's q(z | y) bit for bit this way, which is what bits-back coding of z needs. */
int sga_bb_refine(sga_handle* h, const float* y_hat, int B, int H, int W, float loss_scale, int r_its, double r_lr,
                  uint64_t seed, float* zml_out, void* stream);
/* bb_sga.py:273-275: eval at (y_hat, zml) with one eps draw */
int sga_bb_eval(sga_handle* h, const float* x, int B, int H, int W, const float* y_hat,
                const float* zml, const float* eps, uint64_t seed, float* metrics, void* stream);
/* prior density and its derivative (learned_prior.py:164-185), v is [n_pix, C] */
int sga_op_factorized_density(sga_handle* h, const float* v, int64_t n_pix, float* p, float* dp_dv,
                              void* stream);

/* ---- per-layer operator surface (unit parity; weights are the handle's) ------------------
 * `layer` selects the handle's layer; shapes follow nn_models.py.  in/out NHWC. */
typedef enum sga_layer {
  SGA_GA0 = 0, SGA_GA1, SGA_GA2, SGA_GA3,     /* conv5x5/2 (+GDN for 0..2)   nn_models.py:14-29  */
  SGA_GS0, SGA_GS1, SGA_GS2, SGA_GS3,         /* deconv5x5^2 (+IGDN 0..2)    nn_models.py:48-63  */
  SGA_HA0, SGA_HA1, SGA_HA2,                  /* conv3x3+relu, conv5x5/2+relu, conv5x5/2  :85-96 */
  SGA_HS0, SGA_HS1, SGA_HS2                   /* deconv+relu, deconv+relu, conv3x3 true   :152-163 */
} sga_layer;

/* forward of one layer INCLUDING its activation; (Hin,Win) input spatial size */
int sga_op_layer_fwd(sga_handle* h, int layer, const float* in, int B, int Hin, int Win,
                     float* out, void* stream);
/* data-gradient of one synthesis / hyper-synthesis layer (GS*, HS*): g_out -> g_in.
 * `in` is the layer input of the forward pass (needed by IGDN / ReLU backward). */
int sga_op_layer_bwd(sga_handle* h, int layer, const float* in, const float* g_out,
                     int B, int Hin, int Win, float* g_in, void* stream);

/* SGA relaxation, sga.py:86-98/111-121: v -> v_tilde and d v_tilde / d v (u as sga_step_grads) */
int sga_op_sample(sga_handle* h, const float* v, const float* u, int64_t n, float T,
                  float* v_tilde, float* dvt_dv, void* stream);
/* factorized-prior mass p(v) (tfc EntropyBottleneck._likelihood, sga.py:101) and dp/dv;
 * v is [n_pix, C] */
int sga_op_factorized_likelihood(sga_handle* h, const float* v, int64_t n_pix,
                                 float* p, float* dp_dv, void* stream);
/* box-convolved Gaussian mass (sga.py:130-133, utils.py:80-102) and its partials
 * w.r.t. y, mu and sigma_raw (sigma = max(exp(sigma_raw), the handle's scale_bound)) */
int sga_op_gaussian_likelihood(sga_handle* h, const float* y, const float* mu,
                               const float* sigma_raw, int64_t n,
                               float* p, float* dp_dy, float* dp_dmu, float* dp_dsraw,
                               void* stream);
/* ... with an explicit bound instead of the handle's (entropy-coder tables must not depend on mutable handle state) */
int sga_op_gaussian_likelihood_bound(sga_handle* h, const float* y, const float* mu, const float* sigma_raw, int64_t n,
                                     float scale_bound, float* p, float* dp_dy, float* dp_dmu, float* dp_dsraw,
                                     void* stream);

/* The rate half of the graph with fed intermediates (the reference can feed y_tilde / z_tilde and
 * fetch any tensor, sga.py:219-225): the very kernels the SGA step launches for sga.py:100-104 and
 * :126-146 -- factorized mass of z_tilde [B,zh,zw,C], box-Gaussian mass of y_tilde [B,yh,yw,C] under
 * ms = (mu | sigma_raw) [B,4zh,4zw,2C] (cropped to the y grid, sga.py:126-128), both through
 * lower_bound with its gradient rule (math_ops.py:63-76) -- and the gradients of
 * train_bpp = loss_scale * sum_b (y_bpp[b] + z_bpp[b]) w.r.t. y_tilde, ms and z_tilde.
 * metrics[B][7]: est_bpp, est_y_bpp, est_z_bpp filled (fields 4..6).  Any output may be NULL. */
int sga_op_rate_terms(sga_handle* h, const float* y_tilde, const float* z_tilde, const float* ms, int B,
                      int H, int W, float loss_scale, float* g_yt, float* g_ms, float* g_zt,
                      float* metrics, void* stream);

/* ---- entropy coding of the quantised latents on the device --------------------------------------
 * mbt2018.py:84-85,211-222 turns (y_hat, z_hat) into bytes with tfc's C++ range-coder ops.  Counterpart here: rANS
 * (32-bit state, 16-bit probabilities, escape symbol) over BLOCKED streams -- the symbols are cut into blocks of
 * `block`, each an independent stream, one lane per block -- byte for byte what the host coder
 * (csrc_cpu/rans.c: rans_encode_blocked) writes for the same tables.  All pointers are DEVICE memory (cdf [ntables][stride]
 * uint32 prefix sums with total 65536, lens / offs per table: entropy_coding.EntropyCoder builds them, from the
 * device entropy-model kernels by default); no handle is involved.
 *   sga_ec_y_symbols : sym = y_hat - rint(mu), tab = y_tab0 + level(sigma) * mean_bins + bin(mu - rint(mu)); r0 = rint(mu)
 *                      (optional); y_hat = NULL: tab / r0 only (decoder).  bad (optional) counts non-integer y_hat.
 *   sga_ec_z_symbols : sym = z_hat, tab = channel.
 *   sga_ec_y_symbols_centred / sga_ec_z_symbols_centred (ABI 5): the same for the MEAN- / MEDIAN-centred latents that
 *                      mbt2018.py compress codes (y_hat = round(y - mu) + mu, mbt2018.py:80; z_hat = round(z - median) + median,
 *                      mbt2018.py:69): sym = round(y_hat - mu), tab = y_tab0 + level(sigma) (one zero-offset table per level);
 *                      sym = round(z_hat - median[c]), tab = c.  bad counts elements that are not integers to 1e-3 after centring.
 *   sga_ec_encode    : block b -> the END of slots[b * slot_cap .. (b + 1) * slot_cap), block_bytes[b] bytes (0: overflow;
 *                      slot_cap = 16 + 8 * block always suffices).
 *   sga_ec_compact   : slots -> out + block_off[b] (the caller's exclusive scan of block_bytes).
 *   sga_ec_decode    : the inverse; bad counts corrupt blocks. */
int sga_ec_y_symbols(const float* y_hat, const float* mu, const float* sigma, int64_t n, const double* scale_table,
                     int levels, int mean_bins, int y_tab0, int32_t* sym, int32_t* tab, int32_t* r0, int32_t* bad,
                     void* stream);
int sga_ec_z_symbols(const float* z_hat, int64_t n, int num_filters, int32_t* sym, int32_t* tab, int32_t* bad,
                     void* stream);
int sga_ec_y_symbols_centred(const float* y_hat, const float* mu, const float* sigma, int64_t n, const double* scale_table,
                             int levels, int y_tab0, int32_t* sym, int32_t* tab, int32_t* bad, void* stream);
int sga_ec_z_symbols_centred(const float* z_hat, const float* medians, int64_t n, int num_filters, int32_t* sym, int32_t* tab,
                             int32_t* bad, void* stream);
int sga_ec_encode(const int32_t* sym, const int32_t* tab, int64_t n, int block, const uint32_t* cdf, const int32_t* lens,
                  const int32_t* offs, int stride, uint8_t* slots, int slot_cap, uint32_t* block_bytes, void* stream);
int sga_ec_compact(const uint8_t* slots, int slot_cap, const uint32_t* block_bytes, const uint64_t* block_off,
                   int nblocks, uint8_t* out, void* stream);
int sga_ec_decode(const uint8_t* in, const uint64_t* block_off, const uint32_t* block_bytes, int nblocks,
                  const int32_t* tab, int64_t n, int block, const uint32_t* cdf, const int32_t* lens, const int32_t* offs,
                  int stride, int32_t* sym, int32_t* bad, void* stream);

/* ---- measurement: per-kernel hipEvent timing of the convolution launches -------------------
 * Between sga_profile_begin and sga_profile_end every MFMA convolution launch issued through
 * this handle is bracketed by a hipEvent pair on its own stream (sga_run then launches eagerly
 * instead of replaying its hipGraph).  sga_profile_end synchronises and returns one row per
 * kernel symbol: launches, summed duration, summed ALGORITHMIC flops (useful MACs x 2, no
 * zero-stuffed taps, no channel padding). */
typedef struct sga_kernel_stat {
  char name[64];          /* kernel symbol as rocprofv3 prints it, e.g. conv_mfma_kernel<2,3,2,2,0,false,false,0> */
  int64_t launches;
  double ms_total;
  double flops_total;
} sga_kernel_stat;
int sga_profile_begin(sga_handle* h);
int sga_profile_end(sga_handle* h, sga_kernel_stat* out, int max_out, int* n_out);
/* The same for ONE convolution kernel symbol inside the hipGraph replay that sga_run / sga_run_steps execute (what the
 * headline number times).  HIP cannot time events recorded by graph nodes, so the first launch of `kernel_name` in the
 * next captured iteration is handed a stamp pointer: its workgroups record the earliest entry and the latest exit on
 * the 100 MHz wall clock; the graph is otherwise the production one (both streams), and every replay is followed by a
 * stream synchronisation to read the pair.  sga_profile_graph_end returns launches, summed duration (first wave in to
 * last wave out: what rocprofv3's kernel trace reports, without dispatch latency) and summed algorithmic flops, and
 * drops the instrumented graph. */
int sga_profile_graph_begin(sga_handle* h, const char* kernel_name);
int sga_profile_graph_end(sga_handle* h, sga_kernel_stat* out);
/* The main-chain launch before which the hyper branch of the cached step graph is forked ("start", "gs2.fwd", "gs3.fwd"):
 * chosen by time, once per geometry, by the first sga_run_steps call with >= 100 iterations (it changes no
 * bit of any result); "untimed" while no such call has been made.  Reporting only. */
int sga_get_fork_point(const sga_handle* h, char* name, int name_len);

/* Counters of the handle's cache of executable step graphs (ABI 4; reporting / tests only).  One captured iteration is kept
 * per (kind, B, H, W, relaxation, sigma bound): a geometry that comes back -- a ragged last batch, a service alternating two
 * sizes -- is SELECTED, never re-captured or re-timed; entries live until sga_destroy. */
typedef enum sga_counter {
  SGA_COUNTER_GRAPH_CAPTURES = 0,   /* stream captures + instantiations so far (3 per timed geometry, 1 per untimed one) */
  SGA_COUNTER_GRAPHS_CACHED = 1,    /* live entries */
  SGA_COUNTER_GRAPH_EVICTIONS = 2,  /* entries dropped because the cache was full (16) */
  SGA_COUNTER_GRAPHS_DROPPED = 3    /* executable graphs destroyed in mid-life so far: losing fork-point candidates (two per timed geometry), evictions, stamped graphs (ABI 6; ABI 4-5: graphs retired until sga_destroy) */
} sga_counter;
int sga_debug_counter(const sga_handle* h, int which, long long* value);

#ifdef __cplusplus
}
#endif
#endif /* SGA_HIP_H_ */
