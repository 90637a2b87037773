// Host-callable launchers of the elementwise / entropy-model / reduction kernels
// (elementwise.hip).  All pointers are device memory; every launcher returns hipError_t.
#pragma once
#include "sga_common.h"

constexpr int EB_STRIDE = 44;   // packed floats per channel of the factorized prior

// per-image accumulators (double), zeroed by k_finalize after being consumed
// The distortion sums are accumulated by thousands of workgroups (one f64 atomic pair each): 2048 workgroups adding to
// 8 images x 2 addresses serialise at the L2 atomic unit (~45 ns per same-address atomic: 23 us of a 60-us launch), so
// every image has kSqSlots sub-accumulators, chosen by workgroup index, that the finalize kernels fold.
constexpr int kSqSlots = 16;
struct ImgSums {
  double sq;      // sum (x - x_tilde)^2                       sga.py:150       (folded from sq_p by the finalize kernels)
  double sq_q;    // sum (255x - round(255 clip(x_tilde)))^2   sga.py:170-173   (folded from sqq_p)
  double y_nats;  // sum -ln p(y_tilde | z_tilde)              sga.py:144
  double z_nats;  // sum -ln p(z_tilde)                        sga.py:145
  double q_ln;    // sum  ln q(z_tilde | y)  (bits-back)        bb_sga.py:102,129-130
  double sq_p[kSqSlots];
  double sqq_p[kSqSlots];
};

// SGA relaxation sga.py:86-98 / :111-121 (+ tfp RelaxedOneHotCategorical.sample).
// u != null: injected uniforms [n][2]; else Philox keyed by ctx (stream_id 0 = y, 1 = z).
// mode: sga_relaxation (0 SGA, 1 deterministic annealing, 2 uniform noise, 3 STE, 4 none)
// img_ids (device, [n / per_img] int4) or null: {position of the image in its reference batch, seed lo, seed hi, seed valid};
// the Philox counter of element e of image b is img_ids[b].x * per_img + e (sga_set_image_ids), its key the image's own
// seed when valid (sga_set_image_seeds), else the run's
int launch_sample(const float* v, const float* u, const StepCtx* ctx, int stream_id,
                  float* vt, float* dvt, int64_t n, hipStream_t s, int mode = 0,
                  const int4* img_ids = nullptr, int64_t per_img = 0);

// Factorized prior on z_tilde [B, npix, C]: accumulates -ln p into sums[b].z_nats and writes
// d rd_loss / d z_tilde (rate term) to g_zt.  p_out / dp_out (optional): raw mass and dp/dv.
int launch_factorized(const float* zt, const float* eb_packed, const StepCtx* ctx, int B,
                      int npix, int C, float inv_ln2_hw, ImgSums* sums, float* g_zt, float* p_out,
                      float* dp_out, hipStream_t s);

// Gaussian conditional on y_tilde [B,h,w,C] with (mu | sigma_raw) = ms [B,hs,ws,2C] cropped to
// [h,w] (sga.py:126-136).  Writes g_yt (rate term), g_ms [B,hs,ws,2C] (zero outside the crop).
// scale_bound: lower bound on sigma with the lower_bound gradient rule; 0 = none (sga_config.scale_bound)
int launch_gaussian(const float* yt, const float* ms, const StepCtx* ctx, int B, int h, int w,
                    int hs, int ws, int C, float inv_ln2_hw, float scale_bound, ImgSums* sums, float* g_yt,
                    float* g_ms, hipStream_t s);
// unit-parity form: flat arrays, returns p and partials
int launch_gaussian_op(const float* y, const float* mu, const float* sraw, int64_t n, float scale_bound,
                       float* p, float* dp_dy, float* dp_dmu, float* dp_dsraw, hipStream_t s);

// distortion: sums + gradient image g = lambda*2*255^2*loss_scale/(HW3) * (xt - x) written into
// the zero-bordered buffer gpad [B,Hp,Wp,3] at offset (2,2)  (sga.py:150-161)
// the C -> 3 transposed convolution with k_mse in its epilogue (deconv3.hip; step only: ctx != null, gpad != null)
int launch_deconv3_halo_mse(const float* in, const float* w, const float* bias, float* out, int B,
                            int Hi, int Wi, int C, int Ho, int Wo, const float* x, const StepCtx* ctx,
                            ImgSums* sums, float* gpad, int Hp, int Wp, hipStream_t stream);
// the same layer as a plain GEMM (P = in . W, 80 columns per input pixel) + col2im with the distortion in it
// (deconv3_gemm.hip); x != null: step (sums, gpad as launch_deconv3_halo_mse)
int launch_deconv3_gemm(const float* in, const float* w80, float* P, int B, int Hi, int Wi, int C, hipStream_t s);
int launch_deconv3_col2im(const float* P, const float* bias, float* out, int B, int Hi, int Wi, int Ho, int Wo,
                          const float* x, const StepCtx* ctx, ImgSums* sums, float* gpad, int Hp, int Wp, hipStream_t s);
int launch_mse(const float* x, const float* xt, const StepCtx* ctx, int B, int H, int W, int Hp,
               int Wp, ImgSums* sums, float* gpad, float* xq_out, hipStream_t s);

// x [B,H,W,3] -> zero-bordered [B,Hp,Wp,3] at offset (2,2)
int launch_pad_image(const float* x, int B, int H, int W, int Hp, int Wp, float* xp,
                     hipStream_t s);

// g = (ga + gb) * jac (chain through the sampler); optional Adam update (adam.py:20-59, f32)
int launch_combine_grad(const float* ga, const float* gb, const float* jac, float* g, int64_t n,
                        hipStream_t s);
int launch_adam_latent(float* p, const float* ga, const float* gb, const float* jac, float* m,
                       float* v, int64_t n, const StepCtx* ctx, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, double b1,
                double b2, double eps, hipStream_t s);

// ctx management
int launch_set_ctx(StepCtx* ctx, int it, int its, float T, float lr_t, float lambda,
                   float loss_scale, uint64_t seed, hipStream_t s);
// it += 1; T = Ttab[it]; lr_t = lrtab[it]   (head of every graph replay)
int launch_advance_ctx(StepCtx* ctx, const float* Ttab, const float* lrtab, hipStream_t s);

// consume + zero the per-image sums.  scalars[3] = {rd_loss, train_mse, train_bpp}; psnr[B];
// trace row [it][4] (if trace != null).  Any output may be null.
// Ttab / lrtab != null: also advance the step context to the next iteration (it += 1, T, lr_t)
int launch_finalize_step(ImgSums* sums, StepCtx* ctx, int B, int H, int W, float* scalars,
                         float* psnr, float* trace, hipStream_t s, const float* Ttab = nullptr,
                         const float* lrtab = nullptr);
// Adam of this iteration + relaxation for the next + per-iteration scalars / context advance in one launch
// (k_step_boundary); `ticket`: a zero-initialised device counter owned by the handle
int launch_step_boundary(float* py, const float* gay, const float* gby, float* jy, float* my, float* vy, float* yt,
                         int64_t ny, float* pz, const float* gaz, const float* gbz, float* jz, float* mz, float* vz,
                         float* zt, int64_t nz, StepCtx* ctx, int mode, const int4* img_ids, int B, int H, int W,
                         ImgSums* sums, float* trace, const float* Ttab, const float* lrtab, unsigned* ticket,
                         hipStream_t s);
// the y and z relaxations / Adam updates of one SGA iteration in one launch each (Philox noise only)
int launch_sample_yz(const float* y, float* yt, float* dyt, int64_t ny, const float* z, float* zt, float* dzt,
                     int64_t nz, const StepCtx* ctx, int mode, const int4* img_ids, int B, hipStream_t s);
int launch_adam_latent_yz(float* py, const float* gay, const float* gby, const float* jy, float* my, float* vy,
                          int64_t ny, float* pz, const float* gaz, const float* gbz, const float* jz,
                          float* mz, float* vz, int64_t nz, const StepCtx* ctx, hipStream_t s);
// metrics[B][7] = {mse, psnr, msssim(=nan here), msssim_db, est_bpp, est_y_bpp, est_z_bpp}
int launch_finalize_eval(ImgSums* sums, int B, int H, int W, float* metrics, hipStream_t s);

int launch_round(const float* v, float* out, int64_t n, hipStream_t s);               // rint
// y_hat = round(y - mu) + mu with mu = ms[..., :C] cropped (mbt2018.py:80; cfg 1)
int launch_round_centered(const float* y, const float* ms, int B, int h, int w, int hs, int ws,
                          int C, float* out, hipStream_t s);
// z_hat = round(z - med[c]) + med[c]
int launch_round_median(const float* z, const float* med, int64_t n, int C, float* out,
                        hipStream_t s);
int launch_fill(float* p, float val, int64_t n, hipStream_t s);
int launch_check_iter(const StepCtx* ctx, int expected_it, int* bad, hipStream_t s);
int launch_fence(hipStream_t s);
int launch_checksum(const float* p, int64_t n, unsigned long long* out, hipStream_t s, const StepCtx* ctx = nullptr);
int launch_spin(int us, hipStream_t s);
int launch_mark(unsigned long long* out, const StepCtx* ctx, hipStream_t s);
int launch_probe(unsigned long long* out, const StepCtx* ctx, hipStream_t s);
int launch_set_int(int* p, int v, hipStream_t s);
int launch_check_int(const int* p, int expected, int* bad, hipStream_t s);
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t s);

// ---- bits-back variant (bb_sga.py) -------------------------------------------------------------
// z_tilde = eps*exp(.5 logvar) + mean with (mean | logvar) = zml [B,npix,2C] (bb_sga.py:99-100);
// eps_in != null: injected normals [B,npix,C], else Philox Box-Muller (stream_id).  Writes
// jac_lv = d z_tilde / d logvar and accumulates sum ln q(z_tilde) (utils.py:72-77).
int launch_bb_sample_z(const float* zml, const float* eps_in, const StepCtx* ctx, int stream_id,
                       int B, int npix, int C, float* zt, float* jac_lv, ImgSums* sums,
                       hipStream_t s, const int4* img_ids = nullptr);
// prior DENSITY p(z_tilde) = dCDF/dz (learned_prior.py:164-185) with lower bound, rate gradient
// (needs the second derivative of the CDF network)
int launch_factorized_pdf(const float* zt, const float* eb_packed, const StepCtx* ctx, int B,
                          int npix, int C, float inv_ln2_hw, ImgSums* sums, float* g_zt,
                          float* p_out, float* dp_out, hipStream_t s);
// g_zml[..., c] = ga+gb;  g_zml[..., C+c] = (ga+gb)*jac_lv - 0.5*loss_scale*inv_ln2_hw
int launch_bb_zgrad(const float* ga, const float* gb, const float* jac_lv, const StepCtx* ctx,
                    float inv_ln2_hw, int64_t npix_total, int C, float* g_zml, hipStream_t s);
// Adam with the step size taken from ctx->lr_t (plain gradient)
int launch_adam_ctx(float* p, const float* g, float* m, float* v, int64_t n, const StepCtx* ctx,
                    hipStream_t s);
int launch_finalize_eval_bb(ImgSums* sums, int B, int H, int W, float* metrics8, hipStream_t s);
// out = act > 0 ? g : 0  (ReLU backward; unit-parity op only, the step fuses it into conv epilogues)
int launch_relu_mask(const float* g, const float* act, float* out, int64_t n, hipStream_t s);

// ---- MS-SSIM of the final evaluation (msssim.hip) ----------------------------------------------
int msssim_init();
int launch_msssim(const float* xq, const float* x, int B, int H, int W, float* const* lvlA,
                  float* const* lvlB, double* stats, int* counts_dev, float* metrics, int mstride,
                  hipStream_t s);
