/*
 * spdm.h -- C ABI of libspdm_hip.so: the MI355X (gfx950) denoising hot path of
 * rafaelsoStanford/State_Policy_DiffusionModel.
 *
 * The reference has no FFI/plugin registry: its boundary for this path is the
 * Python object protocol.  Each entry point below names the
 * reference call it stands behind; the Python host mirror lives in
 * state_policy_diffusionmodel_amd/{engine,diffusion,schedulers}.py and binds
 * these symbols with ctypes (INTEGRATION.md shows the stub a reference
 * maintainer would add).
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch / HIP types in signatures
 *    (`stream` is a hipStream_t passed as void*; NULL = the null stream and the
 *    call synchronises before returning, matching the reference's blocking
 *    semantics).
 *  - `d_` pointers are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm
 *    tensors' data_ptr()), `h_` pointers are HOST pointers.
 *  - Tensors at the boundary use the reference's own layouts: trajectories
 *    (B,1,H,D) contiguous fp32 == (B,H,D); cond (B,1,obs_h,obs_dim) == (B,cond_dim).
 *  - Every function returns 0 on success or a negative spdm_status; the text of
 *    the last error is available from spdm_last_error().  Nothing throws across
 *    the ABI and nothing calls exit().
 *  - One handle = one device = one in-flight call (not re-entrant per handle;
 *    distinct handles are independent).
 */
#ifndef SPDM_H
#define SPDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPDM_NAME_MAX 64
#define SPDM_ABI_VERSION 1

typedef enum {
    SPDM_OK = 0,
    SPDM_ERR_INVALID = -1,      /* bad argument / shape */
    SPDM_ERR_HIP = -2,          /* a HIP runtime call failed */
    SPDM_ERR_STATE = -3,        /* call order (weights/schedule not set, ...) */
    SPDM_ERR_MISSING = -4,      /* tensor name not found in the index */
    SPDM_ERR_NOMEM = -5
} spdm_status;

typedef enum { SPDM_DDPM = 0, SPDM_DDIM = 1 } spdm_scheduler_kind;

typedef struct spdm_handle spdm_handle;

/* Shape of one noise predictor instance.  Mirrors the constructor arguments at
 * models/diffusion_ddpm.py:76-82 (UNet_Film(in=1, out=1, noise_steps,
 * global_cond_dim=observation_dim*obs_horizon, time_dim=256)) plus the trajectory
 * geometry sample() uses (pred_horizon + inpaint_horizon rows of prediction_dim,
 * models/diffusion_ddpm.py:252). */
typedef struct {
    int32_t horizon;              /* H: rows of x_t (unpadded)                      */
    int32_t state_dim;            /* D: columns of x_t (unpadded), 1..8             */
    int32_t cond_dim;             /* obs_horizon * observation_dim (flattened y)    */
    int32_t time_dim;             /* sinusoidal embedding width (256)               */
    int32_t attention;            /* 1: UNet_Film, 0: UNet_Film_noAttention         */
    int32_t max_batch;            /* workspace is sized for this many trajectories  */
    int32_t device;               /* HIP device ordinal                             */
    int32_t num_train_timesteps;  /* rows of the time-embedding table (t < this)    */
    int32_t flags;                /* SPDM_FLAG_*                                    */
} spdm_config;

#define SPDM_FLAG_DEBUG_KEEP 1    /* keep every intermediate alive for spdm_debug_tensor */
#define SPDM_FLAG_EXACT_FP32 2    /* contractions on the exact fp32 MFMA path instead of the default split-fp16
                                    path (hi + 2^-11 lo, 3 fp16 MFMAs, fp32 accumulate); env SPDM_PREC=f32 does the same */

/* One entry per tensor of the reference state_dict (names exactly as
 * UNet_Film.state_dict() gives them, e.g. "down1.cond_encoder.2.weight"),
 * torch-native layouts (conv: (Cout,Cin,3,3); linear: (out,in)). */
typedef struct {
    char     name[SPDM_NAME_MAX];
    uint64_t offset;              /* in floats, into the blob */
    uint64_t numel;
    int32_t  ndim;
    int32_t  shape[4];
} spdm_tensor_index;

int  spdm_abi_version(void);
const char* spdm_last_error(void);

/* Replaces: Diffusion_DDPM.__init__'s construction of self.noise_estimator
 * (models/diffusion_ddpm.py:76-82).  Allocates weights + workspace on `device`. */
int  spdm_create(const spdm_config* cfg, spdm_handle** out);
void spdm_destroy(spdm_handle* h);

/* Replaces: load_state_dict of the `noise_estimator.*` tensors
 * (generate.py:25-27 -> Lightning load_from_checkpoint).  `h_blob` is a HOST
 * array; the library re-lays the weights out for its kernels and uploads them. */
int  spdm_load_weights(spdm_handle* h, const float* h_blob, size_t n_floats,
                       const spdm_tensor_index* h_index, int32_t n_index);

/* Optional: overwrite the sinusoidal table pos_encoding(t) for t = 0..T-1
 * (models/Unet_FiLmLayer.py:266-274), (T, time_dim) fp32 on the host.  By
 * default the library computes it itself in fp32; the Python host passes
 * torch's own values so that the table is bit-identical to the reference's. */
int  spdm_set_time_table(spdm_handle* h, const float* h_table, int32_t T);

/* Replaces: DDPMScheduler(...)/DDIMScheduler(...) construction +
 * set_timesteps(n) (models/diffusion_ddpm.py:65-70,268; generate.py:28-35).
 * Linear betas in [beta_start, beta_end], epsilon prediction, no clipping,
 * fixed_small variance / eta = 0. */
int  spdm_set_schedule(spdm_handle* h, int32_t kind, int32_t num_train_timesteps,
                       int32_t num_inference_steps, float beta_start, float beta_end);

/* Same, from caller-computed tables (a caller-assigned scheduler object):
 * h_timesteps[n_steps] in loop order, h_coef[n_steps][6] =
 *   { sqrt(1-abar_t), sqrt(abar_t), k_x0, k_x, k_eps, k_noise } with
 *   x0   = (x - c0*eps) / c1
 *   prev = k_x0*x0 + k_x*x [+ k_noise*z]   (DDPM)
 *   prev = k_x0*x0 + k_eps*eps              (DDIM, eta = 0)                       */
int  spdm_set_schedule_tables(spdm_handle* h, int32_t kind, int32_t n_steps,
                              const int32_t* h_timesteps, const float* h_coef);

/* Pure host helper (no GPU needed): the tables spdm_set_schedule would build. */
int  spdm_schedule_tables(int32_t kind, int32_t num_train_timesteps, int32_t num_inference_steps,
                          float beta_start, float beta_end,
                          int32_t* h_timesteps_out, float* h_coef_out /* [n][6] */);

/* Replaces: self.noise_estimator(x_t, torch.tensor([t]), obs_cond)
 * (models/diffusion_ddpm.py:272 -> UNet_Film.forward, models/Unet_FiLmLayer.py:277-312).
 * d_x (B,H,D), h_t[t_count] with t_count == 1 (broadcast) or B, d_cond (B,cond_dim),
 * d_eps (B,H,D).  cond may be NULL (no FiLM, `y=None`). */
int  spdm_unet_forward(spdm_handle* h, int32_t B, const float* d_x, const int32_t* h_t,
                       int32_t t_count, const float* d_cond, float* d_eps, void* stream);

/* Replaces: the body of Diffusion_DDPM.sample / Diffusion_DDIM.sample after the
 * conditioning vectors are built (models/diffusion_ddpm.py:252-277,
 * models/diffusion_ddim.py:52-74): for t in timesteps: eps = unet(x,t,cond);
 * x = scheduler.step(eps,t,x).prev_sample; x[:, :, :inp_h, :] = inpaint.
 *
 *  d_cond     (B,cond_dim)
 *  d_inpaint  (B,inp_h,D) if inpaint_per_sample else (inp_h,D) broadcast; NULL/inp_h=0: none
 *  d_xT       (B,H,D) initial sample (the reference draws it uniform, ddpm.py:252)
 *  d_noise    (n_steps,B,H,D) pre-drawn N(0,1) (row i used by loop iteration i when t>0),
 *             or NULL: device Philox4x32-10 stream keyed by (seed, sample_offset+b, i)
 *  d_out      (B,H,D) final x_0
 *  d_history  NULL or (n_steps+1,B,H,D): x_T followed by every iterate (option='sample_history')
 */
int  spdm_sample(spdm_handle* h, int32_t B, const float* d_cond,
                 const float* d_inpaint, int32_t inp_h, int32_t inpaint_per_sample,
                 const float* d_xT, const float* d_noise, uint64_t seed, uint64_t sample_offset,
                 float* d_out, float* d_history, void* stream);

/* The same loop in three pieces, so a caller (bench.py) can time an exact range
 * of denoise steps: begin() hoists the step-invariant FiLM projections and
 * loads x_T; run() executes loop iterations [step_begin, step_end); result()
 * copies the current iterate out. */
int  spdm_sample_begin(spdm_handle* h, int32_t B, const float* d_cond,
                       const float* d_inpaint, int32_t inp_h, int32_t inpaint_per_sample,
                       const float* d_xT, const float* d_noise, uint64_t seed,
                       uint64_t sample_offset, float* d_history, void* stream);
int  spdm_sample_run(spdm_handle* h, int32_t step_begin, int32_t step_end, void* stream);
int  spdm_sample_result(spdm_handle* h, float* d_out, void* stream);

/* How many times this handle has captured its denoise step into a hipGraph.  spdm_sample_run replays one captured step
 * for every iteration; the capture is keyed by the session's SHAPE (batch, inpaint horizon, scheduler, switches), not by
 * the addresses of the caller's buffers or the seed -- so the closed-loop caller (run_predictions.py:151-156: one
 * model.sample() per control period, fresh tensors every time) captures once and replays ever after. */
int64_t spdm_graph_captures(const spdm_handle* h);

/* Introspection for tests: copy a named intermediate of the LAST
 * spdm_unet_forward (handle created with SPDM_FLAG_DEBUG_KEEP) to d_out in
 * channels-last (B, H_l*W_l, C) order; shape_out = {B, H_l, W_l, C}.
 * Names: x1 d1 x2 d2 x3 d3 x4 x5 u1 a4 u2 a5 u3 a6. */
int  spdm_debug_tensor(spdm_handle* h, const char* name, float* d_out, size_t cap_floats,
                       int32_t shape_out[4]);

/* Range guard of the split-precision contractions.  The split-fp16 path represents |weight| < 511 and
 * |activation| < 4094.  Weights: spdm_load_weights checks every tensor and keeps a layer whose
 * tensor exceeds the bound on the exact fp32 kernels (spdm_demoted_tensors counts them; results stay within the
 * parity tolerance, that layer runs slower).  Activations: the step-update kernel raises a device flag when an
 * updated iterate (or, for spdm_unet_forward, an eps value) is not finite; spdm_nonfinite synchronises `stream`
 * and returns it (reset by spdm_sample_begin / spdm_unet_forward).  The reference's fp32 torch path
 * (models/diffusion_ddpm.py:261-263) has neither limit; a caller that hits the flag re-creates the handle with
 * SPDM_FLAG_EXACT_FP32. */
int32_t spdm_demoted_tensors(const spdm_handle* h);
int  spdm_nonfinite(spdm_handle* h, int32_t* flag_out, void* stream);

/* Flip one kernel-selection switch ("SPDM_NO_GRAPH", "SPDM_NO_WIDE", ... -- the names the environment is read for,
 * ONCE, at spdm_create) on a live handle.  Test / tuning hook: the product path never calls it.  The workspace is
 * re-planned for the new kernel selection (the unfused fallbacks need more scratch) and grown if it has to be. */
int  spdm_set_switch(spdm_handle* h, const char* name, int32_t on);

/* 1 if the handle's contractions run on the split-fp16 MFMA path, 0 on the exact fp32 MFMA path. */
int32_t spdm_uses_split_precision(const spdm_handle* h);

/* Workspace bytes currently reserved on the device (weights + arena). */
size_t spdm_device_bytes(const spdm_handle* h);

/* Device time of the dominant kernel class, measured with HIP events on the
 * launch stream: when enabled, every conv3x3 implicit-GEMM launch is bracketed
 * by events; spdm_profile_read returns launches, total ms and total FLOPs
 * since the last reset.  For bench.py's roofline leg only (adds sync points).
 * on = 1: instrument from now on (runs are plain launches, not graph replays);
 * on = 0: stop; on = 2: only create the events ahead of time (nothing is
 * instrumented, graph replay stays on) so that a later on = 1 costs nothing. */
int  spdm_profile_enable(spdm_handle* h, int32_t on);
int  spdm_profile_read(spdm_handle* h, int64_t* launches, double* total_ms, double* total_flops);

/* Micro-benchmark of ONE implicit-GEMM launch shape on synthetic data (tools/bench_gemm.py; kernel tuning
 * only, not on the product path): conv taps in {1,3,9} over (B, H*W, Cin) -> (B, H*W, Cout).  pro: 0 none,
 * 1 GroupNorm, 2 GroupNorm+GELU; epi: 0 GN stats, 1 bias, 2 bias+GELU, 3 bias+residual; debug: ablation bits. */
int  spdm_bench_gemm(int32_t device, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t taps,
                     int32_t pro, int32_t epi, int32_t split, int32_t iters, int32_t debug,
                     double* ms_out /* [3]: ms per launch; max |out - exact-fp32 out|; worst deviation of the per-sample
                                        GroupNorm mean (in sigmas) / variance (relative) the launch reported from the
                                        values recomputed from its own output (both: debug == 0, else -1) */);

/* Observation front end (widened scope).  Replaces: self.vision_encoder(img.flatten(end_dim=1))
 * in prepare_obs_cond_vectors (models/diffusion_ddpm.py:317-321), i.e. Autoencoder.encoder of
 * models/encoder/autoencoder.py:11-20: Conv2d(3,16,2,2,p1) ReLU Conv2d(16,32,2,2) ReLU Conv2d(32,64,2,2) ReLU Flatten
 * Linear(9216,128).  Weights: the encoder's own state_dict (names "0.weight" "0.bias" "2.*" "4.*" "7.*", torch layouts)
 * as a host blob + index, like spdm_load_weights.  d_images: (n,3,96,96) fp32 on the device; d_latent: (n,128). */
typedef struct spdm_encoder spdm_encoder;
int  spdm_encoder_create(int32_t device, const float* h_blob, size_t n_floats, const spdm_tensor_index* h_index,
                         int32_t n_index, spdm_encoder** out);
int  spdm_encoder_forward(spdm_encoder* e, int32_t n_images, const float* d_images, float* d_latent, void* stream);
void spdm_encoder_destroy(spdm_encoder* e);

/* Host-only test hook (no GPU call): the launch geometry chosen for a split-precision 3x3 / 3x1 convolution with the
 * statistics epilogue -- out = {m_tile, n_tile, n_tiles, slots, ksplit, kernel, st_m_tile, st_n_tiles, reserved_slots,
 * combine_rows}; kernel: bit 0 = the small-grid kernel (conv_skinny.hip), bit 1 = the register-resident kernel (conv_reg.hip).
 * `switches` = 0 for the defaults. */
int  spdm_debug_geometry(int32_t M, int32_t N, int32_t K, int32_t HW, int32_t W, int32_t taps, uint32_t switches,
                         int32_t out[10]);

/* Op-level test hook: d_y = GELU(d_x) evaluated with the device erf that the conv prologues use
 * (nn.GELU(), models/Unet_FiLmLayer.py:104). */
int  spdm_op_gelu(const float* d_x, float* d_y, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPDM_H */
