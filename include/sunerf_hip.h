/*
 * sunerf_hip.h -- C ABI of the MI355X (gfx950) SuNeRF ray-march renderer.
 *
 * The reference (FrontierDevelopmentLab/2024-HL-SPI3S-SuNeRF) has no FFI / plugin registry: its hot path is
 * the Python class API of sunerf.rendering + sunerf.model + sunerf.train.sampling.  This
 * header is the drop-in boundary *under* that API: every entry point replaces a span of aten ops of one
 * reference function (cited per function, paths relative to the reference root) and is called by the Python
 * mirror classes in 2024-hl-spi3s-sunerf_amd/sunerf/ through ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP, same device as `stream`) unless the name ends in `_host`;
 *   - fp32 row-major contiguous tensors, shapes given per argument;
 *   - no allocation, no host synchronisation, no ownership transfer: callers own inputs, outputs and
 *     workspaces; every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - returns 0 on success, a negative SUNERF_E_* code on argument errors, a positive hipError_t if a launch
 *     failed.  The Python side turns non-zero into RuntimeError / ValueError.
 *   - thread-safe / re-entrant: no global mutable state (evaluation/loader.py:226-229 calls the renderer from
 *     a ThreadPoolExecutor).
 */
#ifndef SUNERF_HIP_H
#define SUNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUNERF_ABI_VERSION 9

#define SUNERF_E_BADARG   (-1)   /* null pointer / non-positive size                               */
#define SUNERF_E_UNSUPPORTED (-2) /* d_filter / n_layers / sample count outside the compiled set     */
#define SUNERF_E_WORKSPACE (-3)  /* workspace too small                                             */

#define SUNERF_MAX_LAYERS 16     /* Linear layers per MLP including in_layer and out_layer          */
#define SUNERF_ENC_DIM    84     /* PositionalEncoding(d_input=4, n_freqs=10).d_output, model.py:109 */

/* forward arithmetic of the MLP products x*w (x, w split into fp16 head + exact fp32 remainder); the packed image and the
 * render call must name the same mode.
 *   FAST  : head*head on the fp16 matrix cores + the two cross terms as block-scaled fp8 products
 *           (v_mfma_scale_f32_32x32x64_f8f6f4); raw MLP output within ~1e-5 |raw| rms (4e-5 |raw| worst sample) of fp32,
 *           i.e. images ~1e-5 rel for |raw| ~ 1: inside the 1e-4 parity gate; 18 % (d_filter 256) / 23 % (512) faster
 *   EXACT : all three terms as fp16 products (fp32-class results, raw within ~1e-7)
 *   HALF  : (opt-in) single fp16 operands, fp32 accumulate: the head product only.  This is the
 *           "bf16 MLP weights on MFMA" class of BASELINE.json config 3 (with fp16's 11-bit instead of bf16's 8-bit
 *           mantissa): results follow an fp16-emulating evaluation to 1e-4 and deviate from fp32 by ~1e-3; it does NOT meet
 *           the 1e-4-vs-fp32 parity gate and is never the default */
#define SUNERF_PRECISION_FAST  0
#define SUNERF_PRECISION_EXACT 1
#define SUNERF_PRECISION_HALF  2

/* sampler kinds -- sunerf/train/sampling.py */
#define SUNERF_SAMPLER_STRATIFIED 0   /* StratifiedSampler.forward  sampling.py:68-102 */
#define SUNERF_SAMPLER_SPHERICAL  1   /* SphericalSampler.forward   sampling.py:16-54  */

int sunerf_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Weight packing.  The fused renderer consumes the MLP weights as fp16 hi/lo pairs (w = hi + lo, |err| <= 2^-22
 * relative) stored in MFMA A-fragment order, plus fp32 biases.  Must be re-run after every optimiser step.
 *
 * Replaces: nothing numerically -- it is a re-layout of the nn.Linear parameters of NeRF (model.py:28-42).
 *
 *   weights_host[i] -> device fp32 W_i [out_i, in_i] row-major (nn.Linear layout), i = 0..n_linear-1
 *   biases_host[i]  -> device fp32 b_i [out_i]
 *   n_linear = n_layers + 1 : in_layer (84 -> d_filter), n_layers-1 hidden (d_filter -> d_filter), out_layer
 *   packed: device buffer of sunerf_packed_mlp_bytes() bytes, 16-byte aligned
 *   precision: SUNERF_PRECISION_FAST / _EXACT (above); selects the stream format of the hidden and out layers
 * ---------------------------------------------------------------------------------------------------------- */
size_t sunerf_packed_mlp_bytes(int d_filter, int n_linear);

int sunerf_pack_mlp(const float* const* weights_host, const float* const* biases_host, int n_linear,
                    int d_filter, int d_out, int precision, void* packed, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sample placement along rays.
 * Replaces StratifiedSampler.forward sampling.py:68-98 / SphericalSampler.forward sampling.py:16-49 (z_vals only;
 * the points o + d*z of :100 / :52 are formed inside the render kernel and never materialised).
 *
 *   rays_o, rays_d [N,3]; t_vals [S] (the module buffer, sampling.py:65-66); t_rand [N,S] or NULL
 *   (perturb=False); distance = sampler.distance buffer, solar_R = sampler.solar_R buffer; z_vals out [N,S]
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_sample_z(int sampler_kind, const float* rays_o, const float* rays_d, const float* t_vals,
                    const float* t_rand, int64_t n_rays, int n_samples, float distance, float solar_R,
                    float* z_vals, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused render pass: points -> time concat -> positional encoding -> MLP -> emission/absorption integral.
 * Replaces, for one coarse or fine pass:
 *   sampling.py:100 (points), base_tracing.py:64-65 / :83-84 (time concat), base_tracing.py:118-129 (_render),
 *   model.py:123-132 (PositionalEncoding.forward), model.py:44-57 (NeRF.forward), emission.py:14-54
 *   (raw2outputs), base_tracing.py:135-156 (cumprod_exclusive), and the epilogues base_tracing.py:99-110
 *   (absorption_map, distance, height_map, regularization with defect D2 resolved to (N,S)).
 *
 *   packed     : sunerf_pack_mlp output for this pass's model
 *   rays_o/d   : [N,3]; times [N] (the (N,1) column); z_vals [N,S]
 *   image      : [N]    sum_S I*T                                  (emission.py:46)
 *   weights    : [N,S]  I*T / (sum + 1e-10)                        (emission.py:49-50)
 *   absorption : [N,S]  exp(-relu(r1)*dists)  'regularizing_quantity' (emission.py:35)
 *   raw        : [N,S,2] MLP output ('inferences'), may be NULL
 *   height_map, absorption_map : [N] or NULL;  regularization : [N,S] or NULL  (base_tracing.py:99-106)
 *   reg_radius : 1.2 / Rs_per_ds (base_tracing.py:44)
 *   act_stash  : NULL for inference; for training a device buffer of sunerf_act_stash_bytes() bytes that
 *                receives the hidden activations for the backward kernels
 *   stash_format: SUNERF_STASH_FP16 -- fp16 sin and fp16 cos of every activation (8.2 KB per sample of an 8 x 256 network): what
 *                sunerf_mlp_dgrad / sunerf_mlp_wgrad read -- or SUNERF_STASH_PHASE -- the 16-bit phase of every pre-activation
 *                (4.1 KB per sample; sin and cos to 4.8e-5 from it): what sunerf_mlp_backward_pipe reads; d_filter = 256 only
 *   workspace  : sunerf_render_workspace_bytes(d_filter) bytes of device scratch (may be NULL when that is 0)
 * ---------------------------------------------------------------------------------------------------------- */
#define SUNERF_STASH_FP16  0
#define SUNERF_STASH_PHASE 1
size_t sunerf_act_stash_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear, int stash_format);
/* d_filter = 512 (the reference's default width, model.py:16): scratch for layer outputs; 0 for narrower nets */
size_t sunerf_render_workspace_bytes(int d_filter);

int sunerf_emission_render_fwd(const void* packed, int d_filter, int n_linear, int precision,
                               const float* rays_o, const float* rays_d, const float* times,
                               const float* z_vals, int64_t n_rays, int n_samples,
                               float* image, float* weights, float* absorption, float* raw,
                               float* height_map, float* absorption_map, float* regularization,
                               float reg_radius, void* act_stash, int stash_format, void* workspace,
                               size_t workspace_bytes, void* stream);

/* NeRF.forward on free-standing query points, model.py:44-57 (positional encoding + sine MLP, no ray, no integral): the fused
 * render kernel fed with explicit points.  points [M,4] = (x, y, z, t), M a multiple of 32 (callers pad); raw [M,2].
 * act_stash (optional): as in sunerf_emission_render_fwd with n_rays = M / 32, n_samples = 32 -- sunerf_mlp_dgrad /
 * sunerf_mlp_wgrad then take g_raw [M/32, 32, 2] (a loss on arbitrary points trains, as the reference's module call does).
 * Serves evaluation/loader.py:load_coords (volume queries) at the kernel's full rate. */
int sunerf_mlp_points_fwd(const void* packed, int d_filter, int n_linear, int precision, const float* points,
                          int64_t n_points, float* raw, void* act_stash, int stash_format, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Backward of the fused render pass (training).  The reference has no backward code of its own: these entry points
 * replace what torch.autograd derives from base_tracing.py:118-129 + emission.py:14-54 + model.py:44-57 for the
 * loss of sunerf/model/sunerf.py:110-120 (gradients w.r.t. the MLP parameters only: sampling.py:120 detaches the
 * resampled z, so nothing flows to the ray geometry).  Call order for one pass:
 *
 *   sunerf_emission_render_fwd(..., act_stash != NULL)       forward, stashes fp16 sin / cos fragments
 *   sunerf_emission_integral_bwd                             g_image (N), g_reg (N,S) -> g_raw (N,S,2), max |g_raw|
 *   sunerf_mlp_dgrad                                         dZ of every layer -> dz_stash (fp16, scaled)
 *   sunerf_mlp_wgrad                                         dW, db of every Linear layer (nn.Linear layouts)
 *
 *   packedT  : sunerf_pack_mlp_t output (transposed fp16 weight image, each layer times a power of two chosen from the layer's
 *              own weights so that the data gradient keeps the scale of g_raw from layer to layer; the same buffer goes to
 *              sunerf_mlp_dgrad and sunerf_mlp_wgrad), re-pack after every optimiser step
 *   g_reg    : (N,S) gradient w.r.t. the 'regularization' output, or NULL with g_reg_const (the usual
 *              lambda / (N*S) of regularization.mean(), sunerf.py:118-119)
 *   g_absmax : 4-byte device scratch (bit pattern of max |g_raw|; selects the fp16 gradient scale on the device)
 *   workspace: sunerf_wgrad_workspace_bytes(d_filter, n_linear, split) bytes; split = number of partial sums per layer
 *   accumulate != 0 adds to grad_* instead of overwriting (autograd .grad accumulation)
 * ---------------------------------------------------------------------------------------------------------- */
size_t sunerf_packed_mlp_t_bytes(int d_filter, int n_linear);
int sunerf_pack_mlp_t(const float* const* weights_host, int n_linear, int d_filter, int d_out, void* packedT,
                      void* stream);
size_t sunerf_dz_stash_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear);
size_t sunerf_wgrad_workspace_bytes(int d_filter, int n_linear, int split);

/* EmissionRadiativeTransfer.raw2outputs (sunerf/rendering/emission.py:14-54, cumprod_exclusive base_tracing.py:135-156) on a
 * GIVEN raw tensor -- the subclass hook SuNeRFRendering._render calls (base_tracing.py:128): raw (N,S,2), z_vals (N,S),
 * rays_d (N,3) -> image (N), weights (N,S), absorption (N,S) = the 'regularizing_quantity'.  (Inside
 * sunerf_emission_render_fwd the same arithmetic is fused behind the MLP.) */
int sunerf_emission_integral_fwd(const float* raw, const float* z_vals, const float* rays_d, int64_t n_rays, int n_samples,
                                 float* image, float* weights, float* absorption, void* stream);

/* g_weights / g_absorption: optional (N,S) gradients w.r.t. the 'weights' and 'regularizing_quantity' outputs of
 * raw2outputs (NULL on the training path, whose loss only reads image and regularization) */
int sunerf_emission_integral_bwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                                 const float* g_image, const float* g_reg, const float* g_weights, const float* g_absorption,
                                 float g_reg_const, float reg_radius, int64_t n_rays, int n_samples, float* g_raw,
                                 void* g_absmax, void* stream);

int sunerf_mlp_dgrad(const void* packedT, int d_filter, int n_linear, const float* g_raw, const void* g_absmax,
                     const void* act_stash, void* dz_stash, int64_t n_rays, int n_samples, void* stream);

/* packedT: the SAME transposed image sunerf_mlp_dgrad ran with -- sunerf_pack_mlp_t folds a power of two per layer into it
 * that keeps the data gradient at the scale of g_raw from layer to layer (fp16 operands), and the sums are divided by
 * those powers here */
int sunerf_mlp_wgrad(int d_filter, int n_linear, int d_out, const void* packedT, const void* act_stash,
                     const void* dz_stash, const float* g_raw, const void* g_absmax, int64_t n_rays, int n_samples,
                     void* workspace, int split, float* const* grad_weights_host, float* const* grad_biases_host,
                     int accumulate, void* stream);

/* Layer-pipelined backward (d_filter = 256, n_linear >= 3, a 256-CU device): sunerf_mlp_dgrad + sunerf_mlp_wgrad in one
 * pass that never writes the hidden layers' dZ to HBM (csrc/bwd_pipe.hip).  Replaces the same autograd span of
 * sunerf/model/model.py:44-57.  A streaming prologue forms dZ of the last activation layer and the out layer's dW / db; then
 * one persistent launch in which pairs of workgroups own one Linear layer each (its dW accumulators and W^T rows stay in
 * registers) and hand dZ from layer to layer through the L2 of the XCD they share.
 *   act_stash: written by the forward with stash_format = SUNERF_STASH_PHASE [ABI 9] (the 16-bit phase of every pre-activation:
 *              half the bytes of the fp16 sin + cos stash the two-kernel backward reads; decoded inside the kernel)
 *   workspace: sunerf_bwd_pipe_workspace_bytes(...) bytes (0 = configuration not supported: use dgrad + wgrad).  Its first 256
 *              bytes (SUNERF_PIPE_WS_STICKY) are a STICKY STATUS block that belongs to the caller: zero it once after the
 *              allocation; word 0 is only ever raised by the library, to the largest launch status seen: 0 = every launch since
 *              the caller last cleared it ran to the end; non-zero = a launch gave up (1: its workgroups were not co-resident,
 *              2: a class of workgroups was not placed on one XCD, 3: a hand-off timed out) -- the gradients of THAT call are
 *              NaN (the optimiser's non-finite guard skips the step) and the caller should fall back to sunerf_mlp_dgrad +
 *              sunerf_mlp_wgrad.  One workspace may serve any number of launches between two looks at the word (the
 *              per-launch control block behind it is cleared in front of every launch; a give-up of an earlier launch
 *              survives later ones here).  Debug counters (flags bit 1): 64 KiB at SUNERF_PIPE_WS_DEBUG.
 *   flags    : bit 0 = single fp16 W^T in the data gradient (default: fp16 head + fp16 remainder, as sunerf_mlp_dgrad);
 *              bit 1 = per-workgroup debug counters; bit 7 = bracket the pipelined kernel of this call with library-owned
 *              HIP events on `stream` (read and released by sunerf_bwd_pipe_kernel_time: bench.py's roofline line times the
 *              dominant kernel without changing the call sequence users run);
 *              bit 8 = TEST HOOK: the placement check of workgroup class 0 fails, so the launch gives up the way a really
 *              misplaced one would (status 2, NaN gradients) -- tests/test_gpu_pipe.py exercises the fallback with it
 * Requires that no other kernel holds CUs of the device while it runs long enough to starve it (all 256 workgroups must
 * become resident; every wait is bounded, so a starved launch gives up instead of hanging).
 * sunerf_bwd_pipe_kernel_time: waits for the timed launches (flags bit 7) issued so far by this process, returns the sum of
 * their kernel durations (ms) and their number, and forgets them. */
#define SUNERF_PIPE_WS_STICKY 0
#define SUNERF_PIPE_WS_DEBUG  256
size_t sunerf_bwd_pipe_workspace_bytes(int64_t n_rays, int n_samples, int d_filter, int n_linear);
int sunerf_mlp_backward_pipe(int d_filter, int n_linear, int d_out, const void* packedT, const void* act_stash,
                             const float* g_raw, const void* g_absmax, int64_t n_rays, int n_samples, void* workspace,
                             size_t workspace_bytes, float* const* grad_weights_host, float* const* grad_biases_host,
                             int accumulate, int flags, void* stream);
int sunerf_bwd_pipe_kernel_time(double* total_ms, int* launches);

/* The same gradients in the REFERENCE's arithmetic, for small batches: every product and sum in fp32 (fp32-input MFMA), the
 * forward activations recomputed in fp32 from the query points (the fp16 activation stash is not read).  Replaces
 * torch.autograd over sunerf/model/model.py:44-57 + 123-132 where the fp16 kernels above are not the right tool: their
 * operands (dZ, cos, H) carry 2^-12 of relative rounding error per term, which a training batch averages away but a sum over a
 * few hundred samples that cancels to a few per cent of its terms does not (bias gradients of tiny batches: 2e-3 ... 3e-2).
 *   weights / biases          : host arrays of n_linear DEVICE pointers, nn.Linear layouts of the kernel shapes
 *                               ([d_filter][84], [d_filter][d_filter] ..., [d_out][d_filter]; fp32)
 *   query points              : either rays (rays_o, rays_d (N,3), times (N), z_vals (N,S); points = o + d z as sampling.py:100
 *                               forms them) or `points` (N*S, 4) given explicitly (then the ray arguments may be NULL)
 *   g_raw (N,S,d_out)         : gradient w.r.t. the raw MLP output (no scaling convention: plain fp32)
 *   workspace                 : sunerf_mlp_backward_exact_workspace_bytes(N*S, d_filter, n_linear) bytes
 *   grad_weights / grad_biases: as sunerf_mlp_wgrad (overwritten, or added to when accumulate != 0)
 * Cost ~ 0.25 us per sample of an 8 x 256 network: meant for <= a few thousand samples (sunerf_hip/ops.py picks it by count). */
size_t sunerf_mlp_backward_exact_workspace_bytes(int64_t n_points, int d_filter, int n_linear);
int sunerf_mlp_backward_exact(const float* const* weights_host, const float* const* biases_host, int n_linear, int d_filter,
                              int d_out, const float* rays_o, const float* rays_d, const float* times, const float* z_vals,
                              const float* points, int64_t n_rays, int n_samples, const float* g_raw, void* workspace,
                              size_t workspace_bytes, float* const* grad_weights_host, float* const* grad_biases_host,
                              int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Density / temperature head (run_density_temperature.py path).
 * Replaces DensityTemperatureRadiativeTransfer.raw2outputs / regularization, density_temperature.py:192-274, the base
 * offsets of NeRF_DT.forward, model.py:181-185, and the DT epilogues of base_tracing.py:99-110; the per-wavelength Python
 * loop with host syncs (density_temperature.py:245-256) is one launch.  The MLP runs in sunerf_emission_render_fwd
 * (its `raw` output is the input here; its emission outputs are ignored).
 *
 *   raw (N,S,2); wavelengths (N,W<=7) in Angstrom, <= 0 = channel absent; table_logt / table_resp (7,101) fp32 =
 *   LOGTE / TRESP x exposure time of aia_temp_resp.genx in channel order 94,131,171,193,211,304,335
 *   (density_temperature.py:131-146); log_abs (7) = log_absortpion parameters in that order; vol_c (1)
 *   image (N,W); weights (N,S) = relu(inf0)/(sum+1e-10); reg_q (N,S) = relu(inf0);
 *   height_map / absorption_map (N) and regularization (N,S) optional; reg_radius = 1.25 / Rs_per_ds
 *   backward: g_image (N,W), g_reg (N,S) or NULL -> g_raw (N,S,2) (feed sunerf_mlp_dgrad / sunerf_mlp_wgrad),
 *   g_log_abs (7), g_vol_c (1) (overwritten), g_absmax as in sunerf_emission_integral_bwd
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_dt_integral_fwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                           const float* wavelengths, int n_wavelengths, const float* table_logt, const float* table_resp,
                           const float* log_abs, const float* vol_c, float base_log_density, float base_log_temperature,
                           float pixel_intensity_factor, float reg_radius, int64_t n_rays, int n_samples, float* image,
                           float* weights, float* reg_q, float* height_map, float* absorption_map, float* regularization,
                           void* stream);

int sunerf_dt_integral_bwd(const float* raw, const float* z_vals, const float* rays_o, const float* rays_d,
                           const float* wavelengths, int n_wavelengths, const float* table_logt, const float* table_resp,
                           const float* log_abs, const float* vol_c, float base_log_density, float base_log_temperature,
                           float pixel_intensity_factor, float reg_radius, int64_t n_rays, int n_samples,
                           const float* g_image, const float* g_reg, float* g_raw, float* g_log_abs, float* g_vol_c,
                           void* g_absmax, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input side of the path: observer rays on the device.
 * Replaces get_rays, sunerf/data/ray_sampling.py:7-36, and the host-side tiling / H2D copy of the rays and of the
 * time column in SuNeRFLoader.render_observer_image, sunerf/evaluation/loader.py:73-92 and :186-214.
 *   tx, ty   : helioprojective angles [rad], fp64, device.  per_pixel = 0: tx[width] (columns) and ty[rows] (axes of a
 *              regular grid; pixel p = row * width + column);  per_pixel != 0: tx[p], ty[p] for every pixel of the frame
 *              (e.g. sunpy's all_coordinates_from_map for a real WCS)
 *   pixels [pix_begin, pix_begin + n_pix) of the frame are produced (one tile)
 *   c2w_host : HOST pointer, 12 floats = rows of pose_spherical(...)[:3, :4] (train/coordinate_transformation.py:36-54)
 *   rays_o, rays_d : [n_pix, 3] out;  times : [n_pix] out, filled with time_value (may be NULL)
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_observer_rays(const double* tx, const double* ty, int per_pixel, int width, int64_t pix_begin, int64_t n_pix,
                         const float* c2w_host, float time_value, float* rays_o, float* rays_d, float* times,
                         void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Output side of the path: training loss and optimiser step without host synchronisation.
 *
 * sunerf_training_loss replaces EmissionSuNeRFModule.training_step's loss section, sunerf/model/sunerf.py:105-125
 * (finite asserts :105-107, ImageAsinhScaling sunerf/train/scaling.py:17-28, 2 x nn.MSELoss, regularization.mean(),
 * psnr) and, with scaling = 0, DensityTemperatureSuNeRFModule.training_step sunerf.py:185-200 (plain MSE).
 *   coarse_image / fine_image / target_image : n = N * W floats each;  regularization : n_reg floats (may be 0)
 *   finite_check_host[n_finite_check <= 8]   : HOST arrays of further device tensors (+ sizes) that only take part in
 *                                              the NaN / Inf count (z_vals, height_map, ...: sunerf.py:105-107)
 *   scaling 1 = asinh(x / vmax / a) / asinh(1 / a) applied to all three images; 0 = none
 *   g_coarse / g_fine : d loss / d coarse_image, d loss / d fine_image (n floats each); d loss / d regularization is
 *                       the constant lambda_regularization / n_reg
 *   stats (8 floats, device): loss, coarse MSE, fine MSE, regularization mean, psnr, number of non-finite values, 0, 0
 *   workspace: sunerf_train_workspace_bytes() bytes, ZERO-INITIALISED once by the caller (the kernels leave it zeroed);
 *              one workspace serves one stream
 *
 * sunerf_clip_adam_step replaces torch.nn.utils.clip_grad_norm_(params, max_norm) (Lightning gradient_clip_val,
 * run_emission.py:72) followed by torch.optim.Adam.step() (sunerf.py:31) on ONE flat fp32 buffer:
 *   g = grads * grad_scale (1 / world size after a sum all-reduce); total = ||g||_2; g *= min(1, max_norm / (total + 1e-6));
 *   m += (1 - beta1)(g - m); v = beta2 v + (1 - beta2) g g; p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *   max_norm <= 0 disables clipping; step counts from 1.
 *   skip_if_positive: optional device float -- the number of non-finite outputs of this step summed over ALL ranks (the
 *     caller carries it as one extra element at the tail of the all-reduced gradient bucket); a value > 0
 *     leaves params and moments untouched, and so does a non-finite gradient norm: every rank takes the same decision
 *     because both inputs are results of the all-reduce (the reference asserts instead, sunerf.py:105-107).
 *   norm_out (4 floats, device): total norm, clip coefficient, skipped (0 / 1), 0.  grads holds the scaled, clipped
 *     gradient afterwards.  norm_out / workspace may be NULL only with max_norm <= 0 and step_counter == NULL (then there
 *     is no norm pass and only skip_if_positive can skip).
 *   step_counter: optional device int64 -- number of APPLIED updates.  When given it is authoritative (`step` is ignored):
 *     the update uses *step_counter + 1 for the bias corrections and advances the counter only if it is not skipped.
 * ---------------------------------------------------------------------------------------------------------- */
size_t sunerf_train_workspace_bytes(void);
int sunerf_training_loss(const float* coarse_image, const float* fine_image, const float* target_image, int64_t n,
                         const float* regularization, int64_t n_reg, const float* const* finite_check_host,
                         const int64_t* finite_check_sizes_host, int n_finite_check, int scaling, float vmax, float a,
                         float lambda_image, float lambda_regularization, float* g_coarse, float* g_fine, float* stats,
                         void* workspace, size_t workspace_bytes, void* stream);
int sunerf_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                          double beta1, double beta2, double eps, float max_norm, float grad_scale, int64_t step,
                          const float* skip_if_positive, float* norm_out, void* workspace, size_t workspace_bytes,
                          void* step_counter, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Analytic field of SimpleStar: replaces SimpleStar.forward, sunerf/model/stellar_model.py:53-102,
 * evaluated at the sample points o + d z (sampling.py:100) of every ray; the result feeds sunerf_dt_integral_fwd with
 * base_log_density = base_log_temperature = 0 exactly as the MLP output of NeRF_DT does
 * (DensityTemperatureRadiativeTransfer(model=SimpleStar), evaluation/image_render.py:266-268).
 *   raw [N, S, 2] out: (ln rho, log10 T);  rho_0 [cm^-3], h0 and Rs [solar radii], T0 and t_photosphere [K]
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_simple_star_field(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays, int n_samples,
                             float rho_0, float h0, float T0, float Rs, float t_photosphere, float* raw, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Hierarchical (inverse-CDF) resampling + merge.
 * Replaces HierarchicalSampler.forward / sample_pdf, sampling.py:111-169 (perturb=False: u = linspace(0,1,S_f),
 * passed in as the tensor `u` [S_f] so that torch.linspace's own fp32 values are used; or a per-ray u [N,S_f]
 * with u_per_ray != 0 for perturb=True).
 *
 *   z_vals [N,S_c], weights [N,S_c] -> new_z [N,S_f], z_comb [N,S_c+S_f] (sorted)
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_hier_resample(const float* z_vals, const float* weights, const float* u, int u_per_ray,
                         int64_t n_rays, int n_coarse, int n_fine, float* new_z, float* z_comb, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Inverse-CDF sampling on given bins.
 * Replaces HierarchicalSampler.sample_pdf called by itself, sampling.py:128-169: pdf = (w + 1e-5) / sum(w + 1e-5),
 * cdf = [0, cumsum(pdf)], searchsorted(cdf, u, right=True), linear interpolation between the neighbouring bins with the
 * reference's `denom < 1e-5 -> 1` rule.  `u` as in sunerf_hier_resample.
 *
 *   bins [N,B], weights [N,B-1] -> samples [N,S_f]
 * ---------------------------------------------------------------------------------------------------------- */
int sunerf_sample_pdf(const float* bins, const float* weights, const float* u, int u_per_ray, int64_t n_rays,
                      int n_bins, int n_fine, float* samples, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUNERF_HIP_H */
