/* segs_train.h -- C ABI of the optimizer step of the mapper/trainer loop (part of libsegs_raster.so).
 *
 * The reference performs the parameter update with LibTorch's torch::optim::Adam over ~20 tensors
 * (src/gaussian_model.cpp:620-872 builds the groups with eps = 1e-15, default betas, no weight decay, no amsgrad;
 * stepped at src/gaussian_mapper.cpp:1027-1030 and src/gaussian_trainer.cpp:115-116).  Here the Gaussian parameters
 * live in ONE flat fp32 bucket (also the RCCL all-reduce operand), updated by one fused HBM-streaming kernel
 * (28 B per parameter: read p,g,m,v, write p,m,v).  Arithmetic follows LibTorch 2.0.1's C++ Adam
 * (the reference's pinned version, README.md:109):
 *     m = m*b1 + g*(1-b1);  v = v*b2 + g*g*(1-b2);
 *     denom = sqrt(v)/sqrt(1 - b2^t) + eps;  p = p - (lr/(1 - b1^t)) * (m/denom)
 * with g = grad * grad_scale (1/world_size for keyframe-parallel averaging).  Like LibTorch, the hyper-parameters are
 * doubles: 1-b1, 1-b2, the bias corrections and lr/(1-b1^t) are formed in double and only then rounded to the tensor's
 * float32 (so 1-b2 is float(0.001), not 1.0f - 0.999f); checked against LibTorch's own C++ Adam (tests/golden/adam_libtorch.npz).
 */
#ifndef SEGS_TRAIN_H_
#define SEGS_TRAIN_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct segs_adam_segment {
  int64_t offset;  /* first element of the segment inside the flat bucket */
  int64_t count;   /* number of elements */
  double lr;       /* learning rate of this parameter group (LibTorch keeps it in double) */
} segs_adam_segment;

/* One Adam step over `nseg` (<= 16) segments of the flat bucket; `segments` is a HOST array.
 * `step` is the 1-based step count after increment (LibTorch increments before use).  If zero_grad != 0 the
 * gradient bucket is cleared in the same pass (zero_grad of src/gaussian_trainer.cpp:116 folded in). */
int segs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                   const segs_adam_segment* segments, int nseg,
                   double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad, void* stream);

/* Same, guarded on the device: if skip_flag is non-NULL and *skip_flag != 0 when the kernel runs, parameters and moments
 * stay untouched and the gradient bucket is only cleared.  Pass the overflow word of the resident rasterizer
 * (status + 3, segs_raster.h) so that an iteration whose instance count outgrew the scratch capacity is dropped without a
 * host synchronisation; the host still counts it as a step (the caller may roll its counter back when it learns of it). */
int segs_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                           const segs_adam_segment* segments, int nseg,
                           double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad,
                           const uint32_t* skip_flag, void* stream);

/* Same, with the step count kept ON THE DEVICE: device_steps points at two int64 words (zero-filled before the first
 * call; the pair is used in turn -- call number `call_index` = 0, 1, 2, ... of this state reads word call_index & 1 as the
 * number of steps taken so far and leaves the count after this call in the other word).  A call whose skip_flag is set does
 * not advance the count, exactly as torch::optim::Adam only counts the steps it takes; with the flag all-reduced over the
 * ranks of a keyframe-parallel job every replica drops the same steps and no rank ever has to synchronise with its device
 * to keep the bias corrections (formed in the kernel, in double: 1 - pow(beta, t)) right. */
int segs_adam_step_device(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                          const segs_adam_segment* segments, int nseg,
                          double beta1, double beta2, double eps, int64_t* device_steps, int call_index,
                          float grad_scale, int zero_grad, const uint32_t* skip_flag, void* stream);

/* hipGraph-capturable form of segs_adam_step_device: nothing in the launch depends on a host value that changes from step to
 * step, so a captured iteration can be replayed.  The learning rates are read from DEVICE memory (device_lr[i] for segment i;
 * segments[i].lr is ignored) and the parity of the step-count pair is a third device word: device_steps3 points at THREE int64
 * words, [0], [1] the pair of segs_adam_step_device and [2] = number of calls so far (its low bit says which word holds the
 * count), flipped by a one-thread kernel behind the update.  A caller that alternates between the two forms keeps [2] equal
 * to its own call count.  segs_set_doubles writes up to 16 host doubles into device memory through kernel arguments (they are
 * copied at launch time: no pinned staging buffer to keep alive, nothing to wait for) -- the learning-rate refresh before a
 * replay. */
int segs_adam_step_graph(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments, int nseg,
                         const double* device_lr, double beta1, double beta2, double eps, int64_t* device_steps3,
                         float grad_scale, int zero_grad, const uint32_t* skip_flag, void* stream);
int segs_set_doubles(double* device_dst, const double* host_values, int n, void* stream);

/* Fused L1 + SSIM loss of the trainer/mapper step and its gradient w.r.t. the rendered image:
 *     loss = (1 - lambda) * mean|img1 - img2| + lambda * (1 - mean(SSIM(img1, img2)))
 * (src/gaussian_trainer.cpp:89-90, src/gaussian_mapper.cpp:924-928 with loss_utils::l1_loss / ssim,
 * include/loss_utils.h:29-32,51-124: 11x11 window, sigma 1.5, zero padding 5, C1 = 1e-4, C2 = 9e-4, mean over all
 * 3*H*W elements).  img1/img2/dL_dimg1 are (3,H,W) planar fp32; loss_out is 3 device floats {loss, l1, ssim};
 * temp holds segs_l1_ssim_temp_bytes(H, W) bytes. */
size_t segs_l1_ssim_temp_bytes(int H, int W);
int segs_l1_ssim_loss(const float* img1, const float* img2, int H, int W, float lambda_dssim, float* loss_out,
                      float* dL_dimg1, char* temp, void* stream);

/* ---- Frequency regulariser of the mapper loss (src/gaussian_mapper.cpp:930-945) ----------------------------------------
 * loss_utils::multi_scale_loss / high_frequency_loss (include/loss_utils.h:126-165, 216-237):
 *     freq = lambda_high * sum_s  s * mean( | |fft2(resize_s(image))| - |fft2(resize_s(gt))| | ),   s = 1, 1/2, 1/4
 * (without Mapper.use_multi_resolution: the s = 1 term alone).  resize_s is torch's bilinear interpolate with
 * align_corners = false and recompute_scale_factor = true, i.e. to floor(size * s) with scale = in / out.  The reference's
 * frequency mask is a no-op for real image sizes and low_freq_loss has a zero gradient (pinned by
 * tests/golden/loss_reference.npz), so this is the whole regulariser.
 *
 * The FFTs stay library calls of the caller (hipFFT through torch.fft / torch::fft, like the reference); these entry points
 * are everything around them, one launch each for all scales ("levels"; at most SEGS_FREQ_MAX_LEVELS):
 *   1. segs_freq_pyramid           image (C,H,W) -> level_out[l] (C,h_l,w_l) for every level smaller than the image (a level
 *                                  of the image's own size is the image: its buffer is not written and may be NULL);
 *   2. caller: spectrum[l] = rfft2(level l)  -- (C, h_l, w_l/2+1) interleaved complex64, unnormalised;
 *   3. segs_freq_spectrum_loss     *freq_loss_out = sum_l level_weight[l] * sum_k m_k | |G_k| - target_magnitude[l][k] |
 *                                  (m_k = 2 for the columns that stand for their mirror image too, else 1), added to
 *                                  *loss_inout if non-NULL; spectrum[l] <- level_weight[l] * sign(|G|-|T|) * G/|G| in place.
 *                                  level_weight[l] = lambda_high * s_l / (C h_l w_l);
 *   4. caller: level_grad[l] = irfft2(spectrum[l], size (h_l,w_l)) WITHOUT the 1/(h w) normalisation (norm = "forward");
 *   5. segs_freq_pyramid_backward_add   dL_dimage += sum_l resize_l^T(level_grad[l])   (plain add for a full-size level).
 * target_magnitude[l] = |rfft2(resize_l(gt))| is constant per keyframe: segs_freq_pyramid + rfft2 + segs_spectrum_magnitude
 * once, then cached by the host.  temp: segs_freq_temp_bytes(...) bytes. */
#define SEGS_FREQ_MAX_LEVELS 4
int segs_freq_pyramid(const float* image, int C, int H, int W, int nlevels, const int* level_h, const int* level_w,
                      float* const* level_out, void* stream);
int segs_spectrum_magnitude(const float* spectrum, size_t n_complex, float* magnitude, void* stream);
size_t segs_freq_temp_bytes(int C, int nlevels, const int* level_h, const int* level_w);
int segs_freq_spectrum_loss(int C, int nlevels, const int* level_h, const int* level_w, float* const* spectrum,
                            const float* const* target_magnitude, const float* level_weight, float* freq_loss_out,
                            float* loss_inout, char* temp, void* stream);
int segs_freq_pyramid_backward_add(float* dL_dimage, int C, int H, int W, int nlevels, const int* level_h, const int* level_w,
                                   const float* const* level_grad, void* stream);

/* The whole regulariser in one call, transforms included ("plan": one per image size; owns its hipFFT plans and scratch).
 * hipFFT is bound at run time by soname (the copy the process already carries, e.g. PyTorch-ROCm's, else the system one);
 * SEGS_ERR_UNSUPPORTED if there is none.  scales: Mapper.scale_num values 1 / 2^i (src/gaussian_mapper.cpp:514-517), or {1}
 * for high_frequency_loss alone.  When H and W are multiples of 4 and the scales are {1, 1/2, 1/4} the plan evaluates all
 * three scales from ONE forward and ONE inverse full-size transform (alias folding, csrc/freq_loss.hip); otherwise one pair
 * per scale as in steps 1-5 above.
 *   segs_freq_target   target_out[segs_freq_target_floats(plan)] = |FFT| tables of a target image (once per keyframe)
 *   segs_freq_loss     *freq_loss_out = the regulariser's value (also added to *loss_inout if non-NULL);
 *                      dL_inout (3,H,W) += its gradient w.r.t. image.  Nothing waits for the device. */
typedef struct segs_freq_plan segs_freq_plan;
int segs_freq_plan_create(int H, int W, int nscales, const float* scales, float lambda_high, segs_freq_plan** out);
void segs_freq_plan_destroy(segs_freq_plan* plan);
int segs_freq_plan_levels(const segs_freq_plan* plan, int* level_h, int* level_w, int* folded);   /* returns the level count */
size_t segs_freq_target_floats(const segs_freq_plan* plan);
int segs_freq_target(segs_freq_plan* plan, const float* gt, float* target_out, void* stream);
int segs_freq_loss(segs_freq_plan* plan, const float* image, const float* target, float* dL_inout, float* freq_loss_out,
                   float* loss_inout, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* SEGS_TRAIN_H_ */
