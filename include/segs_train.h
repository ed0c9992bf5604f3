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

/* Fused L1 + SSIM loss of the trainer/mapper step and its gradient w.r.t. the rendered image:
 *     loss = (1 - lambda) * mean|img1 - img2| + lambda * (1 - mean(SSIM(img1, img2)))
 * (src/gaussian_trainer.cpp:89-90, src/gaussian_mapper.cpp:924-928 with loss_utils::l1_loss / ssim,
 * include/loss_utils.h:29-32,51-124: 11x11 window, sigma 1.5, zero padding 5, C1 = 1e-4, C2 = 9e-4, mean over all
 * 3*H*W elements).  img1/img2/dL_dimg1 are (3,H,W) planar fp32; loss_out is 3 device floats {loss, l1, ssim};
 * temp holds segs_l1_ssim_temp_bytes(H, W) bytes. */
size_t segs_l1_ssim_temp_bytes(int H, int W);
int segs_l1_ssim_loss(const float* img1, const float* img2, int H, int W, float lambda_dssim, float* loss_out,
                      float* dL_dimg1, char* temp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_TRAIN_H_ */
