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
 * with g = grad * grad_scale (1/world_size for keyframe-parallel averaging).
 */
#ifndef SEGS_TRAIN_H_
#define SEGS_TRAIN_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct segs_adam_segment {
  int64_t offset;  /* first element of the segment inside the flat bucket */
  int64_t count;   /* number of elements */
  float lr;        /* learning rate of this parameter group */
} segs_adam_segment;

/* One Adam step over `nseg` (<= 16) segments of the flat bucket; `segments` is a HOST array.
 * `step` is the 1-based step count after increment (LibTorch increments before use).  If zero_grad != 0 the
 * gradient bucket is cleared in the same pass (zero_grad of src/gaussian_trainer.cpp:116 folded in). */
int segs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                   const segs_adam_segment* segments, int nseg,
                   float beta1, float beta2, float eps, int64_t step, float grad_scale, int zero_grad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_TRAIN_H_ */
