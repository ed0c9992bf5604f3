// loss_driver.cpp -- thin C-ABI driver around the REFERENCE's own loss code.  TEST INFRASTRUCTURE ONLY.
//
// Nothing of the reference is copied: this file #includes /root/reference/include/loss_utils.h from where it lies (it
// depends on <torch/torch.h> only) and calls loss_utils::l1_loss / loss_utils::ssim exactly as the trainer does
// (src/gaussian_trainer.cpp:89-90), on the CPU (`device_type = torch::kCPU` is a parameter of the reference's functions),
// with LibTorch's autograd giving dL/dimage.  Built by oracle/Makefile into oracle/_ref/ (git-ignored) in the container
// that has /root/reference; tests/golden/make_loss_golden.py turns its outputs into committed fixtures.
#include <torch/torch.h>

#include "include/loss_utils.h"

extern "C" int ref_l1_ssim(const float* img1, const float* img2, int H, int W, float lambda_dssim, float* out3, float* dL_dimg1) {
  try {
    auto opts = torch::TensorOptions().dtype(torch::kFloat32);
    torch::Tensor a = torch::from_blob(const_cast<float*>(img1), {3, H, W}, opts).clone().requires_grad_(true);
    torch::Tensor b = torch::from_blob(const_cast<float*>(img2), {3, H, W}, opts).clone();
    auto Ll1 = loss_utils::l1_loss(a, b);
    auto s = loss_utils::ssim(a, b, torch::kCPU);
    auto loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - s);   // gaussian_trainer.cpp:89-90
    loss.backward();
    out3[0] = loss.item<float>();
    out3[1] = Ll1.item<float>();
    out3[2] = s.item<float>();
    auto g = a.grad().contiguous();
    std::memcpy(dL_dimg1, g.data_ptr<float>(), sizeof(float) * 3 * (size_t)H * W);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_l1_ssim: %s\n", e.what());
    return 1;
  }
}
