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

// The reference's frequency-domain losses (include/loss_utils.h:126-213) and their autograd gradients on the CPU.
// multi_scale_loss (:216-237) cannot be driven here: it calls high_frequency_loss with its default device (kCUDA).
extern "C" int ref_freq_losses(const float* img1, const float* img2, int H, int W, float* out2 /*high, low*/, float* dL_high,
                               float* dL_low) {
  try {
    auto opts = torch::TensorOptions().dtype(torch::kFloat32);
    torch::Tensor b = torch::from_blob(const_cast<float*>(img2), {3, H, W}, opts).clone();
    {
      torch::Tensor a = torch::from_blob(const_cast<float*>(img1), {3, H, W}, opts).clone().requires_grad_(true);
      auto l = loss_utils::high_frequency_loss(a, b, 0.4, torch::kCPU);
      l.backward();
      out2[0] = l.item<float>();
      std::memcpy(dL_high, a.grad().contiguous().data_ptr<float>(), sizeof(float) * 3 * (size_t)H * W);
    }
    {
      torch::Tensor a = torch::from_blob(const_cast<float*>(img1), {3, H, W}, opts).clone().requires_grad_(true);
      auto l = loss_utils::low_freq_loss(a, b, 0.2, torch::kCPU);
      l.backward();
      out2[1] = l.item<float>();
      std::memcpy(dL_low, a.grad().contiguous().data_ptr<float>(), sizeof(float) * 3 * (size_t)H * W);
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_freq_losses: %s\n", e.what());
    return 1;
  }
}

// multi_scale_loss (include/loss_utils.h:216-237) piece by piece.  The function itself cannot run here -- it calls
// high_frequency_loss with its default device, kCUDA (:234) -- so this driver issues the same two interpolate calls per scale
// (:225-232: bilinear, align_corners = false, recompute_scale_factor = true) and hands their results to the REFERENCE's
// high_frequency_loss with torch::kCPU, summing scale * loss like :234.  out[0] = the sum, out[1 + i] = the i-th scale's
// high_frequency_loss; dL = autograd gradient of the sum w.r.t. gen_img.
extern "C" int ref_multi_scale_loss(const float* gen, const float* target, int H, int W, const float* scales, int nscales,
                                    float* out, float* dL) {
  try {
    namespace F = torch::nn::functional;
    auto opts = torch::TensorOptions().dtype(torch::kFloat32);
    torch::Tensor a = torch::from_blob(const_cast<float*>(gen), {3, H, W}, opts).clone().requires_grad_(true);
    torch::Tensor b = torch::from_blob(const_cast<float*>(target), {3, H, W}, opts).clone();
    torch::Tensor loss = torch::zeros({});
    for (int i = 0; i < nscales; i++) {
      const float scale = scales[i];
      std::vector<double> sf = {static_cast<double>(scale), static_cast<double>(scale)};
      auto io = F::InterpolateFuncOptions().scale_factor(sf).mode(torch::kBilinear).align_corners(false).recompute_scale_factor(true);
      auto ga = F::interpolate(a.unsqueeze(0), io);
      auto gb = F::interpolate(b.unsqueeze(0), io);
      auto l = loss_utils::high_frequency_loss(ga.squeeze(0), gb.squeeze(0), 0.4, torch::kCPU);
      out[1 + i] = l.item<float>();
      loss = loss + scale * l;
    }
    loss.backward();
    out[0] = loss.item<float>();
    std::memcpy(dL, a.grad().contiguous().data_ptr<float>(), sizeof(float) * 3 * (size_t)H * W);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_multi_scale_loss: %s\n", e.what());
    return 1;
  }
}
