// adam_driver.cpp -- thin C-ABI driver around LibTorch's C++ torch::optim::Adam.  TEST INFRASTRUCTURE ONLY.
//
// The reference's optimizer arithmetic is not in the reference tree: it is LibTorch's (pinned 2.0.1+cu118, README.md:109;
// this container links LibTorch 2.10 CPU).  This driver configures the optimizer the way the reference does
// (src/gaussian_model.cpp:632-640: default AdamOptions, lr set per group, eps = 1e-15) and steps it on given gradients,
// so that tests/golden/make_adam_golden.py can commit input/output vectors for the fused HIP Adam.
#include <torch/torch.h>

extern "C" int ref_adam_steps(float* param /*n, in-out*/, const float* grads /*steps x n*/, int n, int steps, double lr,
                              float* exp_avg_out, float* exp_avg_sq_out) {
  try {
    auto opts = torch::TensorOptions().dtype(torch::kFloat32);
    torch::Tensor p = torch::from_blob(param, {n}, opts).clone().requires_grad_(true);
    std::vector<torch::Tensor> group{p};
    torch::optim::AdamOptions adam_options;
    adam_options.set_lr(0.0);
    adam_options.eps() = 1e-15;
    torch::optim::Adam opt(group, adam_options);
    opt.param_groups()[0].options().set_lr(lr);
    for (int s = 0; s < steps; s++) {
      p.mutable_grad() = torch::from_blob(const_cast<float*>(grads) + (size_t)s * n, {n}, opts).clone();
      opt.step();
    }
    std::memcpy(param, p.data_ptr<float>(), sizeof(float) * n);
    auto& st = static_cast<torch::optim::AdamParamState&>(*opt.state().at(p.unsafeGetTensorImpl()));
    std::memcpy(exp_avg_out, st.exp_avg().contiguous().data_ptr<float>(), sizeof(float) * n);
    std::memcpy(exp_avg_sq_out, st.exp_avg_sq().contiguous().data_ptr<float>(), sizeof(float) * n);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_adam_steps: %s\n", e.what());
    return 1;
  }
}
