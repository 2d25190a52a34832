// oracle/_ref/libngpadam_ref.so, part 2 of 2 -- TEST INFRASTRUCTURE ONLY.  oracle/Makefile assembles the translation unit on a pipe: ref_adam_pre.hpp + the text of
// `class VarAdamOptimizer` of /root/reference/include/neural-graphics-primitives/adam_optimizer.h read where it lies (the header's other optimizers need tcnn's rotation helpers,
// absent from the mount) + this file.  The reference's class, driven the way train_nerf drives the per-image latent optimizers (src/testbed_nerf.cu:2860-2878): gradient / LOSS_SCALE, set_learning_rate(network optimizer's rate), step().
// Pins oracle/ora_nerf.hpp var_adam_step (and through it csrc k_extra_dims_adam, tests/test_extra_dims.py).
} // namespace ngp
using namespace ngp;
// n_steps steps of one optimizer of `n` variables created like Testbed::Nerf::Training::reset_extra_dims does (VarAdamOptimizer(n, 1e-4f), variable() = initial values);
// gradients_scaled: n_steps x n (as accumulated on the device, i.e. times the loss scale), learning_rates: n_steps.  Returns the variables after every step (n_steps x n).
extern "C" __attribute__((visibility("default"))) void ref_var_adam_steps(uint32_t n, const float* initial, const float* gradients_scaled, const float* learning_rates, uint32_t n_steps,
		float loss_scale, float* variables_out) {
	VarAdamOptimizer opt(n, 1e-4f);
	for (uint32_t i = 0; i < n; ++i) opt.variable()[i] = initial[i];
	for (uint32_t s = 0; s < n_steps; ++s) {
		std::vector<float> gradient(n);
		for (uint32_t j = 0; j < n; ++j) gradient[j] = gradients_scaled[(size_t)s * n + j] / loss_scale;
		opt.set_learning_rate(learning_rates[s]);
		opt.step(gradient);
		for (uint32_t j = 0; j < n; ++j) variables_out[(size_t)s * n + j] = opt.variable()[j];
	}
}
