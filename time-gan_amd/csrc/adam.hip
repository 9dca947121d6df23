// Adam for many small tensors in ONE launch.
//
// A TimeGAN sub-step updates two networks, about forty parameter tensors of a few hundred to a few
// thousand floats.  torch's fused Adam needs one launch to advance the step counters and one per ~30
// tensors for the update, each shorter than its launch.  Here a job table (tensor pointers in the kernel
// arguments) covers all of them; the arithmetic is torch.optim.Adam's (no weight decay, no amsgrad,
// torch/optim/adam.py with fused=True):
//      step += 1;  m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g^2;
//      p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
// (torch evaluates the bias corrections in double; here they are formed as -expm1(step log b), which
// is as accurate in single precision).
// `skip` (device float, may be NULL): a non-zero value turns the whole launch into a no-op -- the
// on-device form of "if d_loss > threshold: step()" (outdated_files/TimeGANv1.py:805-808).
#include "../../include/tgru.h"
#include "tgru_internal.h"
#include <math.h>

namespace tgru {
namespace adam {

constexpr int kMaxJobs = TGRU_MAX_ADAM_JOBS;
constexpr int kThreads = 256;

struct Table {
    tgru_adam_job job[kMaxJobs];
    float lr, beta1, beta2, eps, log_beta1, log_beta2;
    const float *skip;
};

// One workgroup per tensor: it reads the tensor's step counter, updates the elements and then advances the
// counter itself -- no other workgroup touches it, so no grid-wide ordering is needed.
__global__ __launch_bounds__(kThreads) void adam_jobs_kernel(Table t) {
    if (t.skip && *t.skip != 0.f) return;
    const tgru_adam_job &j = t.job[blockIdx.x];
    // 1 - beta^step = -expm1(step * log(beta)): accurate in single precision also while beta^step is
    // close to 1 (torch evaluates the powers in double); the logarithms come from the host
    const float step = *j.step + 1.f;
    const float step_size = t.lr / -expm1f(step * t.log_beta1);
    const float bc2s = sqrtf(-expm1f(step * t.log_beta2));
    const int n = j.n;
    float *p = j.param, *ma = j.exp_avg, *va = j.exp_avg_sq;
    const float *ga = j.grad;
    for (int idx = threadIdx.x; idx < n; idx += kThreads) {
        const float g = ga[idx];
        float m = ma[idx], v = va[idx];
        m = m + (1.f - t.beta1) * (g - m);
        v = t.beta2 * v + (1.f - t.beta2) * g * g;
        p[idx] -= step_size * m / (sqrtf(v) / bc2s + t.eps);
        ma[idx] = m;
        va[idx] = v;
    }
    __syncthreads();              // every thread has read the counter
    if (threadIdx.x == 0) *j.step = step;
}

}  // namespace adam
}  // namespace tgru

using namespace tgru;

extern "C" int tgru_adam_jobs(const tgru_adam_job *jobs, int n_jobs, float lr, float beta1, float beta2, float eps,
                              const float *skip, tgru_stream_t stream) {
    if (n_jobs < 0 || n_jobs > adam::kMaxJobs) return TGRU_E_DIMS;
    if (n_jobs == 0) return TGRU_OK;
    if (!jobs) return TGRU_E_NULL;
    adam::Table t{};
    for (int k = 0; k < n_jobs; ++k) {
        const tgru_adam_job &j = jobs[k];
        if (!j.param || !j.grad || !j.exp_avg || !j.exp_avg_sq || !j.step) return TGRU_E_NULL;
        if (j.n <= 0 || j.n > TGRU_MAX_ADAM_ELEMS) return TGRU_E_DIMS;
        t.job[k] = j;
    }
    t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps;
    t.log_beta1 = (float)log((double)beta1); t.log_beta2 = (float)log((double)beta2);
    t.skip = skip;
    TGRU_LAUNCH(adam::adam_jobs_kernel, dim3(n_jobs), dim3(adam::kThreads), 0, (hipStream_t)stream, t);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}
