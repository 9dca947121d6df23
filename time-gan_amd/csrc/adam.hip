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
constexpr int kThreads = 1024;     // the largest tensors (64 K elements) are sixteen float4 rounds of one workgroup

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
    // whole float4s where all four arrays are 16-byte aligned (gradients may be views into a flat buffer)
    int done = 0;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ma) | reinterpret_cast<uintptr_t>(va) |
          reinterpret_cast<uintptr_t>(ga)) & 15) == 0) {
        const float omb1 = 1.f - t.beta1, omb2 = 1.f - t.beta2;
        for (int i4 = threadIdx.x; i4 < (n >> 2); i4 += kThreads) {
            const float4 g4 = reinterpret_cast<const float4 *>(ga)[i4];
            float4 m4 = reinterpret_cast<float4 *>(ma)[i4], v4 = reinterpret_cast<float4 *>(va)[i4];
            float4 p4 = reinterpret_cast<float4 *>(p)[i4];
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
            float m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w}, w[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[e] = m[e] + omb1 * (g[e] - m[e]);
                v[e] = t.beta2 * v[e] + omb2 * g[e] * g[e];
                w[e] -= step_size * m[e] / (sqrtf(v[e]) / bc2s + t.eps);
            }
            reinterpret_cast<float4 *>(p)[i4] = make_float4(w[0], w[1], w[2], w[3]);
            reinterpret_cast<float4 *>(ma)[i4] = make_float4(m[0], m[1], m[2], m[3]);
            reinterpret_cast<float4 *>(va)[i4] = make_float4(v[0], v[1], v[2], v[3]);
        }
        done = n & ~3;
    }
    for (int idx = done + threadIdx.x; idx < n; idx += kThreads) {
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
