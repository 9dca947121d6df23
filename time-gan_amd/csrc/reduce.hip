// One launch for the partial-gradient reductions of MANY backward calls.
//
// Every backward kernel of the small-stack path leaves one partial parameter-gradient vector per
// workgroup and a short second kernel sums them.  In a training sub-step a dozen such reductions follow
// each other, each shorter than its launch, and nothing reads their results before the optimiser runs.
// A caller may therefore ask the backward entry points to leave the partials alone (accumulate ==
// TGRU_DEFER_REDUCE), collect the reductions as flat jobs
//      dst[i] = sum over k < count of src[k * stride + i],   i < n
// and run all of them here at once.  Workgroup = 32 elements x 16 groups of partials, summed in a fixed
// order, as in the per-call kernels.
#include "../../include/tgru.h"
#include "tgru_internal.h"

namespace tgru {
namespace red {

constexpr int kMaxJobs = TGRU_MAX_REDUCE_JOBS;
constexpr int kElems = 32, kGroups = 16;

struct JobTable {
    tgru_reduce_job job[kMaxJobs];
    int first[kMaxJobs + 1];     // first workgroup of each job; first[n] = grid size
    int n;
};

__global__ __launch_bounds__(kElems * kGroups) void reduce_jobs_kernel(JobTable t) {
    __shared__ float part[kGroups][kElems];
    int lo = 0, hi = t.n;        // the job this workgroup belongs to (uniform binary search)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const tgru_reduce_job &j = t.job[lo];
    const float *src = j.src;
    const long stride = j.stride;
    const int count = j.count, n = j.n;
    const int e = threadIdx.x & (kElems - 1), grp = threadIdx.x / kElems;
    const int idx = ((int)blockIdx.x - t.first[lo]) * kElems + e;
    const int per = (count + kGroups - 1) / kGroups;
    const int c0 = grp * per, c1 = min(count, c0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (idx < n) {
        int c = c0;
        for (; c + 3 < c1; c += 4) {
            v0 += src[(long)c * stride + idx];
            v1 += src[(long)(c + 1) * stride + idx];
            v2 += src[(long)(c + 2) * stride + idx];
            v3 += src[(long)(c + 3) * stride + idx];
        }
        for (; c < c1; ++c) v0 += src[(long)c * stride + idx];
    }
    part[grp][e] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (grp != 0 || idx >= n) return;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
    for (int g = 0; g < kGroups; g += 4) {
        u0 += part[g][e];
        u1 += part[g + 1][e];
        u2 += part[g + 2][e];
        u3 += part[g + 3][e];
    }
    j.dst[idx] = (u0 + u1) + (u2 + u3);
}

}  // namespace red
}  // namespace tgru

using namespace tgru;

extern "C" int tgru_reduce_jobs(const tgru_reduce_job *jobs, int n_jobs, tgru_stream_t stream) {
    if (n_jobs < 0) return TGRU_E_DIMS;
    if (n_jobs == 0) return TGRU_OK;
    if (!jobs) return TGRU_E_NULL;
    for (int base = 0; base < n_jobs; base += red::kMaxJobs) {
        red::JobTable t{};
        const int n = n_jobs - base < red::kMaxJobs ? n_jobs - base : red::kMaxJobs;
        int blocks = 0;
        for (int k = 0; k < n; ++k) {
            const tgru_reduce_job &j = jobs[base + k];
            if (!j.src || !j.dst) return TGRU_E_NULL;
            if (j.count <= 0 || j.n <= 0) return TGRU_E_DIMS;
            t.job[k] = j;
            t.first[k] = blocks;
            blocks += (j.n + red::kElems - 1) / red::kElems;
        }
        t.first[n] = blocks;
        t.n = n;
        TGRU_LAUNCH(red::reduce_jobs_kernel, dim3(blocks), dim3(red::kElems * red::kGroups), 0,
                           (hipStream_t)stream, t);
        if (hipGetLastError() != hipSuccess) return TGRU_E_LAUNCH;
    }
    return TGRU_OK;
}
