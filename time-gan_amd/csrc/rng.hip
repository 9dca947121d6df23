// Counter-based uniform noise: the z = torch.rand_like(x) of the joint phase (outdated_files/TimeGANv1.py:762,793)
// for data-parallel runs.  torch's device generator hands its stream out by thread and launch, so a rank that
// owns rows [lo, hi) of a global batch cannot draw just those rows of the tensor a single process would draw; it
// had to draw the whole batch and keep a slice (eleven tensors of the global batch per iteration on every rank).
// Here element e of a stream is a pure function of (seed, stream, e) -- Philox4x32-10, counter = e / 4, the
// stream number in the counter's upper words -- so a rank generates exactly its rows, and the concatenation of all
// ranks' rows is bit for bit the tensor one process generates.  HBM-bound: 4 bytes written per element, ten
// rounds of two 32 x 32 -> 64 multiplies per four elements.
#include "../../include/tgru.h"
#include "tgru_internal.h"

namespace tgru {
namespace rng {

constexpr int kThreads = 256;

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// thread i serves the aligned group of four elements q = first / 4 + i; groups at the ends are partial
__global__ __launch_bounds__(kThreads) void uniform_kernel(float *__restrict__ out, long n, unsigned long long seed,
                                                           unsigned long long stream, unsigned long long first) {
    const unsigned long long q0 = first >> 2, nq = ((first + (unsigned long long)n + 3) >> 2) - q0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; i < nq;
         i += (unsigned long long)gridDim.x * kThreads) {
        const unsigned long long q = q0 + i;
        unsigned w[4];
        philox4x32_10((unsigned)q, (unsigned)(q >> 32), (unsigned)stream, (unsigned)(stream >> 32), (unsigned)seed,
                      (unsigned)(seed >> 32), w);
        const long base = (long)(4 * q - first);          // index in `out` of the group's first element
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)(w[j] >> 8) * (1.0f / 16777216.0f);
        if (base >= 0 && base + 4 <= n && ((reinterpret_cast<uintptr_t>(out + base) & 15) == 0)) {
            *reinterpret_cast<float4 *>(out + base) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (base + j >= 0 && base + j < n) out[base + j] = v[j];
        }
    }
}

}  // namespace rng
}  // namespace tgru

extern "C" int tgru_uniform(float *out, long n, unsigned long long seed, unsigned long long stream,
                            unsigned long long first, tgru_stream_t s) {
    if (n < 0) return TGRU_E_DIMS;
    if (n == 0) return TGRU_OK;
    if (!out) return TGRU_E_NULL;
    const long groups = (n + 3) / 4 + 1;
    long blocks = (groups + tgru::rng::kThreads - 1) / tgru::rng::kThreads;
    if (blocks > 8192) blocks = 8192;
    TGRU_LAUNCH(tgru::rng::uniform_kernel, dim3((unsigned)blocks), dim3(tgru::rng::kThreads), 0, (hipStream_t)s, out, n, seed,
                stream, first);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}
