// Issue rate of v_mfma_f32_4x4x1_16b_f32 for one wave per SIMD with 1, 2, 4, 8 independent accumulators
// (256 workgroups x 4 waves).  Prints ns per instruction by events.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int n, float av, float bv) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC> void run(float *out, int n, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, 256, 64 * waves, 0, 0, out, n, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, 256, 64 * waves, 0, 0, out, n, 1.0f, 0.5f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)n * 8 * NACC;
    printf("waves/WG %d NACC %d: %.2f ns per instruction per wave (%.1f cycles at 2.4 GHz)\n", waves, NACC, ms * 1e6 / nm, ms * 1e6 / nm * 2.4);
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<1>(out, 20000, 4); run<2>(out, 20000, 4); run<4>(out, 20000, 4); run<8>(out, 20000, 4);
    run<2>(out, 20000, 8); run<4>(out, 20000, 8);
    return 0;
}
