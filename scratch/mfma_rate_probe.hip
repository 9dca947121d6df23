// How fast does one wave per SIMD issue v_mfma_f32_32x32x2_f32 with 2, 3, 4 independent accumulators, on a
// full chip (256 workgroups x 4 waves)?  Prints s_memtime ticks and wall-clock (100 MHz) time per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *tm, int n, float av, float bv) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { tm[0] = t1 - t0; tm[1] = w1 - w0; }
}
template <int NACC> void run(float *out, unsigned long long *tm, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, 256, 256, 0, 0, out, tm, n, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, 256, 256, 0, 0, out, tm, n, 1.0f, 0.5f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
    const double nm = (double)n * 4 * NACC;
    printf("NACC %d: %.1f s_memtime ticks / MFMA, %.2f ns / MFMA by wall_clock64 (100 MHz), %.2f ns / MFMA by events; "
           "=> %.1f TFLOP/s chip\n", NACC, h[0] / nm, h[1] * 10.0 / nm, ms * 1e6 / nm, 4096.0 * 1024 / (ms * 1e6 / nm) / 1e3);
}
int main() {
    float *out; unsigned long long *tm; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&tm, 16);
    run<2>(out, tm, 20000); run<3>(out, tm, 20000); run<4>(out, tm, 20000); run<8>(out, tm, 20000);
    return 0;
}
