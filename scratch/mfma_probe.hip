// prints the operand/result layout of v_mfma_f32_32x32x1_2b_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v32f __attribute__((ext_vector_type(32)));
__global__ void k(float *outA, float *outB) {
    const int lane = threadIdx.x;
    v32f c = {0};
    v32f da = __builtin_amdgcn_mfma_f32_32x32x1f32((float)(lane + 1), 1.f, c, 0, 0, 0);
    v32f db = __builtin_amdgcn_mfma_f32_32x32x1f32(1.f, (float)(lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 32; ++r) { outA[lane * 32 + r] = da[r]; outB[lane * 32 + r] = db[r]; }
}
int main() {
    float *a, *b; hipMalloc(&a, 64 * 32 * 4); hipMalloc(&b, 64 * 32 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b);
    static float ha[2048], hb[2048];
    hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost); hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
    for (int lane : {0, 1, 31, 32, 33, 63}) {
        printf("lane %d: A-lane per reg:", lane);
        for (int r = 0; r < 32; ++r) printf(" %d", (int)ha[lane * 32 + r] - 1);
        printf("\nlane %d: B-lane per reg:", lane);
        for (int r = 0; r < 32; ++r) printf(" %d", (int)hb[lane * 32 + r] - 1);
        printf("\n");
    }
    return 0;
}
