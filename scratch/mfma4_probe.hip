// layout of v_mfma_f32_4x4x1_16b_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *outA, float *outB) {
    const int lane = threadIdx.x;
    v4f c = {0, 0, 0, 0};
    v4f da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.f, c, 0, 0, 0);
    v4f db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, (float)(lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { outA[lane * 4 + r] = da[r]; outB[lane * 4 + r] = db[r]; }
}
int main() {
    float *a, *b; (void)hipMalloc(&a, 1024); (void)hipMalloc(&b, 1024);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b);
    static float ha[256], hb[256];
    (void)hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost); (void)hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
    for (int lane : {0, 1, 2, 3, 4, 5, 17, 63}) {
        printf("lane %2d: A-lane per reg:", lane);
        for (int r = 0; r < 4; ++r) printf(" %d", (int)ha[lane * 4 + r] - 1);
        printf("   B-lane per reg:");
        for (int r = 0; r < 4; ++r) printf(" %d", (int)hb[lane * 4 + r] - 1);
        printf("\n");
    }
    return 0;
}
