#include <hip/hip_runtime.h>
__global__ void k(const float* src, float* dst, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    // 4-byte and 16-byte direct-to-LDS loads
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem), 4, lane * 4, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 64), 16, lane < 8 ? lane * 16 : 0x7ffffff0, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    dst[lane] = smem[lane];
    for (int i = 0; i < 4; ++i) dst[64 + lane * 4 + i] = smem[64 + lane * 4 + i];
}
int main() {
    float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    hipMemset(d, 0xff, 4096);
    hipLaunchKernelGGL(k, 1, 64, 8192, 0, s, d, 256);
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("b32: %g %g %g ... %g\n", h[0], h[1], h[2], h[63]);
    printf("b128 lane0: %g %g %g %g lane1: %g %g lane7: %g %g lane8 (oob): %g %g\n", h[64], h[65], h[66], h[67], h[68], h[69], h[64+28], h[64+29], h[64+32], h[64+33]);
    return 0;
}
