// Weight gradients of one layer for mid-size hidden states (32 < H <= 128), both matrices in one pass:
//      dW_ih (3H x In) = dgi^T . X          dW_hh (3H x H) = dgh^T . H_prev
// contracted over all K = T*B rows.  dgi = dg[:, 0:3H], dgh = dg[:, 0:2H] | dg[:, 3H:4H].
//
// A workgroup owns a contiguous range of rows and ALL output tiles: wave ti holds the 32-row block
// ti of the 3H output rows against every 32-column block of [X | H_prev] (up to eight 32x32
// accumulators).  Rows are streamed through LDS in stages of 16 with direct-to-LDS loads, double
// buffered; every input byte is fetched once per workgroup (a tiled split-K GEMM re-reads dg once
// per column tile and X once per row tile).  Each workgroup leaves one partial [dW_ih | dW_hh];
// a second kernel sums the partials in a fixed order.
#include "tgru_internal.h"

namespace tgru {
namespace midw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kRows = 16;          // rows per stage
constexpr int kMaxNT = 6;          // 32-column blocks of [X | H_prev]: In + H <= 192
constexpr int kOOB = 0x7ffffff0;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_load_b128(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, off, 0, 0, 0);
}

}  // namespace midw

struct MidWgradArgs {
    const float *dg;               // (K, 4H) time-major
    const float *xin; int In;      // (K, In) rows; x_bt: caller tensor is (B,T,In) and row (t,b) is at b*T+t
    int x_bt;
    const float *y;                // (K, H) this layer's output, time-major: H_prev row k = y row k - B
    const float *h0;               // (B, H) or null: H_prev rows k < B
    float *slabs; long slab_stride;
    int rows_per_wg;               // multiple of kRows
    int B, T, H;
};

__global__ __launch_bounds__(576) void gru_mid_wgrad_kernel(MidWgradArgs a) {
    using namespace midw;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x ( dgs[16][4H] | xs[16][In] | hs[16][H] )
    const int H = a.H, In = a.In, B = a.B;
    const int K = a.T * B;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;                 // = ceil(3H / 32) row blocks
    const int ntx = (In + 31) / 32, nth = (H + 31) / 32;
    const int stage_floats = kRows * (4 * H + In + H);
    const int kbeg = blockIdx.x * a.rows_per_wg, kend = min(K, kbeg + a.rows_per_wg);
    const __amdgpu_buffer_rsrc_t rs_dg = make_rsrc(a.dg, (long)K * 4 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.xin, (long)K * In * 4);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.y, (long)K * H * 4);
    const __amdgpu_buffer_rsrc_t rs_h0 = make_rsrc(a.h0, (long)B * H * 4);

    // stage loader: 16-byte chunks of the stage, flattened dgs | xs | hs, dealt round-robin to the waves
    const int cdg = kRows * H, cx = kRows * In / 4, ch = kRows * H / 4;   // chunks per part
    const int ninstr = (cdg + cx + ch + 63) / 64;
    auto issue = [&](int k0, float *buf) {
        const bool first_rows = k0 < B;            // B is a multiple of kRows: a stage is all-initial-state or none
        for (int i = wave; i < ninstr; i += nwaves) {
            const int c = i * 64 + lane;
            if (c < cdg) {
                const int r = c / H, o = c - r * H;                       // row, 16-byte column
                lds_load_b128(rs_dg, buf + i * 256, (k0 + r < kend) ? ((k0 + r) * 4 * H + 4 * o) * 4 : kOOB);
            } else if (c < cdg + cx) {
                const int cc = c - cdg, r = cc / (In / 4), o = cc - r * (In / 4);
                const int k = k0 + r;
                int row = k;
                if (a.x_bt) { const int t = k / B, b = k - t * B; row = b * a.T + t; }
                lds_load_b128(rs_x, buf + i * 256, (k < kend) ? (row * In + 4 * o) * 4 : kOOB);
            } else {
                const int cc = c - cdg - cx, r = cc / (H / 4), o = cc - r * (H / 4);
                const int k = k0 + r;
                const bool ok = cc < ch && k < kend;
                if (first_rows) lds_load_b128(rs_h0, buf + i * 256, ok ? (k * H + 4 * o) * 4 : kOOB);
                else lds_load_b128(rs_y, buf + i * 256, ok ? ((k - B) * H + 4 * o) * 4 : kOOB);
            }
        }
    };
    // NOTE: a 64-lane instruction must not straddle two parts (different descriptors): the parts are
    // padded to whole instructions by the host-side choice of sizes (H, In multiples of 16)

    // ---- this wave's output rows and its lane constants --------------------------------------------
    const int i32 = lane & 31, kk = lane >> 5;
    const int m = wave * 32 + i32;                       // output row (gate-major, 0..3H)
    const bool mrow = m < 3 * H;
    const int acol_x = mrow ? m : 0;
    const int acol_h = mrow ? (m < 2 * H ? m : m + H) : 0;
    const float amask = mrow ? 1.f : 0.f;
    f32x16 acc[kMaxNT];
    // per column block q of [X | H_prev]: where its element of a stage row lives, and which A operand it pairs with
    int boff[kMaxNT], bstr[kMaxNT];
    float bmask[kMaxNT];
    bool useh[kMaxNT];
#pragma unroll
    for (int q = 0; q < kMaxNT; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        const bool isx = q < ntx;
        const int c = (isx ? q : q - ntx) * 32 + i32;
        const int width = isx ? In : H;
        const bool ok = q < ntx + nth && c < width;
        useh[q] = !isx;
        bstr[q] = width;
        boff[q] = (isx ? kRows * 4 * H : kRows * (4 * H + In)) + (ok ? c : 0);
        bmask[q] = ok ? 1.f : 0.f;
    }

    int cur = 0;
    if (kbeg < kend) issue(kbeg, smem);
    for (int k0 = kbeg; k0 < kend; k0 += kRows) {
        wait_vm0();
        lds_barrier();                                   // stage `cur` has landed for every wave
        if (k0 + kRows < kend) issue(k0 + kRows, smem + (cur ^ 1) * stage_floats);
        const float *st = smem + cur * stage_floats;
#pragma unroll
        for (int s2 = 0; s2 < kRows / 2; ++s2) {
            const int r = 2 * s2 + kk;
            const float ax = st[r * 4 * H + acol_x] * amask;
            const float ah = st[r * 4 * H + acol_h] * amask;
#pragma unroll
            for (int q = 0; q < kMaxNT; ++q) {
                const float b = st[boff[q] + r * bstr[q]] * bmask[q];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(useh[q] ? ah : ax, b, acc[q], 0, 0, 0);
            }
        }
        lds_barrier();                                   // everyone is done reading stage `cur`
        cur ^= 1;
    }
    mfma_drain();

    // ---- partial [dW_ih | dW_hh] of this workgroup: register r of lane (hh, j) is row 8*(r/4)+r%4+4*hh
    float *slab = a.slabs + (long)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int q = 0; q < kMaxNT; ++q) {
        if (q >= ntx + nth) break;
        const bool isx = q < ntx;
        const int N = isx ? In : H;
        const int c = (isx ? q : q - ntx) * 32 + i32;
        float *dst = slab + (isx ? 0 : (long)3 * H * In);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + 8 * (r / 4) + (r % 4) + 4 * kk;
            if (row < 3 * H && c < N) dst[(long)row * N + c] = acc[q][r];
        }
    }
}

// out_a[i] / out_b[i] (=|+=) sum over slabs: slab = [a (na) | b (nb)]; 32 elements x 8 slab groups per workgroup
__global__ __launch_bounds__(256) void mid_wgrad_reduce_kernel(const float *slabs, long stride, int nslabs,
                                                               float *out_a, long na, float *out_b, long nb,
                                                               int accumulate) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + e;
    const int per = (nslabs + 7) / 8;
    const int c0 = grp * per, c1 = min(nslabs, c0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (idx < na + nb) {
        int c = c0;
        for (; c + 3 < c1; c += 4) {
            v0 += slabs[(long)c * stride + idx];
            v1 += slabs[(long)(c + 1) * stride + idx];
            v2 += slabs[(long)(c + 2) * stride + idx];
            v3 += slabs[(long)(c + 3) * stride + idx];
        }
        for (; c < c1; ++c) v0 += slabs[(long)c * stride + idx];
    }
    part[grp][e] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (grp != 0 || idx >= na + nb) return;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][e];
    float *o = idx < na ? out_a + idx : out_b + (idx - na);
    *o = accumulate ? *o + v : v;
}

// Shapes the kernel covers: the parts of a stage must fill whole 64-lane load instructions
// (H and In multiples of 16), a stage must not mix initial-state rows with later ones (B multiple of
// 16), byte offsets must fit 31 bits.
bool gru_mid_wgrad_supported(int B, int T, int In, int H) {
    if (!gru_mid_supported(H) || H > 96 || In > 96 || (H & 15) || (In & 15) || (B & 15)) return false;   // 9 waves x 6 accumulator tiles
    const long widest = 4L * H > In ? 4L * H : In;
    return (long)T * B * widest * 4 < (1L << 31);
}
int gru_mid_wgrad_workgroups(int B, int T) {
    const int stages = (B * T + midw::kRows - 1) / midw::kRows;
    return stages < 256 ? stages : 256;      // one workgroup per CU measured best (128: +12 %, 512: +8 %)
}
long gru_mid_wgrad_slab_floats(int In, int H) { return 3L * H * (In + H); }

hipError_t launch_gru_mid_wgrad(const float *dg, const float *xin, int In, int x_bt, const float *y,
                                const float *h0, float *slabs, float *dw_ih, float *dw_hh, int B, int T, int H,
                                int accumulate, hipStream_t s) {
    MidWgradArgs a{};
    a.dg = dg; a.xin = xin; a.In = In; a.x_bt = x_bt; a.y = y; a.h0 = h0; a.slabs = slabs;
    a.slab_stride = gru_mid_wgrad_slab_floats(In, H);
    a.B = B; a.T = T; a.H = H;
    const int nwg = gru_mid_wgrad_workgroups(B, T);
    const int K = B * T;
    a.rows_per_wg = ((K + nwg - 1) / nwg + midw::kRows - 1) / midw::kRows * midw::kRows;
    const int nwaves = (3 * H + 31) / 32;
    const size_t lds = (size_t)2 * midw::kRows * (4 * H + In + H) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_mid_wgrad_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int used = (K + a.rows_per_wg - 1) / a.rows_per_wg;     // workgroups that own rows
    TGRU_LAUNCH(gru_mid_wgrad_kernel, dim3(used), dim3(64 * nwaves), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long na = 3L * H * In, nb = 3L * H * H;
    TGRU_LAUNCH(mid_wgrad_reduce_kernel, dim3((unsigned)((na + nb + 31) / 32)), dim3(256), 0, s, slabs,
                       a.slab_stride, used, dw_ih, na, dw_hh, nb, accumulate);
    return hipGetLastError();
}

}  // namespace tgru
