// Weight and bias gradients of ALL layers of a small-H stack from the dg buffers the whole-stack
// backward kernel wrote:
//   dW_ih[m][n] = sum_k dg[k][m] * xin[k][n]                     m < 3H, k over all (t,b) rows
//   dW_hh[m][n] = sum_k dg[k][cm(m)] * h_{t-1}[k][n]             cm(m) = m < 2H ? m : m + H
// Stage 1 (grid = row chunks x layers): a workgroup streams its rows of dg / xin / h_prev through
// LDS with fully coalesced loads (every byte is fetched once per workgroup) and multiplies all
// output tiles of its layer on the MFMA; partial tiles go to a slab per chunk.
// Stage 2: sums the slabs (and the backward kernel's per-workgroup bias partials) per gradient
// element in a fixed order.
#include "tgru_internal.h"

namespace tgru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgRows = 64;      // rows staged per iteration
constexpr int kWgLdx = 48;       // LDS row stride of the xin / h_prev tiles (widths <= 32)

__host__ __device__ inline int wg_ldg(int H) { return ((4 * H + 31) / 32) * 32 + 16; }   // = 16 mod 32: conflict-free

__global__ __launch_bounds__(512) void gru_wgrad_partial_kernel(SmallWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B;
    const int l = blockIdx.y, chunk = blockIdx.x;
    const int In_l = (l == 0) ? a.In : H;
    const int K = a.T * B;
    const int ldg = wg_ldg(H);
    float *lds_dg = smem, *lds_x = smem + kWgRows * ldg, *lds_h = lds_x + kWgRows * kWgLdx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, gq = lane >> 4;
    const int rows_per_chunk = ((K + a.nchunks - 1) / a.nchunks + kWgRows - 1) / kWgRows * kWgRows;
    const int kbeg = chunk * rows_per_chunk, kend = min(K, kbeg + rows_per_chunk);
    const int Mt = (3 * H + 15) / 16, Nti = (In_l + 15) / 16, NtH = (H + 15) / 16;
    const int ntiles = Mt * (Nti + NtH);
    const float *dg = a.dg[l];
    const float *xsrc = (l == 0) ? a.x : a.y_int[l - 1];
    const float *hsrc = a.y_int[l];
    const bool mapped = l == 0 && a.xmap.mode != 0;

    // up to three output tiles per wave
    int kind[3], mt[3], nt[3], acol[3];
    bool have[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tl = wave + 8 * j;
        have[j] = tl < ntiles;
        const int ih = Mt * Nti;
        kind[j] = tl < ih ? 0 : 1;
        const int r = kind[j] ? tl - ih : tl;
        const int nn = kind[j] ? NtH : Nti;
        mt[j] = r / nn; nt[j] = r % nn;
        const int m = 16 * mt[j] + i;
        acol[j] = (kind[j] == 1 && m >= 2 * H) ? m + H : m;
        if (acol[j] >= ldg) acol[j] = ldg - 1;      // rows past 3H are discarded; keep the read in bounds
    }
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += kWgRows) {
        for (int idx = tid; idx < kWgRows * H; idx += 512) {       // dg: H float4 per row
            const int row = idx / H, c4 = idx - row * H;
            const int k = k0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) v = *reinterpret_cast<const float4 *>(dg + (long)k * 4 * H + c4 * 4);
            *reinterpret_cast<float4 *>(lds_dg + row * ldg + c4 * 4) = v;
        }
        for (int idx = tid; idx < kWgRows * In_l; idx += 512) {
            const int row = idx / In_l, c = idx - row * In_l;
            const int k = k0 + row;
            float v = 0.f;
            if (k < kend) {
                const long er = mapped ? map_row(a.xmap, k) : (long)k;
                if (er >= 0) v = xsrc[er * In_l + c];
            }
            lds_x[row * kWgLdx + c] = v;
        }
        for (int idx = tid; idx < kWgRows * H; idx += 512) {
            const int row = idx / H, c = idx - row * H;
            const int k = k0 + row;
            float v = 0.f;
            if (k < kend) {
                if (k >= B) v = hsrc[(long)(k - B) * H + c];
                else if (a.h0) v = a.h0[((long)l * B + k) * H + c];
            }
            lds_h[row * kWgLdx + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (have[j]) {
                const float *bsrc = (kind[j] ? lds_h : lds_x) + 16 * nt[j] + i;
                f32x4 c0 = acc[j], c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < kWgRows / 4; s4 += 2) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(lds_dg[(4 * s4 + gq) * ldg + acol[j]],
                                                              bsrc[(4 * s4 + gq) * kWgLdx], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(lds_dg[(4 * s4 + 4 + gq) * ldg + acol[j]],
                                                              bsrc[(4 * s4 + 4 + gq) * kWgLdx], c1, 0, 0, 0);
                }
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                acc[j] = (f32x4){c0[0] + c1[0], c0[1] + c1[1], c0[2] + c1[2], c0[3] + c1[3]};
            }
        }
        __syncthreads();
    }

    // partial tiles -> slab of this chunk, laid out like the gradients: per layer dW_ih then dW_hh
    long off = 0;
    for (int q = 0; q < l; ++q) off += 3L * H * ((q == 0 ? a.In : H) + H);
    float *slab = a.slabs + (long)chunk * a.slab_stride + off;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (!have[j]) continue;
        const int N = kind[j] ? H : In_l;
        float *out = slab + (kind[j] ? 3L * H * In_l : 0);
        const int n = 16 * nt[j] + i;          // D: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * mt[j] + 4 * gq + r;
            if (m < 3 * H && n < N) out[(long)m * N + n] = acc[j][r];
        }
    }
}

constexpr int kRedElems = 32;    // gradient elements per workgroup of the reduce kernel
constexpr int kRedGroups = 32;   // slab groups summed in parallel per element

// Stage 2: workgroup = 32 gradient elements x 32 groups of slabs.  Each thread sums its group's
// contiguous range of partials -- a handful, all loads in flight together, so the kernel costs about
// one memory round trip -- and the group sums are combined in a fixed order: the result does not
// depend on timing.
__global__ __launch_bounds__(kRedElems * kRedGroups) void gru_wgrad_reduce_kernel(VecPair<SmallWgradArgs> ga) {
    __shared__ float part[kRedGroups][kRedElems];
    const int set = (int)blockIdx.x >= ga.nwg0;
    const SmallWgradArgs &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x;
    const int H = a.H, L = a.L;
    const int e = threadIdx.x & (kRedElems - 1), grp = threadIdx.x / kRedElems;
    const long idx = (long)wg * kRedElems + e;
    const long nb = (long)L * 4 * H;
    const bool is_w = idx < a.slab_stride;
    const bool valid = idx < a.slab_stride + nb;
    // partial k of this element is at base[k * stride], k < count
    const float *base = is_w ? a.slabs + idx : a.db_part + (idx - a.slab_stride);
    const long stride = is_w ? a.slab_stride : nb;
    const int count = is_w ? a.nchunks : a.nblk;
    const int per = (count + kRedGroups - 1) / kRedGroups;
    const int c0 = grp * per, c1 = min(count, c0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (valid) {
        int c = c0;
        for (; c + 3 < c1; c += 4) {
            v0 += base[(long)c * stride];
            v1 += base[(long)(c + 1) * stride];
            v2 += base[(long)(c + 2) * stride];
            v3 += base[(long)(c + 3) * stride];
        }
        for (; c < c1; ++c) v0 += base[(long)c * stride];
    }
    part[grp][e] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (grp != 0 || !valid) return;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
    for (int g = 0; g < kRedGroups; g += 4) {
        u0 += part[g][e];
        u1 += part[g + 1][e];
        u2 += part[g + 2][e];
        u3 += part[g + 3][e];
    }
    const float v = (u0 + u1) + (u2 + u3);
    if (is_w) {
        long rem = idx;
        for (int l = 0; l < L; ++l) {
            const long n_ih = 3L * H * (l == 0 ? a.In : H), n_hh = 3L * H * H;
            if (rem < n_ih) { a.dw_ih[l][rem] = a.accumulate ? a.dw_ih[l][rem] + v : v; return; }
            rem -= n_ih;
            if (rem < n_hh) { a.dw_hh[l][rem] = a.accumulate ? a.dw_hh[l][rem] + v : v; return; }
            rem -= n_hh;
        }
        return;
    }
    const long bi = idx - a.slab_stride;
    const int l = (int)(bi / (4 * H)), i = (int)(bi % (4 * H));
    if (i < 3 * H) {
        a.db_ih[l][i] = a.accumulate ? a.db_ih[l][i] + v : v;
        if (i < 2 * H) a.db_hh[l][i] = a.accumulate ? a.db_hh[l][i] + v : v;
    } else {
        a.db_hh[l][i - H] = a.accumulate ? a.db_hh[l][i - H] + v : v;
    }
}

long small_wgrad_slab_floats(int In, int H, int L) {
    long n = 0;
    for (int l = 0; l < L; ++l) n += 3L * H * ((l == 0 ? In : H) + H);
    return n;
}

int small_wgrad_chunks(int B, int T) {
    long c = ((long)B * T + kWgRows - 1) / kWgRows;     // one staged tile per workgroup while that gives <= 64 chunks
    return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}

hipError_t launch_gru_small_wgrad(const SmallWgradArgs &a, hipStream_t s) {
    const size_t lds = (size_t)(kWgRows * wg_ldg(a.H) + 2 * kWgRows * kWgLdx) * sizeof(float);
    TGRU_LAUNCH(gru_wgrad_partial_kernel, dim3(a.nchunks, a.L), dim3(512), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_gru_small_wgrad_reduce(a, nullptr, s);
}

static int reduce_workgroups(const SmallWgradArgs &a) {
    const long items = a.slab_stride + (long)a.L * 4 * a.H;
    return (int)((items + kRedElems - 1) / kRedElems);
}

hipError_t launch_gru_small_wgrad_reduce(const SmallWgradArgs &a, const SmallWgradArgs *second, hipStream_t s) {
    VecPair<SmallWgradArgs> ga{};
    ga.g[0] = a;
    ga.nwg0 = reduce_workgroups(a);
    if (second) {
        ga.g[1] = *second;
        ga.nwg1 = reduce_workgroups(*second);
    }
    TGRU_LAUNCH(gru_wgrad_reduce_kernel, dim3((unsigned)(ga.nwg0 + ga.nwg1)), dim3(kRedElems * kRedGroups), 0,
                       s, ga);
    return hipGetLastError();
}

}  // namespace tgru
