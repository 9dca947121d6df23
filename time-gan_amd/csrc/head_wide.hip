// Fused per-row head of the TimeGAN networks at widths beyond head.hip's register-resident weights
// (outdated_files/TimeGANv1.py:39-50,55,67-68,119-128,145-148,226-227,276-287,294,305-306,382-383 at
// dim_hidden = 96 and the like):
//
//      y = [ LayerNorm0 ] -> Linear1 -> [ LayerNorm1 -> ReLU -> Linear2 ]
//
// input / output up to 128 wide, the middle up to 192, forward and backward one kernel each plus the two
// weight-gradient products.
//
// The register scheme is head.hip's: a wave owns 16 rows, row j lives in the four lanes (g, j), which hold
// features 16*ft + 4*g + r of every 16-feature tile ft, and a Linear is the transposed product
// Y^T = W . X^T on v_mfma_f32_16x16x4_f32 with W as the A operand, so that the result registers of one
// Linear are the B operands of the next and LayerNorm / ReLU run in registers between them.  What differs
// is where the weights live: both matrices sit in LDS, rows padded to a stride of 16*m + 4 floats, which
// makes the 16-byte fragment reads of W (forward products) and the 4-byte fragment reads of W^T (backward
// products) both bank-conflict free.  Four waves share a workgroup's copy.
//
// Backward recomputes the forward from x, writes dx, and leaves the operands of the two weight-gradient
// products (LayerNorm0 output, ReLU output, gradient at the Linear1 output) in a workspace: a 96 x 96
// gradient over tens of thousands of rows is a product to be split over the ROWS, which the split-K GEMM
// of gemm.hip does; 2 x 144 accumulator registers per wave would not fit beside the row state.  Bias and
// LayerNorm-parameter gradients leave the kernel as one partial vector per workgroup; all partials are
// summed in a fixed order (deterministic, no atomics).
#include "../../include/tgru.h"
#include "tgru_internal.h"

namespace tgru {
namespace headw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;                 // rows per wave
constexpr int kWaves = 4;                 // waves per workgroup
constexpr int kThreads = 64 * kWaves;
constexpr int kMaxIo = 128, kMaxMid = 192;
constexpr int kMaxLdsBytes = 150 * 1024;

__host__ __device__ inline int rup16(int n) { return (n + 15) & ~15; }

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// MFMA results read behind a branch: the compiler pads that hazard only inside a basic block
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

// sum of v over the four lanes that share a row; the same bits in all four
__device__ __forceinline__ float row_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

struct Args {
    const float *x; float *y;                 // forward
    const float *dy; float *dx;               // backward (dx may be null)
    const float *g0, *b0;                     // LayerNorm0 (null: absent)
    const float *w1, *c1;                     // Linear1 (dmid x din), bias
    const float *g1, *b1, *w2, *c2;           // LayerNorm1, Linear2 (dout x dmid), bias (w2 null: shallow)
    float *xn, *act, *dh;                     // backward: operands of the weight-gradient products
    float *slabs; long slab_stride;           // backward: one partial vector per workgroup (VecLayout)
    float *zero; int nzero;                   // backward, shallow head: the absent stages' gradient entries
    int M, din, dmid, dout;
    int vec_in, vec_out;                      // rows of x/dx (y/dy) may be moved as aligned float4
    float eps;
};

// LDS image of the parameters (float offsets): both matrices zero-padded to whole tiles, rows of
// 16*m + 4 floats; the vectors zero-padded to whole tiles
struct Lds { int w1, ld1, w2, ld2, c1, g1, b1, g0, b0, c2, total; };
__host__ __device__ inline Lds lds_layout(int din, int dmid, int dout, bool deep) {
    Lds l;
    int o = 0;
    l.ld1 = rup16(din) + 4; l.w1 = o; o += rup16(dmid) * l.ld1;
    l.ld2 = rup16(dmid) + 4; l.w2 = o; if (deep) o += rup16(dout) * l.ld2;
    l.c1 = o; o += rup16(dmid);
    l.g1 = o; o += rup16(dmid);
    l.b1 = o; o += rup16(dmid);
    l.g0 = o; o += rup16(din);
    l.b0 = o; o += rup16(din);
    l.c2 = o; o += rup16(dout);
    l.total = o;
    return l;
}

// per-workgroup partial vector: db1 | dln0_w | dln0_b | dln1_w | dln1_b | db2 (the order of the flat
// gradient buffer, so that the first five are one contiguous reduction)
struct VecLayout { int c1, g0, b0, g1, b1, c2, total; };
__host__ __device__ inline VecLayout vec_layout(int din, int dmid, int dout) {
    VecLayout v;
    int o = 0;
    v.c1 = o; o += dmid;
    v.g0 = o; o += din;
    v.b0 = o; o += din;
    v.g1 = o; o += dmid;
    v.b1 = o; o += dmid;
    v.c2 = o; o += dout;
    v.total = o;
    return v;
}

// ---- parameters -> LDS ------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_matrix(float *dst, int ld, const float *w, int rows, int cols, int rows_p) {
    const int n4 = ld >> 2;
    const bool vec = (cols & 3) == 0;
    for (int idx = threadIdx.x; idx < rows_p * n4; idx += blockDim.x) {
        const int r = idx / n4, c = 4 * (idx - r * n4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            const float *p = w + (long)r * cols + c;
            if (vec) {
                if (c < cols) v = *(const float4 *)p;
            } else {
                if (c < cols) v.x = p[0];
                if (c + 1 < cols) v.y = p[1];
                if (c + 2 < cols) v.z = p[2];
                if (c + 3 < cols) v.w = p[3];
            }
        }
        *(float4 *)(dst + r * ld + c) = v;
    }
}
__device__ __forceinline__ void stage_vec(float *dst, const float *p, int n, int n_p) {
    for (int i = threadIdx.x; i < n_p; i += blockDim.x) dst[i] = (p && i < n) ? p[i] : 0.f;
}
__device__ __forceinline__ void stage_params(const Args &a, const Lds &l, float *lds, bool deep) {
    stage_matrix(lds + l.w1, l.ld1, a.w1, a.dmid, a.din, rup16(a.dmid));
    if (deep) stage_matrix(lds + l.w2, l.ld2, a.w2, a.dout, a.dmid, rup16(a.dout));
    stage_vec(lds + l.c1, a.c1, a.dmid, rup16(a.dmid));
    stage_vec(lds + l.g1, a.g1, a.dmid, rup16(a.dmid));
    stage_vec(lds + l.b1, a.b1, a.dmid, rup16(a.dmid));
    stage_vec(lds + l.g0, a.g0, a.din, rup16(a.din));
    stage_vec(lds + l.b0, a.b0, a.din, rup16(a.din));
    stage_vec(lds + l.c2, a.c2, a.dout, rup16(a.dout));
}

// ---- registers <-> memory -----------------------------------------------------------------------------------
__device__ __forceinline__ bool live(int q, int g, int n) { return 16 * (q >> 2) + 4 * g + (q & 3) < n; }

// v[4*ft + r] = p[16*ft + 4*g + r] from a zero-padded LDS vector; tiles at or beyond n are zeros
template <int NT>
__device__ __forceinline__ void lds_vec(const float *p, int n, int g, float (&v)[4 * NT]) {
#pragma unroll
    for (int ft = 0; ft < NT; ++ft) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (16 * ft < n) t = *(const float4 *)(p + 16 * ft + 4 * g);
        v[4 * ft + 0] = t.x; v[4 * ft + 1] = t.y; v[4 * ft + 2] = t.z; v[4 * ft + 3] = t.w;
    }
}
// one row per lane quartet: v = p[row*d + feature], zeros for absent rows / features
template <int NT>
__device__ __forceinline__ void load_rows(const float *p, int M, int d, long row, int g, bool vec, float (&v)[4 * NT]) {
    const bool ok = row < M;
#pragma unroll
    for (int ft = 0; ft < NT; ++ft) {
        const int f0 = 16 * ft + 4 * g;
        if (vec) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && f0 < d) t = *(const float4 *)(p + row * d + f0);
            v[4 * ft + 0] = t.x; v[4 * ft + 1] = t.y; v[4 * ft + 2] = t.z; v[4 * ft + 3] = t.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * ft + r] = (ok && f0 + r < d) ? p[row * d + f0 + r] : 0.f;
        }
    }
}
template <int NT>
__device__ __forceinline__ void store_rows(float *p, int M, int d, long row, int g, bool vec, const float (&v)[4 * NT]) {
    if (row >= M) return;
#pragma unroll
    for (int ft = 0; ft < NT; ++ft) {
        const int f0 = 16 * ft + 4 * g;
        if (vec) {
            if (f0 < d) *(float4 *)(p + row * d + f0) = make_float4(v[4 * ft], v[4 * ft + 1], v[4 * ft + 2], v[4 * ft + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (f0 + r < d) p[row * d + f0 + r] = v[4 * ft + r];
        }
    }
}

// ---- Linear layers on LDS-resident weights ------------------------------------------------------------------
// out = W in, W (rows x cols) in LDS with row stride ld: `in` covers NI tiles of the cols, `out` NO tiles of
// the rows.  A operand of step (ft, r), output tile ot: lane (kg, i) -> W[16*ot + i][16*ft + 4*kg + r], four
// steps per 16-byte read.
template <int NI, int NO>
__device__ __forceinline__ void apply_w(const float *w, int ld, int nout, int nin, int lane, const float (&in)[4 * NI],
                                        float (&out)[4 * NO]) {
    const float *base = w + (lane & 15) * ld + 4 * (lane >> 4);
#pragma unroll
    for (int ot = 0; ot < NO; ++ot) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (16 * ot < nout) {
#pragma unroll
            for (int ft = 0; ft < NI; ++ft)
                if (16 * ft < nin) {
                    const float4 f = *(const float4 *)(base + 16 * ot * ld + 16 * ft);
                    acc = mfma(f.x, in[4 * ft + 0], acc);
                    acc = mfma(f.y, in[4 * ft + 1], acc);
                    acc = mfma(f.z, in[4 * ft + 2], acc);
                    acc = mfma(f.w, in[4 * ft + 3], acc);
                }
        }
        mfma_drain();
#pragma unroll
        for (int r = 0; r < 4; ++r) out[4 * ot + r] = acc[r];
    }
}
// out = W^T in, W (rows x cols) in LDS: `in` covers NI tiles of the rows, `out` NO tiles of the cols.
// A operand of step (ft, r), output tile kt: lane (ig, k) -> W[16*ft + 4*ig + r][16*kt + k].
template <int NI, int NO>
__device__ __forceinline__ void apply_wt(const float *w, int ld, int nout, int nin, int lane, const float (&in)[4 * NI],
                                         float (&out)[4 * NO]) {
    const float *base = w + 4 * (lane >> 4) * ld + (lane & 15);
#pragma unroll
    for (int kt = 0; kt < NO; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (16 * kt < nout) {
#pragma unroll
            for (int ft = 0; ft < NI; ++ft)
                if (16 * ft < nin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma(base[(16 * ft + r) * ld + 16 * kt], in[4 * ft + r], acc);
                }
        }
        mfma_drain();
#pragma unroll
        for (int r = 0; r < 4; ++r) out[4 * kt + r] = acc[r];
    }
}

// LayerNorm statistics of one row spread over its four lanes; entries beyond d are zero on entry
template <int NT>
__device__ __forceinline__ void ln_stats(const float (&v)[4 * NT], int g, int d, float eps, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) s += v[q];
    mean = row_sum(s) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
        const float c = live(q, g, d) ? v[q] - mean : 0.f;
        sq += c * c;
    }
    rstd = rsqrtf(row_sum(sq) / (float)d + eps);
}

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
// NW waves per workgroup: two per SIMD, so that one wave's row loads and LayerNorm chains hide under the
// other's products
template <int NI, int NM, int NO, int NW>
__global__ __launch_bounds__(64 * NW) void head_wide_fwd_kernel(Args a) {
    extern __shared__ __align__(16) float lds[];
    const bool deep = a.w2 != nullptr;
    const Lds l = lds_layout(a.din, a.dmid, a.dout, deep);
    stage_params(a, l, lds, deep);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int ntiles = (a.M + kRows - 1) / kRows;
    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
        const long row = (long)tile * kRows + j;
        float x[4 * NI];
        load_rows<NI>(a.x, a.M, a.din, row, g, a.vec_in, x);
        if (a.g0) {
            float g0[4 * NI], b0[4 * NI], mean, rstd;
            lds_vec<NI>(lds + l.g0, a.din, g, g0);
            lds_vec<NI>(lds + l.b0, a.din, g, b0);
            ln_stats<NI>(x, g, a.din, a.eps, mean, rstd);
#pragma unroll
            for (int q = 0; q < 4 * NI; ++q) x[q] = (x[q] - mean) * rstd * g0[q] + b0[q];   // padding: 0 * 0 + 0
        }
        float h[4 * NM];
        apply_w<NI, NM>(lds + l.w1, l.ld1, a.dmid, a.din, lane, x, h);
        {
            float c1[4 * NM];
            lds_vec<NM>(lds + l.c1, a.dmid, g, c1);
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) h[q] += c1[q];
        }
        if (deep) {
            float g1[4 * NM], b1[4 * NM], mean, rstd;
            lds_vec<NM>(lds + l.g1, a.dmid, g, g1);
            lds_vec<NM>(lds + l.b1, a.dmid, g, b1);
            ln_stats<NM>(h, g, a.dmid, a.eps, mean, rstd);
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) h[q] = fmaxf((h[q] - mean) * rstd * g1[q] + b1[q], 0.f);
            float y[4 * NO], c2[4 * NO];
            apply_w<NM, NO>(lds + l.w2, l.ld2, a.dout, a.dmid, lane, h, y);
            lds_vec<NO>(lds + l.c2, a.dout, g, c2);
#pragma unroll
            for (int q = 0; q < 4 * NO; ++q) y[q] += c2[q];
            store_rows<NO>(a.y, a.M, a.dout, row, g, a.vec_out, y);
        } else {
            store_rows<NM>(a.y, a.M, a.dmid, row, g, a.vec_out, h);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
// Column sums without an accumulator register per feature.  Lane (g, j) holds, for row j, N values (one per
// feature slot q); what is wanted is the sum of slot q over the 16 lanes j of the row group.  A butterfly that
// halves the slots while it doubles the lanes summed (keep one slot of a pair, hand the other to the partner
// lane) leaves the total of slot q in lane j = q: 16 slots cost 15 adds, 30 selects and 15 lane exchanges,
// and a whole accumulator set packs into N / 16 registers.
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float swz_xor4(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }
__device__ __forceinline__ float swz_xor8(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x201F)); }

struct LaneBits { bool b0, b1, b2, b3; };

// lane j of every 16-lane row gets the sum of v[j] over the row's lanes
__device__ __forceinline__ float reduce16(const float (&v)[16], const LaneBits &lb) {
    float t1[8], t2[4], t3[2];
#pragma unroll
    for (int p = 0; p < 8; ++p) t1[p] = (lb.b0 ? v[2 * p + 1] : v[2 * p]) + dpp_xor1(lb.b0 ? v[2 * p] : v[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) t2[p] = (lb.b1 ? t1[2 * p + 1] : t1[2 * p]) + dpp_xor2(lb.b1 ? t1[2 * p] : t1[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 2; ++p) t3[p] = (lb.b2 ? t2[2 * p + 1] : t2[2 * p]) + swz_xor4(lb.b2 ? t2[2 * p] : t2[2 * p + 1]);
    return (lb.b3 ? t3[1] : t3[0]) + swz_xor8(lb.b3 ? t3[0] : t3[1]);
}
// P[grp] (lane j) += column total of slot 16*grp + j
template <int N>
__device__ __forceinline__ void add_columns(const float (&v)[N], float (&P)[(N + 15) / 16], const LaneBits &lb) {
#pragma unroll
    for (int grp = 0; grp < (N + 15) / 16; ++grp) {
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = (16 * grp + i < N) ? v[16 * grp + i] : 0.f;
        P[grp] += reduce16(t, lb);
    }
}
// packed totals -> this wave's vector in LDS: slot q = 4*ft + r of lane group g is feature 16*ft + 4*g + r
template <int N>
__device__ __forceinline__ void put_columns(float *dst, const float (&P)[(N + 15) / 16], int g, int j) {
#pragma unroll
    for (int grp = 0; grp < (N + 15) / 16; ++grp) {
        const int q = 16 * grp + j;
        if (q < N) dst[16 * (q >> 2) + 4 * g + (q & 3)] = P[grp];
    }
}

// NW waves per workgroup (eight where the registers allow two waves per SIMD)
template <int NI, int NM, int NO, int NW>
__global__ __launch_bounds__(64 * NW) void head_wide_bwd_kernel(Args a) {
    extern __shared__ __align__(16) float lds[];
    const bool deep = a.w2 != nullptr;
    const Lds l = lds_layout(a.din, a.dmid, a.dout, deep);
    stage_params(a, l, lds, deep);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const LaneBits lb{(j & 1) != 0, (j & 2) != 0, (j & 4) != 0, (j & 8) != 0};
    const bool vec_mid = (a.dmid & 3) == 0;      // workspace rows of dmid floats start 16-byte aligned then
    constexpr int PM = (4 * NM + 15) / 16, PI = (4 * NI + 15) / 16, PO = (4 * NO + 15) / 16;
    float pc1[PM], pg1[PM], pb1[PM], pg0[PI], pb0[PI], pc2[PO];     // packed column totals, see add_columns
#pragma unroll
    for (int q = 0; q < PM; ++q) pc1[q] = pg1[q] = pb1[q] = 0.f;
#pragma unroll
    for (int q = 0; q < PI; ++q) pg0[q] = pb0[q] = 0.f;
#pragma unroll
    for (int q = 0; q < PO; ++q) pc2[q] = 0.f;

    const int ntiles = (a.M + kRows - 1) / kRows;
    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
        const long row = (long)tile * kRows + j;
        // ---- forward again, keeping what the backward needs --------------------------------------
        float x1[4 * NI], xh0[4 * NI], rstd0 = 1.f;
        load_rows<NI>(a.x, a.M, a.din, row, g, a.vec_in, x1);
        if (a.g0) {
            float g0[4 * NI], b0[4 * NI], mean;
            lds_vec<NI>(lds + l.g0, a.din, g, g0);
            lds_vec<NI>(lds + l.b0, a.din, g, b0);
            ln_stats<NI>(x1, g, a.din, a.eps, mean, rstd0);
#pragma unroll
            for (int q = 0; q < 4 * NI; ++q) {
                xh0[q] = live(q, g, a.din) ? (x1[q] - mean) * rstd0 : 0.f;
                x1[q] = xh0[q] * g0[q] + b0[q];
            }
            if (a.xn) store_rows<NI>(a.xn, a.M, a.din, row, g, (a.din & 3) == 0, x1);
        }
        float dh[4 * NM];             // gradient at the output of Linear1
        if (deep) {
            float hn[4 * NM], g1[4 * NM], b1[4 * NM], rstd1;
            apply_w<NI, NM>(lds + l.w1, l.ld1, a.dmid, a.din, lane, x1, hn);
            {
                float c1[4 * NM], mean1;
                lds_vec<NM>(lds + l.c1, a.dmid, g, c1);
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) hn[q] += c1[q];
                ln_stats<NM>(hn, g, a.dmid, a.eps, mean1, rstd1);
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) hn[q] = live(q, g, a.dmid) ? (hn[q] - mean1) * rstd1 : 0.f;
            }
            lds_vec<NM>(lds + l.g1, a.dmid, g, g1);
            lds_vec<NM>(lds + l.b1, a.dmid, g, b1);
            {
                float act[4 * NM];
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) act[q] = fmaxf(hn[q] * g1[q] + b1[q], 0.f);
                if (a.act) store_rows<NM>(a.act, a.M, a.dmid, row, g, vec_mid, act);
            }
            // Linear2: dc2, da = W2^T . dy   (dW2 = dy^T . act is the caller's product)
            float da[4 * NM];
            {
                float dy[4 * NO];
                load_rows<NO>(a.dy, a.M, a.dout, row, g, a.vec_out, dy);
                apply_wt<NO, NM>(lds + l.w2, l.ld2, a.dmid, a.dout, lane, dy, da);
                add_columns<4 * NO>(dy, pc2, lb);
            }
            // ReLU and LayerNorm1
            float s1 = 0.f, s2 = 0.f;
            {
                float tg[4 * NM];
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) {
                    da[q] = (hn[q] * g1[q] + b1[q] > 0.f) ? da[q] : 0.f;
                    tg[q] = da[q] * hn[q];
                }
                add_columns<4 * NM>(da, pb1, lb);
                add_columns<4 * NM>(tg, pg1, lb);
            }
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) {
                da[q] *= g1[q];
                s1 += da[q];
                s2 += da[q] * hn[q];
            }
            const float m1 = row_sum(s1) / (float)a.dmid, m2 = row_sum(s2) / (float)a.dmid;
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) dh[q] = live(q, g, a.dmid) ? rstd1 * (da[q] - m1 - hn[q] * m2) : 0.f;
            if (a.dh) store_rows<NM>(a.dh, a.M, a.dmid, row, g, vec_mid, dh);
        } else {
            load_rows<NM>(a.dy, a.M, a.dmid, row, g, a.vec_out, dh);
        }
        // ---- Linear1: dc1, dx1 = W1^T . dh   (dW1 = dh^T . x1 is the caller's product) ----------------
        if (a.dx || a.g0) {
            float dx[4 * NI];
            apply_wt<NM, NI>(lds + l.w1, l.ld1, a.din, a.dmid, lane, dh, dx);
            add_columns<4 * NM>(dh, pc1, lb);
            if (a.g0) {
                float g0[4 * NI];
                lds_vec<NI>(lds + l.g0, a.din, g, g0);
                {
                    float tg[4 * NI];
#pragma unroll
                    for (int q = 0; q < 4 * NI; ++q) tg[q] = dx[q] * xh0[q];
                    add_columns<4 * NI>(dx, pb0, lb);
                    add_columns<4 * NI>(tg, pg0, lb);
                }
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 4 * NI; ++q) {
                    dx[q] *= g0[q];
                    s1 += dx[q];
                    s2 += dx[q] * xh0[q];
                }
                const float m1 = row_sum(s1) / (float)a.din, m2 = row_sum(s2) / (float)a.din;
#pragma unroll
                for (int q = 0; q < 4 * NI; ++q) dx[q] = live(q, g, a.din) ? rstd0 * (dx[q] - m1 - xh0[q] * m2) : 0.f;
            }
            if (a.dx) store_rows<NI>(a.dx, a.M, a.din, row, g, a.vec_in, dx);
        } else {
            add_columns<4 * NM>(dh, pc1, lb);
        }
    }

    // ---- this workgroup's partial bias / LayerNorm-parameter gradients ---------------------------------
    // the weights in LDS are done with: the same memory carries one vector of totals per wave
    const VecLayout vl = vec_layout(a.din, a.dmid, a.dout);
    float *slab = a.slabs + (long)blockIdx.x * a.slab_stride;
    constexpr int oc1 = 0, og0 = 16 * NM, ob0 = og0 + 16 * NI, og1 = ob0 + 16 * NI, ob1 = og1 + 16 * NM,
                  oc2 = ob1 + 16 * NM, kStride = oc2 + 16 * NO;
    __syncthreads();
    float *mine = lds + wave * kStride;
    put_columns<4 * NM>(mine + oc1, pc1, g, j);
    put_columns<4 * NI>(mine + og0, pg0, g, j);
    put_columns<4 * NI>(mine + ob0, pb0, g, j);
    put_columns<4 * NM>(mine + og1, pg1, g, j);
    put_columns<4 * NM>(mine + ob1, pb1, g, j);
    put_columns<4 * NO>(mine + oc2, pc2, g, j);
    __syncthreads();
    const int src[6] = {oc1, og0, ob0, og1, ob1, oc2};
    const int dst[6] = {vl.c1, vl.g0, vl.b0, vl.g1, vl.b1, vl.c2};
    const int len[6] = {a.dmid, a.din, a.din, a.dmid, a.dmid, deep ? a.dout : 0};
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (int f = threadIdx.x; f < len[k]; f += blockDim.x) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += lds[w * kStride + src[k] + f];
            slab[dst[k] + f] = sum;
        }
    if (a.zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.nzero; i += blockDim.x) a.zero[i] = 0.f;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NI, int NM, int NO>
size_t lds_bytes(int din, int dmid, int dout, bool deep, bool backward) {
    size_t b = (size_t)lds_layout(din, dmid, dout, deep).total * sizeof(float);
    if (backward) {      // one vector of column totals per wave, in the weights' place
        const size_t r = (size_t)8 * 16 * (3 * NM + 2 * NI + NO) * sizeof(float);
        if (r > b) b = r;
    }
    return b;
}

constexpr int kFwdWaves = 8;
int workgroups(int M, int waves) {
    const int groups = ((M + kRows - 1) / kRows + waves - 1) / waves;
    return groups < 256 ? groups : 256;
}
// waves per workgroup of the backward kernel: two per SIMD where its registers allow (every width <= 96)
int bwd_waves(int din, int dmid, int dout) { return ((din <= 96 && dmid <= 96 && dout <= 96) || (din <= 32 && dout <= 96)) ? 8 : kWaves; }
int bwd_workgroups(int M, int din, int dmid, int dout) { return workgroups(M, bwd_waves(din, dmid, dout)); }

// workspace (floats): partial vectors | LayerNorm0 output | ReLU output | gradient at the Linear1 output |
// split-K partials of dW1 | of dW2.  Sized without knowing which stages a call has.
struct Ws { size_t slabs, xn, act, dh, p1, p2, total; int ns1, kc1, ns2, kc2; };
inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }
Ws ws_layout(int M, int din, int dmid, int dout) {
    Ws w;
    size_t o = 0;
    w.slabs = o; o += align64((size_t)bwd_workgroups(M, din, dmid, dout) * vec_layout(din, dmid, dout).total);
    w.xn = o; o += align64((size_t)M * din);
    w.act = o; o += align64((size_t)M * dmid);
    w.dh = o; o += align64((size_t)M * dmid);
    w.ns1 = gemm_split_for(dmid, din, M, &w.kc1);
    w.ns2 = gemm_split_for(dout, dmid, M, &w.kc2);
    w.p1 = o; o += align64((size_t)w.ns1 * dmid * din);
    w.p2 = o; o += align64((size_t)w.ns2 * dout * dmid);
    w.total = o;
    return w;
}

template <int NI, int NM, int NO, int NWB>
hipError_t launch(const Args &a, bool backward, hipStream_t s) {
    const bool deep = a.w2 != nullptr;
    const size_t lds = lds_bytes<NI, NM, NO>(a.din, a.dmid, a.dout, deep, backward);
    const void *fn = backward ? reinterpret_cast<const void *>(head_wide_bwd_kernel<NI, NM, NO, NWB>)
                              : reinterpret_cast<const void *>(head_wide_fwd_kernel<NI, NM, NO, kFwdWaves>);
    // the LDS limit is raised once per device, template instance and direction
    static std::atomic<unsigned long long> raised[2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!(raised[backward].load(std::memory_order_relaxed) >> dev & 1ULL)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
        if (e != hipSuccess) return e;
        raised[backward].fetch_or(1ULL << dev, std::memory_order_relaxed);
    }
    if (backward)
        TGRU_LAUNCH((head_wide_bwd_kernel<NI, NM, NO, NWB>), dim3(workgroups(a.M, NWB)), dim3(64 * NWB), lds, s, a);
    else
        TGRU_LAUNCH((head_wide_fwd_kernel<NI, NM, NO, kFwdWaves>), dim3(workgroups(a.M, kFwdWaves)),
                    dim3(64 * kFwdWaves), lds, s, a);
    return hipGetLastError();
}

hipError_t dispatch(const Args &a, bool backward, hipStream_t s) {
    if (a.din <= 96 && a.dmid <= 96 && a.dout <= 96) return launch<6, 6, 6, 8>(a, backward, s);   // see bwd_waves
    if (a.din <= 32 && a.dout <= 96) return launch<2, 12, 6, 8>(a, backward, s);
    return launch<8, 12, 8, kWaves>(a, backward, s);
}

void fill(const tgru_head_args &c, bool backward, Args &a) {
    a.x = c.x; a.g0 = c.ln0_w; a.b0 = c.ln0_b; a.w1 = c.w1; a.c1 = c.b1; a.g1 = c.ln1_w; a.b1 = c.ln1_b;
    a.w2 = c.w2; a.c2 = c.b2; a.M = c.M; a.din = c.din; a.dmid = c.dmid; a.dout = c.dout; a.eps = c.eps;
    if (backward) {
        a.dy = c.dy; a.dx = c.dx;
        a.vec_in = (c.din % 4 == 0) && aligned16(c.x) && (!c.dx || aligned16(c.dx));
        a.vec_out = (c.dout % 4 == 0) && aligned16(c.dy);
        const Ws w = ws_layout(c.M, c.din, c.dmid, c.dout);
        a.slabs = c.workspace + w.slabs; a.slab_stride = vec_layout(c.din, c.dmid, c.dout).total;
        if (c.dparams) {     // operands of the weight-gradient products (no parameter gradients wanted: not written)
            a.xn = c.workspace + w.xn; a.act = c.workspace + w.act; a.dh = c.workspace + w.dh;
        }
        if (c.dparams && !c.w2 && c.accumulate != 1) {      // dW2 | db2 of a head without Linear2 read as zeros
            a.zero = c.dparams + c.dmid * c.din + vec_layout(c.din, c.dmid, c.dout).c2;
            a.nzero = c.dout * c.dmid + c.dout;
        }
    } else {
        a.y = c.y;
        a.vec_in = (c.din % 4 == 0) && aligned16(c.x);
        a.vec_out = (c.dout % 4 == 0) && aligned16(c.y);
    }
}

// the reductions a backward call owes: dW1, the five vectors next to it, dW2, db2 (flat gradient layout of
// tgru_head_backward)
int reduce_jobs(const tgru_head_args &c, tgru_reduce_job *jobs) {
    const Ws w = ws_layout(c.M, c.din, c.dmid, c.dout);
    const VecLayout vl = vec_layout(c.din, c.dmid, c.dout);
    const bool deep = c.w2 != nullptr;
    const int n1 = c.dmid * c.din, nvec = vl.c2, n2 = c.dout * c.dmid;
    float *d = c.dparams;
    int n = 0;
    jobs[n++] = tgru_reduce_job{c.workspace + w.p1, (int64_t)n1, w.ns1, n1, d};
    jobs[n++] = tgru_reduce_job{c.workspace + w.slabs, (int64_t)vl.total, bwd_workgroups(c.M, c.din, c.dmid, c.dout), nvec, d + n1};
    if (!deep) return n;        // no Linear2: the backward kernel has written zeros there
    jobs[n++] = tgru_reduce_job{c.workspace + w.p2, (int64_t)n2, w.ns2, n2, d + n1 + nvec};
    jobs[n++] = tgru_reduce_job{c.workspace + w.slabs + vl.c2, (int64_t)vl.total, bwd_workgroups(c.M, c.din, c.dmid, c.dout), c.dout,
                                d + n1 + nvec + n2};
    return n;
}

}  // namespace headw

bool head_wide_supported(int din, int dmid, int dout, bool deep) {
    if (switches().no_wide_head) return false;
    if (din <= 0 || dmid <= 0 || dout <= 0) return false;
    if (din > headw::kMaxIo || dout > headw::kMaxIo || dmid > headw::kMaxMid) return false;
    if (!deep && dmid > headw::kMaxIo) return false;
    return (size_t)headw::lds_layout(din, dmid, dout, deep).total * sizeof(float) <= (size_t)headw::kMaxLdsBytes;
}

size_t head_wide_workspace_floats(int M, int din, int dmid, int dout) {
    return headw::ws_layout(M, din, dmid, dout).total;
}

int head_wide_forward(const tgru_head_args &c, hipStream_t s) {
    headw::Args a{};
    headw::fill(c, false, a);
    return headw::dispatch(a, false, s) == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int head_wide_backward(const tgru_head_args &c, hipStream_t s) {
    headw::Args a{};
    headw::fill(c, true, a);
    const bool deep = c.w2 != nullptr;
    if (headw::dispatch(a, true, s) != hipSuccess) return TGRU_E_LAUNCH;
    if (!c.dparams) return TGRU_OK;          // dx only
    const headw::Ws w = headw::ws_layout(c.M, c.din, c.dmid, c.dout);
    // dW1 (dmid x din) = dh^T . (Linear1 input), dW2 (dout x dmid) = dy^T . (ReLU output): split over the rows
    GemmArgs g{};
    g.amap = RowMap{0, 1, 1, nullptr};
    g.bmap = g.amap;
    g.cmap = g.amap;
    g.bias = nullptr; g.accumulate = 0;
    g.A = deep ? a.dh : c.dy; g.lda = c.dmid; g.Bm = c.ln0_w ? a.xn : c.x; g.ldb = c.din;
    g.C = c.workspace + w.p1; g.ldc = c.din; g.M = c.dmid; g.N = c.din; g.K = c.M; g.kchunk = w.kc1;
    g.slab_stride = (long)c.dmid * c.din;
    if (launch_gemm(g, false, false, w.ns1, s) != hipSuccess) return TGRU_E_LAUNCH;
    if (deep) {
        g.A = c.dy; g.lda = c.dout; g.Bm = a.act; g.ldb = c.dmid;
        g.C = c.workspace + w.p2; g.ldc = c.dmid; g.M = c.dout; g.N = c.dmid; g.K = c.M; g.kchunk = w.kc2;
        g.slab_stride = (long)c.dout * c.dmid;
        if (launch_gemm(g, false, false, w.ns2, s) != hipSuccess) return TGRU_E_LAUNCH;
    }
    if (c.accumulate == TGRU_DEFER_REDUCE) return TGRU_OK;
    tgru_reduce_job jobs[4];
    const int n = headw::reduce_jobs(c, jobs);
    if (c.accumulate == 0) return tgru_reduce_jobs(jobs, n, (tgru_stream_t)s);
    for (int k = 0; k < n; ++k)
        if (launch_slab_reduce(jobs[k].src, jobs[k].stride, jobs[k].count, jobs[k].dst, jobs[k].n, 1, s) != hipSuccess)
            return TGRU_E_LAUNCH;
    return TGRU_OK;
}

int head_wide_reduce_jobs(const tgru_head_args &c, tgru_reduce_job *jobs, int max_jobs) {
    if (max_jobs < 4) return TGRU_E_DIMS;
    return headw::reduce_jobs(c, jobs);
}

}  // namespace tgru
