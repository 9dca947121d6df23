// LayerNorm over the last dimension, forward and backward.  First the kernels for rows of up to 128 floats
// (a multiple of 4): the normalisations of the dense heads around the GRU (outdated_files/TimeGANv1.py:45,67)
// at hidden sizes beyond what the fused head kernels of head.hip cover (e.g. H = 96).
//
// HBM-bound by construction: forward reads x and writes y once (plus mean / rstd per row), backward reads
// x and dy once and writes dx once.  Half a wave (32 lanes x 4 consecutive floats) owns a row, the row
// sums are five butterfly exchanges inside the half; a lane keeps the same four columns for every row it
// visits, so the column sums for d(gamma) / d(beta) accumulate in registers, are combined over the
// workgroup's eight row slots through LDS and leave ONE partial per workgroup; a second kernel sums
// the partials in a fixed order (deterministic, no atomics).
#include "tgru_internal.h"
#include "../../include/tgru.h"

namespace tgru {
namespace ln {

constexpr int kThreads = 256;
constexpr int kRowsPerPass = kThreads / 32;   // 8 rows per workgroup and pass
constexpr int kMaxWidth = 128;
constexpr int kMaxWorkgroups = 256;     // one per CU; also the number of partial column sums

__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ y,
                                                          float *__restrict__ stats, long rows, int W, float eps) {
    const int c = (threadIdx.x & 31) * 4, slot = threadIdx.x >> 5;
    const bool on = c < W;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = (on && gamma) ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b = (on && beta) ? *reinterpret_cast<const float4 *>(beta + c) : zero4;
    const float inv = 1.f / (float)W;
    for (long r0 = (long)blockIdx.x * kRowsPerPass; r0 < rows; r0 += (long)gridDim.x * kRowsPerPass) {
        const long r = r0 + slot;
        const long rc = r < rows ? r : rows - 1;       // rows past the end are computed on the last row, not stored
        const int cc = on ? c : 0;
        float4 v = *reinterpret_cast<const float4 *>(x + rc * W + cc);
        if (!on) v = zero4;
        const float mean = half_wave_sum(v.x + v.y + v.z + v.w) * inv;
        const float4 d = on ? make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean) : zero4;
        const float var = half_wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * inv;
        const float rstd = rsqrtf(var + eps);
        if (r < rows) {
            if (on)
                *reinterpret_cast<float4 *>(y + r * W + c) =
                    make_float4(d.x * rstd * g.x + b.x, d.y * rstd * g.y + b.y, d.z * rstd * g.z + b.z, d.w * rstd * g.w + b.w);
            if (c == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
        }
    }
}

__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                          const float *__restrict__ stats, const float *__restrict__ gamma,
                                                          float *__restrict__ dx, float *__restrict__ part, long rows, int W) {
    __shared__ float red[kRowsPerPass][2][kMaxWidth];
    const int c = (threadIdx.x & 31) * 4, slot = threadIdx.x >> 5;
    const bool on = c < W;
    const int cc = on ? c : 0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = (on && gamma) ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float inv = 1.f / (float)W;
    float4 sg = zero4, sb = zero4;     // column sums of dy * xhat and of dy over this lane's rows
    for (long r0 = (long)blockIdx.x * kRowsPerPass; r0 < rows; r0 += (long)gridDim.x * kRowsPerPass) {
        const long r = r0 + slot;
        const bool live = r < rows;
        const long rc = live ? r : rows - 1;
        float4 v = *reinterpret_cast<const float4 *>(x + rc * W + cc);
        float4 d = *reinterpret_cast<const float4 *>(dy + rc * W + cc);
        const float mean = stats[2 * rc], rstd = stats[2 * rc + 1];
        if (!on || !live) { v = make_float4(mean, mean, mean, mean); d = zero4; }
        const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        const float4 gd = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        const float m1 = half_wave_sum(gd.x + gd.y + gd.z + gd.w) * inv;
        const float m2 = half_wave_sum(gd.x * xh.x + gd.y * xh.y + gd.z * xh.z + gd.w * xh.w) * inv;
        if (live && on && dx)
            *reinterpret_cast<float4 *>(dx + r * W + c) =
                make_float4(rstd * (gd.x - m1 - xh.x * m2), rstd * (gd.y - m1 - xh.y * m2),
                            rstd * (gd.z - m1 - xh.z * m2), rstd * (gd.w - m1 - xh.w * m2));
        sg.x += d.x * xh.x; sg.y += d.y * xh.y; sg.z += d.z * xh.z; sg.w += d.w * xh.w;
        sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
    }
    if (on) {
        *reinterpret_cast<float4 *>(&red[slot][0][c]) = sg;
        *reinterpret_cast<float4 *>(&red[slot][1][c]) = sb;
    }
    __syncthreads();
    // one partial per workgroup: part[blockIdx][2][W], the eight row slots added in order
    for (int i = threadIdx.x; i < 2 * W; i += kThreads) {
        const int which = i / W, col = i - which * W;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kRowsPerPass; ++k) s += red[k][which][col];
        part[(long)blockIdx.x * 2 * W + i] = s;
    }
}

// dgamma / dbeta = column sums of the per-workgroup partials part[nparts][2][W].  Workgroup = 32 columns x
// 8 groups of partials; a group adds a contiguous range with four interleaved accumulators, the groups
// are combined in fixed order.
__global__ __launch_bounds__(kThreads) void ln_bwd_reduce_kernel(const float *__restrict__ part, int nparts, int W,
                                                                 float *dgamma, float *dbeta, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    const int per = (nparts + 7) / 8;
    const int k0 = grp * per, k1 = min(nparts, k0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < 2 * W) {
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            s0 += part[(long)k * 2 * W + i];
            s1 += part[(long)(k + 1) * 2 * W + i];
            s2 += part[(long)(k + 2) * 2 * W + i];
            s3 += part[(long)(k + 3) * 2 * W + i];
        }
        for (; k < k1; ++k) s0 += part[(long)k * 2 * W + i];
    }
    red[grp][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || i >= 2 * W) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][c];
    float *dst = i < W ? dgamma : dbeta;
    const int col = i < W ? i : i - W;
    if (dst) dst[col] = accumulate ? dst[col] + s : s;
}

// ----------------------------------------------------------------------------------------------
// Rows of up to 1024 floats (a multiple of 4): the normalisations of the heads at hidden sizes above 128
// (the stress configuration: 512, and 1024 inside the generator's feature MLP).  A WAVE owns a row: lane l
// holds columns 4l + 256j, j < NV (one coalesced 1 KiB load per j), the row sums are six butterfly steps
// over the wave; the ReLU that follows the inner LayerNorm of a head (outdated_files/TimeGANv1.py:45-47) is
// applied in the same pass (forward: y = max(., 0); backward: dy is masked where the recomputed
// pre-activation is not positive), so neither the normalised tensor nor a mask crosses HBM on its own.
// Column sums as above: in registers over the rows a lane visits, over the workgroup's four waves through
// LDS, one partial per workgroup, fixed-order reduce.  column_sum_kernel is the same traversal for a bare
// column sum (the bias gradient of a Linear layer: the sum of dy over the rows).
// ----------------------------------------------------------------------------------------------
constexpr int kWideMaxWidth = 1024;
constexpr int kWideWaves = 4;               // rows per workgroup and pass
constexpr int kWideMaxWorkgroups = 2048;    // eight per CU keep enough loads in flight for HBM

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    return half_wave_sum(v);
}

template <int NV>
__global__ __launch_bounds__(64 * kWideWaves) void lnw_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float *__restrict__ y,
                                                                  float *__restrict__ stats, long rows, int W, float eps,
                                                                  int relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[NV], b[NV];
    bool on[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * lane + 256 * j;
        on[j] = c < W;
        g[j] = (on[j] && gamma) ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        b[j] = (on[j] && beta) ? *reinterpret_cast<const float4 *>(beta + c) : zero4;
    }
    const float inv = 1.f / (float)W;
    for (long r = (long)blockIdx.x * kWideWaves + wave; r < rows; r += (long)gridDim.x * kWideWaves) {
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = on[j] ? *reinterpret_cast<const float4 *>(x + r * W + 4 * lane + 256 * j) : zero4;
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = on[j] ? make_float4(v[j].x - mean, v[j].y - mean, v[j].z - mean, v[j].w - mean) : zero4;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rstd = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (!on[j]) continue;
            float4 o = make_float4(v[j].x * rstd * g[j].x + b[j].x, v[j].y * rstd * g[j].y + b[j].y,
                                   v[j].z * rstd * g[j].z + b[j].z, v[j].w * rstd * g[j].w + b[j].w);
            if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            *reinterpret_cast<float4 *>(y + r * W + 4 * lane + 256 * j) = o;
        }
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
}

template <int NV>
__global__ __launch_bounds__(64 * kWideWaves) void lnw_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                  const float *__restrict__ stats, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float *__restrict__ dx,
                                                                  float *__restrict__ part, long rows, int W, int relu) {
    __shared__ float red[kWideWaves][2][kWideMaxWidth];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[NV], b[NV], sg[NV], sb[NV];
    bool on[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * lane + 256 * j;
        on[j] = c < W;
        g[j] = (on[j] && gamma) ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        b[j] = (on[j] && beta) ? *reinterpret_cast<const float4 *>(beta + c) : zero4;
        sg[j] = zero4; sb[j] = zero4;
    }
    const float inv = 1.f / (float)W;
    for (long r = (long)blockIdx.x * kWideWaves + wave; r < rows; r += (long)gridDim.x * kWideWaves) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float4 xh[NV], gd[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long o = r * W + 4 * lane + 256 * j;
            const float4 v = on[j] ? *reinterpret_cast<const float4 *>(x + o) : make_float4(mean, mean, mean, mean);
            float4 d = on[j] ? *reinterpret_cast<const float4 *>(dy + o) : zero4;
            xh[j] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
            if (relu) {     // the gradient passes where the forward's pre-activation was positive
                d.x = xh[j].x * g[j].x + b[j].x > 0.f ? d.x : 0.f;
                d.y = xh[j].y * g[j].y + b[j].y > 0.f ? d.y : 0.f;
                d.z = xh[j].z * g[j].z + b[j].z > 0.f ? d.z : 0.f;
                d.w = xh[j].w * g[j].w + b[j].w > 0.f ? d.w : 0.f;
            }
            gd[j] = make_float4(d.x * g[j].x, d.y * g[j].y, d.z * g[j].z, d.w * g[j].w);
            s1 += (gd[j].x + gd[j].y) + (gd[j].z + gd[j].w);
            s2 += (gd[j].x * xh[j].x + gd[j].y * xh[j].y) + (gd[j].z * xh[j].z + gd[j].w * xh[j].w);
            sg[j].x += d.x * xh[j].x; sg[j].y += d.y * xh[j].y; sg[j].z += d.z * xh[j].z; sg[j].w += d.w * xh[j].w;
            sb[j].x += d.x; sb[j].y += d.y; sb[j].z += d.z; sb[j].w += d.w;
        }
        const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
        if (dx) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (on[j])
                    *reinterpret_cast<float4 *>(dx + r * W + 4 * lane + 256 * j) =
                        make_float4(rstd * (gd[j].x - m1 - xh[j].x * m2), rstd * (gd[j].y - m1 - xh[j].y * m2),
                                    rstd * (gd[j].z - m1 - xh[j].z * m2), rstd * (gd[j].w - m1 - xh[j].w * m2));
        }
    }
    if (!part) return;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (on[j]) {
            *reinterpret_cast<float4 *>(&red[wave][0][4 * lane + 256 * j]) = sg[j];
            *reinterpret_cast<float4 *>(&red[wave][1][4 * lane + 256 * j]) = sb[j];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * W; i += 64 * kWideWaves) {
        const int which = i / W, col = i - which * W;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kWideWaves; ++k) s += red[k][which][col];
        part[(long)blockIdx.x * 2 * W + i] = s;
    }
}

// part[blockIdx][W] = sum of this workgroup's rows of a (rows, W)
template <int NV>
__global__ __launch_bounds__(64 * kWideWaves) void column_sum_kernel(const float *__restrict__ a, float *__restrict__ part,
                                                                     long rows, int W) {
    __shared__ float red[kWideWaves][kWideMaxWidth];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) s[j] = zero4;
    for (long r = (long)blockIdx.x * kWideWaves + wave; r < rows; r += (long)gridDim.x * kWideWaves) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * lane + 256 * j;
            const float4 v = c < W ? *reinterpret_cast<const float4 *>(a + r * W + c) : zero4;
            s[j].x += v.x; s[j].y += v.y; s[j].z += v.z; s[j].w += v.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (4 * lane + 256 * j < W) *reinterpret_cast<float4 *>(&red[wave][4 * lane + 256 * j]) = s[j];
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 64 * kWideWaves) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kWideWaves; ++k) t += red[k][i];
        part[(long)blockIdx.x * W + i] = t;
    }
}

// out[i] (=|+=) sum_k part[k][i], i < n: 32 elements x 8 groups of partials per workgroup, fixed order
__global__ __launch_bounds__(kThreads) void partial_sum_kernel(const float *__restrict__ part, int nparts, int n,
                                                               float *out, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    const int per = (nparts + 7) / 8;
    const int k0 = grp * per, k1 = min(nparts, k0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            s0 += part[(long)k * n + i];
            s1 += part[(long)(k + 1) * n + i];
            s2 += part[(long)(k + 2) * n + i];
            s3 += part[(long)(k + 3) * n + i];
        }
        for (; k < k1; ++k) s0 += part[(long)k * n + i];
    }
    red[grp][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][c];
    out[i] = accumulate ? out[i] + s : s;
}

static int wide_workgroups(long rows) {
    const long n = (rows + kWideWaves - 1) / kWideWaves;
    return (int)(n < kWideMaxWorkgroups ? n : kWideMaxWorkgroups);
}

static int workgroups(long rows) {
    const long n = (rows + kRowsPerPass - 1) / kRowsPerPass;
    return (int)(n < kMaxWorkgroups ? n : kMaxWorkgroups);
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ln
}  // namespace tgru

using namespace tgru;

extern "C" {

int tgru_layer_norm_supported(int width) { return width >= 4 && width <= ln::kWideMaxWidth && (width & 3) == 0; }

size_t tgru_layer_norm_backward_workspace_bytes(long rows, int width) {
    if (rows <= 0 || !tgru_layer_norm_supported(width)) return 0;
    const int nwg = width <= ln::kMaxWidth ? ln::workgroups(rows) : ln::wide_workgroups(rows);
    return (size_t)nwg * 2 * width * sizeof(float);
}

int tgru_layer_norm_act_forward(const float *x, const float *gamma, const float *beta, float *y, float *stats,
                                long rows, int width, float eps, int relu, tgru_stream_t stream) {
    if (rows <= 0 || !tgru_layer_norm_supported(width) || (relu != 0 && relu != 1)) return TGRU_E_DIMS;
    if (!x || !y || !stats) return TGRU_E_NULL;
    if (!ln::aligned16(x) || !ln::aligned16(y) || (gamma && !ln::aligned16(gamma)) || (beta && !ln::aligned16(beta)))
        return TGRU_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (width <= ln::kMaxWidth && !relu) {
        TGRU_LAUNCH(ln::ln_fwd_kernel, dim3(ln::workgroups(rows)), dim3(ln::kThreads), 0, s, x, gamma, beta, y, stats, rows,
                    width, eps);
    } else {
        const dim3 grid(ln::wide_workgroups(rows)), block(64 * ln::kWideWaves);
        switch ((width + 255) / 256) {
        case 1: TGRU_LAUNCH(ln::lnw_fwd_kernel<1>, grid, block, 0, s, x, gamma, beta, y, stats, rows, width, eps, relu); break;
        case 2: TGRU_LAUNCH(ln::lnw_fwd_kernel<2>, grid, block, 0, s, x, gamma, beta, y, stats, rows, width, eps, relu); break;
        case 3: TGRU_LAUNCH(ln::lnw_fwd_kernel<3>, grid, block, 0, s, x, gamma, beta, y, stats, rows, width, eps, relu); break;
        default: TGRU_LAUNCH(ln::lnw_fwd_kernel<4>, grid, block, 0, s, x, gamma, beta, y, stats, rows, width, eps, relu); break;
        }
    }
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int tgru_layer_norm_forward(const float *x, const float *gamma, const float *beta, float *y, float *stats, long rows,
                            int width, float eps, tgru_stream_t stream) {
    return tgru_layer_norm_act_forward(x, gamma, beta, y, stats, rows, width, eps, 0, stream);
}

int tgru_layer_norm_act_backward(const float *x, const float *dy, const float *stats, const float *gamma,
                                 const float *beta, float *dx, float *dgamma, float *dbeta, float *workspace, long rows,
                                 int width, int relu, int accumulate, tgru_stream_t stream) {
    if (rows <= 0 || !tgru_layer_norm_supported(width) || (accumulate != 0 && accumulate != 1) || (relu != 0 && relu != 1))
        return TGRU_E_DIMS;
    if (!x || !dy || !stats || !workspace) return TGRU_E_NULL;
    if (!ln::aligned16(x) || !ln::aligned16(dy) || (dx && !ln::aligned16(dx)) || (gamma && !ln::aligned16(gamma)) ||
        (beta && !ln::aligned16(beta)))
        return TGRU_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    int nwg;
    if (width <= ln::kMaxWidth && !relu) {
        nwg = ln::workgroups(rows);
        TGRU_LAUNCH(ln::ln_bwd_kernel, dim3(nwg), dim3(ln::kThreads), 0, s, x, dy, stats, gamma, dx, workspace, rows, width);
    } else {
        nwg = ln::wide_workgroups(rows);
        const dim3 grid(nwg), block(64 * ln::kWideWaves);
        float *part = (dgamma || dbeta) ? workspace : nullptr;
        switch ((width + 255) / 256) {
        case 1: TGRU_LAUNCH(ln::lnw_bwd_kernel<1>, grid, block, 0, s, x, dy, stats, gamma, beta, dx, part, rows, width, relu); break;
        case 2: TGRU_LAUNCH(ln::lnw_bwd_kernel<2>, grid, block, 0, s, x, dy, stats, gamma, beta, dx, part, rows, width, relu); break;
        case 3: TGRU_LAUNCH(ln::lnw_bwd_kernel<3>, grid, block, 0, s, x, dy, stats, gamma, beta, dx, part, rows, width, relu); break;
        default: TGRU_LAUNCH(ln::lnw_bwd_kernel<4>, grid, block, 0, s, x, dy, stats, gamma, beta, dx, part, rows, width, relu); break;
        }
    }
    if (dgamma || dbeta)
        TGRU_LAUNCH(ln::ln_bwd_reduce_kernel, dim3((2 * width + 31) / 32),
                           dim3(ln::kThreads), 0, s, workspace, nwg, width, dgamma, dbeta, accumulate);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int tgru_layer_norm_backward(const float *x, const float *dy, const float *stats, const float *gamma, float *dx,
                             float *dgamma, float *dbeta, float *workspace, long rows, int width, int accumulate,
                             tgru_stream_t stream) {
    return tgru_layer_norm_act_backward(x, dy, stats, gamma, nullptr, dx, dgamma, dbeta, workspace, rows, width, 0,
                                        accumulate, stream);
}

size_t tgru_column_sum_workspace_bytes(long rows, int width) {
    if (rows <= 0 || !tgru_layer_norm_supported(width)) return 0;
    return (size_t)ln::wide_workgroups(rows) * width * sizeof(float);
}

int tgru_column_sum(const float *a, float *out, float *workspace, long rows, int width, int accumulate,
                    tgru_stream_t stream) {
    if (rows <= 0 || !tgru_layer_norm_supported(width) || (accumulate != 0 && accumulate != 1)) return TGRU_E_DIMS;
    if (!a || !out || !workspace) return TGRU_E_NULL;
    if (!ln::aligned16(a)) return TGRU_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int nwg = ln::wide_workgroups(rows);
    const dim3 grid(nwg), block(64 * ln::kWideWaves);
    switch ((width + 255) / 256) {
    case 1: TGRU_LAUNCH(ln::column_sum_kernel<1>, grid, block, 0, s, a, workspace, rows, width); break;
    case 2: TGRU_LAUNCH(ln::column_sum_kernel<2>, grid, block, 0, s, a, workspace, rows, width); break;
    case 3: TGRU_LAUNCH(ln::column_sum_kernel<3>, grid, block, 0, s, a, workspace, rows, width); break;
    default: TGRU_LAUNCH(ln::column_sum_kernel<4>, grid, block, 0, s, a, workspace, rows, width); break;
    }
    TGRU_LAUNCH(ln::partial_sum_kernel, dim3((width + 31) / 32), dim3(ln::kThreads), 0, s, workspace, nwg, width, out,
                accumulate);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

}  // extern "C"
