// Fused per-row "head" of the TimeGAN networks: the small dense stages that sit before and after
// every GRU stack (outdated_files/TimeGANv1.py:55,67-68,145-148,226-227,294,305-306,382-383),
//
//      y = [ LayerNorm0 ] -> Linear1 -> [ LayerNorm1 -> ReLU -> Linear2 ]
//
// over M independent rows of at most 32 features, forward and backward, one kernel each (PyTorch
// runs 5-8 kernels forward and about 20 backward for the same thing, each shorter than its launch).
//
// One wave owns 32 rows.  A Linear is computed TRANSPOSED on v_mfma_f32_32x32x2_f32:
//      Y^T (feature i, row j) = W (i, k) . X^T (k, row j)
// with W as the A operand and the activations as the B operand.  Lane (hh, j) = (lane >> 5,
// lane & 31) then holds, for row j, the 16 features f(r, hh) = 8*(r/4) + r%4 + 4*hh, r = 0..15, of
// the result -- and exactly that register set is the B operand of the next Linear when its k-step m
// pairs the features f(m, 0) (lower half) and f(m, 1) (upper half).  So a whole chain of Linears,
// LayerNorms (sum over a lane's 16 registers + one swap of the halves) and ReLUs runs in registers
// with no shuffles.  Backward recomputes the forward from x, pushes gradients back through the
// same scheme with transposed weight fragments, and forms the weight gradients dW = dY^T . A
// as MFMA products contracted over the 32 rows (operands transposed through LDS).  Parameter
// gradients leave the kernel as one partial per workgroup, summed by a second kernel in a fixed
// order (deterministic, no atomics).
#include "../../include/tgru.h"
#include "tgru_internal.h"

namespace tgru {
namespace head {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kF = 32;          // feature tile
constexpr int kTileLd = 33;     // LDS row stride of a 32x32 transpose tile

__device__ __forceinline__ int feat(int r, int hh) { return 8 * (r >> 2) + (r & 3) + 4 * hh; }

__device__ __forceinline__ void both_halves(float v, float &lo, float &hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_halves(float v) {
    float lo, hi;
    both_halves(v, lo, hi);
    return lo + hi;
}
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

struct Args {
    const float *x; float *y;                 // forward
    const float *dy; float *dx;               // backward (dx may be null)
    const float *g0, *b0;                     // LayerNorm0 (null: absent)
    const float *w1, *c1;                     // Linear1 (dmid x din), bias
    const float *g1, *b1, *w2, *c2;           // LayerNorm1, Linear2 (dout x dmid), bias (w2 null: shallow)
    float *slabs; long slab_stride;           // backward: one partial gradient vector per workgroup
    int M, din, dmid, dout;
    float eps;
};

// flat parameter-gradient layout (floats)
struct GradLayout { int w1, c1, g0, b0, g1, b1, w2, c2, total; };
__host__ __device__ inline GradLayout grad_layout(int din, int dmid, int dout) {
    GradLayout g;
    int o = 0;
    g.w1 = o; o += dmid * din;
    g.c1 = o; o += dmid;
    g.g0 = o; o += din;
    g.b0 = o; o += din;
    g.g1 = o; o += dmid;
    g.b1 = o; o += dmid;
    g.w2 = o; o += dout * dmid;
    g.c2 = o; o += dout;
    g.total = o;
    return g;
}

// A-operand fragments of y = W x (W is rows x cols, row-major): k-step m, lane (kh, i) -> W[i][f(m,kh)]
__device__ __forceinline__ void load_frag(const float *w, int rows, int cols, int lane, float (&f)[16]) {
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int k = feat(m, kh);
        f[m] = (w && i < rows && k < cols) ? w[i * cols + k] : 0.f;
    }
}
// A-operand fragments of x = W^T y: k-step m, lane (ih, k) -> W[f(m,ih)][k]
__device__ __forceinline__ void load_frag_t(const float *w, int rows, int cols, int lane, float (&f)[16]) {
    const int k = lane & 31, ih = lane >> 5;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = feat(m, ih);
        f[m] = (w && i < rows && k < cols) ? w[i * cols + k] : 0.f;
    }
}
// per-lane copy of a feature vector in register order: v[r] = p[f(r,hh)] (fill when absent / out of range)
__device__ __forceinline__ void load_vec(const float *p, int n, int hh, float fill, float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = feat(r, hh);
        v[r] = (p && f < n) ? p[f] : fill;
    }
}

__device__ __forceinline__ f32x16 linear(const float (&wf)[16], const float (&v)[16]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = mfma(wf[m], v[m], acc);
    return acc;
}

// LayerNorm statistics of one row spread over the two lane halves; entries beyond d are zero
__device__ __forceinline__ void ln_stats(const float (&v)[16], const float (&valid)[16], int d, float eps,
                                         float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    mean = sum_halves(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float c = (v[r] - mean) * valid[r];
        q += c * c;
    }
    rstd = rsqrtf(sum_halves(q) / (float)d + eps);
}

// rows of a tile: 32 per wave; row of lane = tile * 32 + (lane & 31)
__device__ __forceinline__ void load_rows(const float *p, int M, int d, long row, int hh, float (&v)[16]) {
    const bool ok = row < M;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = feat(r, hh);
        v[r] = (ok && f < d) ? p[row * d + f] : 0.f;
    }
}
__device__ __forceinline__ void store_rows(float *p, int M, int d, long row, int hh, const float (&v)[16]) {
    if (row >= M) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = feat(r, hh);
        if (f < d) p[row * d + f] = v[r];
    }
}

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void head_fwd_kernel(Args a) {
    const int lane = threadIdx.x, hh = lane >> 5, j = lane & 31;
    const bool deep = a.w2 != nullptr;
    float w1f[16], w2f[16], g0[16], b0[16], c1[16], g1[16], b1[16], c2[16], vin[16], vmid[16];
    load_frag(a.w1, a.dmid, a.din, lane, w1f);
    load_frag(a.w2, a.dout, a.dmid, lane, w2f);
    load_vec(a.g0, a.din, hh, 0.f, g0);
    load_vec(a.b0, a.din, hh, 0.f, b0);
    load_vec(a.c1, a.dmid, hh, 0.f, c1);
    load_vec(a.g1, a.dmid, hh, 0.f, g1);
    load_vec(a.b1, a.dmid, hh, 0.f, b1);
    load_vec(a.c2, a.dout, hh, 0.f, c2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        vin[r] = feat(r, hh) < a.din ? 1.f : 0.f;
        vmid[r] = feat(r, hh) < a.dmid ? 1.f : 0.f;
    }
    const int ntiles = (a.M + 31) / 32;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row = (long)tile * 32 + j;
        float x[16];
        load_rows(a.x, a.M, a.din, row, hh, x);
        if (a.g0) {
            float mean, rstd;
            ln_stats(x, vin, a.din, a.eps, mean, rstd);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = ((x[r] - mean) * rstd * g0[r] + b0[r]) * vin[r];
        }
        f32x16 h = linear(w1f, x);
        mfma_drain();
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = h[r] + c1[r];
        if (deep) {
            float mean, rstd;
            ln_stats(y, vmid, a.dmid, a.eps, mean, rstd);
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = fmaxf(((y[r] - mean) * rstd * g1[r] + b1[r]) * vmid[r], 0.f);
            const f32x16 o = linear(w2f, y);
            mfma_drain();
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = o[r] + c2[r];
        }
        store_rows(a.y, a.M, a.dout, row, hh, y);
    }
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
// F-layout registers of 32 rows -> LDS tile[row][feature]
__device__ __forceinline__ void tile_put(float *tile, int j, int hh, const float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[j * kTileLd + feat(r, hh)] = v[r];
}
// operand of a product contracted over the rows: k-step m, lane (kh, c) -> tile[2m + kh][c]
__device__ __forceinline__ void tile_get_t(const float *tile, int lane, float (&v)[16]) {
    const int c = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = tile[(2 * m + kh) * kTileLd + c];
}
// dW (o, k) += sum over rows of u[row][o] * v[row][k]; both given as LDS tiles
__device__ __forceinline__ f32x16 outer_acc(const float *tu, const float *tv, int lane, f32x16 acc) {
    float ua[16], vb[16];
    tile_get_t(tu, lane, ua);
    tile_get_t(tv, lane, vb);
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = mfma(ua[m], vb[m], acc);
    return acc;
}
// column sums of a per-lane accumulator set: lane (hh, j) holds feature f(r,hh) of "row" j; result:
// every lane c < 32 of the lower half returns sum over j of feature c
__device__ __forceinline__ float colsum(float *tile, int lane, const float (&v)[16]) {
    const int hh = lane >> 5, j = lane & 31;
    tile_put(tile, j, hh, v);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += tile[q * kTileLd + j];
    return s;
}

__global__ __launch_bounds__(64) void head_bwd_kernel(Args a) {
    __shared__ float tiles[2][32 * kTileLd];
    const int lane = threadIdx.x, hh = lane >> 5, j = lane & 31;
    const bool deep = a.w2 != nullptr;
    float w1f[16], w1t[16], w2f[16], w2t[16], g0[16], b0[16], c1[16], g1[16], b1[16], vin[16], vmid[16];
    load_frag(a.w1, a.dmid, a.din, lane, w1f);
    load_frag_t(a.w1, a.dmid, a.din, lane, w1t);
    load_frag(a.w2, a.dout, a.dmid, lane, w2f);
    load_frag_t(a.w2, a.dout, a.dmid, lane, w2t);
    load_vec(a.g0, a.din, hh, 0.f, g0);
    load_vec(a.b0, a.din, hh, 0.f, b0);
    load_vec(a.c1, a.dmid, hh, 0.f, c1);
    load_vec(a.g1, a.dmid, hh, 0.f, g1);
    load_vec(a.b1, a.dmid, hh, 0.f, b1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        vin[r] = feat(r, hh) < a.din ? 1.f : 0.f;
        vmid[r] = feat(r, hh) < a.dmid ? 1.f : 0.f;
    }
    f32x16 dW1, dW2;
    float dc1[16], dg0[16], db0[16], dg1[16], db1[16], dc2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dW1[r] = dW2[r] = 0.f;
        dc1[r] = dg0[r] = db0[r] = dg1[r] = db1[r] = dc2[r] = 0.f;
    }

    const int ntiles = (a.M + 31) / 32;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row = (long)tile * 32 + j;
        // ---- forward again, keeping what the backward needs --------------------------------------
        float x1[16], xh0[16], rstd0 = 1.f;
        load_rows(a.x, a.M, a.din, row, hh, x1);
        if (a.g0) {
            float mean;
            ln_stats(x1, vin, a.din, a.eps, mean, rstd0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xh0[r] = (x1[r] - mean) * rstd0 * vin[r];
                x1[r] = (xh0[r] * g0[r] + b0[r]) * vin[r];
            }
        }
        float dh[16];             // gradient at the output of Linear1
        load_rows(a.dy, a.M, a.dout, row, hh, dh);
        if (deep) {
            const f32x16 hacc = linear(w1f, x1);
            mfma_drain();
            float hn[16], act[16], rstd1, mean1;
#pragma unroll
            for (int r = 0; r < 16; ++r) hn[r] = hacc[r] + c1[r];
            ln_stats(hn, vmid, a.dmid, a.eps, mean1, rstd1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hn[r] = (hn[r] - mean1) * rstd1 * vmid[r];
                act[r] = fmaxf((hn[r] * g1[r] + b1[r]) * vmid[r], 0.f);
            }
            // Linear2: dc2, dW2 = dy^T . act, da = W2^T . dy
#pragma unroll
            for (int r = 0; r < 16; ++r) dc2[r] += dh[r];
            tile_put(tiles[0], j, hh, dh);
            tile_put(tiles[1], j, hh, act);
            dW2 = outer_acc(tiles[0], tiles[1], lane, dW2);
            const f32x16 da = linear(w2t, dh);
            mfma_drain();
            // ReLU and LayerNorm1
            float s1 = 0.f, s2 = 0.f, dn[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dpre = act[r] > 0.f ? da[r] : 0.f;
                dg1[r] += dpre * hn[r];
                db1[r] += dpre;
                dn[r] = dpre * g1[r];
                s1 += dn[r];
                s2 += dn[r] * hn[r];
            }
            const float m1 = sum_halves(s1) / (float)a.dmid, m2 = sum_halves(s2) / (float)a.dmid;
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[r] = rstd1 * (dn[r] - m1 - hn[r] * m2) * vmid[r];
        }
        // ---- Linear1: dc1, dW1 = dh^T . x1, dx1 = W1^T . dh ------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) dc1[r] += dh[r];
        tile_put(tiles[0], j, hh, dh);
        tile_put(tiles[1], j, hh, x1);
        dW1 = outer_acc(tiles[0], tiles[1], lane, dW1);
        if (a.dx) {
            const f32x16 dxa = linear(w1t, dh);
            mfma_drain();
            float dx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[r] = dxa[r];
            if (a.g0) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dg0[r] += dx[r] * xh0[r];
                    db0[r] += dx[r];
                    dx[r] *= g0[r];
                    s1 += dx[r];
                    s2 += dx[r] * xh0[r];
                }
                const float m1 = sum_halves(s1) / (float)a.din, m2 = sum_halves(s2) / (float)a.din;
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[r] = rstd0 * (dx[r] - m1 - xh0[r] * m2) * vin[r];
            }
            store_rows(a.dx, a.M, a.din, row, hh, dx);
        } else if (a.g0) {
            // no dx wanted, but LayerNorm0's own parameters still need the gradient at its output
            const f32x16 dxa = linear(w1t, dh);
            mfma_drain();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dg0[r] += dxa[r] * xh0[r];
                db0[r] += dxa[r];
            }
        }
    }
    mfma_drain();

    // ---- this workgroup's partial parameter gradients -------------------------------------------------
    const GradLayout gl = grad_layout(a.din, a.dmid, a.dout);
    float *slab = a.slabs + (long)blockIdx.x * a.slab_stride;
    // dW (o, k): lane (hh, k), register r -> o = f(r, hh)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = feat(r, hh);
        if (o < a.dmid && j < a.din) slab[gl.w1 + o * a.din + j] = dW1[r];
        if (o < a.dout && j < a.dmid) slab[gl.w2 + o * a.dmid + j] = dW2[r];   // zeros for a shallow head
    }
    float s;
    s = colsum(tiles[0], lane, dc1); if (hh == 0 && j < a.dmid) slab[gl.c1 + j] = s;
    s = colsum(tiles[1], lane, dg0); if (hh == 0 && j < a.din) slab[gl.g0 + j] = s;
    s = colsum(tiles[0], lane, db0); if (hh == 0 && j < a.din) slab[gl.b0 + j] = s;
    s = colsum(tiles[1], lane, dg1); if (hh == 0 && j < a.dmid) slab[gl.g1 + j] = s;
    s = colsum(tiles[0], lane, db1); if (hh == 0 && j < a.dmid) slab[gl.b1 + j] = s;
    s = colsum(tiles[1], lane, dc2); if (hh == 0 && j < a.dout) slab[gl.c2 + j] = s;
}

// out[i] (=|+=) sum over workgroups of slabs[k * stride + i]: 32 elements x 8 slab groups per workgroup
__global__ __launch_bounds__(256) void head_reduce_kernel(const float *slabs, long stride, int nslabs, float *out,
                                                          int n, int accumulate) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    const int per = (nslabs + 7) / 8;
    const int c0 = grp * per, c1 = min(nslabs, c0 + per);
    float v0 = 0.f, v1 = 0.f;
    if (idx < n) {
        int c = c0;
        for (; c + 1 < c1; c += 2) {
            v0 += slabs[(long)c * stride + idx];
            v1 += slabs[(long)(c + 1) * stride + idx];
        }
        if (c < c1) v0 += slabs[(long)c * stride + idx];
    }
    part[grp][e] = v0 + v1;
    __syncthreads();
    if (grp != 0 || idx >= n) return;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][e];
    out[idx] = accumulate ? out[idx] + v : v;
}

int workgroups(int M) {
    const int tiles = (M + 31) / 32;
    return tiles < 1024 ? tiles : 1024;
}

}  // namespace head
}  // namespace tgru

using namespace tgru;

extern "C" {

static int head_check(int M, int din, int dmid, int dout, const void *w1, const void *c1, const void *w2) {
    if (M <= 0 || din <= 0 || dmid <= 0 || dout <= 0) return TGRU_E_DIMS;
    if (din > head::kF || dmid > head::kF || dout > head::kF) return TGRU_E_DIMS;
    if (!w2 && dout != dmid) return TGRU_E_DIMS;
    if ((long long)M * 32 >= (1LL << 31)) return TGRU_E_DIMS;
    if (!w1 || !c1) return TGRU_E_NULL;
    return TGRU_OK;
}

size_t tgru_head_grad_floats(int din, int dmid, int dout) {
    return (size_t)head::grad_layout(din, dmid, dout).total;
}

size_t tgru_head_backward_workspace_bytes(int M, int din, int dmid, int dout) {
    if (M <= 0) return 0;
    return (size_t)head::workgroups(M) * head::grad_layout(din, dmid, dout).total * sizeof(float);
}

int tgru_head_forward(const float *x, float *y, int M, int din, int dmid, int dout, const float *ln0_w,
                      const float *ln0_b, const float *w1, const float *b1, const float *ln1_w,
                      const float *ln1_b, const float *w2, const float *b2, float eps, tgru_stream_t stream) {
    int rc = head_check(M, din, dmid, dout, w1, b1, w2);
    if (rc != TGRU_OK) return rc;
    if (!x || !y || (ln0_w && !ln0_b) || (w2 && (!ln1_w || !ln1_b || !b2))) return TGRU_E_NULL;
    head::Args a{};
    a.x = x; a.y = y; a.g0 = ln0_w; a.b0 = ln0_b; a.w1 = w1; a.c1 = b1; a.g1 = ln1_w; a.b1 = ln1_b;
    a.w2 = w2; a.c2 = b2; a.M = M; a.din = din; a.dmid = dmid; a.dout = dout; a.eps = eps;
    hipLaunchKernelGGL(head::head_fwd_kernel, dim3(head::workgroups(M)), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int tgru_head_backward(const float *x, const float *dy, float *dx, int M, int din, int dmid, int dout,
                       const float *ln0_w, const float *ln0_b, const float *w1, const float *b1,
                       const float *ln1_w, const float *ln1_b, const float *w2, const float *b2, float eps,
                       float *dparams, float *workspace, int accumulate, tgru_stream_t stream) {
    int rc = head_check(M, din, dmid, dout, w1, b1, w2);
    if (rc != TGRU_OK) return rc;
    if (!x || !dy || !dparams || !workspace || (ln0_w && !ln0_b) || (w2 && (!ln1_w || !ln1_b || !b2)))
        return TGRU_E_NULL;
    const head::GradLayout gl = head::grad_layout(din, dmid, dout);
    head::Args a{};
    a.x = x; a.dy = dy; a.dx = dx; a.g0 = ln0_w; a.b0 = ln0_b; a.w1 = w1; a.c1 = b1; a.g1 = ln1_w;
    a.b1 = ln1_b; a.w2 = w2; a.c2 = b2; a.M = M; a.din = din; a.dmid = dmid; a.dout = dout; a.eps = eps;
    a.slabs = workspace; a.slab_stride = gl.total;
    const int nwg = head::workgroups(M);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head::head_bwd_kernel, dim3(nwg), dim3(64), 0, s, a);
    if (hipGetLastError() != hipSuccess) return TGRU_E_LAUNCH;
    hipLaunchKernelGGL(head::head_reduce_kernel, dim3((gl.total + 31) / 32), dim3(256), 0, s, workspace,
                       (long)gl.total, nwg, dparams, gl.total, accumulate);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

}  // extern "C"
