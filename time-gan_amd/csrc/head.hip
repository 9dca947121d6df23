// Fused per-row "head" of the TimeGAN networks: the small dense stages that sit before and after
// every GRU stack (outdated_files/TimeGANv1.py:39-50,55,67-68,119-128,145-148,226-227,276-287,294,
// 305-306,382-383),
//
//      y = [ LayerNorm0 ] -> Linear1 -> [ LayerNorm1 -> ReLU -> Linear2 ]
//
// over M independent rows, input and output at most 32 wide, the middle at most 64, forward and
// backward, one kernel each (PyTorch runs 5-8 kernels forward and about 20 backward for the same
// thing, each shorter than its launch).
//
// A wave here is alone on its SIMD and issues one instruction every ~5 cycles, so its time is its
// instruction count.  One wave therefore owns only 16 rows and spreads each row over FOUR lanes:
// lane (g, j) = (lane >> 4, lane & 15) holds, for row j, the features 16*ft + 4*g + r (r = 0..3) of
// every 16-feature tile ft -- a 32-wide vector is 8 registers, and every elementwise / LayerNorm
// step costs 8 instructions.  A Linear is computed TRANSPOSED on v_mfma_f32_16x16x4_f32,
//      Y^T (feature i, row j) = W (i, k) . X^T (k, row j),
// with W as the A operand and the activations as the B operand.  The reduction index of an MFMA
// step is free to be permuted, and step s of lane group kg is made to stand for feature
// 16*(s/4) + 4*kg + s%4 -- which is exactly register s of that lane.  So the result registers of one
// Linear ARE the B operands of the next, and a chain of Linears, LayerNorms (in-lane sum + two
// lane-group swaps) and ReLUs runs in registers with no shuffles.  Backward recomputes the forward
// from x, pushes gradients back through the same scheme with transposed weight fragments, and forms
// the weight gradients dW = dY^T . A as MFMA products contracted over the 16 rows (operands
// transposed through LDS).  Parameter gradients leave the kernel as one partial per workgroup,
// summed by a second kernel in a fixed order (deterministic, no atomics).
#include "../../include/tgru.h"
#include "tgru_internal.h"

namespace tgru {
namespace head {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIo = 32;         // widest input / output
constexpr int kMid = 64;        // widest middle
constexpr int kRows = 16;       // rows per wave
constexpr int kLd = 68;         // LDS row stride of a 16 x 64 tile (16-byte aligned rows)

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

// sum of v over the four lanes (g = 0..3) that share a row; the same bits in all four
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
    float *slabs; long slab_stride;           // backward: one partial gradient vector per workgroup
    int M, din, dmid, dout;
    int vec_in, vec_out;                      // rows of x/dx (y/dy) may be moved as aligned float4
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

// ---- registers <-> memory ------------------------------------------------------------------------------
// v[4*ft + r] = p[16*ft + 4*g + r] for features below n, `fill` beyond (p may be null)
template <int NT>
__device__ __forceinline__ void load_vec(const float *p, int n, int g, float fill, float (&v)[4 * NT]) {
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
        const int f = 16 * (q >> 2) + 4 * g + (q & 3);
        v[q] = (p && f < n) ? p[f] : fill;
    }
}
template <int NT>
__device__ __forceinline__ void valid_mask(int n, int g, float (&v)[4 * NT]) {
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) v[q] = (16 * (q >> 2) + 4 * g + (q & 3)) < n ? 1.f : 0.f;
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

// ---- Linear layers ---------------------------------------------------------------------------------------
// Weight fragments are fetched apart from their use, so that a kernel can issue every parameter load
// before the first row arrives and have the latencies overlap.
// out = W in, W (rows x cols) row-major; `in` covers NI tiles of the cols, `out` NO tiles of the rows.
// A operand of step s, output tile ot: lane (kg, i) -> W[16*ot + i][16*(s/4) + 4*kg + s%4].
template <int NI, int NO>
__device__ __forceinline__ void load_w(const float *w, int rows, int cols, int lane, float (&f)[NO][4 * NI]) {
    const int i = lane & 15, kg = lane >> 4;
    const bool vec = (cols & 3) == 0;     // weight rows then start 16-byte aligned (the base pointer is)
#pragma unroll
    for (int ot = 0; ot < NO; ++ot) {
        const int o = 16 * ot + i;
#pragma unroll
        for (int ft = 0; ft < NI; ++ft) {
            const int k0 = 16 * ft + 4 * kg;
            if (vec) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (w && o < rows && k0 < cols) t = *(const float4 *)(w + o * cols + k0);
                f[ot][4 * ft + 0] = t.x; f[ot][4 * ft + 1] = t.y; f[ot][4 * ft + 2] = t.z; f[ot][4 * ft + 3] = t.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f[ot][4 * ft + r] = (w && o < rows && k0 + r < cols) ? w[o * cols + k0 + r] : 0.f;
            }
        }
    }
}
// fragments of out = W^T in, W (rows x cols): `in` covers NI tiles of the rows, `out` NO tiles of the cols.
// A operand of step s, output tile kt: lane (ig, k) -> W[16*(s/4) + 4*ig + s%4][16*kt + k].
template <int NI, int NO>
__device__ __forceinline__ void load_wt(const float *w, int rows, int cols, int lane, float (&f)[NO][4 * NI]) {
    const int k = lane & 15, ig = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < NO; ++kt) {
        const int c = 16 * kt + k;
#pragma unroll
        for (int q = 0; q < 4 * NI; ++q) {
            const int i = 16 * (q >> 2) + 4 * ig + (q & 3);
            f[kt][q] = (w && i < rows && c < cols) ? w[i * cols + c] : 0.f;
        }
    }
}
// out tile t = sum over steps of fragment x in; nout / nin: live widths (whole tiles beyond them are skipped)
template <int NI, int NO>
__device__ __forceinline__ void apply_w(const float (&f)[NO][4 * NI], int nout, int nin, const float (&in)[4 * NI],
                                        float (&out)[4 * NO]) {
#pragma unroll
    for (int ot = 0; ot < NO; ++ot) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (16 * ot < nout) {
#pragma unroll
            for (int ft = 0; ft < NI; ++ft)
                if (16 * ft < nin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma(f[ot][4 * ft + r], in[4 * ft + r], acc);
                }
        }
        mfma_drain();
#pragma unroll
        for (int r = 0; r < 4; ++r) out[4 * ot + r] = acc[r];
    }
}

// LayerNorm statistics of one row spread over its four lanes; entries beyond d are zero on entry
template <int NT>
__device__ __forceinline__ void ln_stats(const float (&v)[4 * NT], const float (&valid)[4 * NT], int d, float eps,
                                         float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) s += v[q];
    mean = row_sum(s) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
        const float c = (v[q] - mean) * valid[q];
        sq += c * c;
    }
    rstd = rsqrtf(row_sum(sq) / (float)d + eps);
}

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
// A launch serves one head or a PAIR of independent heads: workgroups [0, nwg0) the first, the others
// the second (a single head is compiled apart so its argument reads stay at fixed offsets).
template <int NM, bool PAIR>
__global__ __launch_bounds__(64) void head_fwd_kernel(VecPair<Args> ga) {
    const int set = PAIR && (int)blockIdx.x >= ga.nwg0;
    const Args &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x, nwg = set ? ga.nwg1 : ga.nwg0;
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    const bool deep = a.w2 != nullptr;
    // every parameter is requested before the first row: their latencies overlap with the row's
    float w1f[NM][8], w2f[2][4 * NM], g0[8], b0[8], c1[4 * NM], g1[4 * NM], b1[4 * NM], c2[8];
    load_w<2, NM>(a.w1, a.dmid, a.din, lane, w1f);
    load_w<NM, 2>(a.w2, a.dout, a.dmid, lane, w2f);
    load_vec<2>(a.g0, a.din, g, 0.f, g0);
    load_vec<2>(a.b0, a.din, g, 0.f, b0);
    load_vec<NM>(a.c1, a.dmid, g, 0.f, c1);
    load_vec<NM>(a.g1, a.dmid, g, 0.f, g1);
    load_vec<NM>(a.b1, a.dmid, g, 0.f, b1);
    load_vec<2>(a.c2, a.dout, g, 0.f, c2);
    const int ntiles = (a.M + kRows - 1) / kRows;
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const long row = (long)tile * kRows + j;
        float x[8];
        load_rows<2>(a.x, a.M, a.din, row, g, a.vec_in, x);
        if (a.g0) {
            float vin[8], mean, rstd;
            valid_mask<2>(a.din, g, vin);
            ln_stats<2>(x, vin, a.din, a.eps, mean, rstd);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = ((x[q] - mean) * rstd * g0[q] + b0[q]) * vin[q];
        }
        float h[4 * NM];
        apply_w<2, NM>(w1f, a.dmid, a.din, x, h);
#pragma unroll
        for (int q = 0; q < 4 * NM; ++q) h[q] += c1[q];
        float y[8];
        if (deep) {
            float vmid[4 * NM], mean, rstd;
            valid_mask<NM>(a.dmid, g, vmid);
            ln_stats<NM>(h, vmid, a.dmid, a.eps, mean, rstd);
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) h[q] = fmaxf(((h[q] - mean) * rstd * g1[q] + b1[q]) * vmid[q], 0.f);
            apply_w<NM, 2>(w2f, a.dout, a.dmid, h, y);
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] += c2[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = h[q];   // a shallow head is at most 32 wide
        }
        store_rows<2>(a.y, a.M, a.dout, row, g, a.vec_out, y);
    }
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
// registers of 16 rows -> LDS tile[row][feature]
template <int NT>
__device__ __forceinline__ void tile_put(float *tile, int j, int g, const float (&v)[4 * NT]) {
#pragma unroll
    for (int ft = 0; ft < NT; ++ft)
        *(float4 *)(tile + j * kLd + 16 * ft + 4 * g) = make_float4(v[4 * ft], v[4 * ft + 1], v[4 * ft + 2], v[4 * ft + 3]);
}
// acc[ot][kt] (o, k) += sum over the 16 rows of u[row][o] * v[row][k]; u, v given as LDS tiles.
// step s contracts rows 4s..4s+3: lane (kg, i) reads row 4s + kg, feature 16*tile + i.
template <int NU, int NV>
__device__ __forceinline__ void outer_acc(const float *tu, const float *tv, int du, int dv, int lane,
                                          f32x4 (&acc)[NU][NV]) {
    const int i = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float ua[NU], vb[NV];
#pragma unroll
        for (int t = 0; t < NU; ++t) ua[t] = tu[(4 * s + kg) * kLd + 16 * t + i];
#pragma unroll
        for (int t = 0; t < NV; ++t) vb[t] = tv[(4 * s + kg) * kLd + 16 * t + i];
#pragma unroll
        for (int ot = 0; ot < NU; ++ot)
#pragma unroll
            for (int kt = 0; kt < NV; ++kt)
                if (16 * ot < du && 16 * kt < dv) acc[ot][kt] = mfma(ua[ot], vb[kt], acc[ot][kt]);
    }
    mfma_drain();
}
// dW (o, k): lane (g, jj), register r of acc[ot][kt] -> o = 16*ot + 4*g + r, k = 16*kt + jj.
// Buffer stores: the lane part of the address is one register, the (ot, kt, r) part a scalar offset,
// and an element outside the matrix gets an out-of-range lane offset, which the hardware drops.
template <int NU, int NV>
__device__ __forceinline__ void store_dw(float *dst, int rows, int cols, int lane, const f32x4 (&acc)[NU][NV]) {
    constexpr int kOOB = 0x7ffffff0;
    const int jj = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, rows * cols * 4, 0x00020000);
    const int lane_off = (4 * g * cols + jj) * 4;
#pragma unroll
    for (int ot = 0; ot < NU; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ook = 16 * ot + 4 * g + r < rows;
#pragma unroll
            for (int kt = 0; kt < NV; ++kt) {
                const bool ok = ook && 16 * kt + jj < cols;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[ot][kt][r]), rs, ok ? lane_off : kOOB,
                                                      ((16 * ot + r) * cols + 16 * kt) * 4, 0);
            }
        }
}

template <int NM, bool PAIR>
__global__ __launch_bounds__(64) void head_bwd_kernel(VecPair<Args> ga) {
    const int set = PAIR && (int)blockIdx.x >= ga.nwg0;
    const Args &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x, nwg = set ? ga.nwg1 : ga.nwg0;
    __shared__ __align__(16) float tiles[6][kRows * kLd];   // two operand tiles; all six for the final column sums
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    const bool deep = a.w2 != nullptr;
    const bool wantp = a.slabs != nullptr;       // false: dx only (no parameter gradients wanted by the caller)
    f32x4 dW1[NM][2], dW2[2][NM];
    float dc1[4 * NM], dg1[4 * NM], db1[4 * NM], dg0[8], db0[8], dc2[8];
#pragma unroll
    for (int t = 0; t < NM; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) dW1[t][u] = dW2[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4 * NM; ++q) dc1[q] = dg1[q] = db1[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) dg0[q] = db0[q] = dc2[q] = 0.f;

    // With a middle of at most 32 every weight fragment fits in registers next to the accumulators and is
    // requested before the first row, so the latencies overlap; the 64-wide variant fetches each
    // fragment set where it is used.
    constexpr bool kPre = NM == 2;
    float w1f[NM][8], w1t[2][4 * NM], w2t[NM][8], g0[8], b0[8], vin[8];
    if (kPre) {
        load_w<2, NM>(a.w1, a.dmid, a.din, lane, w1f);
        load_wt<NM, 2>(a.w1, a.dmid, a.din, lane, w1t);
        load_wt<2, NM>(a.w2, a.dout, a.dmid, lane, w2t);
    }
    load_vec<2>(a.g0, a.din, g, 0.f, g0);
    load_vec<2>(a.b0, a.din, g, 0.f, b0);
    valid_mask<2>(a.din, g, vin);

    const int ntiles = (a.M + kRows - 1) / kRows;
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const long row = (long)tile * kRows + j;
        // ---- forward again, keeping what the backward needs --------------------------------------
        float x1[8], xh0[8], rstd0 = 1.f;
        load_rows<2>(a.x, a.M, a.din, row, g, a.vec_in, x1);
        if (a.g0) {
            float mean;
            ln_stats<2>(x1, vin, a.din, a.eps, mean, rstd0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                xh0[q] = (x1[q] - mean) * rstd0 * vin[q];
                x1[q] = (xh0[q] * g0[q] + b0[q]) * vin[q];
            }
        }
        float dh[4 * NM];             // gradient at the output of Linear1
        if (deep) {
            float hn[4 * NM], act[4 * NM], g1[4 * NM], vmid[4 * NM], rstd1, mean1;
            {
                float c1[4 * NM], b1[4 * NM];
                if (!kPre) load_w<2, NM>(a.w1, a.dmid, a.din, lane, w1f);
                apply_w<2, NM>(w1f, a.dmid, a.din, x1, hn);
                load_vec<NM>(a.c1, a.dmid, g, 0.f, c1);
                load_vec<NM>(a.g1, a.dmid, g, 0.f, g1);
                load_vec<NM>(a.b1, a.dmid, g, 0.f, b1);
                valid_mask<NM>(a.dmid, g, vmid);
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) hn[q] += c1[q];
                ln_stats<NM>(hn, vmid, a.dmid, a.eps, mean1, rstd1);
#pragma unroll
                for (int q = 0; q < 4 * NM; ++q) {
                    hn[q] = (hn[q] - mean1) * rstd1 * vmid[q];
                    act[q] = fmaxf((hn[q] * g1[q] + b1[q]) * vmid[q], 0.f);
                }
            }
            // Linear2: dc2, dW2 = dy^T . act, da = W2^T . dy
            float dy[8], da[4 * NM];
            load_rows<2>(a.dy, a.M, a.dout, row, g, a.vec_out, dy);
#pragma unroll
            for (int q = 0; q < 8; ++q) dc2[q] += dy[q];
            if (wantp) {
                tile_put<2>(tiles[0], j, g, dy);
                tile_put<NM>(tiles[1], j, g, act);
                outer_acc<2, NM>(tiles[0], tiles[1], a.dout, a.dmid, lane, dW2);
            }
            if (!kPre) load_wt<2, NM>(a.w2, a.dout, a.dmid, lane, w2t);
            apply_w<2, NM>(w2t, a.dmid, a.dout, dy, da);
            // ReLU and LayerNorm1
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) {
                const float dpre = act[q] > 0.f ? da[q] : 0.f;
                dg1[q] += dpre * hn[q];
                db1[q] += dpre;
                da[q] = dpre * g1[q];
                s1 += da[q];
                s2 += da[q] * hn[q];
            }
            const float m1 = row_sum(s1) / (float)a.dmid, m2 = row_sum(s2) / (float)a.dmid;
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) dh[q] = rstd1 * (da[q] - m1 - hn[q] * m2) * vmid[q];
        } else {
            float dy[8];
            load_rows<2>(a.dy, a.M, a.dout, row, g, a.vec_out, dy);
#pragma unroll
            for (int q = 0; q < 4 * NM; ++q) dh[q] = q < 8 ? dy[q] : 0.f;
        }
        // ---- Linear1: dc1, dW1 = dh^T . x1, dx1 = W1^T . dh ------------------------------------------
#pragma unroll
        for (int q = 0; q < 4 * NM; ++q) dc1[q] += dh[q];
        if (wantp) {
            tile_put<NM>(tiles[0], j, g, dh);
            tile_put<2>(tiles[1], j, g, x1);
            outer_acc<NM, 2>(tiles[0], tiles[1], a.dmid, a.din, lane, dW1);
        }
        if (a.dx || a.g0) {
            float dx[8];
            if (!kPre) load_wt<NM, 2>(a.w1, a.dmid, a.din, lane, w1t);
            apply_w<NM, 2>(w1t, a.din, a.dmid, dh, dx);
            if (a.g0) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    dg0[q] += dx[q] * xh0[q];
                    db0[q] += dx[q];
                    dx[q] *= g0[q];
                    s1 += dx[q];
                    s2 += dx[q] * xh0[q];
                }
                const float m1 = row_sum(s1) / (float)a.din, m2 = row_sum(s2) / (float)a.din;
#pragma unroll
                for (int q = 0; q < 8; ++q) dx[q] = rstd0 * (dx[q] - m1 - xh0[q] * m2) * vin[q];
            }
            if (a.dx) store_rows<2>(a.dx, a.M, a.din, row, g, a.vec_in, dx);
        }
    }

    // ---- this workgroup's partial parameter gradients -------------------------------------------------
    if (!wantp) return;
    const GradLayout gl = grad_layout(a.din, a.dmid, a.dout);
    float *slab = a.slabs + (long)wg * a.slab_stride;
    store_dw<NM, 2>(slab + gl.w1, a.dmid, a.din, lane, dW1);
    store_dw<2, NM>(slab + gl.w2, a.dout, a.dmid, lane, dW2);     // zeros for a shallow head
    // bias / LayerNorm-parameter gradients: the six per-lane accumulator sets go to LDS together and every
    // lane f sums feature f over the 16 rows (one LDS round trip for all of them)
    tile_put<NM>(tiles[0], j, g, dc1);
    tile_put<NM>(tiles[1], j, g, dg1);
    tile_put<NM>(tiles[2], j, g, db1);
    tile_put<2>(tiles[3], j, g, dg0);
    tile_put<2>(tiles[4], j, g, db0);
    tile_put<2>(tiles[5], j, g, dc2);
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kRows; ++q)
#pragma unroll
        for (int v = 0; v < 6; ++v) s[v] += tiles[v][q * kLd + lane];
    if (lane < a.dmid) {
        slab[gl.c1 + lane] = s[0];
        slab[gl.g1 + lane] = s[1];
        slab[gl.b1 + lane] = s[2];
    }
    if (lane < a.din) {
        slab[gl.g0 + lane] = s[3];
        slab[gl.b0 + lane] = s[4];
    }
    if (lane < a.dout) slab[gl.c2 + lane] = s[5];
}

// out[i] (=|+=) sum over workgroups of slabs[k * stride + i]: 32 elements x 32 slab groups per workgroup,
// so a thread sums only a handful of partials with all its loads in flight together; fixed order.
constexpr int kRedGroups = 32;
struct ReduceArgs {
    const float *slabs; long stride; int nslabs;
    float *out; int n, accumulate;
};
__global__ __launch_bounds__(32 * kRedGroups) void head_reduce_kernel(VecPair<ReduceArgs> ga) {
    __shared__ float part[kRedGroups][32];
    const int set = (int)blockIdx.x >= ga.nwg0;
    const ReduceArgs &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x;
    const float *slabs = a.slabs;
    const long stride = a.stride;
    const int nslabs = a.nslabs, n = a.n;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = wg * 32 + e;
    const int per = (nslabs + kRedGroups - 1) / kRedGroups;
    const int c0 = grp * per, c1 = min(nslabs, c0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (idx < n) {
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
    if (grp != 0 || idx >= n) return;
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
    for (int g = 0; g < kRedGroups; g += 4) {
        u0 += part[g][e];
        u1 += part[g + 1][e];
        u2 += part[g + 2][e];
        u3 += part[g + 3][e];
    }
    const float v = (u0 + u1) + (u2 + u3);
    a.out[idx] = a.accumulate ? a.out[idx] + v : v;
}

int workgroups(int M) {
    const int tiles = (M + kRows - 1) / kRows;
    return tiles < 2048 ? tiles : 2048;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace head
}  // namespace tgru

using namespace tgru;

namespace {

// beyond the widths of the register-resident kernels: head_wide.hip
bool is_wide(const tgru_head_args &c) { return c.din > head::kIo || c.dout > head::kIo || c.dmid > head::kMid; }

int head_check(const tgru_head_args &c, bool backward) {
    if (c.M <= 0 || c.din <= 0 || c.dmid <= 0 || c.dout <= 0) return TGRU_E_DIMS;
    if (is_wide(c) && !head_wide_supported(c.din, c.dmid, c.dout, c.w2 != nullptr)) return TGRU_E_DIMS;
    if (!c.w2 && c.dout != c.dmid) return TGRU_E_DIMS;
    if ((long long)c.M * 64 >= (1LL << 31)) return TGRU_E_DIMS;
    if (!c.w1 || !c.b1 || !c.x) return TGRU_E_NULL;
    if ((c.ln0_w && !c.ln0_b) || (c.w2 && (!c.ln1_w || !c.ln1_b || !c.b2))) return TGRU_E_NULL;
    if (backward ? (!c.dy || !c.workspace) : !c.y) return TGRU_E_NULL;     // dparams NULL: dx only
    if (!head::aligned16(c.w1) || (c.w2 && !head::aligned16(c.w2))) return TGRU_E_ALIGN;
    if (backward && is_wide(c) && !head::aligned16(c.workspace)) return TGRU_E_ALIGN;   // rows moved as float4
    return TGRU_OK;
}

void head_fill(const tgru_head_args &c, bool backward, head::Args &a) {
    a.x = c.x; a.g0 = c.ln0_w; a.b0 = c.ln0_b; a.w1 = c.w1; a.c1 = c.b1; a.g1 = c.ln1_w; a.b1 = c.ln1_b;
    a.w2 = c.w2; a.c2 = c.b2; a.M = c.M; a.din = c.din; a.dmid = c.dmid; a.dout = c.dout; a.eps = c.eps;
    if (backward) {
        a.dy = c.dy; a.dx = c.dx;
        a.vec_in = (c.din % 4 == 0) && head::aligned16(c.x) && (!c.dx || head::aligned16(c.dx));
        a.vec_out = (c.dout % 4 == 0) && head::aligned16(c.dy);
        a.slabs = c.dparams ? c.workspace : nullptr; a.slab_stride = head::grad_layout(c.din, c.dmid, c.dout).total;
    } else {
        a.y = c.y;
        a.vec_in = (c.din % 4 == 0) && head::aligned16(c.x);
        a.vec_out = (c.dout % 4 == 0) && head::aligned16(c.y);
    }
}

template <bool BWD>
hipError_t head_launch(const head::Args &a, const head::Args *second, hipStream_t s) {
    VecPair<head::Args> ga{};
    ga.g[0] = a;
    ga.nwg0 = head::workgroups(a.M);
    int dmid = a.dmid;
    if (second) {
        ga.g[1] = *second;
        ga.nwg1 = head::workgroups(second->M);
        if (second->dmid > dmid) dmid = second->dmid;
    }
    const dim3 grid(ga.nwg0 + ga.nwg1), block(64);
    const bool wide = dmid > 32;
#define TGRU_HEAD_LAUNCH(K)                                                                       \
    do {                                                                                          \
        if (wide && second) TGRU_LAUNCH((head::K<4, true>), grid, block, 0, s, ga);        \
        else if (wide) TGRU_LAUNCH((head::K<4, false>), grid, block, 0, s, ga);            \
        else if (second) TGRU_LAUNCH((head::K<2, true>), grid, block, 0, s, ga);           \
        else TGRU_LAUNCH((head::K<2, false>), grid, block, 0, s, ga);                      \
    } while (0)
    if (BWD) TGRU_HEAD_LAUNCH(head_bwd_kernel);
    else TGRU_HEAD_LAUNCH(head_fwd_kernel);
#undef TGRU_HEAD_LAUNCH
    return hipGetLastError();
}

hipError_t head_reduce_launch(const tgru_head_args &a, const tgru_head_args *second, hipStream_t s) {
    VecPair<head::ReduceArgs> ga{};
    const tgru_head_args *both[2] = {&a, second};
    int nwg[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        if (!both[k]) continue;
        const tgru_head_args &c = *both[k];
        const int total = head::grad_layout(c.din, c.dmid, c.dout).total;
        ga.g[k] = head::ReduceArgs{c.workspace, (long)total, head::workgroups(c.M), c.dparams, total, c.accumulate};
        nwg[k] = (total + 31) / 32;
    }
    ga.nwg0 = nwg[0]; ga.nwg1 = nwg[1];
    TGRU_LAUNCH(head::head_reduce_kernel, dim3(nwg[0] + nwg[1]), dim3(32 * head::kRedGroups), 0, s, ga);
    return hipGetLastError();
}

int head_forward_impl(const tgru_head_args *a, const tgru_head_args *b, hipStream_t s) {
    int rc = head_check(*a, false);
    if (rc == TGRU_OK && b) rc = head_check(*b, false);
    if (rc != TGRU_OK) return rc;
    if (is_wide(*a) || (b && is_wide(*b))) {     // no paired launch: one call after the other
        const tgru_head_args *both[2] = {a, b};
        for (int k = 0; k < 2 && both[k]; ++k) {
            rc = is_wide(*both[k]) ? head_wide_forward(*both[k], s) : head_forward_impl(both[k], nullptr, s);
            if (rc != TGRU_OK) return rc;
        }
        return TGRU_OK;
    }
    head::Args fa{}, fb{};
    head_fill(*a, false, fa);
    if (b) head_fill(*b, false, fb);
    return head_launch<false>(fa, b ? &fb : nullptr, s) == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int head_backward_impl(const tgru_head_args *a, const tgru_head_args *b, hipStream_t s) {
    int rc = head_check(*a, true);
    if (rc == TGRU_OK && b) rc = head_check(*b, true);
    if (rc != TGRU_OK) return rc;
    if (is_wide(*a) || (b && is_wide(*b))) {
        const tgru_head_args *both[2] = {a, b};
        for (int k = 0; k < 2 && both[k]; ++k) {
            rc = is_wide(*both[k]) ? head_wide_backward(*both[k], s) : head_backward_impl(both[k], nullptr, s);
            if (rc != TGRU_OK) return rc;
        }
        return TGRU_OK;
    }
    head::Args fa{}, fb{};
    head_fill(*a, true, fa);
    if (b) head_fill(*b, true, fb);
    if (head_launch<true>(fa, b ? &fb : nullptr, s) != hipSuccess) return TGRU_E_LAUNCH;
    // reductions that are not deferred to a later tgru_reduce_jobs launch
    const tgru_head_args *now[2] = {nullptr, nullptr};
    int n = 0;
    if (a->accumulate != TGRU_DEFER_REDUCE && a->dparams) now[n++] = a;
    if (b && b->accumulate != TGRU_DEFER_REDUCE && b->dparams) now[n++] = b;
    if (n == 0) return TGRU_OK;
    return head_reduce_launch(*now[0], now[1], s) == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

}  // namespace

extern "C" {

size_t tgru_head_grad_floats(int din, int dmid, int dout) {
    return (size_t)head::grad_layout(din, dmid, dout).total;
}

size_t tgru_head_backward_workspace_bytes(int M, int din, int dmid, int dout) {
    if (M <= 0) return 0;
    if (din > head::kIo || dout > head::kIo || dmid > head::kMid)
        return head_wide_workspace_floats(M, din, dmid, dout) * sizeof(float);
    return (size_t)head::workgroups(M) * head::grad_layout(din, dmid, dout).total * sizeof(float);
}

int tgru_head_supported(int din, int dmid, int dout, int deep) {
    if (din <= 0 || dmid <= 0 || dout <= 0 || (!deep && dmid != dout)) return 0;
    if (din <= head::kIo && dout <= head::kIo && dmid <= head::kMid) return 1;
    return head_wide_supported(din, dmid, dout, deep != 0) ? 1 : 0;
}

int tgru_head_forward(const float *x, float *y, int M, int din, int dmid, int dout, const float *ln0_w,
                      const float *ln0_b, const float *w1, const float *b1, const float *ln1_w,
                      const float *ln1_b, const float *w2, const float *b2, float eps, tgru_stream_t stream) {
    tgru_head_args c{};
    c.x = x; c.y = y; c.M = M; c.din = din; c.dmid = dmid; c.dout = dout; c.ln0_w = ln0_w; c.ln0_b = ln0_b;
    c.w1 = w1; c.b1 = b1; c.ln1_w = ln1_w; c.ln1_b = ln1_b; c.w2 = w2; c.b2 = b2; c.eps = eps;
    return head_forward_impl(&c, nullptr, (hipStream_t)stream);
}

int tgru_head_backward(const float *x, const float *dy, float *dx, int M, int din, int dmid, int dout,
                       const float *ln0_w, const float *ln0_b, const float *w1, const float *b1,
                       const float *ln1_w, const float *ln1_b, const float *w2, const float *b2, float eps,
                       float *dparams, float *workspace, int accumulate, tgru_stream_t stream) {
    tgru_head_args c{};
    c.x = x; c.dy = dy; c.dx = dx; c.M = M; c.din = din; c.dmid = dmid; c.dout = dout; c.ln0_w = ln0_w;
    c.ln0_b = ln0_b; c.w1 = w1; c.b1 = b1; c.ln1_w = ln1_w; c.ln1_b = ln1_b; c.w2 = w2; c.b2 = b2; c.eps = eps;
    c.dparams = dparams; c.workspace = workspace; c.accumulate = accumulate;
    return head_backward_impl(&c, nullptr, (hipStream_t)stream);
}

int tgru_head_reduce_jobs(const tgru_head_args *c, tgru_reduce_job *jobs, int max_jobs) {
    if (!c || !jobs) return TGRU_E_NULL;
    const int rc = head_check(*c, true);
    if (rc != TGRU_OK) return rc;
    if (c->accumulate != TGRU_DEFER_REDUCE || !c->dparams) return 0;
    if (is_wide(*c)) return head_wide_reduce_jobs(*c, jobs, max_jobs);
    if (max_jobs < 1) return TGRU_E_DIMS;
    const int total = head::grad_layout(c->din, c->dmid, c->dout).total;
    jobs[0] = tgru_reduce_job{c->workspace, (int64_t)total, head::workgroups(c->M), total, c->dparams};
    return 1;
}

int tgru_head_forward_pair(const tgru_head_args *a, const tgru_head_args *b, tgru_stream_t stream) {
    if (!a || !b) return TGRU_E_NULL;
    return head_forward_impl(a, b, (hipStream_t)stream);
}

int tgru_head_backward_pair(const tgru_head_args *a, const tgru_head_args *b, tgru_stream_t stream) {
    if (!a || !b) return TGRU_E_NULL;
    return head_backward_impl(a, b, (hipStream_t)stream);
}

}  // extern "C"
