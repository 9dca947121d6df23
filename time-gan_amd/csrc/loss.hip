// The five loss expressions of TimeGAN training (outdated_files/TimeGANv1.py:410-553), each as
//      partial sums  ->  [centered second moments]  ->  final scalar(s)  ->  gradients
// in two or three short kernels forward and one backward, where PyTorch's composition of mse_loss /
// binary_cross_entropy_with_logits / mean / std / abs / sqrt runs 30 to 100 kernels per loss, every
// one shorter than its launch.
//
// A loss is a sum of at most three elementwise TERMS plus, for the generator, the moment term:
//   kSq       sum (a - b)^2                                   mse_loss(a, b)
//   kSqShift  sum (a[:, t, :] - b[:, t+1, :])^2, t < T-1      mse_loss(h_sup[:, :-1, :], h[:, 1:, :])
//   kBceOne   sum softplus(-y)                                BCE-with-logits against ones
//   kBceZero  sum softplus(+y)                                BCE-with-logits against zeros
//   moment    per column c of the (B, T*F) views of x_hat and x: mean and population std over the
//             batch -> mean_c |(s_hat + 1e-6) - s_x + 1e-6| + mean_c |m_hat - m_x|   (:543-548)
//
// All sums are taken in a fixed order (256 strided partial sums per term, then a tree), so results
// are deterministic.  The buffer `buf` carries everything between the kernels:
//      part A  counts[4] | partial[3][256] | colsum[2][C]      (element counts are stored as sums,
//      part B  colsq[2][C]                                       so that adding the part-A / part-B
//      stats   out[3] | mean-terms[4] | colstat[4][C]            arrays of several data-parallel
// ranks -- an all-reduce between the kernels -- yields the loss of the concatenated batch; that is
// also why the layout does not depend on B).  C = T*F for the generator loss, 0 otherwise.
#include "../../include/tgru.h"
#include "tgru_internal.h"
#include <algorithm>

namespace tgru {
namespace loss {

constexpr int kSlots = 256;          // partial sums (= workgroups) per term
constexpr int kThreads = 256;
constexpr int kCols = 16;            // columns per moment workgroup (x 16 row lanes)

enum TermType { kNone = 0, kSq = 1, kSqShift = 2, kBceOne = 3, kBceZero = 4 };

// e / d for e < 2^30 and d <= 2^9 by a multiplication with m = ceil(2^40 / d) (exact while e * d < 2^40): the
// kernels divide an element index by a width per ELEMENT, and a hardware-less 32-bit division is ~40
// instructions, more than everything else they do per element
__host__ __device__ inline unsigned long long magic_of(unsigned d) { return ((1ULL << 40) + d - 1) / d; }
__device__ __forceinline__ unsigned fast_div(unsigned e, unsigned long long m) { return (unsigned)(((unsigned long long)e * m) >> 40); }

struct Layout {
    long cnt, partial, colsum, colsq, out, mean, colstat, nA, nB, total;
};

__host__ __device__ inline Layout layout_of(int C) {
    Layout l;
    l.cnt = 0;
    l.partial = 4;
    l.colsum = l.partial + 3 * kSlots;
    l.nA = l.colsum + 2L * C;
    l.colsq = l.nA;
    l.nB = 2L * C;
    l.out = l.colsq + l.nB;
    l.mean = l.out + 3;
    l.colstat = l.mean + 5;    // 4 mean-terms + 1 pad: colstat starts on a multiple of 4 floats (float4 reads)
    l.total = l.colstat + 4L * C;
    return l;
}

struct Args {
    int kind, pass, B, T, H, C;
    int small;               // every element index fits in 30 bits, widths in 9: 32-bit indices, divisions by multiplication
    unsigned long long mH, mT, mTH1, mC;     // magic_of(H), (T), ((T-1)*H), (C)
    int type[3];
    const float *a[3];
    const float *b[3];
    float *da[3];
    float *db[3];
    long n[3];               // forward element count of the term
    const float *xh, *x;     // moment term: (B, C) each
    float *dxh;
    float gamma;
    float *buf;
    const float *g[3];       // upstream gradients of the outputs (backward), NULL = none
};

__device__ __forceinline__ float softplus_neg(float y) {   // -log_sigmoid(y) = softplus(-y)
    return log1pf(expf(-fabsf(y))) - fminf(y, 0.f);
}

// I: the index type -- 32-bit wherever the tensors allow (a 64-bit division per element costs more than
// everything else these kernels do)
template <typename I>
__device__ __forceinline__ float term_value(int type, const float *a, const float *b, I e, int T, int H) {
    if (type == kSq) {
        const float d = a[e] - b[e];
        return d * d;
    }
    if (type == kSqShift) {
        const I th1 = (I)(T - 1) * (I)H;
        const I bb = e / th1;
        const I ia = bb * (I)T * (I)H + (e - bb * th1);
        const float d = a[ia] - b[ia + H];
        return d * d;
    }
    const float y = a[e];
    return type == kBceOne ? softplus_neg(y) : softplus_neg(-y);
}

// sum over the 256 threads of a workgroup, same association every time; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// pass 0: term sums + column sums + centered column squares (one process holds the whole batch)
// pass 1: term sums + column sums only
// pass 2: centered column squares about the mean colsum / counts[3] read from buf (moment blocks only)
__global__ __launch_bounds__(kThreads) void loss_partial_kernel(Args p) {
    __shared__ float red[4];
    __shared__ float cred[2][16][kCols + 1];
    const Layout l = layout_of(p.C);
    const int tid = threadIdx.x;
    const int nterm_blocks = p.pass == 2 ? 0 : 3 * kSlots;
    if ((int)blockIdx.x < nterm_blocks) {
        const int k = blockIdx.x / kSlots, j = blockIdx.x % kSlots;
        float s = 0.f;
        if (p.type[k] != kNone) {
            if (p.small && p.type[k] == kSqShift) {
                const unsigned n = (unsigned)p.n[k], th1 = (unsigned)(p.T - 1) * p.H, th = (unsigned)p.T * p.H;
                const float *a = p.a[k], *b = p.b[k];
                for (unsigned e = (unsigned)j * kThreads + tid; e < n; e += kSlots * kThreads) {
                    const unsigned bb = fast_div(e, p.mTH1);
                    const unsigned ia = bb * th + (e - bb * th1);
                    const float d = a[ia] - b[ia + p.H];
                    s += d * d;
                }
            } else if (p.small) {
                const unsigned n = (unsigned)p.n[k];
                for (unsigned e = (unsigned)j * kThreads + tid; e < n; e += kSlots * kThreads)
                    s += term_value<unsigned>(p.type[k], p.a[k], p.b[k], e, p.T, p.H);
            } else {
                for (long e = (long)j * kThreads + tid; e < p.n[k]; e += (long)kSlots * kThreads)
                    s += term_value<long>(p.type[k], p.a[k], p.b[k], e, p.T, p.H);
            }
        }
        s = block_sum(s, red);
        if (tid == 0) p.buf[l.partial + blockIdx.x] = s;
        if (blockIdx.x == 0 && tid < 4)
            p.buf[l.cnt + tid] = tid < 3 ? (p.type[tid] != kNone ? (float)p.n[tid] : 0.f) : (p.C ? (float)p.B : 0.f);
        return;
    }
    // moment block: 16 columns x 16 row lanes
    const int cg = blockIdx.x - nterm_blocks;
    const int cl = tid & (kCols - 1), rl = tid >> 4;
    const int col = cg * kCols + cl;
    const bool ok = col < p.C;
    float mh, mx;
    if (p.pass != 2) {
        float sh = 0.f, sx = 0.f;
        if (ok) {      // four rows per round: eight loads in flight, fixed association
            float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
            int r = rl;
            for (; r + 48 < p.B; r += 64) {
                const long o = (long)r * p.C + col, d = 16L * p.C;
                h0 += p.xh[o]; h1 += p.xh[o + d]; h2 += p.xh[o + 2 * d]; h3 += p.xh[o + 3 * d];
                x0 += p.x[o]; x1 += p.x[o + d]; x2 += p.x[o + 2 * d]; x3 += p.x[o + 3 * d];
            }
            for (; r < p.B; r += 16) {
                h0 += p.xh[(long)r * p.C + col];
                x0 += p.x[(long)r * p.C + col];
            }
            sh = (h0 + h1) + (h2 + h3);
            sx = (x0 + x1) + (x2 + x3);
        }
        cred[0][rl][cl] = sh;
        cred[1][rl][cl] = sx;
        __syncthreads();
        sh = 0.f, sx = 0.f;
        for (int i = 0; i < 16; ++i) {
            sh += cred[0][i][cl];
            sx += cred[1][i][cl];
        }
        __syncthreads();
        if (ok && rl == 0) {
            p.buf[l.colsum + col] = sh;
            p.buf[l.colsum + p.C + col] = sx;
        }
        if (p.pass == 1) return;
        mh = sh / (float)p.B;
        mx = sx / (float)p.B;
    } else {
        const float nb = p.buf[l.cnt + 3];
        mh = ok ? p.buf[l.colsum + col] / nb : 0.f;
        mx = ok ? p.buf[l.colsum + p.C + col] / nb : 0.f;
    }
    float qh = 0.f, qx = 0.f;
    if (ok) {
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
        int r = rl;
        for (; r + 48 < p.B; r += 64) {
            const long o = (long)r * p.C + col, d = 16L * p.C;
            const float a0 = p.xh[o] - mh, a1 = p.xh[o + d] - mh, a2 = p.xh[o + 2 * d] - mh, a3 = p.xh[o + 3 * d] - mh;
            const float b0 = p.x[o] - mx, b1 = p.x[o + d] - mx, b2 = p.x[o + 2 * d] - mx, b3 = p.x[o + 3 * d] - mx;
            h0 += a0 * a0; h1 += a1 * a1; h2 += a2 * a2; h3 += a3 * a3;
            x0 += b0 * b0; x1 += b1 * b1; x2 += b2 * b2; x3 += b3 * b3;
        }
        for (; r < p.B; r += 16) {
            const float dh = p.xh[(long)r * p.C + col] - mh;
            const float dx = p.x[(long)r * p.C + col] - mx;
            h0 += dh * dh;
            x0 += dx * dx;
        }
        qh = (h0 + h1) + (h2 + h3);
        qx = (x0 + x1) + (x2 + x3);
    }
    cred[0][rl][cl] = qh;
    cred[1][rl][cl] = qx;
    __syncthreads();
    if (ok && rl == 0) {
        qh = 0.f, qx = 0.f;
        for (int i = 0; i < 16; ++i) {
            qh += cred[0][i][cl];
            qx += cred[1][i][cl];
        }
        p.buf[l.colsq + col] = qh;
        p.buf[l.colsq + p.C + col] = qx;
    }
}

// One workgroup: partial sums -> loss values, plus what the gradient kernel needs.
//   out[0..2]   the loss tensors the reference's function returns (unused ones 0)
//   mean[k]     S_k / N_k of term k;  mean[3] = number of batch rows N_3
//   colstat     per column: m_hat, s_hat, sign of the std difference, sign of the mean difference
__global__ __launch_bounds__(kThreads) void loss_final_kernel(Args p) {
    __shared__ float red[4];
    __shared__ float sh_s[3];
    const Layout l = layout_of(p.C);
    const int tid = threadIdx.x;
    for (int k = 0; k < 3; ++k) {
        const float s = block_sum(p.buf[l.partial + k * kSlots + tid], red);
        if (tid == 0) sh_s[k] = s;
    }
    float v1 = 0.f, v2 = 0.f;
    if (p.C) {
        const float nb = p.buf[l.cnt + 3];
        for (int c = tid; c < p.C; c += kThreads) {
            const float m_hat = p.buf[l.colsum + c] / nb, m_x = p.buf[l.colsum + p.C + c] / nb;
            const float s_hat = sqrtf(p.buf[l.colsq + c] / nb), s_x = sqrtf(p.buf[l.colsq + p.C + c] / nb);
            const float d1 = ((s_hat + 1e-6f) - s_x) + 1e-6f;   // precedence as written at TimeGANv1.py:543
            const float d2 = m_hat - m_x;
            v1 += fabsf(d1);
            v2 += fabsf(d2);
            p.buf[l.colstat + 4L * c + 0] = m_hat;
            p.buf[l.colstat + 4L * c + 1] = s_hat;
            p.buf[l.colstat + 4L * c + 2] = (float)((d1 > 0.f) - (d1 < 0.f));
            p.buf[l.colstat + 4L * c + 3] = (float)((d2 > 0.f) - (d2 < 0.f));
        }
        v1 = block_sum(v1, red);
        v2 = block_sum(v2, red);
    }
    __syncthreads();
    if (tid != 0) return;
    float m[3];
    for (int k = 0; k < 3; ++k) {
        m[k] = p.type[k] != kNone ? sh_s[k] / p.buf[l.cnt + k] : 0.f;
        p.buf[l.mean + k] = m[k];
    }
    p.buf[l.mean + 3] = p.buf[l.cnt + 3];
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    switch (p.kind) {
    case TGRU_LOSS_EMBED_SIDE:      // 10 sqrt(mse(x_tilde, x))                          (:420-421)
        o0 = 10.f * sqrtf(m[0]);
        break;
    case TGRU_LOSS_SUPERVISOR:      // mse(h_sup[:, :-1], h[:, 1:])                       (:470)
        o0 = m[0];
        break;
    case TGRU_LOSS_EMBED_MAIN:      // e_loss0 + 0.1 g_loss_sup, e_loss0, e_loss_t0      (:444-451)
        o2 = m[0];
        o1 = 10.f * sqrtf(m[0]);
        o0 = o1 + 0.1f * m[1];
        break;
    case TGRU_LOSS_DISCRIMINATOR:   // fake + real + gamma fake_e                         (:497-503)
        o0 = (m[1] + m[0]) + p.gamma * m[2];
        break;
    default:                        // generator                                          (:533-553)
        o0 = ((m[0] + p.gamma * m[1]) + 100.f * sqrtf(m[2])) + 1000.f * (v1 / (float)p.C + v2 / (float)p.C);
        break;
    }
    p.buf[l.out + 0] = o0;
    p.buf[l.out + 1] = o1;
    p.buf[l.out + 2] = o2;
}

// d out / d (sum of term k), times the upstream gradients
__device__ __forceinline__ float term_coef(const Args &p, const Layout &l, int k) {
    const float g0 = p.g[0] ? p.g[0][0] : 0.f, g1 = p.g[1] ? p.g[1][0] : 0.f, g2 = p.g[2] ? p.g[2][0] : 0.f;
    const float nk = p.buf[l.cnt + k], mk = p.buf[l.mean + k];
    switch (p.kind) {
    case TGRU_LOSS_EMBED_SIDE: return g0 * 5.f / (sqrtf(mk) * nk);
    case TGRU_LOSS_SUPERVISOR: return g0 / nk;
    case TGRU_LOSS_EMBED_MAIN: return k == 0 ? ((g0 + g1) * 5.f / sqrtf(mk) + g2) / nk : 0.1f * g0 / nk;
    case TGRU_LOSS_DISCRIMINATOR: return k == 2 ? g0 * p.gamma / nk : g0 / nk;
    default: return k == 0 ? g0 / nk : k == 1 ? g0 * p.gamma / nk : g0 * 50.f / (sqrtf(mk) * nk);
    }
}

// Gradients of every input that wants one.  Blocks [k*kSlots, (k+1)*kSlots) serve term k, the rest the
// moment term.  A shifted term covers all B*T*H elements of both operands (zeros on the edge steps).
__global__ __launch_bounds__(kThreads) void loss_backward_kernel(Args p) {
    const Layout l = layout_of(p.C);
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < 3 * kSlots) {
        const int k = blockIdx.x / kSlots, j = blockIdx.x % kSlots;
        const int type = p.type[k];
        if (type == kNone || (!p.da[k] && !p.db[k])) return;
        const float c = term_coef(p, l, k);
        // no output aliases an input: loads of later elements may pass the stores of earlier ones
        const float *__restrict__ a = p.a[k], *__restrict__ b = p.b[k];
        float *__restrict__ da = p.da[k], *__restrict__ db = p.db[k];
        const long stride = (long)kSlots * kThreads;
        if (type == kSq) {
#pragma unroll 4
            for (long e = (long)j * kThreads + tid; e < p.n[k]; e += stride) {
                const float v = 2.f * c * (a[e] - b[e]);
                if (da) da[e] = v;
                if (db) db[e] = -v;
            }
        } else if (type == kSqShift) {
            const long n = (long)p.B * p.T * p.H;
            if (p.small) {
                const unsigned H = p.H, T = p.T;
#pragma unroll 4
                for (unsigned e = (unsigned)j * kThreads + tid; e < (unsigned)n; e += (unsigned)stride) {
                    const unsigned row = fast_div(e, p.mH);
                    const unsigned t = row - fast_div(row, p.mT) * T;
                    if (da) da[e] = t < T - 1 ? 2.f * c * (a[e] - b[e + H]) : 0.f;
                    if (db) db[e] = t >= 1 ? -2.f * c * (a[e - H] - b[e]) : 0.f;
                }
            } else {
                for (long e = (long)j * kThreads + tid; e < n; e += stride) {
                    const int t = (int)((e / p.H) % p.T);
                    if (da) da[e] = t < p.T - 1 ? 2.f * c * (a[e] - b[e + p.H]) : 0.f;
                    if (db) db[e] = t >= 1 ? -2.f * c * (a[e - p.H] - b[e]) : 0.f;
                }
            }
        } else if (da) {
            const float tgt = type == kBceOne ? 1.f : 0.f;
            for (long e = (long)j * kThreads + tid; e < p.n[k]; e += stride)
                da[e] = c * (1.f / (1.f + expf(-a[e])) - tgt);
        }
        return;
    }
    if (!p.dxh) return;
    // d/dx_hat[b,c] of 1000/C sum_c (|s_hat - s_x + 2e-6| + |m_hat - m_x|), population std over the batch
    const float g0 = p.g[0] ? p.g[0][0] : 0.f;
    const float nb = p.buf[l.mean + 3];
    const float cm = g0 * 1000.f / (float)p.C / nb;
    const long n = (long)p.B * p.C;
    const int nmb = gridDim.x - 3 * kSlots;
    if (p.small) {
        for (unsigned e = (blockIdx.x - 3 * kSlots) * kThreads + tid; e < (unsigned)n; e += (unsigned)nmb * kThreads) {
            const float4 st = *(const float4 *)(p.buf + l.colstat + 4L * (e - fast_div(e, p.mC) * (unsigned)p.C));
            p.dxh[e] = cm * (st.w + st.z * (p.xh[e] - st.x) / st.y);
        }
        return;
    }
    for (long e = (long)(blockIdx.x - 3 * kSlots) * kThreads + tid; e < n; e += (long)nmb * kThreads) {
        const float4 st = *(const float4 *)(p.buf + l.colstat + 4L * (e % p.C));
        p.dxh[e] = cm * (st.w + st.z * (p.xh[e] - st.x) / st.y);
    }
}

static int fill_args(Args &a, int kind, const float *const *in, float *const *din, int B, int T, int H, int F) {
    if (B <= 0 || T <= 0 || H <= 0 || F <= 0) return TGRU_E_DIMS;
    if ((long long)B * T * (H > F ? H : F) >= (1LL << 40)) return TGRU_E_DIMS;
    a.kind = kind; a.B = B; a.T = T; a.H = H; a.C = 0;
    {   // index < 2^30 (room for the grid-stride increment), index * divisor < 2^40 for every divisor used (the
        // quotient is then exact), and index * ceil(2^40 / divisor) < 2^64 (the product in fast_div does not wrap:
        // index / divisor < 2^24 for the smallest divisor)
        const long long nmax = (long long)B * T * (H > F ? H : F);
        const long long dmax = std::max<long long>(std::max<long long>((long long)(T - 1) * H, (long long)T * F),
                                                   std::max<long long>(H, T));
        const long long dmin = std::min<long long>(H, T);      // (T - 1) * H and T * F are not smaller than these
        a.small = nmax < (1LL << 30) && nmax * dmax < (1LL << 40) && nmax < (dmin << 24);
    }
    a.mH = loss::magic_of(H); a.mT = loss::magic_of(T);
    a.mTH1 = loss::magic_of(T > 1 ? (unsigned)(T - 1) * H : 1u);
    const long nx = (long)B * T * F, nshift = (long)B * (T - 1) * H, ny = (long)B * T;
    int need = 0;
    auto term = [&](int k, int type, int ia, int ib, long n) {
        a.type[k] = type;
        a.a[k] = in[ia];
        a.b[k] = ib >= 0 ? in[ib] : nullptr;
        a.da[k] = din ? din[ia] : nullptr;
        a.db[k] = (din && ib >= 0) ? din[ib] : nullptr;
        a.n[k] = n;
        need = need > ia + 1 ? need : ia + 1;
        if (ib >= 0) need = need > ib + 1 ? need : ib + 1;
    };
    switch (kind) {
    case TGRU_LOSS_EMBED_SIDE: term(0, kSq, 0, 1, nx); break;
    case TGRU_LOSS_SUPERVISOR:
        if (T < 2) return TGRU_E_DIMS;
        term(0, kSqShift, 0, 1, nshift);
        break;
    case TGRU_LOSS_EMBED_MAIN:
        if (T < 2) return TGRU_E_DIMS;
        term(0, kSq, 0, 1, nx);
        term(1, kSqShift, 2, 3, nshift);
        break;
    case TGRU_LOSS_DISCRIMINATOR:
        term(0, kBceOne, 0, -1, ny);
        term(1, kBceZero, 1, -1, ny);
        term(2, kBceZero, 2, -1, ny);
        break;
    case TGRU_LOSS_GENERATOR:
        if (T < 2) return TGRU_E_DIMS;
        term(0, kBceOne, 0, -1, ny);
        term(1, kBceOne, 1, -1, ny);
        term(2, kSqShift, 2, 3, nshift);
        a.C = T * F;
        a.mC = loss::magic_of((unsigned)a.C);
        a.xh = in[4]; a.x = in[5];
        a.dxh = din ? din[4] : nullptr;
        need = 6;
        break;
    default: return TGRU_E_DIMS;
    }
    for (int i = 0; i < need; ++i)
        if (!in[i]) return TGRU_E_NULL;
    return TGRU_OK;
}

}  // namespace loss
}  // namespace tgru

using namespace tgru;

extern "C" {

int tgru_loss_layout(int kind, int T, int F, int64_t out[4]) {
    if (!out) return TGRU_E_NULL;
    if (kind < TGRU_LOSS_EMBED_SIDE || kind > TGRU_LOSS_GENERATOR || T <= 0 || F <= 0) return TGRU_E_DIMS;
    const loss::Layout l = loss::layout_of(kind == TGRU_LOSS_GENERATOR ? T * F : 0);
    out[0] = l.nA; out[1] = l.nB; out[2] = l.out; out[3] = l.total;
    return TGRU_OK;
}

int tgru_loss_partial(int kind, int pass, const float *const *in, int B, int T, int H, int F, float *buf,
                      tgru_stream_t stream) {
    if (!in || !buf) return TGRU_E_NULL;
    if (pass < 0 || pass > 2 || (pass == 2 && kind != TGRU_LOSS_GENERATOR)) return TGRU_E_DIMS;
    loss::Args a{};
    const int rc = loss::fill_args(a, kind, in, nullptr, B, T, H, F);
    if (rc != TGRU_OK) return rc;
    a.pass = pass; a.buf = buf;
    const int nmb = (a.C + loss::kCols - 1) / loss::kCols;
    const int grid = (pass == 2 ? 0 : 3 * loss::kSlots) + nmb;
    TGRU_LAUNCH(loss::loss_partial_kernel, dim3(grid), dim3(loss::kThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int tgru_loss_final(int kind, int B, int T, int H, int F, float gamma, float *buf, tgru_stream_t stream) {
    if (!buf) return TGRU_E_NULL;
    if (kind < TGRU_LOSS_EMBED_SIDE || kind > TGRU_LOSS_GENERATOR || B <= 0 || T <= 0 || H <= 0 || F <= 0)
        return TGRU_E_DIMS;
    loss::Args a{};
    const float *none[6] = {buf, buf, buf, buf, buf, buf};   // the final kernel reads only `buf`
    const int rc = loss::fill_args(a, kind, none, nullptr, B, T, H, F);
    if (rc != TGRU_OK) return rc;
    a.gamma = gamma; a.buf = buf;
    TGRU_LAUNCH(loss::loss_final_kernel, dim3(1), dim3(loss::kThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

int tgru_loss_backward(int kind, const float *const *in, float *const *din, const float *const *dout, int B,
                       int T, int H, int F, float gamma, const float *buf, tgru_stream_t stream) {
    if (!in || !din || !dout || !buf) return TGRU_E_NULL;
    loss::Args a{};
    const int rc = loss::fill_args(a, kind, in, din, B, T, H, F);
    if (rc != TGRU_OK) return rc;
    a.gamma = gamma; a.buf = const_cast<float *>(buf);
    for (int i = 0; i < 3; ++i) a.g[i] = dout[i];
    const long nm = (long)B * a.C;
    long nmb = (nm + loss::kThreads - 1) / loss::kThreads;
    if (nmb > 1024) nmb = 1024;
    TGRU_LAUNCH(loss::loss_backward_kernel, dim3(3 * loss::kSlots + (int)nmb), dim3(loss::kThreads), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? TGRU_OK : TGRU_E_LAUNCH;
}

}  // extern "C"
