// Whole-stack GRU kernels for small hidden sizes (H, In <= 32) and small-to-medium batches:
// ONE SEQUENCE PER WORKGROUP, matrix-vector products on the vector ALU.
//
// Why not MFMA here: an MFMA tile needs 16 sequences per workgroup, so a batch of 128 occupies 8 of
// the 256 CUs and every time step pays for a 16-column tile (about 2 us per step).  The fp32 packed
// FMA rate of the vector ALU equals the fp32 MFMA rate on this chip, and with one sequence per
// workgroup the same batch spreads over 128 CUs, each doing a 72x48 matrix-vector product per layer
// and step (36 packed FMAs per lane).  The time step, which is what bounds a recurrence this small,
// becomes several times shorter.  Above a few thousand sequences the MFMA kernels win again.
//
// Layout of one workgroup: one wave per layer (layers run as a wavefront: wave l works on time step
// tick - l).  In a wave, lane = (p, j): j = lane & 31 is a hidden unit (forward) or an input /
// hidden column (backward), p = lane >> 5 selects the half of the work:
//   forward   p = 0: rows j of W_ir, W_iz, W_in times the layer input x_t
//             p = 1: rows j of W_hr, W_hz, W_hn times h_{t-1}
//   backward  p = 0: column j of W_ih (dx_t[j]);  p = 1: column j of W_hh (dh_{t-1}[j])
// Weights stay in VGPRs for the whole kernel; the vectors they multiply are broadcast through LDS
// (all lanes of a half read the same 16 bytes); one v_permlane32_swap per value joins the two
// halves.  Global operands of a step are loaded several steps ahead into a register ring.
// Backward also accumulates the weight gradients: dW += dg (x) input is a rank-1 update per step,
// issued as three v_mfma_f32_32x32x1 (two independent 32x32 blocks: block 0 = dW_ih from the lower
// half's operands, block 1 = dW_hh from the upper half's), accumulating in AGPRs off the critical
// path; each workgroup leaves one partial gradient, reduced by a second kernel in a fixed order.
#include "tgru_internal.h"

namespace tgru {
namespace vec {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lo: the value lane (l & 31) holds, hi: the value lane (l & 31) + 32 holds -- in every lane
__device__ __forceinline__ void both_halves(float v, float &lo, float &hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Branch-free predication: every global access in the time loops is a raw buffer access; a lane
// (or a step) that must not touch memory uses an offset past the end of the buffer, for which loads
// return 0 and stores are dropped.  An absent tensor gets a descriptor with zero records.  Without
// control flow around them the loads of a later step stay in flight across the steps in between.
constexpr int kOOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, int off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, int off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
// row of (t, b) in a caller tensor, -1 if there is none; t must be a valid step of sequence b
__device__ __forceinline__ int ext_row(const RowMap &m, int t, int b) {
    if (m.mode == 0) return t * m.B + b;
    if (m.mode == 1) return b * m.T + t;
    const int o = m.toff[t];
    return (b < m.toff[t + 1] - o) ? o + b : -1;
}

constexpr int kFwdRing = 8;   // time steps the layer-0 input is loaded ahead
constexpr int kBwdRing = 4;   // time steps the backward operands are loaded ahead

}  // namespace vec

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(512) void gru_vec_fwd_kernel(SmallFwdArgs a) {
    using namespace vec;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hb[2][L][KP] | xb[2][KP]
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int l = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane >> 5, j = lane & 31;
    const int In_l = (l == 0) ? a.In : H;
    float *xb = smem + 2 * L * KP;
    const long plane = (long)T * B * H;
    const bool unit = j < H;

    // ---- weights: rows j of the three gate blocks, zero-padded to KP columns ----------------------
    f32x2 w[3][KP / 2];
    {
        const float *wsrc = p ? a.w_hh[l] : a.w_ih[l];
        const int K = p ? H : In_l;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float *row = wsrc + (long)(g * H + j) * K;
            if ((K & 3) == 0) {
#pragma unroll
                for (int q = 0; q < KP / 4; ++q) {
                    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (unit && 4 * q < K) u = *reinterpret_cast<const float4 *>(row + 4 * q);
                    w[g][2 * q] = (f32x2){u.x, u.y};
                    w[g][2 * q + 1] = (f32x2){u.z, u.w};
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < KP / 2; ++k2) {
                    const float lo = (unit && 2 * k2 < K) ? row[2 * k2] : 0.f;
                    const float hi = (unit && 2 * k2 + 1 < K) ? row[2 * k2 + 1] : 0.f;
                    w[g][k2] = (f32x2){lo, hi};
                }
            }
        }
    }
    const float *bi = a.b_ih[l], *bh = a.b_hh[l];
    const float b_r = unit ? bi[j] + bh[j] : 0.f;
    const float b_z = unit ? bi[H + j] + bh[H + j] : 0.f;
    const float b_in = unit ? bi[2 * H + j] : 0.f;
    const float b_hn = unit ? bh[2 * H + j] : 0.f;
    const bool top = l == L - 1;
    const int nticks = T + L - 1;
    const int tbh = (int)(plane * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(l == 0 ? a.x : nullptr, (long)T * B * a.In * 4);
    const __amdgpu_buffer_rsrc_t rs_yint = make_rsrc(a.y_int[l], tbh);
    const __amdgpu_buffer_rsrc_t rs_yext = make_rsrc(top ? a.y_ext : nullptr, tbh);
    const __amdgpu_buffer_rsrc_t rs_gates = make_rsrc(a.gates[l], 4L * tbh);
    const __amdgpu_buffer_rsrc_t rs_hn = make_rsrc(a.h_n, (long)L * B * H * 4);

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int len = a.lengths ? a.lengths[b] : T;
        float hreg = (unit && a.h0) ? a.h0[((long)l * B + b) * H + j] : 0.f;

        // layer-0 input ring: xq[u] holds x[t][j] for the next t = u (mod kFwdRing) still to be handed
        // to LDS; x[0] goes there directly, so slot 0 starts with x[kFwdRing]
        float xq[kFwdRing];
        auto load_x = [&](int t) -> float {
            int er = -1;
            if (l == 0 && t < len) er = ext_row(a.xmap, t, b);
            return buf_ld(rs_x, (er >= 0 && p == 0 && j < a.In) ? (er * a.In + j) * 4 : kOOB);
        };
#pragma unroll
        for (int u = 0; u < kFwdRing; ++u) xq[u] = load_x(u == 0 ? kFwdRing : u);
        const float x0 = load_x(0);

        if (p == 0 && j < KP) {
            smem[(0 * L + l) * KP + j] = hreg;
            smem[(1 * L + l) * KP + j] = hreg;
            if (l == 0) xb[j] = x0;
        }
        lds_barrier();

        for (int k0 = 0; k0 < nticks; k0 += kFwdRing) {
#pragma unroll
            for (int u = 0; u < kFwdRing; ++u) {
                const int k = k0 + u;
                if (k >= nticks) break;
                const int par = k & 1;
                const int t = k - l;
                if (t >= 0 && t < T) {
                    const bool active = t < len;
                    const float *src = p ? smem + (par * L + l) * KP
                                         : (l == 0 ? xb + par * KP : smem + (par * L + l - 1) * KP);
                    f32x2 acc0 = {0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
                    for (int q = 0; q < KP / 4; ++q) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(src + 4 * q);
                        const f32x2 va = {v4.x, v4.y}, vb = {v4.z, v4.w};
                        acc0 = pk_fma(w[0][2 * q], va, acc0); acc1 = pk_fma(w[1][2 * q], va, acc1);
                        acc2 = pk_fma(w[2][2 * q], va, acc2);
                        acc0 = pk_fma(w[0][2 * q + 1], vb, acc0); acc1 = pk_fma(w[1][2 * q + 1], vb, acc1);
                        acc2 = pk_fma(w[2][2 * q + 1], vb, acc2);
                    }
                    float gi_r, gh_r, gi_z, gh_z, gi_n, gh_n;
                    both_halves(acc0.x + acc0.y, gi_r, gh_r);
                    both_halves(acc1.x + acc1.y, gi_z, gh_z);
                    both_halves(acc2.x + acc2.y, gi_n, gh_n);
                    const float r = fast_sigmoid(gi_r + gh_r + b_r);
                    const float z = fast_sigmoid(gi_z + gh_z + b_z);
                    const float ghn = gh_n + b_hn;
                    const float n = fast_tanh(gi_n + b_in + r * ghn);
                    const float hnew = n + z * (hreg - n);
                    if (active && unit) hreg = hnew;
                    const float yv = active ? hreg : 0.f;
                    if (p == 0 && j < KP) smem[((par ^ 1) * L + l) * KP + j] = hreg;
                    {
                        // the two halves hold the same values: each stores half of them
                        const int oi = ((t * B + b) * H + j) * 4;
                        const int og = (unit && active) ? oi + (p ? 2 * tbh : 0) : kOOB;
                        buf_st(rs_gates, og, p ? n : r);                 // planes 0 (r) and 2 (n)
                        buf_st(rs_gates, og == kOOB ? kOOB : og + tbh, p ? ghn : z);   // planes 1 (z) and 3 (gh_n)
                        buf_st(rs_yint, (unit && p == 0) ? oi : kOOB, yv);
                        int er = -1;
                        if (top && (active || a.ymap.mode != 2)) er = ext_row(a.ymap, t, b);
                        buf_st(rs_yext, (unit && p == 1 && er >= 0) ? (er * H + j) * 4 : kOOB, yv);
                        if (t == len - 1) buf_st(rs_hn, (unit && p == 0) ? ((l * B + b) * H + j) * 4 : kOOB, yv);
                    }
                }
                if (l == 0) {   // hand x[k+1] to the next tick, refill the ring slot
                    const int un = (u + 1) % kFwdRing;
                    if (p == 0 && j < KP) xb[(par ^ 1) * KP + j] = xq[un];
                    xq[un] = load_x(k + 1 + kFwdRing);
                }
                lds_barrier();
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// backward (dx, dh0 and per-workgroup partial weight / bias gradients)
// ----------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void gru_vec_bwd_kernel(SmallBwdArgs a) {
    using namespace vec;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // gv[L][4][KP] | dxb[2][L][KP]
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int l = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane >> 5, j = lane & 31;
    const int In_l = (l == 0) ? a.In : H;
    float *gv = smem + l * 4 * KP;
    float *dxb = smem + L * 4 * KP;
    const long plane = (long)T * B * H;
    const bool unit = j < H;
    const bool top = l == L - 1;
    const int ncol = p ? H : In_l;           // columns this half owns
    const bool col = j < ncol;

    // ---- column j of the three gate blocks of W_ih (p = 0) / W_hh (p = 1), zero-padded to KP rows --
    f32x2 wT[3][KP / 2];
    {
        const float *wsrc = p ? a.w_hh[l] : a.w_ih[l];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k2 = 0; k2 < KP / 2; ++k2) {
                const float lo = (col && 2 * k2 < H) ? wsrc[(long)(g * H + 2 * k2) * ncol + j] : 0.f;
                const float hi = (col && 2 * k2 + 1 < H) ? wsrc[(long)(g * H + 2 * k2 + 1) * ncol + j] : 0.f;
                wT[g][k2] = (f32x2){lo, hi};
            }
    }
    // dW accumulators: for gate g, block p of dWa[g] is sum over steps of dg_p[g][k] * in_p[j]
    f32x32 dWa[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 32; ++r) dWa[g][r] = 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};

    const int nticks = T + L - 1;
    const int tbh = (int)(plane * 4);
    const __amdgpu_buffer_rsrc_t rs_gates = make_rsrc(a.gates[l], 4L * tbh);
    const __amdgpu_buffer_rsrc_t rs_yint = make_rsrc(a.y_int[l], tbh);
    const __amdgpu_buffer_rsrc_t rs_h0 = make_rsrc(a.h0, (long)L * B * H * 4);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(top ? a.dy_ext : nullptr, tbh);
    const __amdgpu_buffer_rsrc_t rs_dhn = make_rsrc(a.dh_n, (long)L * B * H * 4);
    const __amdgpu_buffer_rsrc_t rs_xin = l == 0 ? make_rsrc(a.x, (long)T * B * a.In * 4) : make_rsrc(a.y_int[l - 1], tbh);
    const __amdgpu_buffer_rsrc_t rs_dx = make_rsrc(l == 0 ? a.dx_ext : nullptr, (long)T * B * a.In * 4);

    struct Ops { float r, z, n, hn, hp, dy, xin; };

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int len = a.lengths ? a.lengths[b] : T;
        // the gradient of h_n enters as the initial carry: steps past the length pass it through
        const int sb = ((l * B + b) * H + j) * 4;
        float dhc = buf_ld(rs_dhn, unit ? sb : kOOB);
        const float h0v = buf_ld(rs_h0, unit ? sb : kOOB);

        auto load_ops = [&](int t) -> Ops {
            const bool ok = t >= 0 && t < len;
            const int oi = ((t * B + b) * H + j) * 4;
            const int ou = (ok && unit) ? oi : kOOB;
            Ops o;
            o.r = buf_ld(rs_gates, ou);
            o.z = buf_ld(rs_gates, ou == kOOB ? kOOB : ou + tbh);
            o.n = buf_ld(rs_gates, ou == kOOB ? kOOB : ou + 2 * tbh);
            o.hn = buf_ld(rs_gates, ou == kOOB ? kOOB : ou + 3 * tbh);
            // h_{t-1}: both halves (p = 1 multiplies it into dW_hh)
            o.hp = buf_ld(rs_yint, (ok && unit && t > 0) ? oi - B * H * 4 : kOOB);
            int er = -1;
            if (ok && top) er = ext_row(a.dymap, t, b);
            o.dy = buf_ld(rs_dy, (er >= 0 && unit) ? (er * H + j) * 4 : kOOB);
            int xr = ok ? t * B + b : -1;
            if (ok && l == 0) xr = ext_row(a.xmap, t, b);
            o.xin = buf_ld(rs_xin, (xr >= 0 && col && p == 0) ? (xr * In_l + j) * 4 : kOOB);
            return o;
        };
        auto tick_time = [&](int k) { return T - 1 - k + (L - 1 - l); };

        Ops ring[kBwdRing];
#pragma unroll
        for (int u = 0; u < kBwdRing; ++u) ring[u] = load_ops(tick_time(u));

        for (int k0 = 0; k0 < nticks; k0 += kBwdRing) {
#pragma unroll
            for (int u = 0; u < kBwdRing; ++u) {
                const int k = k0 + u;
                if (k >= nticks) break;
                const int par = k & 1;
                const int t = tick_time(k);
                if (t >= 0 && t < T) {
                    const bool active = t < len;
                    Ops o = ring[u];
                    if (t == 0) o.hp = h0v;
                    float dyv = o.dy;
                    if (!top && j < KP) dyv += dxb[((par ^ 1) * L + (l + 1)) * KP + j];
                    const bool on = active && unit;      // both halves run the gate arithmetic
                    const float d = on ? dyv + dhc : 0.f;
                    const float dn = d * (1.f - o.z);
                    const float dzz = d * (o.hp - o.n);
                    const float an = dn * (1.f - o.n * o.n);
                    const float dar = an * o.hn * o.r * (1.f - o.r);
                    const float daz = dzz * o.z * (1.f - o.z);
                    const float dgn = an * o.r;
                    const float keep = on ? d * o.z : dhc;
                    if (j < KP) {   // each half publishes two of the four vectors
                        gv[(p ? 1 : 0) * KP + j] = p ? daz : dar;
                        gv[(p ? 3 : 2) * KP + j] = p ? dgn : an;
                    }
                    dbs[0] += dar; dbs[1] += daz; dbs[2] += an; dbs[3] += dgn;
                    // rank-1 weight-gradient updates: rows from this half's dg, columns from its input
                    const float s = p ? o.hp : o.xin;
                    dWa[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(dar, s, dWa[0], 0, 0, 0);
                    dWa[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(daz, s, dWa[1], 0, 0, 0);
                    dWa[2] = __builtin_amdgcn_mfma_f32_32x32x1f32(p ? dgn : an, s, dWa[2], 0, 0, 0);
                    // same wave wrote the vectors: LDS executes a wave's accesses in order
                    f32x2 acc0 = {0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
                    for (int q = 0; q < KP / 4; ++q) {
                        const float4 u0 = *reinterpret_cast<const float4 *>(gv + 0 * KP + 4 * q);
                        const float4 u1 = *reinterpret_cast<const float4 *>(gv + 1 * KP + 4 * q);
                        const float4 u2 = *reinterpret_cast<const float4 *>(gv + (p ? 3 : 2) * KP + 4 * q);
                        const f32x2 a0 = {u0.x, u0.y}, b0 = {u0.z, u0.w};
                        const f32x2 a1 = {u1.x, u1.y}, b1 = {u1.z, u1.w};
                        const f32x2 a2 = {u2.x, u2.y}, b2 = {u2.z, u2.w};
                        acc0 = pk_fma(wT[0][2 * q], a0, acc0); acc0 = pk_fma(wT[0][2 * q + 1], b0, acc0);
                        acc1 = pk_fma(wT[1][2 * q], a1, acc1); acc1 = pk_fma(wT[1][2 * q + 1], b1, acc1);
                        acc2 = pk_fma(wT[2][2 * q], a2, acc2); acc2 = pk_fma(wT[2][2 * q + 1], b2, acc2);
                    }
                    const f32x2 sum = acc0 + acc1 + acc2;
                    float dxv, dhv;
                    both_halves(sum.x + sum.y, dxv, dhv);
                    dhc = keep + (on ? dhv : 0.f);
                    if (l > 0 && p == 0 && j < KP) dxb[(par * L + l) * KP + j] = dxv;
                    if (l == 0) {
                        int er = -1;
                        if (active || a.dxmap.mode != 2) er = ext_row(a.dxmap, t, b);
                        buf_st(rs_dx, (er >= 0 && p == 0 && j < a.In) ? (er * a.In + j) * 4 : kOOB, dxv);
                    }
                }
                ring[u] = load_ops(tick_time(k + kBwdRing));
                lds_barrier();
            }
        }
        if (a.dh0 && p == 0 && unit) a.dh0[((long)l * B + b) * H + j] = dhc;
    }

    // ---- per-workgroup partial gradients: slab laid out like the gradients themselves -------------
    // result layout of the two-block 32x32x1 MFMA: register 16*blk + rr of lane (hh, jc) holds
    // D[blk][i][jc] with i = 8*(rr/4) + rr%4 + 4*hh
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    long off = 0;
    for (int q = 0; q < l; ++q) off += 3L * H * ((q == 0 ? a.In : H) + H);
    float *slab = a.slabs + (long)blockIdx.x * a.slab_stride + off;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int nc = blk ? H : In_l;
        float *dst = slab + (blk ? 3L * H * In_l : 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int i = 8 * (rr / 4) + (rr % 4) + 4 * p;
                if (i < H && j < nc) dst[(long)(g * H + i) * nc + j] = dWa[g][16 * blk + rr];
            }
    }
    if (p == 0 && unit) {
        float *dbp = a.db_part + ((long)blockIdx.x * L + l) * 4 * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) dbp[g * H + j] = dbs[g];
    }
}

int gru_vec_workgroups(int B) { return B < 512 ? B : 512; }

// Up to 2048 sequences and 4 layers; byte offsets into every per-(t,b) buffer must fit 31 bits.
bool gru_vec_supported(int B, int T, int In, int H, int L) {
    if (H > 32 || In > 32 || L > 4 || B > 2048) return false;   // 4 waves: the backward kernel needs > 256 registers
    const long widest = H > In ? H : In;
    return (long)T * B * widest * 16 < (1L << 31);
}

hipError_t launch_gru_vec_fwd(const SmallFwdArgs &a, hipStream_t s) {
    const bool k24 = a.H <= 24 && a.In <= 24;
    const int KP = k24 ? 24 : 32;
    dim3 grid(gru_vec_workgroups(a.B)), block(64 * a.L);
    const size_t lds = (size_t)(2 * a.L + 2) * KP * sizeof(float);
    if (k24) hipLaunchKernelGGL(gru_vec_fwd_kernel<24>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(gru_vec_fwd_kernel<32>, grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_vec_bwd(const SmallBwdArgs &a, hipStream_t s) {
    const bool k24 = a.H <= 24 && a.In <= 24;
    const int KP = k24 ? 24 : 32;
    dim3 grid(gru_vec_workgroups(a.B)), block(64 * a.L);
    const size_t lds = (size_t)(4 * a.L + 2 * a.L) * KP * sizeof(float);
    if (k24) hipLaunchKernelGGL(gru_vec_bwd_kernel<24>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(gru_vec_bwd_kernel<32>, grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace tgru
