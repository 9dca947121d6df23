// Whole-stack GRU kernels for small hidden sizes (H <= 32, In <= 32, L <= 8) on gfx950.
//
// At these sizes a stack is latency-bound: every step is a 16 x 3H x 2H product per workgroup and
// there are T*L of them in sequence.  These kernels therefore run the WHOLE stack in one launch:
//
//   * one workgroup owns 16 sequences, one wave owns (layer, hidden tile of 16);
//   * layers run as a wavefront: at tick k layer l works on step t = k - l, so the critical
//     path is T + L - 1 ticks instead of T * L steps;
//   * W_ih and W_hh fragments stay in VGPRs for the whole kernel (MFMA A operands), the hidden
//     state and the layer-to-layer activations move between waves as MFMA B fragments through
//     LDS (the D fragment of a hidden tile is the B fragment of the matching k-block, so one
//     lane-linear 16-byte store / load per lane, no transposition);
//   * the input projection of every layer is fused into the step (no gi buffer at all);
//   * every global load / store of a tick is issued at the top of the tick (results of the
//     previous tick are stored one tick late, inputs of the next tick are loaded one tick early),
//     so memory latency hides behind the tick's MFMA chain instead of adding to it.
//
// The backward kernel mirrors this in reverse time / reverse layer order and also produces dx;
// gru_wgrad.hip then turns the dg buffers into all weight and bias gradients of all layers
// (deterministic: fixed reduction order, no atomics).
#include "tgru_internal.h"
#include <stdlib.h>

namespace tgru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() {
    // one asm statement with a memory clobber: the compiler must not move LDS accesses across it
    // (the s_barrier builtin alone is not a compiler-level memory barrier)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 4 consecutive floats p[0..3], element c valid iff c < nvalid.  VEC: nvalid is 0 or >= 4 and p
// is 16-byte aligned, so the access is one dwordx4; otherwise element-wise with guards.
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float *p, int nvalid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
        if (nvalid > 0) v = *reinterpret_cast<const float4 *>(p);
    } else {
        if (nvalid > 0) v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
        if (nvalid > 3) v.w = p[3];
    }
    return v;
}
template <bool VEC>
__device__ __forceinline__ void st4(float *p, float4 v, int nvalid) {
    if (VEC) {
        if (nvalid > 0) *reinterpret_cast<float4 *>(p) = v;
    } else {
        if (nvalid > 0) p[0] = v.x;
        if (nvalid > 1) p[1] = v.y;
        if (nvalid > 2) p[2] = v.z;
        if (nvalid > 3) p[3] = v.w;
    }
}

// Wait states between the last MFMA of a block and the first VALU read of its result.  hipcc pads
// this hazard inside a basic block, but a reader that is reached through a branch taken right
// after the MFMA was observed to get the upper result registers before they were written, so
// every MFMA block that is followed by control flow ends with this.
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#define F4ARR(name, v) const float name[4] = {(v).x, (v).y, (v).z, (v).w}

// Diagnostic build only (make stamps): per-wave cycle shares of the forward tick, read back with
// tgru_debug_stamps().  No stamp executes in the product library.
#ifdef TGRU_STAMPS
__device__ unsigned long long tgru_stamps[16 * 8];
#define STAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(k)                                                   \
    do {                                                           \
        __builtin_amdgcn_sched_barrier(0);                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += now_ - st_last; st_last = now_;               \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)
#define STAMP_FLUSH                                                          \
    if (blockIdx.x == 0 && lane == 0)                                        \
        for (int k_ = 0; k_ < 6; ++k_) tgru_stamps[wave * 8 + k_] = st_acc[k_];
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int NT, int MAXT, bool VEC>
__global__ __launch_bounds__(MAXT) void gru_small_fwd_kernel(SmallFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hbuf[2][L][NT][256]
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = wave / NT, tile = wave % NT;
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const int nQin = (In_l + 15) / 16;
    const int j0 = 16 * tile + 4 * gq;            // first hidden unit of this lane
    const int jn = min(4, max(0, H - j0));        // how many of its 4 units exist
    const float *wi = a.w_ih[l], *wh = a.w_hh[l], *bi = a.b_ih[l], *bhp = a.b_hh[l];
    float *yint = a.y_int[l], *gates = a.gates[l];

    // weight fragments: lane (i = lane&15, gq) holds W[g*H + 16*tile + i][16*q + 4*gq + c]
    float4 wih[3][2], whh[3][NT];
    {
        const int r = 16 * tile + (lane & 15);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c0 = 16 * q + 4 * gq;
                wih[g][q] = ld4<VEC>(wi + (long)(g * H + r) * In_l + c0, (r < H && q < nQin) ? In_l - c0 : 0);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int c0 = 16 * q + 4 * gq;
                whh[g][q] = ld4<VEC>(wh + (long)(g * H + r) * H + c0, (r < H) ? H - c0 : 0);
            }
        }
    }
    float brz[2][4], bin[4], bhn[4], hreg[4];
    {
        const float4 i0 = ld4<VEC>(bi + j0, jn), i1 = ld4<VEC>(bi + H + j0, jn), i2 = ld4<VEC>(bi + 2 * H + j0, jn);
        const float4 h0_ = ld4<VEC>(bhp + j0, jn), h1 = ld4<VEC>(bhp + H + j0, jn), h2 = ld4<VEC>(bhp + 2 * H + j0, jn);
        const float4 hi = (a.h0 && bvalid) ? ld4<VEC>(a.h0 + ((long)l * B + bglob) * H + j0, jn)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        F4ARR(i0a, i0); F4ARR(i1a, i1); F4ARR(i2a, i2); F4ARR(h0a, h0_); F4ARR(h1a, h1); F4ARR(h2a, h2); F4ARR(hia, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            brz[0][c] = i0a[c] + h0a[c]; brz[1][c] = i1a[c] + h1a[c];
            bin[c] = i2a[c]; bhn[c] = h2a[c]; hreg[c] = hia[c];
        }
    }
    // both parities start from h0 so that step 0 of every layer reads its initial state
    {
        const float4 h4 = make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
        *reinterpret_cast<float4 *>(smem + (((0 * L + l) * NT + tile) * 64 + lane) * 4) = h4;
        *reinterpret_cast<float4 *>(smem + (((1 * L + l) * NT + tile) * 64 + lane) * 4) = h4;
    }
    lds_barrier();

    const long plane = (long)T * B * H;
    // layer 0 reads x straight from global memory as B fragments, one step ahead
    float4 xn[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    auto load_x = [&](int t) {
        if (t < len) {   // len <= T
            const long er = map_tb(a.xmap, t, bglob);
            if (er >= 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c0 = 16 * q + 4 * gq;
                    if (q < nQin) xn[q] = ld4<VEC>(a.x + er * a.In + c0, a.In - c0);
                }
            }
        }
    };
    if (l == 0) load_x(0);

    // results of the previous tick, stored at the top of the next one
    float4 py = xn[0], pr = xn[0], pz = xn[0], pn = xn[0], phn = xn[0];
    int pt = 0;
    bool pact = false, pend = false;
    auto flush = [&]() {
        if (bvalid && jn > 0) {
            const long row = (long)pt * B + bglob;
            if (yint) st4<VEC>(yint + row * H + j0, py, jn);
            if (l == L - 1 && a.y_ext) {
                const long er = map_tb(a.ymap, pt, bglob);
                if (er >= 0) st4<VEC>(a.y_ext + er * H + j0, py, jn);
            }
            if (gates && pact) {
                float *gp = gates + row * H + j0;
                st4<VEC>(gp, pr, jn);
                st4<VEC>(gp + plane, pz, jn);
                st4<VEC>(gp + 2 * plane, pn, jn);
                st4<VEC>(gp + 3 * plane, phn, jn);
            }
            if (a.h_n && pt == len - 1) st4<VEC>(a.h_n + ((long)l * B + bglob) * H + j0, py, jn);
        }
    };

    int p = 0;
    const int nticks = T + L - 1;
    STAMP_DECL
    for (int tick = 0; tick < nticks; ++tick) {
        const int t = tick - l;
        const bool work = t >= 0 && t < T;
        STAMP(5);
        // order matters for the memory counter: first take last tick's loads (everything older is
        // long done), then issue this tick's stores and next tick's loads
        float4 xin[2] = {xn[0], xn[1]};
        if (l == 0) asm volatile("" : "+v"(xin[0].x), "+v"(xin[0].y), "+v"(xin[0].z), "+v"(xin[0].w),
                                      "+v"(xin[1].x), "+v"(xin[1].y), "+v"(xin[1].z), "+v"(xin[1].w));
        if (pend) flush();
        STAMP(0);
        if (work) {
            const bool active = t < len;
            const float *hb = smem + (long)p * L * NT * 256;
            float4 hp[NT];
            if (l == 0) {
                if (t + 1 < T) load_x(t + 1);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    xin[q] = (q < NT) ? *reinterpret_cast<const float4 *>(hb + (((l - 1) * NT + q) * 64 + lane) * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q)
                hp[q] = *reinterpret_cast<const float4 *>(hb + ((l * NT + q) * 64 + lane) * 4);

            // r, z accumulate both products; the candidate keeps W_in x and W_hn h apart.
            // The chains are interleaved so dependent MFMAs are never back to back.
            STAMP(1);
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ain = ar, ahn = ar;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < nQin) {
                    F4ARR(xv, xin[q]); F4ARR(w0, wih[0][q]); F4ARR(w1, wih[1][q]); F4ARR(w2, wih[2][q]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ar = mf(w0[c], xv[c], ar);
                        az = mf(w1[c], xv[c], az);
                        ain = mf(w2[c], xv[c], ain);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                F4ARR(hv, hp[q]); F4ARR(w0, whh[0][q]); F4ARR(w1, whh[1][q]); F4ARR(w2, whh[2][q]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ar = mf(w0[c], hv[c], ar);
                    az = mf(w1[c], hv[c], az);
                    ahn = mf(w2[c], hv[c], ahn);
                }
            }

            mfma_drain();
            STAMP(2);
            float rv[4], zv[4], nv[4], hnv[4], yv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ghn = ahn[c] + bhn[c];
                const float r = fast_sigmoid(ar[c] + brz[0][c]);
                const float z = fast_sigmoid(az[c] + brz[1][c]);
                const float n = fast_tanh(ain[c] + bin[c] + r * ghn);
                const float hnew = n + z * (hreg[c] - n);
                rv[c] = r; zv[c] = z; nv[c] = n; hnv[c] = ghn;
                if (active && c < jn) hreg[c] = hnew;
                yv[c] = active ? hreg[c] : 0.f;
            }
            float *hw = smem + (long)(p ^ 1) * L * NT * 256;
            *reinterpret_cast<float4 *>(hw + ((l * NT + tile) * 64 + lane) * 4) =
                make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
            py = make_float4(yv[0], yv[1], yv[2], yv[3]);
            pr = make_float4(rv[0], rv[1], rv[2], rv[3]);
            pz = make_float4(zv[0], zv[1], zv[2], zv[3]);
            pn = make_float4(nv[0], nv[1], nv[2], nv[3]);
            phn = make_float4(hnv[0], hnv[1], hnv[2], hnv[3]);
            pt = t; pact = active;
        }
        pend = work;
        STAMP(3);
        lds_barrier();
        STAMP(4);
        p ^= 1;
    }
    if (pend) flush();
    STAMP_FLUSH
}

#ifdef TGRU_STAMPS
extern "C" int tgru_debug_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps), sizeof(unsigned long long) * 16 * 8);
}
#endif

template <int NT, int MAXT>
static hipError_t launch_fwd_t(const SmallFwdArgs &a, dim3 grid, dim3 block, size_t lds, bool vec, hipStream_t s) {
    if (vec) TGRU_LAUNCH((gru_small_fwd_kernel<NT, MAXT, true>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_small_fwd_kernel<NT, MAXT, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

static bool all_vec(int In, int H) { return (In & 3) == 0 && (H & 3) == 0; }

static bool h24_shape(int In, int H) {
    // TGRU_NO_H24=1: the tests use it to reach the generic small-H kernels at H = 24
    return H > 16 && H <= 24 && In <= 24 && !switches().no_h24;
}

hipError_t launch_gru_small_fwd(const SmallFwdArgs &a, hipStream_t s) {
    if (h24_shape(a.In, a.H)) return launch_gru_h24_fwd(a, s);
    const int NT = (a.H + 15) / 16;
    dim3 grid(gru_batch_blocks(a.B)), block(64 * NT * a.L);
    const size_t lds = (size_t)2 * a.L * NT * 1024;
    const bool big = block.x > 512, vec = all_vec(a.In, a.H);
    if (NT == 1) return big ? launch_fwd_t<1, 1024>(a, grid, block, lds, vec, s) : launch_fwd_t<1, 512>(a, grid, block, lds, vec, s);
    return big ? launch_fwd_t<2, 1024>(a, grid, block, lds, vec, s) : launch_fwd_t<2, 512>(a, grid, block, lds, vec, s);
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
template <int NT, int MAXT, bool VEC>
__global__ __launch_bounds__(MAXT) void gru_small_bwd_kernel(SmallBwdArgs a) {
    // gbuf[L][4][NT][256] : da_r | da_z | da_n | da_n*r fragments, dxbuf[2][L][NT][256]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = wave / NT, tile = wave % NT;
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const int nDx = (In_l + 15) / 16;             // dx tiles of this layer
    const bool two_dx = nDx > NT;                 // layer 0 with more input tiles than hidden tiles
    const int j0 = 16 * tile + 4 * gq;
    const int jn = min(4, max(0, H - j0));
    const long plane = (long)T * B * H;
    float *gb = smem + (long)l * 4 * NT * 256;
    float *dxbuf = smem + (long)L * 4 * NT * 256;
    const float *wi = a.w_ih[l], *wh = a.w_hh[l], *yint = a.y_int[l], *gates = a.gates[l];
    float *dgl = a.dg[l];

    // A fragments of the transposed weights.
    //   whhT[g][q]: lane (i, gq), c -> W_hh[g*H + 16q + 4gq + c][16*tile + i]
    //   wihT[d][g][q]: lane (i, gq), c -> W_ih[g*H + 16q + 4gq + c][16*(tile + d*NT) + i]
    float4 whhT[3][NT], wihT[2][3][NT];
    {
        const int i = lane & 15;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kr = 16 * q + 4 * gq + c, oc = 16 * tile + i;
                    v[c] = (kr < H && oc < H) ? wh[(long)(g * H + kr) * H + oc] : 0.f;
                }
                whhT[g][q] = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    float u[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int kr = 16 * q + 4 * gq + c, oc = 16 * (tile + d * NT) + i;
                        u[c] = (kr < H && oc < In_l && (d == 0 || two_dx)) ? wi[(long)(g * H + kr) * In_l + oc] : 0.f;
                    }
                    wihT[d][g][q] = make_float4(u[0], u[1], u[2], u[3]);
                }
            }
    }

    float dhc[4] = {0.f, 0.f, 0.f, 0.f};
    float dbs[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) dbs[g][c] = 0.f;

    // saved activations of the step this wave handles next, loaded one tick ahead
    float4 pr, pz, pn, phn, php, pdy;
    auto prefetch = [&](int t) {
        pr = pz = pn = phn = php = pdy = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < len && jn > 0) {
            const long row = (long)t * B + bglob;
            const float *gp = gates + row * H + j0;
            pr = ld4<VEC>(gp, jn);
            pz = ld4<VEC>(gp + plane, jn);
            pn = ld4<VEC>(gp + 2 * plane, jn);
            phn = ld4<VEC>(gp + 3 * plane, jn);
            if (t > 0) php = ld4<VEC>(yint + (row - B) * H + j0, jn);
            else if (a.h0) php = ld4<VEC>(a.h0 + ((long)l * B + bglob) * H + j0, jn);
            if (l == L - 1 && a.dy_ext) {
                const long er = map_tb(a.dymap, t, bglob);
                if (er >= 0) pdy = ld4<VEC>(a.dy_ext + er * H + j0, jn);
            }
            if (a.dh_n && t == len - 1) {
                const float4 e = ld4<VEC>(a.dh_n + ((long)l * B + bglob) * H + j0, jn);
                pdy.x += e.x; pdy.y += e.y; pdy.z += e.z; pdy.w += e.w;
            }
        }
    };
    prefetch(T - 1);   // every layer starts at t = T-1

    // layer 0: dx of the previous tick, stored at the top of the next one
    float4 qx0 = make_float4(0.f, 0.f, 0.f, 0.f), qx1 = qx0;
    int qt = 0;
    bool qpend = false;
    auto flush_dx = [&]() {
        if (a.dx_ext && bvalid) {
            const long er = map_tb(a.dxmap, qt, bglob);
            if (er >= 0) {
                const int c0 = 16 * tile + 4 * gq;
                if (c0 < a.In) st4<VEC>(a.dx_ext + er * a.In + c0, qx0, a.In - c0);
                const int c1 = 16 * (tile + NT) + 4 * gq;
                if (two_dx && c1 < a.In) st4<VEC>(a.dx_ext + er * a.In + c1, qx1, a.In - c1);
            }
        }
    };

    int p = 0;
    const int nticks = T + L - 1;
    for (int tick = 0; tick < nticks; ++tick) {
        const int t = T - 1 - tick + (L - 1 - l);
        const bool work = t >= 0 && t < T;
        const bool active = work && t < len;
        float keep[4] = {dhc[0], dhc[1], dhc[2], dhc[3]};
        if (qpend) { flush_dx(); qpend = false; }
        if (work) {
            float dyv[4] = {pdy.x, pdy.y, pdy.z, pdy.w};
            if (l < L - 1) {   // gradient from the layer above, produced at the previous tick
                const float4 u = *reinterpret_cast<const float4 *>(
                    dxbuf + (long)(p ^ 1) * L * NT * 256 + (((l + 1) * NT + tile) * 64 + lane) * 4);
                dyv[0] += u.x; dyv[1] += u.y; dyv[2] += u.z; dyv[3] += u.w;
            }
            F4ARR(rr, pr); F4ARR(zz, pz); F4ARR(nn, pn); F4ARR(hh, phn); F4ARR(hq, php);
            float dar[4], daz[4], dan[4], dgn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool on = active && c < jn;
                const float d = on ? dyv[c] + dhc[c] : 0.f;
                const float dn = d * (1.f - zz[c]);
                const float dzz = d * (hq[c] - nn[c]);
                const float an = dn * (1.f - nn[c] * nn[c]);
                dan[c] = on ? an : 0.f;
                dar[c] = on ? an * hh[c] * rr[c] * (1.f - rr[c]) : 0.f;
                daz[c] = on ? dzz * zz[c] * (1.f - zz[c]) : 0.f;
                dgn[c] = on ? an * rr[c] : 0.f;
                keep[c] = active ? d * zz[c] : dhc[c];
                dbs[0][c] += dar[c]; dbs[1][c] += daz[c]; dbs[2][c] += dan[c]; dbs[3][c] += dgn[c];
            }
            const float4 f0 = make_float4(dar[0], dar[1], dar[2], dar[3]);
            const float4 f1 = make_float4(daz[0], daz[1], daz[2], daz[3]);
            const float4 f2 = make_float4(dan[0], dan[1], dan[2], dan[3]);
            const float4 f3 = make_float4(dgn[0], dgn[1], dgn[2], dgn[3]);
            *reinterpret_cast<float4 *>(gb + ((0 * NT + tile) * 64 + lane) * 4) = f0;
            *reinterpret_cast<float4 *>(gb + ((1 * NT + tile) * 64 + lane) * 4) = f1;
            *reinterpret_cast<float4 *>(gb + ((2 * NT + tile) * 64 + lane) * 4) = f2;
            *reinterpret_cast<float4 *>(gb + ((3 * NT + tile) * 64 + lane) * 4) = f3;
            if (bvalid && jn > 0) {
                float *dst = dgl + ((long)t * B + bglob) * 4 * H + j0;
                st4<VEC>(dst, f0, jn);
                st4<VEC>(dst + H, f1, jn);
                st4<VEC>(dst + 2 * H, f2, jn);
                st4<VEC>(dst + 3 * H, f3, jn);
            }
            prefetch(t - 1);
        }
        lds_barrier();
        if (work) {
            // dh_{t-1} += W_hh^T dgh ; dx = W_ih^T dgi.  Four chains, interleaved.
            f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, ax0 = ah0, ax2 = ah0, ay0 = ah0, ay2 = ah0;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 fr4 = *reinterpret_cast<const float4 *>(gb + ((0 * NT + q) * 64 + lane) * 4);
                const float4 fz4 = *reinterpret_cast<const float4 *>(gb + ((1 * NT + q) * 64 + lane) * 4);
                const float4 fn4 = *reinterpret_cast<const float4 *>(gb + ((2 * NT + q) * 64 + lane) * 4);
                const float4 fg4 = *reinterpret_cast<const float4 *>(gb + ((3 * NT + q) * 64 + lane) * 4);
                F4ARR(fr, fr4); F4ARR(fz, fz4); F4ARR(fn, fn4); F4ARR(fg, fg4);
                F4ARR(h0w, whhT[0][q]); F4ARR(h1w, whhT[1][q]); F4ARR(h2w, whhT[2][q]);
                F4ARR(x0w, wihT[0][0][q]); F4ARR(x1w, wihT[0][1][q]); F4ARR(x2w, wihT[0][2][q]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ah0 = mf(h0w[c], fr[c], ah0);
                    ax0 = mf(x0w[c], fr[c], ax0);
                    ah1 = mf(h1w[c], fz[c], ah1);
                    ax2 = mf(x1w[c], fz[c], ax2);
                    ah0 = mf(h2w[c], fg[c], ah0);
                    ax0 = mf(x2w[c], fn[c], ax0);
                }
                mfma_drain();
                if (two_dx) {
                    F4ARR(y0w, wihT[1][0][q]); F4ARR(y1w, wihT[1][1][q]); F4ARR(y2w, wihT[1][2][q]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ay0 = mf(y0w[c], fr[c], ay0);
                        ay2 = mf(y1w[c], fz[c], ay2);
                        ay0 = mf(y2w[c], fn[c], ay0);
                    }
                }
            }
            mfma_drain();
#pragma unroll
            for (int c = 0; c < 4; ++c) dhc[c] = keep[c] + (active ? ah0[c] + ah1[c] : 0.f);
            const float4 dx0 = make_float4(ax0[0] + ax2[0], ax0[1] + ax2[1], ax0[2] + ax2[2], ax0[3] + ax2[3]);
            if (l > 0) {
                *reinterpret_cast<float4 *>(dxbuf + (long)p * L * NT * 256 + ((l * NT + tile) * 64 + lane) * 4) = dx0;
            } else {
                qx0 = dx0;
                qx1 = make_float4(ay0[0] + ay2[0], ay0[1] + ay2[1], ay0[2] + ay2[2], ay0[3] + ay2[3]);
                qt = t; qpend = true;
            }
        }
        lds_barrier();
        p ^= 1;
    }
    if (qpend) flush_dx();

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a.dh0 && bvalid && c < jn) a.dh0[((long)l * B + bglob) * H + j0 + c] = dhc[c];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = dbs[g][c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (b == 0 && c < jn) a.db_part[((long)blockIdx.x * L + l) * 4 * H + g * H + j0 + c] = v;
        }
    }
}

template <int NT, int MAXT>
static hipError_t launch_bwd_t(const SmallBwdArgs &a, dim3 grid, dim3 block, size_t lds, bool vec, hipStream_t s) {
    const void *fn = vec ? (const void *)gru_small_bwd_kernel<NT, MAXT, true>
                         : (const void *)gru_small_bwd_kernel<NT, MAXT, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (vec) TGRU_LAUNCH((gru_small_bwd_kernel<NT, MAXT, true>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_small_bwd_kernel<NT, MAXT, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_small_bwd(const SmallBwdArgs &a, hipStream_t s) {
    if (h24_shape(a.In, a.H)) return launch_gru_h24_bwd(a, s);
    const int NT = (a.H + 15) / 16;
    dim3 grid(gru_batch_blocks(a.B)), block(64 * NT * a.L);
    const size_t lds = ((size_t)a.L * 4 * NT + (size_t)2 * a.L * NT) * 1024;
    const bool big = block.x > 512, vec = all_vec(a.In, a.H);
    if (NT == 1) return big ? launch_bwd_t<1, 1024>(a, grid, block, lds, vec, s) : launch_bwd_t<1, 512>(a, grid, block, lds, vec, s);
    return big ? launch_bwd_t<2, 1024>(a, grid, block, lds, vec, s) : launch_bwd_t<2, 512>(a, grid, block, lds, vec, s);
}

}  // namespace tgru
