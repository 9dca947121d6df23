// Whole-stack GRU kernels specialised for 16 < H <= 24 and In <= 24 (TimeGAN's H = 24 configs).
//
// Same structure as gru_small.hip (one launch per stack, one workgroup per 16 sequences, layers as a
// wavefront, weights in VGPRs, state exchanged as MFMA fragments through LDS) but without the MFMA
// work that padding H = 24 to 32 costs there:
//   * the reduction dimension is walked as k-block 0 (k 0..15, four k-steps, k = 4g + c) and
//     k-block 1 (k 16..23, TWO k-steps, k = 16 + 4c' + g), so no k-step multiplies padding;
//   * hidden units 16..23 have only 8 rows per gate, so the second wave of a layer packs two gates
//     into one 16-row MFMA tile: P = [r | z] and Q = [W_in x | W_hn h]; lanes 0-31 then hold
//     (r, gi_n) and lanes 32-63 hold (z, gh_n) of the same hidden unit, and two
//     v_permlane32_swap per register bring them together;
//   * MFMAs per layer and step: 36 (units 0..15) + 24 (units 16..23) instead of 96, and waves are
//     numbered so that the two waves sharing a SIMD are one of each kind.
// Backward mirrors this: the second wave packs [dh | dx] of units / inputs 16..23 into one tile.
#include "tgru_internal.h"

namespace tgru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef TGRU_STAMPS
__device__ unsigned long long tgru_stamps_h24[16 * 8];
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
        for (int k_ = 0; k_ < 6; ++k_) tgru_stamps_h24[wave * 8 + k_] = st_acc[k_];
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif

namespace h24 {

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

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

// value held by lane l+32 delivered to lane l (l < 32); upper lanes get an unspecified value
__device__ __forceinline__ float from_upper(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[1]);
}

// One operand row of W (row-major, K columns) as A fragments: block 0 -> 4 values k = 4g+c,
// block 1 -> 2 values k = 16+4c'+g.  `ok` false gives zeros.
struct AFrag { float a[4]; float b[2]; };
__device__ __forceinline__ AFrag load_row_frag(const float *w_row, int K, int g, bool ok) {
    AFrag f;
#pragma unroll
    for (int c = 0; c < 4; ++c) f.a[c] = (ok && 4 * g + c < K) ? w_row[4 * g + c] : 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) f.b[c] = (ok && 16 + 4 * c + g < K) ? w_row[16 + 4 * c + g] : 0.f;
    return f;
}
// Same for a COLUMN of W (the transposed products of the backward pass): element k is W[k][col].
__device__ __forceinline__ AFrag load_col_frag(const float *w, int ld, int nrows, int col, int g, bool ok) {
    AFrag f;
#pragma unroll
    for (int c = 0; c < 4; ++c) f.a[c] = (ok && 4 * g + c < nrows) ? w[(long)(4 * g + c) * ld + col] : 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) f.b[c] = (ok && 16 + 4 * c + g < nrows) ? w[(long)(16 + 4 * c + g) * ld + col] : 0.f;
    return f;
}

// B fragments of one 24-wide vector set: block 0 float4 (lane-linear), block 1 float2 (lane-linear)
struct BFrag { float a[4]; float b[2]; };
__device__ __forceinline__ BFrag lds_read_frag(const float *base, int lane) {   // base: 384 floats
    const float4 u = *reinterpret_cast<const float4 *>(base + lane * 4);
    const float2 v = *reinterpret_cast<const float2 *>(base + 256 + lane * 2);
    BFrag f;
    f.a[0] = u.x; f.a[1] = u.y; f.a[2] = u.z; f.a[3] = u.w; f.b[0] = v.x; f.b[1] = v.y;
    return f;
}
// block 1 writer: lane (b, g1) with g1 < 2 holds units 16 + 4*g1 + reg; element goes to the slot
// that reader lane (b, g = reg) takes at k-step c' = g1
__device__ __forceinline__ void lds_write_block1(float *base, int b, int g1, const float v[4]) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) base[256 + (reg * 16 + b) * 2 + g1] = v[reg];
}

#define ACC6(acc, A, Bf)                                              \
    do {                                                              \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) acc = mf((A).a[c_], (Bf).a[c_], acc); \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) acc = mf((A).b[c_], (Bf).b[c_], acc); \
    } while (0)

constexpr int kFrag = 384;   // floats per 24-wide fragment set (256 + 128)

// Branch-free predication for the FAST variants: every global access is a raw buffer access; a lane
// that must not touch memory uses an offset past the end of the buffer (loads return 0, stores are
// dropped by the bounds check) and an absent tensor gets a descriptor with zero records.  With no
// control flow around them the compiler schedules these accesses and their address arithmetic into
// the shadow of the MFMAs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int off, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, int off, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, int off, int soff, float4 v) {
    const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, off, soff, 0);
}

}  // namespace h24

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int MAXT, bool VEC, bool FAST>
__global__ __launch_bounds__(MAXT) void gru_h24_fwd_kernel(SmallFwdArgs a) {
    using namespace h24;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hbuf[2][L][kFrag]
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave >= L ? 1 : 0;           // waves [0,L): units 0..15, waves [L,2L): units 16..H-1
    const int l = tile ? wave - L : wave;
    const int b = lane & 15, gq = lane >> 4;
    const int g1 = gq & 1;
    const bool upper = gq >= 2;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const float *wi = a.w_ih[l], *wh = a.w_hh[l], *bi = a.b_ih[l], *bh = a.b_hh[l];
    float *yint = a.y_int[l], *gates = a.gates[l];
    const long plane = (long)T * B * H;
    const int i16 = lane & 15;

    // ---- weights as A fragments -----------------------------------------------------------------
    // tile 0: X[g], Hh[g] for g = r, z, n on rows g*H + i.
    // tile 1: P (rows i<8: r unit 16+i, i>=8: z unit 16+i-8) and Q (i<8: W_in x only, i>=8: W_hn h only)
    AFrag X[3], Hh[3];
    if (tile == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            X[g] = load_row_frag(wi + (long)(g * H + i16) * In_l, In_l, gq, true);
            Hh[g] = load_row_frag(wh + (long)(g * H + i16) * H, H, gq, true);
        }
    } else {
        const int u = 16 + (i16 & 7);
        const bool uok = u < H;
        const int gP = i16 < 8 ? 0 : 1;
        X[0] = load_row_frag(wi + (long)(gP * H + u) * In_l, In_l, gq, uok);
        Hh[0] = load_row_frag(wh + (long)(gP * H + u) * H, H, gq, uok);
        X[1] = load_row_frag(wi + (long)(2 * H + u) * In_l, In_l, gq, uok && i16 < 8);
        Hh[1] = load_row_frag(wh + (long)(2 * H + u) * H, H, gq, uok && i16 >= 8);
        X[2] = X[1]; Hh[2] = Hh[1];   // unused
    }

    // ---- per-lane output units, biases, state ---------------------------------------------------
    // tile 0: lane (b,gq) owns units 4gq..4gq+3.  tile 1: lanes gq<2 own units 16+4gq..; lanes gq>=2
    // only carry the z / gh_n halves of the packed tiles.
    const int j0 = tile ? 16 + 4 * g1 : 4 * gq;
    const int jn = min(4, max(0, H - j0));
    const bool owner = tile == 0 || !upper;
    float b0[4], b1[4], b2[4], b3[4], hreg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool ok = c < jn;
        const int j = j0 + c;
        if (tile == 0) {
            b0[c] = ok ? bi[j] + bh[j] : 0.f;             // r
            b1[c] = ok ? bi[H + j] + bh[H + j] : 0.f;     // z
            b2[c] = ok ? bi[2 * H + j] : 0.f;             // b_in
            b3[c] = ok ? bh[2 * H + j] : 0.f;             // b_hn
        } else {
            b0[c] = ok ? (upper ? bi[H + j] + bh[H + j] : bi[j] + bh[j]) : 0.f;   // P: z (upper) / r (lower)
            b1[c] = ok ? (upper ? bh[2 * H + j] : bi[2 * H + j]) : 0.f;           // Q: b_hn / b_in
            b2[c] = 0.f; b3[c] = 0.f;
        }
        hreg[c] = (ok && owner && bvalid && a.h0) ? a.h0[((long)l * B + bglob) * H + j] : 0.f;
    }
    // initial state into both parities
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        float *hb = smem + ((long)par * L + l) * kFrag;
        if (tile == 0) *reinterpret_cast<float4 *>(hb + lane * 4) = make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
        else if (!upper) lds_write_block1(hb, b, g1, hreg);
    }
    lds_barrier();

    // ---- layer 0 input straight from global memory, one step ahead --------------------------------
    BFrag xn;
#pragma unroll
    for (int c = 0; c < 4; ++c) xn.a[c] = 0.f;
    xn.b[0] = xn.b[1] = 0.f;
    // FAST: descriptors and per-lane byte offsets (row t adds a wave-uniform term)
    const long tbh = (long)T * B * H * 4;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(l == 0 ? a.x : nullptr, (long)T * B * a.In * 4);
    const __amdgpu_buffer_rsrc_t rs_yint = make_rsrc(yint, tbh);
    const __amdgpu_buffer_rsrc_t rs_yext = make_rsrc(l == L - 1 ? a.y_ext : nullptr, tbh);
    const __amdgpu_buffer_rsrc_t rs_gates = make_rsrc(gates, 4 * tbh);
    const __amdgpu_buffer_rsrc_t rs_hn = make_rsrc(a.h_n, (long)L * B * H * 4);
    const bool lane_ok = bvalid && owner && jn > 0;
    const int x_sb = a.xmap.mode == 1 ? T : 1, x_st = a.xmap.mode == 1 ? 1 : B;    // row = t*st + b*sb
    const int y_sb = a.ymap.mode == 1 ? T : 1, y_st = a.ymap.mode == 1 ? 1 : B;
    const int xoff0 = bglob * x_sb * a.In * 4, xstep = x_st * a.In * 4;
    const int ioff0 = (bglob * H + j0) * 4, istep = B * H * 4;
    const int eoff0 = (bglob * y_sb * H + j0) * 4, estep = y_st * H * 4;
    const int hnoff = ((l * B + bglob) * H + j0) * 4;

    auto load_x = [&](int t) {
        if (FAST) {
            const int base = (bvalid && t < len) ? xoff0 + t * xstep : kOOB;
            const float4 u = buf_ld4(rs_x, (4 * gq < a.In) ? base + 16 * gq : kOOB, 0);
            xn.a[0] = u.x; xn.a[1] = u.y; xn.a[2] = u.z; xn.a[3] = u.w;
            xn.b[0] = buf_ld1(rs_x, (16 + gq < a.In) ? base + (16 + gq) * 4 : kOOB, 0);
            xn.b[1] = buf_ld1(rs_x, (20 + gq < a.In) ? base + (20 + gq) * 4 : kOOB, 0);
            return;
        }
        if (t < len) {
            const long er = map_tb(a.xmap, t, bglob);
            if (er >= 0) {
                const float *xr = a.x + er * a.In;
                const float4 u = ld4<VEC>(xr + 4 * gq, a.In - 4 * gq);
                xn.a[0] = u.x; xn.a[1] = u.y; xn.a[2] = u.z; xn.a[3] = u.w;
                xn.b[0] = (16 + gq < a.In) ? xr[16 + gq] : 0.f;
                xn.b[1] = (20 + gq < a.In) ? xr[20 + gq] : 0.f;
            }
        }
    };
    if (l == 0) load_x(0);

    // ---- results of the previous tick, stored at the top of the next one ---------------------------
    float4 py = make_float4(0.f, 0.f, 0.f, 0.f), pr = py, pz = py, pn = py, phn = py;
    int pt = 0;
    bool pact = false, pend = false;
    auto flush = [&]() {
        if (FAST) {
            const bool ok = lane_ok && pend;
            const int oi = ok ? ioff0 + pt * istep : kOOB;
            buf_st4(rs_yint, oi, 0, py);
            buf_st4(rs_yext, ok ? eoff0 + pt * estep : kOOB, 0, py);
            const int og = (ok && pact) ? oi : kOOB;
            buf_st4(rs_gates, og, 0, pr);
            buf_st4(rs_gates, og, (int)tbh, pz);
            buf_st4(rs_gates, og, (int)(2 * tbh), pn);
            buf_st4(rs_gates, og, (int)(3 * tbh), phn);
            buf_st4(rs_hn, (ok && pt == len - 1) ? hnoff : kOOB, 0, py);
            return;
        }
        if (!pend) return;
        if (bvalid && owner && jn > 0) {
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
        BFrag xin = xn;
        if (l == 0) asm volatile("" : "+v"(xin.a[0]), "+v"(xin.a[1]), "+v"(xin.a[2]), "+v"(xin.a[3]),
                                      "+v"(xin.b[0]), "+v"(xin.b[1]));
        if (!FAST || !work) flush();
        STAMP(0);
        if (work) {
            const bool active = t < len;
            const float *hbase = smem + (long)p * L * kFrag;
            if (l == 0) {
                if (t + 1 < T) load_x(t + 1);
            } else {
                xin = lds_read_frag(hbase + (l - 1) * kFrag, lane);
            }
            const BFrag hp = lds_read_frag(hbase + l * kFrag, lane);
            float rv[4], zv[4], nv[4], hnv[4], yv[4];
            STAMP(1);
            if (tile == 0) {
                f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ain = ar, ahn = ar;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ar = mf(X[0].a[c], xin.a[c], ar); az = mf(X[1].a[c], xin.a[c], az); ain = mf(X[2].a[c], xin.a[c], ain);
                    ar = mf(Hh[0].a[c], hp.a[c], ar); az = mf(Hh[1].a[c], hp.a[c], az); ahn = mf(Hh[2].a[c], hp.a[c], ahn);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    ar = mf(X[0].b[c], xin.b[c], ar); az = mf(X[1].b[c], xin.b[c], az); ain = mf(X[2].b[c], xin.b[c], ain);
                    ar = mf(Hh[0].b[c], hp.b[c], ar); az = mf(Hh[1].b[c], hp.b[c], az); ahn = mf(Hh[2].b[c], hp.b[c], ahn);
                }
                if (FAST) flush();   // independent of the MFMAs above: scheduled into their shadow
                mfma_drain();
                STAMP(2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float ghn = ahn[c] + b3[c];
                    const float r = fast_sigmoid(ar[c] + b0[c]);
                    const float z = fast_sigmoid(az[c] + b1[c]);
                    const float n = fast_tanh(ain[c] + b2[c] + r * ghn);
                    const float hnew = n + z * (hreg[c] - n);
                    rv[c] = r; zv[c] = z; nv[c] = n; hnv[c] = ghn;
                    if (active && c < jn) hreg[c] = hnew;
                    yv[c] = active ? hreg[c] : 0.f;
                }
                *reinterpret_cast<float4 *>(smem + ((long)(p ^ 1) * L + l) * kFrag + lane * 4) =
                    make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
            } else {
                f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aQ = aP;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    aP = mf(X[0].a[c], xin.a[c], aP); aQ = mf(X[1].a[c], xin.a[c], aQ);
                    aP = mf(Hh[0].a[c], hp.a[c], aP); aQ = mf(Hh[1].a[c], hp.a[c], aQ);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    aP = mf(X[0].b[c], xin.b[c], aP); aQ = mf(X[1].b[c], xin.b[c], aQ);
                    aP = mf(Hh[0].b[c], hp.b[c], aP); aQ = mf(Hh[1].b[c], hp.b[c], aQ);
                }
                if (FAST) flush();
                mfma_drain();
                STAMP(2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float s = fast_sigmoid(aP[c] + b0[c]);   // lower: r, upper: z
                    const float q = aQ[c] + b1[c];                 // lower: W_in x + b_in, upper: W_hn h + b_hn
                    const float ghn = from_upper(q);
                    const float z = from_upper(s);
                    const float n = fast_tanh(q + s * ghn);
                    const float hnew = n + z * (hreg[c] - n);
                    rv[c] = s; zv[c] = z; nv[c] = n; hnv[c] = ghn;
                    if (active && c < jn && !upper) hreg[c] = hnew;
                    yv[c] = active ? hreg[c] : 0.f;
                }
                if (!upper) lds_write_block1(smem + ((long)(p ^ 1) * L + l) * kFrag, b, g1, hreg);
            }
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
    flush();
    STAMP_FLUSH
}

#ifdef TGRU_STAMPS
extern "C" int tgru_debug_stamps_h24(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_h24), sizeof(unsigned long long) * 16 * 8);
}
#endif

hipError_t launch_gru_h24_fwd(const SmallFwdArgs &a, hipStream_t s) {
    dim3 grid(gru_batch_blocks(a.B)), block(128 * a.L);
    const size_t lds = (size_t)2 * a.L * h24::kFrag * sizeof(float);
    const bool big = block.x > 512, vec = (a.In & 3) == 0 && (a.H & 3) == 0;
    // FAST needs closed-form row addressing (no packed rows) and 32-bit byte offsets
    const bool fast = vec && !big && a.xmap.mode != 2 && a.ymap.mode != 2 &&
                      (long)a.T * a.B * a.H * 16 < (1L << 30) && (long)a.T * a.B * a.In * 4 < (1L << 30);
    if (fast) TGRU_LAUNCH((gru_h24_fwd_kernel<512, true, true>), grid, block, lds, s, a);
    else if (!big && vec) TGRU_LAUNCH((gru_h24_fwd_kernel<512, true, false>), grid, block, lds, s, a);
    else if (!big) TGRU_LAUNCH((gru_h24_fwd_kernel<512, false, false>), grid, block, lds, s, a);
    else if (vec) TGRU_LAUNCH((gru_h24_fwd_kernel<1024, true, false>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_h24_fwd_kernel<1024, false, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
template <int MAXT, bool VEC, bool FAST>
__global__ __launch_bounds__(MAXT) void gru_h24_bwd_kernel(SmallBwdArgs a) {
    using namespace h24;
    // gbuf[L][4 kinds][kFrag] : da_r | da_z | da_n | da_n*r ; dxbuf[2][L][512] : tile-0 dx, tile-1 packed D
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave >= L ? 1 : 0;
    const int l = tile ? wave - L : wave;
    const int b = lane & 15, gq = lane >> 4;
    const int g1 = gq & 1;
    const bool upper = gq >= 2;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const int i16 = lane & 15;
    const long plane = (long)T * B * H;
    float *gb = smem + (long)l * 4 * kFrag;
    float *dxbuf = smem + (long)L * 4 * kFrag;
    const float *wi = a.w_ih[l], *wh = a.w_hh[l], *yint = a.y_int[l], *gates = a.gates[l];
    float *dgl = a.dg[l];

    // ---- transposed weights as A fragments --------------------------------------------------------
    // tile 0: HT[g] = column i of W_hh's gate-g block (k = unit inside the gate), XT[g] same for W_ih.
    // tile 1: rows i<8 -> dh of unit 16+i (W_hh column), rows i>=8 -> dx of input 16+i-8 (W_ih column):
    //         K0 pairs with da_r, K1 with da_z, K2 with da_n*r (dh rows only), K3 with da_n (dx rows only)
    AFrag HT[3], XT[3], K3;
    if (tile == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            HT[g] = load_col_frag(wh + (long)g * H * H, H, H, i16, gq, true);
            XT[g] = load_col_frag(wi + (long)g * H * In_l, In_l, H, i16, gq, i16 < In_l);
        }
        K3 = HT[0];   // unused
    } else {
        const int col = 16 + (i16 & 7);
        const bool hrow = i16 < 8;
        const bool okh = hrow && col < H, okx = !hrow && col < In_l;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const AFrag fh = load_col_frag(wh + (long)g * H * H, H, H, col, gq, okh);
            const AFrag fx = load_col_frag(wi + (long)g * H * In_l, In_l, H, col, gq, okx);
            AFrag m;
#pragma unroll
            for (int c = 0; c < 4; ++c) m.a[c] = hrow ? fh.a[c] : fx.a[c];
            m.b[0] = hrow ? fh.b[0] : fx.b[0]; m.b[1] = hrow ? fh.b[1] : fx.b[1];
            HT[g] = m;
        }
        HT[2] = load_col_frag(wh + (long)2 * H * H, H, H, col, gq, okh);          // pairs with da_n*r
        K3 = load_col_frag(wi + (long)2 * H * In_l, In_l, H, col, gq, okx);       // pairs with da_n
        XT[0] = XT[1] = XT[2] = K3;   // unused
    }

    const int j0 = tile ? 16 + 4 * g1 : 4 * gq;
    const int jn = min(4, max(0, H - j0));
    const bool owner = tile == 0 || !upper;

    float dhc[4] = {0.f, 0.f, 0.f, 0.f};
    float dbs[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) dbs[g][c] = 0.f;

    // FAST: descriptors and per-lane byte offsets
    const long tbh = (long)T * B * H * 4;
    const __amdgpu_buffer_rsrc_t rs_gates = make_rsrc(gates, 4 * tbh);
    const __amdgpu_buffer_rsrc_t rs_yint = make_rsrc(yint, tbh);
    const __amdgpu_buffer_rsrc_t rs_h0 = make_rsrc(a.h0, (long)L * B * H * 4);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(l == L - 1 ? a.dy_ext : nullptr, tbh);
    const __amdgpu_buffer_rsrc_t rs_dg = make_rsrc(dgl, 4 * tbh);
    const __amdgpu_buffer_rsrc_t rs_dx = make_rsrc(l == 0 ? a.dx_ext : nullptr, (long)T * B * a.In * 4);
    const bool lane_ok = bvalid && owner && jn > 0;
    const int ioff0 = (bglob * H + j0) * 4, istep = B * H * 4;
    const int y_sb = a.dymap.mode == 1 ? T : 1, y_st = a.dymap.mode == 1 ? 1 : B;
    const int eoff0 = (bglob * y_sb * H + j0) * 4, estep = y_st * H * 4;
    const int goff0 = (bglob * 4 * H + j0) * 4, gstep = B * 4 * H * 4;
    const int x_sb = a.dxmap.mode == 1 ? T : 1, x_st = a.dxmap.mode == 1 ? 1 : B;
    const int dxc0 = tile ? 16 + 4 * g1 : 4 * gq;
    const bool dx_lane = bvalid && (tile == 0 || upper) && dxc0 < a.In;
    const int dxoff0 = (bglob * x_sb * a.In + dxc0) * 4, dxstep = x_st * a.In * 4;
    const int hnoff = ((l * B + bglob) * H + j0) * 4;

    float4 pr, pz, pn, phn, php, pdy;
    auto prefetch = [&](int t) {
        if (FAST) {
            const bool ok = lane_ok && t >= 0 && t < len;
            const int oi = ok ? ioff0 + t * istep : kOOB;
            pr = buf_ld4(rs_gates, oi, 0);
            pz = buf_ld4(rs_gates, oi, (int)tbh);
            pn = buf_ld4(rs_gates, oi, (int)(2 * tbh));
            phn = buf_ld4(rs_gates, oi, (int)(3 * tbh));
            if (t > 0) php = buf_ld4(rs_yint, ok ? oi - istep : kOOB, 0);
            else php = buf_ld4(rs_h0, ok ? hnoff : kOOB, 0);
            pdy = buf_ld4(rs_dy, ok ? eoff0 + t * estep : kOOB, 0);
            if (a.dh_n) {
                float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && t == len - 1) e = *reinterpret_cast<const float4 *>(a.dh_n + ((long)l * B + bglob) * H + j0);
                pdy.x += e.x; pdy.y += e.y; pdy.z += e.z; pdy.w += e.w;
            }
            return;
        }
        pr = pz = pn = phn = php = pdy = make_float4(0.f, 0.f, 0.f, 0.f);
        if (owner && t >= 0 && t < len && jn > 0) {
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
    prefetch(T - 1);

    // layer 0: dx of the previous tick, stored at the top of the next one
    float4 qx = make_float4(0.f, 0.f, 0.f, 0.f);
    int qt = 0;
    bool qpend = false;
    auto flush_dx = [&]() {
        if (FAST) {
            buf_st4(rs_dx, (dx_lane && qpend) ? dxoff0 + qt * dxstep : kOOB, 0, qx);
            return;
        }
        if (!qpend) return;
        if (a.dx_ext && bvalid) {
            const long er = map_tb(a.dxmap, qt, bglob);
            // tile 0: every lane holds inputs 4gq..; tile 1: the upper lanes hold inputs 16+4*g1..
            const int c0 = tile ? 16 + 4 * g1 : 4 * gq;
            if (er >= 0 && (tile == 0 || upper) && c0 < a.In) st4<VEC>(a.dx_ext + er * a.In + c0, qx, a.In - c0);
        }
    };

    int p = 0;
    const int nticks = T + L - 1;
    for (int tick = 0; tick < nticks; ++tick) {
        const int t = T - 1 - tick + (L - 1 - l);
        const bool work = t >= 0 && t < T;
        const bool active = work && t < len;
        float keep[4] = {dhc[0], dhc[1], dhc[2], dhc[3]};
        flush_dx(); qpend = false;
        if (work) {
            float dyv[4] = {pdy.x, pdy.y, pdy.z, pdy.w};
            if (l < L - 1) {   // dx of the layer above, produced at the previous tick
                const float *src = dxbuf + ((long)(p ^ 1) * L + (l + 1)) * 512;
                // tile 0 reads the tile-0 fragment at its own lane; tile 1 (lower lanes) reads the packed
                // tile's upper half, i.e. the fragment written by lane + 32
                const float4 u = *reinterpret_cast<const float4 *>(src + (tile ? 256 + ((lane + 32) & 63) * 4 : lane * 4));
                dyv[0] += u.x; dyv[1] += u.y; dyv[2] += u.z; dyv[3] += u.w;
            }
            const float rr[4] = {pr.x, pr.y, pr.z, pr.w}, zz[4] = {pz.x, pz.y, pz.z, pz.w};
            const float nn[4] = {pn.x, pn.y, pn.z, pn.w}, hh[4] = {phn.x, phn.y, phn.z, phn.w};
            const float hq[4] = {php.x, php.y, php.z, php.w};
            float dar[4], daz[4], dan[4], dgn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool on = active && owner && c < jn;
                const float d = on ? dyv[c] + dhc[c] : 0.f;
                const float dn = d * (1.f - zz[c]);
                const float dzz = d * (hq[c] - nn[c]);
                const float an = dn * (1.f - nn[c] * nn[c]);
                dan[c] = on ? an : 0.f;
                dar[c] = on ? an * hh[c] * rr[c] * (1.f - rr[c]) : 0.f;
                daz[c] = on ? dzz * zz[c] * (1.f - zz[c]) : 0.f;
                dgn[c] = on ? an * rr[c] : 0.f;
                keep[c] = (active && owner) ? d * zz[c] : dhc[c];
                dbs[0][c] += dar[c]; dbs[1][c] += daz[c]; dbs[2][c] += dan[c]; dbs[3][c] += dgn[c];
            }
            if (tile == 0) {
                *reinterpret_cast<float4 *>(gb + 0 * kFrag + lane * 4) = make_float4(dar[0], dar[1], dar[2], dar[3]);
                *reinterpret_cast<float4 *>(gb + 1 * kFrag + lane * 4) = make_float4(daz[0], daz[1], daz[2], daz[3]);
                *reinterpret_cast<float4 *>(gb + 2 * kFrag + lane * 4) = make_float4(dan[0], dan[1], dan[2], dan[3]);
                *reinterpret_cast<float4 *>(gb + 3 * kFrag + lane * 4) = make_float4(dgn[0], dgn[1], dgn[2], dgn[3]);
            } else if (!upper) {
                lds_write_block1(gb + 0 * kFrag, b, g1, dar);
                lds_write_block1(gb + 1 * kFrag, b, g1, daz);
                lds_write_block1(gb + 2 * kFrag, b, g1, dan);
                lds_write_block1(gb + 3 * kFrag, b, g1, dgn);
            }
            if (FAST) {
                const int og = lane_ok ? goff0 + t * gstep : kOOB;
                buf_st4(rs_dg, og, 0, make_float4(dar[0], dar[1], dar[2], dar[3]));
                buf_st4(rs_dg, og, H * 4, make_float4(daz[0], daz[1], daz[2], daz[3]));
                buf_st4(rs_dg, og, 2 * H * 4, make_float4(dan[0], dan[1], dan[2], dan[3]));
                buf_st4(rs_dg, og, 3 * H * 4, make_float4(dgn[0], dgn[1], dgn[2], dgn[3]));
            } else if (bvalid && owner && jn > 0) {
                float *dst = dgl + ((long)t * B + bglob) * 4 * H + j0;
                st4<VEC>(dst, make_float4(dar[0], dar[1], dar[2], dar[3]), jn);
                st4<VEC>(dst + H, make_float4(daz[0], daz[1], daz[2], daz[3]), jn);
                st4<VEC>(dst + 2 * H, make_float4(dan[0], dan[1], dan[2], dan[3]), jn);
                st4<VEC>(dst + 3 * H, make_float4(dgn[0], dgn[1], dgn[2], dgn[3]), jn);
            }
            prefetch(t - 1);
        }
        lds_barrier();
        if (work) {
            const BFrag fr = lds_read_frag(gb + 0 * kFrag, lane);
            const BFrag fz = lds_read_frag(gb + 1 * kFrag, lane);
            const BFrag fn = lds_read_frag(gb + 2 * kFrag, lane);
            const BFrag fg = lds_read_frag(gb + 3 * kFrag, lane);
            float *dxw = dxbuf + ((long)p * L + l) * 512;
            if (tile == 0) {
                f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, ax0 = ah0, ax1 = ah0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ah0 = mf(HT[0].a[c], fr.a[c], ah0); ax0 = mf(XT[0].a[c], fr.a[c], ax0);
                    ah1 = mf(HT[1].a[c], fz.a[c], ah1); ax1 = mf(XT[1].a[c], fz.a[c], ax1);
                    ah0 = mf(HT[2].a[c], fg.a[c], ah0); ax0 = mf(XT[2].a[c], fn.a[c], ax0);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    ah0 = mf(HT[0].b[c], fr.b[c], ah0); ax0 = mf(XT[0].b[c], fr.b[c], ax0);
                    ah1 = mf(HT[1].b[c], fz.b[c], ah1); ax1 = mf(XT[1].b[c], fz.b[c], ax1);
                    ah0 = mf(HT[2].b[c], fg.b[c], ah0); ax0 = mf(XT[2].b[c], fn.b[c], ax0);
                }
                mfma_drain();
#pragma unroll
                for (int c = 0; c < 4; ++c) dhc[c] = keep[c] + (active ? ah0[c] + ah1[c] : 0.f);
                const float4 dx0 = make_float4(ax0[0] + ax1[0], ax0[1] + ax1[1], ax0[2] + ax1[2], ax0[3] + ax1[3]);
                if (l > 0) *reinterpret_cast<float4 *>(dxw + lane * 4) = dx0;
                else { qx = dx0; qt = t; qpend = true; }
            } else {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a0 = mf(HT[0].a[c], fr.a[c], a0); a1 = mf(HT[1].a[c], fz.a[c], a1);
                    a0 = mf(HT[2].a[c], fg.a[c], a0); a1 = mf(K3.a[c], fn.a[c], a1);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    a0 = mf(HT[0].b[c], fr.b[c], a0); a1 = mf(HT[1].b[c], fz.b[c], a1);
                    a0 = mf(HT[2].b[c], fg.b[c], a0); a1 = mf(K3.b[c], fn.b[c], a1);
                }
                mfma_drain();
                // lower lanes: dh of units 16+4*g1..; upper lanes: dx of inputs 16+4*g1..
                const float4 d4 = make_float4(a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]);
                const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) dhc[c] = upper ? 0.f : keep[c] + (active ? dv[c] : 0.f);
                if (l > 0) *reinterpret_cast<float4 *>(dxw + 256 + lane * 4) = d4;
                else { qx = d4; qt = t; qpend = true; }
            }
        }
        lds_barrier();
        p ^= 1;
    }
    flush_dx();

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a.dh0 && bvalid && owner && c < jn) a.dh0[((long)l * B + bglob) * H + j0 + c] = dhc[c];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = dbs[g][c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (b == 0 && owner && c < jn) a.db_part[((long)blockIdx.x * L + l) * 4 * H + g * H + j0 + c] = v;
        }
    }
}

hipError_t launch_gru_h24_bwd(const SmallBwdArgs &a, hipStream_t s) {
    dim3 grid(gru_batch_blocks(a.B)), block(128 * a.L);
    const size_t lds = ((size_t)a.L * 4 * h24::kFrag + (size_t)2 * a.L * 512) * sizeof(float);
    const bool big = block.x > 512, vec = (a.In & 3) == 0 && (a.H & 3) == 0;
    const bool fast = vec && !big && a.dymap.mode != 2 && a.dxmap.mode != 2 &&
                      (long)a.T * a.B * a.H * 16 < (1L << 30) && (long)a.T * a.B * a.In * 4 < (1L << 30);
    if (fast) TGRU_LAUNCH((gru_h24_bwd_kernel<512, true, true>), grid, block, lds, s, a);
    else if (!big && vec) TGRU_LAUNCH((gru_h24_bwd_kernel<512, true, false>), grid, block, lds, s, a);
    else if (!big) TGRU_LAUNCH((gru_h24_bwd_kernel<512, false, false>), grid, block, lds, s, a);
    else if (vec) TGRU_LAUNCH((gru_h24_bwd_kernel<1024, true, false>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_h24_bwd_kernel<1024, false, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace tgru
