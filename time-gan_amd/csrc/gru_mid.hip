// Per-layer recurrent kernels for mid-size hidden states, 32 < H <= 128 (H % 4 == 0): the energy
// config's H = 96.
//
// A 16-column MFMA tile needs 16 sequences per workgroup: a batch of 1024 then occupies 64 of the 256
// CUs.  These kernels take FOUR sequences per workgroup (256 workgroups at B = 1024) and use
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction, at the same
// MAC rate as the large tiles.  Lane 4b+i supplies A_b[i], lane 4b+j supplies B_b[j], and register i
// of lane 4b+j receives D_b[i][j] += A_b[i] * B_b[j].  With
//      A_b[i] = W[64*wave + 4b + i][k]   (one weight-matrix row per lane, the row lives in VGPRs)
//      B_b[j] = v[k][j]                  (the step's vector of sequence j, the same in every block)
// one instruction per k accumulates a 64-row x 4-sequence slice of W . v, and a wave never touches
// memory for its weights.  Per step: MFMA phase (rows spread over ceil(3H/64) waves) -> results to
// LDS -> barrier -> gate phase ((unit, sequence) pairs spread over all threads) -> new state to LDS
// -> barrier.  The operands that come from global memory (the precomputed input projection gi of
// the step in forward; saved gates and incoming gradients in backward) are fetched several steps
// ahead by a loader wave with direct-to-LDS loads, so compute waves only ever store.
#include "tgru_internal.h"

namespace tgru {
namespace mid {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSeq = 4;            // sequences per workgroup
constexpr int kAhead = 4;          // steps the loader runs ahead
constexpr int kSlots = kAhead + 1;
constexpr int kOOB = 0x7ffffff0;
// LDS row stride of an HP-wide vector that the matrix phase reads as 16-byte pieces, lanes of the four
// sequences side by side: four floats of padding put the four rows' pieces in different banks (with rows a
// multiple of 32 floats apart every such read was a four-way bank conflict)
constexpr int padded(int hp) { return hp + 4; }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void wait_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_load_b128(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, off, 0, 0, 0);
}
// x[lane] + x[lane ^ 32]
__device__ __forceinline__ float add_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

}  // namespace mid

#ifdef TGRU_STAMPS   // diagnostic build only: time (100 MHz ticks) wave 0 of workgroup 0 spends per phase
__device__ unsigned long long tgru_stamps_mid[16];
#define MSTAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MSTAMP(k)                                                     \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += now_ - st_last; st_last = now_;                  \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define MSTAMP_FLUSH(base)                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0)                          \
        for (int k_ = 0; k_ < 8; ++k_) tgru_stamps_mid[(base) + k_] = st_acc[k_];
extern "C" int tgru_debug_stamps_mid(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_mid), sizeof(unsigned long long) * 16);
}
#else
#define MSTAMP_DECL
#define MSTAMP(k)
#define MSTAMP_FLUSH(base)
#endif

// ----------------------------------------------------------------------------------------------
// forward: h_t from gi_t (input projection incl. b_ih, precomputed for all steps) and h_{t-1}
// ----------------------------------------------------------------------------------------------
// BAL (H = 96 only): 3H = 288 rows are four and a half 64-row slices.  Five waves would put two on one SIMD
// and double the step's matrix time; instead four waves (one per SIMD) take a full slice each and share the
// last 32 rows: wave w covers k in [24w, 24w + 24) of them, its lanes 0-31 the first twelve k of that range
// and lanes 32-63 the second twelve (a 4x4x1 block may use any k: the two halves are added afterwards).
// Two more waves only take part in the gate phase, so that its 4H = 384 (sequence, unit) pairs are one per
// thread (with 256 threads two waves ran the phase twice); the loader is wave 6.
template <int HP, bool BAL>
__global__ __launch_bounds__(448) void gru_mid_fwd_kernel(FwdArgs a) {
    using namespace mid;
    constexpr int NI = (3 * HP + 63) / 64;        // 16-byte direct-to-LDS loads per step (4 rows of 3H floats)
    constexpr int HS = padded(HP);
    // hs[2][kSeq][HS] | ghs[kSeq][GS] | ghx[4 waves][kSeq][32] | gis[kSlots][NI*256]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = BAL ? HP : a.H, B = a.B, T = a.T;
    const int nw = BAL ? 4 : (3 * H + 63) / 64;   // waves with matrix rows
    const int ngw = BAL ? 6 : nw;                 // waves in the gate phase; wave ngw is the loader
    const int GS = nw * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *hs = smem;
    float *ghs = hs + 2 * kSeq * HS;
    float *ghx = ghs + kSeq * GS;
    float *gis = ghx + 4 * kSeq * 32;
    const int b0 = blockIdx.x * kSeq;

    if (wave == ngw) {
        // ---- loader: gi rows of the workgroup's four sequences, flattened [seq][3H], kAhead steps ahead --
        const __amdgpu_buffer_rsrc_t rs_gi = make_rsrc(a.gi, (long)T * B * 3 * H * 4);
        auto issue = [&](int t) {
            float *slot = gis + (t % kSlots) * NI * 256;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int e = (i * 64 + lane) * 4;              // element of the flattened [kSeq][3H] block
                const int j = e / (3 * H), off = e - j * 3 * H;
                const bool ok = t < T && j < kSeq && b0 + j < B;
                lds_load_b128(rs_gi, slot + i * 256, ok ? (((t * B + b0 + j) * 3 * H) + off) * 4 : kOOB);
            }
        };
        for (int t = 0; t < kAhead; ++t) issue(t);
        wait_vmcnt<(kAhead - 1) * NI>();
        lds_barrier();
        for (int t = 0; t < T; ++t) {
            issue(t + kAhead);
            lds_barrier();                        // A: end of the MFMA phase
            wait_vmcnt<(kAhead - 1) * NI>();      // gi of step t+1 has landed
            lds_barrier();                        // B: end of the gate phase
        }
        wait_vmcnt<0>();
        return;
    }

    // ---- one row of W_hh per lane ------------------------------------------------------------------
    const int row = wave * 64 + lane;
    float w[HP];
    {
        const bool ok = row < 3 * H;
        const float *src = a.w_hh + (long)row * H;
#pragma unroll
        for (int q = 0; q < HP / 4; ++q) {
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && 4 * q < H) u = *reinterpret_cast<const float4 *>(src + 4 * q);
            w[4 * q] = u.x; w[4 * q + 1] = u.y; w[4 * q + 2] = u.z; w[4 * q + 3] = u.w;
        }
    }
    // shared rows: row 256 + (lane & 31), k = kx .. kx + 11
    const int kx = 24 * wave + 12 * (lane >> 5);
    float wx[12] = {};
    if (BAL && wave < nw) {
        const float *src = a.w_hh + (long)(256 + (lane & 31)) * H + kx;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 u = *reinterpret_cast<const float4 *>(src + 4 * q);
            wx[4 * q] = u.x; wx[4 * q + 1] = u.y; wx[4 * q + 2] = u.z; wx[4 * q + 3] = u.w;
        }
    }
    // ---- gate phase: (sequence, unit) pairs p = j*H + u of this thread -------------------------------
    const int nthreads = ngw * 64;
    const int tid = threadIdx.x;
    constexpr int kPairs = BAL ? 1 : 2;           // 4*H pairs over >= 3*H threads
    int pj[kPairs], pu[kPairs], plen[kPairs];
    float bh_r[kPairs], bh_z[kPairs], bh_n[kPairs];
    bool pok[kPairs];
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
        const int p = tid + s * nthreads;
        const bool in = p < kSeq * H;
        pj[s] = in ? p / H : 0;
        pu[s] = in ? p % H : 0;
        pok[s] = in && b0 + pj[s] < B;            // a real sequence (the last workgroup may have fewer than four)
        plen[s] = pok[s] ? (a.lengths ? a.lengths[b0 + pj[s]] : T) : 0;
        bh_r[s] = a.b_hh[pu[s]];
        bh_z[s] = a.b_hh[H + pu[s]];
        bh_n[s] = a.b_hh[2 * H + pu[s]];
        const float h0v = (pok[s] && a.h0) ? a.h0[(long)(b0 + pj[s]) * H + pu[s]] : 0.f;
        if (p < kSeq * H) {
            hs[(0 * kSeq + p / H) * HS + p % H] = h0v;
            hs[(1 * kSeq + p / H) * HS + p % H] = h0v;
        }
    }
    // zero the padding of the state vectors (units H..HP-1)
    for (int i = tid; i < 2 * kSeq * (HP - H); i += nthreads) {
        const int v = i / (HP - H), c = i % (HP - H);
        hs[v * HS + H + c] = 0.f;
    }
    const long plane = (long)T * B * H;
    const int ysb = a.ymap.mode == 1 ? T : 1, yst = a.ymap.mode == 1 ? 1 : B;
    const int jl = lane & 3, bl = lane >> 2;
    wait_loads();
    lds_barrier();
    MSTAMP_DECL

    for (int t = 0; t < T; ++t) {
        const int par = t & 1;
        MSTAMP(7);
        // ---- MFMA phase: gh[row][j] = sum_k W_hh[row][k] * h[j][k] -------------------------------
        if (wave < nw) {
            const float *hv = hs + (par * kSeq + jl) * HS;
            // Four independent accumulation chains (with one wave per SIMD nothing else hides the latency between
            // dependent products), and the state vector's 16-byte reads kept four pieces ahead of the products
            // that use them.  The scheduling barriers pin that order.
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            f32x4 ax0 = acc0, ax1 = acc0, ax2 = acc0, ax3 = acc0;
            constexpr int NQ = HP / 4 + (BAL ? 3 : 0), kRing = 5;
            float4 ring[kRing];
            auto piece = [&](int q) {
                return *reinterpret_cast<const float4 *>(q < HP / 4 ? hv + 4 * q : hv + kx + 4 * (q - HP / 4));
            };
#pragma unroll
            for (int q = 0; q < kRing - 1; ++q) ring[q] = piece(q);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + kRing - 1 < NQ) ring[(q + kRing - 1) % kRing] = piece(q + kRing - 1);
                __builtin_amdgcn_sched_barrier(0);
                const float4 v = ring[q % kRing];
                if (q < HP / 4) {
                    acc0 = mfma4(w[4 * q], v.x, acc0);
                    acc1 = mfma4(w[4 * q + 1], v.y, acc1);
                    acc2 = mfma4(w[4 * q + 2], v.z, acc2);
                    acc3 = mfma4(w[4 * q + 3], v.w, acc3);
                } else {
                    const int e = q - HP / 4;
                    ax0 = mfma4(wx[(4 * e) % 12], v.x, ax0);
                    ax1 = mfma4(wx[(4 * e + 1) % 12], v.y, ax1);
                    ax2 = mfma4(wx[(4 * e + 2) % 12], v.z, ax2);
                    ax3 = mfma4(wx[(4 * e + 3) % 12], v.w, ax3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc0 += acc2; acc1 += acc3;
            ax0 += ax2; ax1 += ax3;
            MSTAMP(0);
            mfma_drain();
            // lane 4b+j holds rows 64*wave + 4b .. +3 of sequence j
            *reinterpret_cast<float4 *>(ghs + jl * GS + wave * 64 + 4 * bl) =
                make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
            if constexpr (BAL) {
                // lanes l and l + 32 hold the two k halves of rows 256 + 4*(bl & 7) .. + 3 of sequence j
                const float4 sx = make_float4(add_halves(ax0[0] + ax1[0]), add_halves(ax0[1] + ax1[1]),
                                              add_halves(ax0[2] + ax1[2]), add_halves(ax0[3] + ax1[3]));
                if (lane < 32) *reinterpret_cast<float4 *>(ghx + (wave * kSeq + jl) * 32 + 4 * bl) = sx;
            }
        }
        MSTAMP(1);
        lds_barrier();   // A
        MSTAMP(2);
        // ---- gate phase ----------------------------------------------------------------------------
        const float *gi = gis + (t % kSlots) * NI * 256;
#pragma unroll
        for (int s = 0; s < kPairs; ++s) {
            if (tid + s * nthreads < kSeq * H) {
                const int j = pj[s], u = pu[s];
                const float *g = ghs + j * GS;
                const float *x = gi + j * 3 * H;
                const float r = fast_sigmoid(x[u] + g[u] + bh_r[s]);
                const float z = fast_sigmoid(x[H + u] + g[H + u] + bh_z[s]);
                float ghn_raw;
                if (BAL && u >= 64) {             // the shared rows: the four waves' partial sums, in wave order
                    const float *gx = ghx + j * 32 + (u - 64);
                    ghn_raw = ((gx[0] + gx[kSeq * 32]) + gx[2 * kSeq * 32]) + gx[3 * kSeq * 32];
                } else {
                    ghn_raw = g[2 * H + u];
                }
                const float ghn = ghn_raw + bh_n[s];
                const float n = fast_tanh(x[2 * H + u] + r * ghn);
                const float hp = hs[(par * kSeq + j) * HS + u];
                const bool active = t < plen[s];
                const float h = active ? n + z * (hp - n) : hp;
                hs[((par ^ 1) * kSeq + j) * HS + u] = h;
                if (pok[s]) {
                    const float yv = active ? h : 0.f;
                    const long rowi = ((long)t * B + b0 + j) * H + u;
                    if (a.y_int) a.y_int[rowi] = yv;
                    if (a.y_ext) {
                        const long er = a.ymap.mode == 2 ? map_tb(a.ymap, t, b0 + j) : (long)t * yst + (long)(b0 + j) * ysb;
                        if (er >= 0) a.y_ext[er * H + u] = yv;
                    }
                    if (a.gates && active) {
                        a.gates[rowi] = r;
                        a.gates[plane + rowi] = z;
                        a.gates[2 * plane + rowi] = n;
                        a.gates[3 * plane + rowi] = ghn;
                    }
                    if (a.h_n && t == plen[s] - 1) a.h_n[(long)(b0 + j) * H + u] = yv;
                }
            }
        }
        MSTAMP(3);
        lds_barrier();   // B
        MSTAMP(4);
    }
    MSTAMP_FLUSH(0)
}

// ----------------------------------------------------------------------------------------------
// backward: dg_t (gradients of the gate pre-activations, written for the weight-gradient and dX
// GEMMs) and the state-gradient recurrence  dh_{t-1} = keep_t + W_hh^T . dgh_t
// ----------------------------------------------------------------------------------------------
// BAL (H = 96 only): as in forward, the 3 x 96 (gate, unit) output rows of W_hh^T . dgh are four and a half
// 64-row slices; four waves (one per SIMD) take rows 64w .. 64w + 63 of the flattened (gate, unit) list and
// share the last 32 (gate 2, units 64-95) by splitting the contraction in eight; waves 4 and 5 only take part
// in the elementwise phase.  Without it six waves hold (gate, 64-unit slice) each, half of them half empty.
template <int HP, bool BAL>
__global__ __launch_bounds__(448) void gru_mid_bwd_kernel(BwdArgs a) {
    using namespace mid;
    // direct-to-LDS loads per step: the four gate planes [4][kSeq][H], then h_{t-1} [kSeq][H], then dy [kSeq][H]
    constexpr int NIG = (4 * HP + 63) / 64, NIY = (HP + 63) / 64, NI = NIG + 2 * NIY;
    constexpr int HS = padded(HP);
    // ops[kSlots][NI*256] | dgs[kSeq][3][HS] | part[3][kSeq][RS] | partx[4 waves][kSeq][32]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = BAL ? HP : a.H, B = a.B, T = a.T;
    const int nhs = (H + 63) / 64;                // 64-row slices of the H output rows
    const int nwc = 3 * nhs;                      // compute waves (gate, slice); wave nwc is the loader
    const int RS = nhs * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ops = smem;
    float *dgs = ops + kSlots * NI * 256;
    float *part = dgs + kSeq * 3 * HS;
    float *partx = part + 3 * kSeq * RS;
    const int b0 = blockIdx.x * kSeq;
    const long plane = (long)T * B * H;

    if (wave == nwc) {
        // ---- loader: step k handles time t = T-1-k ----------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.gates, 4 * plane * 4);
        const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.y_int, plane * 4);
        const float *dyp = a.dy_int ? a.dy_int : a.dy_ext;
        const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyp, plane * 4);
        const bool dy_bt = !a.dy_int && a.dymap.mode == 1;              // caller's dy is (B,T,H)
        auto issue = [&](int k) {
            const int t = T - 1 - k;
            float *slot = ops + (k % kSlots) * NI * 256;
#pragma unroll
            for (int i = 0; i < NIG; ++i) {
                const int e = (i * 64 + lane) * 4;          // element of the flattened [4 planes][kSeq][H] block
                const int q = e / (kSeq * H), rem = e - q * kSeq * H;
                const int j = rem / H, off = rem - j * H;
                const bool ok = t >= 0 && q < 4 && b0 + j < B;
                lds_load_b128(rs_g, slot + i * 256, ok ? (int)(q * plane * 4) + ((t * B + b0 + j) * H + off) * 4 : kOOB);
            }
#pragma unroll
            for (int i = 0; i < NIY; ++i) {
                const int e = (i * 64 + lane) * 4;          // element of a [kSeq][H] block
                const int j = e / H, off = e - j * H;
                const bool ok = t >= 0 && j < kSeq && b0 + j < B;
                lds_load_b128(rs_y, slot + (NIG + i) * 256, (ok && t > 0) ? (((t - 1) * B + b0 + j) * H + off) * 4 : kOOB);
                const int drow = dy_bt ? (b0 + j) * T + t : t * B + b0 + j;
                lds_load_b128(rs_dy, slot + (NIG + NIY + i) * 256, ok ? (drow * H + off) * 4 : kOOB);
            }
        };
        for (int k = 0; k < kAhead; ++k) issue(k);
        wait_vmcnt<(kAhead - 1) * NI>();
        lds_barrier();
        for (int k = 0; k < T; ++k) {
            issue(k + kAhead);
            lds_barrier();                        // A: end of the elementwise phase
            wait_vmcnt<(kAhead - 1) * NI>();      // operands of step k+1 have landed
            lds_barrier();                        // B: end of the MFMA phase
        }
        wait_vmcnt<0>();
        return;
    }

    // ---- wave (g, s): column u = 64 s + lane of gate block g of W_hh, i.e. W_hh[g*H + i][u] for all i --
    // (BAL: lane-row 64 wave + lane of the flattened (gate, unit) list instead)
    const int flat = wave * 64 + lane;
    const int g = BAL ? flat / HP : wave / nhs, sl = wave - g * nhs;
    const int ucol = BAL ? flat - g * HP : sl * 64 + lane;
    float w[HP];
    {
        const bool ok = ucol < H && g < 3;
        const float *src = a.w_hh + (long)(ok ? g : 0) * H * H + (ok ? ucol : 0);
#pragma unroll
        for (int i = 0; i < HP; ++i) w[i] = (ok && i < H) ? src[(long)i * H] : 0.f;
    }
    // shared rows: gate 2, unit 64 + (lane & 31), contraction index kx .. kx + 11
    const int kx = 24 * wave + 12 * (lane >> 5);
    float wx[12] = {};
    if (BAL && wave < 4) {
        const float *src = a.w_hh + ((long)2 * H + kx) * H + 64 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 12; ++e) wx[e] = src[(long)e * H];
    }
    // ---- elementwise phase: (sequence, unit) pairs of this thread ---------------------------------------
    const int nthreads = nwc * 64;
    const int tid = threadIdx.x;
    constexpr int kPairs = 2;                     // 4*H pairs over >= 3*64*ceil(H/64) threads
    int pj[kPairs], pu[kPairs], plen[kPairs];
    float keep[kPairs], h0v[kPairs], dbs[kPairs][4];
    bool pin[kPairs], pok[kPairs];
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
        const int p = tid + s * nthreads;
        pin[s] = p < kSeq * H;
        pj[s] = pin[s] ? p / H : 0;
        pu[s] = pin[s] ? p % H : 0;
        pok[s] = pin[s] && b0 + pj[s] < B;
        plen[s] = pok[s] ? (a.lengths ? a.lengths[b0 + pj[s]] : T) : 0;
        const long sb = (long)(b0 + pj[s]) * H + pu[s];
        // the gradient of h_n enters as the initial carry: steps past the length pass it through
        keep[s] = (pok[s] && a.dh_n) ? a.dh_n[sb] : 0.f;
        h0v[s] = (pok[s] && a.h0) ? a.h0[sb] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) dbs[s][q] = 0.f;
    }
    // zero the padding of the dgh vectors (entries H..HP-1) once
    for (int i = tid; i < kSeq * 3 * (HP - H); i += nthreads) {
        const int v = i / (HP - H), c = i % (HP - H);
        dgs[v * HS + H + c] = 0.f;
    }
    const int jl = lane & 3, bl = lane >> 2;
    // gate 2's product for (sequence j, unit u); the shared rows come as four partial sums, added in wave order
    auto part2 = [&](int j, int u) {
        if (BAL && u >= 64) {
            const float *px = partx + j * 32 + (u - 64);
            return ((px[0] + px[kSeq * 32]) + px[2 * kSeq * 32]) + px[3 * kSeq * 32];
        }
        return part[(2 * kSeq + j) * RS + u];
    };
    wait_loads();
    lds_barrier();

    MSTAMP_DECL
    for (int k = 0; k < T; ++k) {
        const int t = T - 1 - k;
        const float *slot = ops + (k % kSlots) * NI * 256;
        MSTAMP(7);
        // ---- elementwise phase ----------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < kPairs; ++s) {
            if (pin[s]) {
                const int j = pj[s], u = pu[s];
                const int o = j * H + u;
                const float r = slot[o], z = slot[kSeq * H + o], n = slot[2 * kSeq * H + o], hn = slot[3 * kSeq * H + o];
                const float hp = t > 0 ? slot[NIG * 256 + o] : h0v[s];
                const float dy = slot[(NIG + NIY) * 256 + o];
                float dhc = keep[s];
                if (k > 0 && t + 1 < plen[s])     // the previous step was active: add W_hh^T . dgh of that step
                    dhc += part[(0 * kSeq + j) * RS + u] + part[(1 * kSeq + j) * RS + u] + part2(j, u);
                const bool on = t < plen[s];
                const float d = on ? dy + dhc : 0.f;
                const float dn = d * (1.f - z);
                const float dzz = d * (hp - n);
                const float an = dn * (1.f - n * n);
                const float dar = an * hn * r * (1.f - r);
                const float daz = dzz * z * (1.f - z);
                const float dgn = an * r;
                keep[s] = on ? d * z : dhc;
                dgs[(j * 3 + 0) * HS + u] = dar;
                dgs[(j * 3 + 1) * HS + u] = daz;
                dgs[(j * 3 + 2) * HS + u] = dgn;
                dbs[s][0] += dar; dbs[s][1] += daz; dbs[s][2] += an; dbs[s][3] += dgn;
                if (pok[s]) {
                    float *dg = a.dg + ((long)t * B + b0 + j) * 4 * H + u;
                    dg[0] = dar; dg[H] = daz; dg[2 * H] = an; dg[3 * H] = dgn;
                }
            }
        }
        MSTAMP(0);
        lds_barrier();   // A
        MSTAMP(1);
        // ---- MFMA phase: part[g][j][u] = sum_i W_hh[g*H + i][u] * dgh_g[j][i] ----------------------------
        if (!BAL) {
            const float *v = dgs + (jl * 3 + g) * HS;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int q = 0; q < HP / 4; ++q) {
                const float4 x = *reinterpret_cast<const float4 *>(v + 4 * q);
                acc0 = mfma4(w[4 * q], x.x, acc0);
                acc1 = mfma4(w[4 * q + 1], x.y, acc1);
                acc0 = mfma4(w[4 * q + 2], x.z, acc0);
                acc1 = mfma4(w[4 * q + 3], x.w, acc1);
            }
            mfma_drain();
            *reinterpret_cast<float4 *>(part + (g * kSeq + jl) * RS + sl * 64 + 4 * bl) =
                make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
        } else if (wave < 4) {
            // four accumulation chains, the vector's 16-byte reads four pieces ahead of their products (one
            // wave per SIMD: nothing else hides the latencies); the scheduling barriers pin that order
            const int fb = wave * 64 + 4 * bl, gb = fb / HP, ub = fb - gb * HP;   // this 4-lane block's rows
            const float *v = dgs + (jl * 3 + gb) * HS;
            const float *vx = dgs + (jl * 3 + 2) * HS + kx;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            f32x4 ax0 = acc0, ax1 = acc0, ax2 = acc0, ax3 = acc0;
            constexpr int NQ = HP / 4 + 3, kRing = 5;
            float4 ring[kRing];
            auto piece = [&](int q) {
                return *reinterpret_cast<const float4 *>(q < HP / 4 ? v + 4 * q : vx + 4 * (q - HP / 4));
            };
#pragma unroll
            for (int q = 0; q < kRing - 1; ++q) ring[q] = piece(q);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + kRing - 1 < NQ) ring[(q + kRing - 1) % kRing] = piece(q + kRing - 1);
                __builtin_amdgcn_sched_barrier(0);
                const float4 x = ring[q % kRing];
                if (q < HP / 4) {
                    acc0 = mfma4(w[4 * q], x.x, acc0);
                    acc1 = mfma4(w[4 * q + 1], x.y, acc1);
                    acc2 = mfma4(w[4 * q + 2], x.z, acc2);
                    acc3 = mfma4(w[4 * q + 3], x.w, acc3);
                } else {
                    const int e = q - HP / 4;
                    ax0 = mfma4(wx[(4 * e) % 12], x.x, ax0);
                    ax1 = mfma4(wx[(4 * e + 1) % 12], x.y, ax1);
                    ax2 = mfma4(wx[(4 * e + 2) % 12], x.z, ax2);
                    ax3 = mfma4(wx[(4 * e + 3) % 12], x.w, ax3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc0 += acc2; acc1 += acc3;
            ax0 += ax2; ax1 += ax3;
            mfma_drain();
            *reinterpret_cast<float4 *>(part + (gb * kSeq + jl) * RS + ub) =
                make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
            // lanes l and l + 32 hold the two contraction halves of units 64 + 4*(bl & 7) .. + 3 of sequence j
            const float4 sx = make_float4(add_halves(ax0[0] + ax1[0]), add_halves(ax0[1] + ax1[1]),
                                          add_halves(ax0[2] + ax1[2]), add_halves(ax0[3] + ax1[3]));
            if (lane < 32) *reinterpret_cast<float4 *>(partx + (wave * kSeq + jl) * 32 + 4 * bl) = sx;
        }
        MSTAMP(2);
        lds_barrier();   // B
        MSTAMP(3);
    }
    MSTAMP_FLUSH(8)
    // ---- dh0 and the per-sequence bias-gradient partial sums --------------------------------------------
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
        if (!pok[s]) continue;
        const int j = pj[s], u = pu[s];
        float dh = keep[s];
        if (0 < plen[s]) dh += part[(0 * kSeq + j) * RS + u] + part[(1 * kSeq + j) * RS + u] + part2(j, u);
        if (a.dh0) a.dh0[(long)(b0 + j) * H + u] = dh;
        float *dbp = a.db_part + (long)(b0 + j) * 4 * H + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dbp[q * H] = dbs[s][q];
    }
}

bool gru_mid_supported(int H) { return H > 32 && H <= 128 && (H & 3) == 0; }

hipError_t launch_gru_mid_bwd(const BwdArgs &a, hipStream_t s) {
    const int nhs = (a.H + 63) / 64, nwc = 3 * nhs;
    dim3 grid((a.B + mid::kSeq - 1) / mid::kSeq), block(64 * (nwc + 1));
    const int HP = a.H <= 64 ? 64 : (a.H <= 96 ? 96 : 128);
    const int NI = (4 * HP + 63) / 64 + 2 * ((HP + 63) / 64);
    const size_t lds = (size_t)(mid::kSlots * NI * 256 + mid::kSeq * 3 * mid::padded(HP) + 3 * mid::kSeq * nhs * 64 +
                                4 * mid::kSeq * 32) * sizeof(float);
    if (HP == 64) TGRU_LAUNCH((gru_mid_bwd_kernel<64, false>), grid, block, lds, s, a);
    else if (a.H == 96) TGRU_LAUNCH((gru_mid_bwd_kernel<96, true>), grid, block, lds, s, a);
    else if (HP == 96) TGRU_LAUNCH((gru_mid_bwd_kernel<96, false>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_mid_bwd_kernel<128, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gru_mid_fwd(const FwdArgs &a, hipStream_t s) {
    const bool bal = a.H == 96;
    const int nw = bal ? 4 : (3 * a.H + 63) / 64;
    dim3 grid((a.B + mid::kSeq - 1) / mid::kSeq), block(64 * ((bal ? 6 : nw) + 1));
    const int HP = a.H <= 64 ? 64 : (a.H <= 96 ? 96 : 128);
    const int NI = (3 * HP + 63) / 64;
    const size_t lds = (size_t)(2 * mid::kSeq * mid::padded(HP) + mid::kSeq * nw * 64 + 4 * mid::kSeq * 32 +
                                mid::kSlots * NI * 256) * sizeof(float);
    if (HP == 64) TGRU_LAUNCH((gru_mid_fwd_kernel<64, false>), grid, block, lds, s, a);
    else if (bal) TGRU_LAUNCH((gru_mid_fwd_kernel<96, true>), grid, block, lds, s, a);
    else if (HP == 96) TGRU_LAUNCH((gru_mid_fwd_kernel<96, false>), grid, block, lds, s, a);
    else TGRU_LAUNCH((gru_mid_fwd_kernel<128, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace tgru
