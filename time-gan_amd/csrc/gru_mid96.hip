// Forward recurrent kernel for H = In = 96 with the input projection inside it.
//
// The per-layer kernels of gru_mid.hip read gi = X . W_ih^T + b_ih from memory, written by a GEMM launched
// before them: 3H floats per (step, sequence) out to HBM and back in.  Here the same workgroup (four
// sequences, time loop inside) computes gi itself: a second set of four matrix waves holds W_ih in registers,
// exactly as the recurrent waves hold W_hh, and multiplies the step's input rows beside the recurrent product
// (two matrix waves per SIMD keep the matrix pipe full where one left it waiting on its own dependent
// products).  What comes from memory per step is the input row (H floats instead of 3H), fetched a step
// ahead.
//
// Waves: 0-3 recurrent product (one per SIMD, 64 rows of W_hh each + a share of the last 32, see below), 4-7
// input projection.  Eight waves are two per SIMD: 256 VGPRs each, of which a wave's share of its matrix is
// 108.  The gate phase's 4H = 384 (sequence, unit) pairs are one per thread of waves 0-5; in that phase
// waves 6 and 7 move the input rows instead: they put x_{t+1} (in registers since the step before) into LDS
// and start the loads of x_{t+2}.  Per step: [0-3: gh_t = W_hh . h_{t-1} | 4-7: gi_t = W_ih . x_t] ->
// barrier A -> [0-5: gates, h_t, stores | 6-7: input rows] -> barrier B.
//
// A product is v_mfma_f32_4x4x1_16b_f32 with one matrix row per lane in VGPRs (gru_mid.hip).  3H = 288 rows
// are four and a half 64-row slices: the four waves of a product take a full slice each and share the last 32
// rows, wave w covering k in [24w, 24w + 24) of them, lanes 0-31 the first twelve k of that range and lanes
// 32-63 the second twelve; the halves are added across the wave, the four waves' sums by the gate phase.
#include "tgru_internal.h"

namespace tgru {
namespace m96 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 96;
constexpr int kSeq = 4;            // sequences per workgroup
constexpr int HS = H + 4;          // LDS stride of a state row
constexpr int kWaves = 8;
constexpr int kMain = kSeq * 256, kPart = 4 * kSeq * 32;      // a product's result: rows 0-255, then the shared rows' partials

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// x[lane] + x[lane ^ 32]
__device__ __forceinline__ float add_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace m96

#ifdef TGRU_STAMPS   // diagnostic build only: shader cycles wave 0 (slots 0-7) and wave 5 (8-15) of workgroup 0 spend per phase
__device__ unsigned long long tgru_stamps_mid96[16];
#define FSTAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FSTAMP(k)                                                     \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += now_ - st_last; st_last = now_;                  \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define FSTAMP_FLUSH                                                  \
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 320))  \
        for (int k_ = 0; k_ < 8; ++k_) tgru_stamps_mid96[(threadIdx.x ? 8 : 0) + k_] = st_acc[k_];
extern "C" int tgru_debug_stamps_mid96(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_mid96), sizeof(unsigned long long) * 16);
}
#else
#define FSTAMP_DECL
#define FSTAMP(k)
#define FSTAMP_FLUSH
#endif

__global__ __launch_bounds__(64 * m96::kWaves) void gru_mid96_fwd_kernel(FwdArgs a) {
    using namespace m96;
#ifdef TGRU_STAMPS
    const unsigned long long st_kernel0 = __builtin_amdgcn_s_memtime();
#endif
    // hs[2][kSeq][HS] | gh[kMain + kPart] | gi[kMain + kPart] | xs[kSeq][H]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int B = a.B, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *hs = smem;
    float *gh = hs + 2 * kSeq * HS;
    float *gi = gh + kMain + kPart;
    float *xs = gi + kMain + kPart;
    const int b0 = blockIdx.x * kSeq;

    // ---- waves 6 and 7 move the input rows: 16-byte chunk c = 0 .. 95 of the step's [kSeq][96] block per lane
    const int xc = (wave - 6) * 64 + lane;
    const bool x_mover = wave >= 6 && xc < kSeq * (H / 4);
    const int xseq = x_mover ? xc / (H / 4) : 0, xq = x_mover ? xc - xseq * (H / 4) : 0;
    const bool x_ok = x_mover && b0 + xseq < B;
    const long x_row0 = a.xmap.mode == 1 ? (long)(b0 + xseq) * T : b0 + xseq;      // row of step 0
    const long x_step = a.xmap.mode == 1 ? 1 : B;                                   // rows from a step to the next
    // Rows are 96 floats = 3 x 32 banks: the four sequences' rows would start in the same bank, and the products
    // read the same 16-byte chunk of all four rows at once.  Row j is therefore stored rotated by j chunks.
    float *const x_dst = xs + xseq * H + 4 * ((xq + xseq) % (H / 4));
    float4 xreg = make_float4(0.f, 0.f, 0.f, 0.f);
    auto x_load = [&](int t) {
        if (x_ok && t < T) xreg = *reinterpret_cast<const float4 *>(a.x + (x_row0 + t * x_step) * H + 4 * xq);
    };
    auto x_put = [&]() {
        if (x_mover) *reinterpret_cast<float4 *>(x_dst) = xreg;
    };
    x_load(0);
    x_put();
    x_load(1);

    // ---- matrix waves: one row of W_hh (waves 0-3) or W_ih (waves 4-7) per lane, plus the shared rows ----
    const bool is_rec = wave < 4, is_proj = !is_rec;
    const int mw = wave & 3;                               // 64-row slice of the matrix
    const int kx = 24 * mw + 12 * (lane >> 5);             // shared rows: row 256 + (lane & 31), k = kx .. kx + 11
    float w[H], wx[12];
    {
        const float *mat = is_proj ? a.w_ih : a.w_hh;
        const float *src = mat + (long)(mw * 64 + lane) * H;
        const float *srx = mat + (long)(256 + (lane & 31)) * H + kx;
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 u = *reinterpret_cast<const float4 *>(src + 4 * q);
            w[4 * q] = u.x; w[4 * q + 1] = u.y; w[4 * q + 2] = u.z; w[4 * q + 3] = u.w;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 u = *reinterpret_cast<const float4 *>(srx + 4 * q);
            wx[4 * q] = u.x; wx[4 * q + 1] = u.y; wx[4 * q + 2] = u.z; wx[4 * q + 3] = u.w;
        }
    }
    const int jl = lane & 3, bl = lane >> 2;
    FSTAMP_DECL
#ifdef TGRU_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (slot 7: the prologue, matrix rows in registers)
    st_acc[7] = __builtin_amdgcn_s_memtime() - st_kernel0;
    st_last = __builtin_amdgcn_s_memtime();
#endif
    // out[kMain | kPart] = M . v for the four vectors v (`vec`: this lane's sequence jl), M the wave set's matrix.
    // Lane 4b+i gives A_b[i] = its row's element k, lane 4b+j gives B_b[j] = v_j[k]; register i of lane 4b+j
    // receives rows 4b .. 4b+3 of sequence j.  The vector's 16-byte reads are kept a few pieces ahead of the
    // products that use them (the scheduling barriers pin that order; hoisting all of them does not fit the
    // register file).  rot: chunk q of the vector sits at chunk position (q + rot) mod 24 (0: not rotated)
    auto product = [&](const float *vec, int rot, float *out) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0, ax0 = acc0, ax1 = acc0;
        constexpr int NQ = H / 4 + 3, kRing = 6;
        float4 ring[kRing];
        const int cx = kx / 4 + rot;                       // first chunk of the shared rows' k range, rotated
        auto piece = [&](int q) {
            int c = q < H / 4 ? q + rot : cx + (q - H / 4);
            if (c >= H / 4) c -= H / 4;
            return *reinterpret_cast<const float4 *>(vec + 4 * c);
        };
#pragma unroll
        for (int q = 0; q < kRing - 1; ++q) ring[q] = piece(q);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + kRing - 1 < NQ) ring[(q + kRing - 1) % kRing] = piece(q + kRing - 1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 v = ring[q % kRing];
            if (q < H / 4) {
                acc0 = mfma4(w[4 * q], v.x, acc0);
                acc1 = mfma4(w[4 * q + 1], v.y, acc1);
                acc2 = mfma4(w[4 * q + 2], v.z, acc2);
                acc3 = mfma4(w[4 * q + 3], v.w, acc3);
            } else {
                const int e = q - H / 4;
                ax0 = mfma4(wx[4 * e], v.x, ax0);
                ax1 = mfma4(wx[4 * e + 1], v.y, ax1);
                ax0 = mfma4(wx[4 * e + 2], v.z, ax0);
                ax1 = mfma4(wx[4 * e + 3], v.w, ax1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        FSTAMP(5);
        acc0 += acc2; acc1 += acc3;
        // lane 4b+j holds rows 64*mw + 4b .. +3 of sequence j
        *reinterpret_cast<float4 *>(out + jl * 256 + mw * 64 + 4 * bl) =
            make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
        // lanes l and l + 32 hold the two k halves of rows 256 + 4*(bl & 7) .. + 3 of sequence j
        const float4 sx = make_float4(add_halves(ax0[0] + ax1[0]), add_halves(ax0[1] + ax1[1]),
                                      add_halves(ax0[2] + ax1[2]), add_halves(ax0[3] + ax1[3]));
        if (lane < 32) *reinterpret_cast<float4 *>(out + kMain + (mw * kSeq + jl) * 32 + 4 * bl) = sx;
        FSTAMP(6);
    };
    // the n-gate element of unit u of a product's result for sequence j (rows 2H + u; from 256 on they come
    // as the four waves' partial sums, added in wave order)
    auto result_n = [&](const float *out, int j, int u) {
        if (u >= 64) {
            const float *px = out + kMain + j * 32 + (u - 64);
            return ((px[0] + px[kSeq * 32]) + px[2 * kSeq * 32]) + px[3 * kSeq * 32];
        }
        return out[j * 256 + 2 * H + u];
    };

    // ---- gate phase: (sequence, unit) pair p of this thread (waves 0-5) --------------------------------------
    const int p = threadIdx.x;
    const bool pin = p < kSeq * H;
    const int pj = pin ? p / H : 0, pu = pin ? p % H : 0;
    const bool pok = pin && b0 + pj < B;                   // a real sequence (the last workgroup may have fewer than four)
    const int plen = pok ? (a.lengths ? a.lengths[b0 + pj] : T) : 0;
    const float b_r = a.b_hh[pu] + a.b_ih[pu], b_z = a.b_hh[H + pu] + a.b_ih[H + pu];
    const float bh_n = a.b_hh[2 * H + pu], bi_n = a.b_ih[2 * H + pu];
    if (pin) {
        const float h0v = (pok && a.h0) ? a.h0[(long)(b0 + pj) * H + pu] : 0.f;
        hs[(0 * kSeq + pj) * HS + pu] = h0v;
        hs[(1 * kSeq + pj) * HS + pu] = h0v;
    }
    const long plane = (long)T * B * H;
    const int ysb = a.ymap.mode == 1 ? T : 1, yst = a.ymap.mode == 1 ? 1 : B;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the weight loads
    lds_barrier();                                         // x_0 has landed, the states are set

    for (int t = 0; t < T; ++t) {
        const int par = t & 1;
        FSTAMP(0);
        if (is_rec) product(hs + (par * kSeq + jl) * HS, 0, gh);
        else product(xs + jl * H, jl, gi);
        FSTAMP(1);
        lds_barrier();   // A
        FSTAMP(2);
        if (pin) {
            const int j = pj, u = pu;
            const float r = fast_sigmoid((gi[j * 256 + u] + gh[j * 256 + u]) + b_r);
            const float z = fast_sigmoid((gi[j * 256 + H + u] + gh[j * 256 + H + u]) + b_z);
            const float ghn = result_n(gh, j, u) + bh_n;
            const float n = fast_tanh((result_n(gi, j, u) + bi_n) + r * ghn);
            const float hp = hs[(par * kSeq + j) * HS + u];
            const bool active = t < plen;
            const float h = active ? n + z * (hp - n) : hp;
            hs[((par ^ 1) * kSeq + j) * HS + u] = h;
            if (pok) {
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
                if (a.h_n && t == plen - 1) a.h_n[(long)(b0 + j) * H + u] = yv;
            }
        }
        if (wave >= 6) {
            x_put();                                       // x_{t+1}, loaded during the step before
            x_load(t + 2);
        }
        FSTAMP(3);
        lds_barrier();   // B
        FSTAMP(4);
    }
    FSTAMP_FLUSH
}

// ----------------------------------------------------------------------------------------------
// Backward recurrent kernel for H = In = 96 with dX = dgi . W_ih inside it.
//
// gru_mid.hip's backward kernel writes dg = (da_r | da_z | da_n | da_n*r) for the weight-gradient kernel and
// for a GEMM that forms the gradient of the layer's input.  Here a second set of four matrix waves holds the
// columns of W_ih as the first holds those of W_hh and multiplies the step's (da_r, da_z, da_n) beside the
// recurrent product W_hh^T . (da_r, da_z, da_n*r): the same shape, the same code.  Its result goes through
// LDS like the recurrent one: the elementwise phase of the NEXT step adds the per-gate parts and stores the
// row of dX (the last row after the loop).
//
// Waves: 0-3 recurrent product + elementwise phase, 4-7 input-gradient product (4 and 5 also elementwise: 384
// (sequence, unit) pairs, one per thread of waves 0-5), 8 loader (saved gates, h_{t-1}, dy; direct-to-LDS,
// four steps ahead; it alone loads, so its load counter is exact).  Nine waves: three on one SIMD, 168 VGPRs.
// Per step: [0-5: gate gradients of step t, dX row of step t+1] -> barrier A -> [0-3: W_hh^T . dgh | 4-7:
// W_ih^T . dgi] -> barrier B.
//
// Products: the 3 x 96 (gate, unit) output rows are four and a half 64-row slices; wave w takes rows 64w ..
// 64w + 63 of the flattened (gate, unit) list (one column of the matrix per lane: 96 VGPRs) and a share of
// the last 32 (gate 2, units 64-95): contraction indices [24w, 24w + 24), lanes 0-31 the first twelve, lanes
// 32-63 the second twelve.
// ----------------------------------------------------------------------------------------------
namespace m96 {
constexpr int kBwdWaves = 9;
constexpr int kAhead = 4, kSlots = kAhead + 1;
constexpr int NIG = 4 * H / 64, NIY = (H + 63) / 64, NIB = NIG + 2 * NIY;    // loads per step: gates, h_{t-1}, dy
constexpr int RS = 128;                                                        // stride of a (gate, sequence) row of parts
constexpr int kParts = 3 * kSeq * RS + 4 * kSeq * 32;                          // [3][kSeq][RS] | shared rows [4 waves][kSeq][32]
constexpr int kOOB = 0x7ffffff0;
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
}  // namespace m96

__global__ __launch_bounds__(64 * m96::kBwdWaves) void gru_mid96_bwd_kernel(BwdArgs a) {
    using namespace m96;
    // ops[kSlots][NIB*256] | dgs[kSeq][4][HS] (da_r, da_z, da_n*r, da_n) | parts of W_hh^T.dgh | parts of W_ih^T.dgi
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int B = a.B, T = a.T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ops = smem;
    float *dgs = ops + kSlots * NIB * 256;
    float *part = dgs + kSeq * 4 * HS;
    float *dpart = part + kParts;
    const int b0 = blockIdx.x * kSeq;
    const long plane = (long)T * B * H;

    if (wave == kBwdWaves - 1) {
        // ---- loader: step k handles time t = T-1-k ----------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.gates, 4 * plane * 4);
        const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.y_int, plane * 4);
        const float *dyp = a.dy_int ? a.dy_int : a.dy_ext;
        const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyp, plane * 4);
        const bool dy_bt = !a.dy_int && a.dymap.mode == 1;              // caller's dy is (B,T,H)
        auto issue = [&](int k) {
            const int t = T - 1 - k;
            float *slot = ops + (k % kSlots) * NIB * 256;
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
        wait_vmcnt<(kAhead - 1) * NIB>();
        lds_barrier();
        for (int k = 0; k < T; ++k) {
            issue(k + kAhead);
            lds_barrier();                        // A: end of the elementwise phase
            wait_vmcnt<(kAhead - 1) * NIB>();     // operands of step k+1 have landed
            lds_barrier();                        // B: end of the matrix phase
        }
        wait_vmcnt<0>();
        lds_barrier();                            // the last dX row is stored after this one
        return;
    }

    // ---- matrix waves: column `ucol` of gate block g of W_hh (waves 0-3) or W_ih (waves 4-7) per lane --------
    const bool is_dx = wave >= 4;
    const int mw = wave & 3;
    const int flat = mw * 64 + lane;                      // row of the flattened (gate, unit) list
    const int g = flat / H, ucol = flat - g * H;
    const int kx = 24 * mw + 12 * (lane >> 5);            // shared rows: gate 2, unit 64 + (lane & 31), index kx .. kx + 11
    float w[H], wx[12];
    {
        const float *mat = is_dx ? a.w_ih : a.w_hh;       // both (3H, 96) row-major
        const float *src = mat + (long)g * H * H + ucol;
#pragma unroll
        for (int i = 0; i < H; ++i) w[i] = src[(long)i * H];
        const float *srx = mat + ((long)2 * H + kx) * H + 64 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 12; ++e) wx[e] = srx[(long)e * H];
    }
    const int jl = lane & 3, bl = lane >> 2;
    // out = M^T . (the three gate-gradient vectors of each sequence); vn: which vector stands for gate 2
    // (2: da_n*r for the recurrent product, 3: da_n for the input gradient)
    auto product = [&](int vn, float *out) {
        const int fb = mw * 64 + 4 * bl, gb = fb / H, ub = fb - gb * H;      // this 4-lane block's rows
        const float *v = dgs + (jl * 4 + (gb == 2 ? vn : gb)) * HS;
        const float *vx = dgs + (jl * 4 + vn) * HS + kx;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, ax0 = acc0, ax1 = acc0;
        constexpr int NQ = H / 4 + 3, kRing = 4;
        float4 ring[kRing];
        auto piece = [&](int q) {
            return *reinterpret_cast<const float4 *>(q < H / 4 ? v + 4 * q : vx + 4 * (q - H / 4));
        };
#pragma unroll
        for (int q = 0; q < kRing - 1; ++q) ring[q] = piece(q);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + kRing - 1 < NQ) ring[(q + kRing - 1) % kRing] = piece(q + kRing - 1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 x = ring[q % kRing];
            if (q < H / 4) {
                acc0 = mfma4(w[4 * q], x.x, acc0);
                acc1 = mfma4(w[4 * q + 1], x.y, acc1);
                acc0 = mfma4(w[4 * q + 2], x.z, acc0);
                acc1 = mfma4(w[4 * q + 3], x.w, acc1);
            } else {
                const int e = q - H / 4;
                ax0 = mfma4(wx[4 * e], x.x, ax0);
                ax1 = mfma4(wx[4 * e + 1], x.y, ax1);
                ax0 = mfma4(wx[4 * e + 2], x.z, ax0);
                ax1 = mfma4(wx[4 * e + 3], x.w, ax1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        *reinterpret_cast<float4 *>(out + (gb * kSeq + jl) * RS + ub) =
            make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
        // lanes l and l + 32 hold the two contraction halves of units 64 + 4*(bl & 7) .. + 3 of sequence j
        const float4 sx = make_float4(add_halves(ax0[0] + ax1[0]), add_halves(ax0[1] + ax1[1]),
                                      add_halves(ax0[2] + ax1[2]), add_halves(ax0[3] + ax1[3]));
        if (lane < 32) *reinterpret_cast<float4 *>(out + 3 * kSeq * RS + (mw * kSeq + jl) * 32 + 4 * bl) = sx;
    };
    // the sum over the three gates of a product's result for (sequence j, unit u); gate 2's shared rows come as
    // four partial sums, added in wave order
    auto gates_sum = [&](const float *out, int j, int u) {
        float g2;
        if (u >= 64) {
            const float *px = out + 3 * kSeq * RS + j * 32 + (u - 64);
            g2 = ((px[0] + px[kSeq * 32]) + px[2 * kSeq * 32]) + px[3 * kSeq * 32];
        } else {
            g2 = out[(2 * kSeq + j) * RS + u];
        }
        return out[(0 * kSeq + j) * RS + u] + out[(1 * kSeq + j) * RS + u] + g2;
    };

    // ---- elementwise phase: (sequence, unit) pair p = tid of waves 0-5 -----------------------------------------
    const int tid = threadIdx.x;
    const bool pin = tid < kSeq * H;
    const int pj = pin ? tid / H : 0, pu = pin ? tid % H : 0;
    const bool pok = pin && b0 + pj < B;
    const int plen = pok ? (a.lengths ? a.lengths[b0 + pj] : T) : 0;
    const long sb = (long)(b0 + pj) * H + pu;
    // the gradient of h_n enters as the initial carry: steps past the length pass it through
    float keep = (pok && a.dh_n) ? a.dh_n[sb] : 0.f;
    const float h0v = (pok && a.h0) ? a.h0[sb] : 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};
    // row of dX for (t, this pair's sequence): caller layouts (T,B) and (B,T), or the time-major buffer below
    const long dx_row0 = a.dxmap.mode == 1 ? (long)(b0 + pj) * T : b0 + pj;
    const long dx_step = a.dxmap.mode == 1 ? 1 : B;
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the weight loads
    lds_barrier();

    FSTAMP_DECL
    for (int k = 0; k < T; ++k) {
        const int t = T - 1 - k;
        const float *slot = ops + (k % kSlots) * NIB * 256;
        FSTAMP(0);
        if (pin) {
            const int j = pj, u = pu;
            const int o = j * H + u;
            const float r = slot[o], z = slot[kSeq * H + o], n = slot[2 * kSeq * H + o], hn = slot[3 * kSeq * H + o];
            const float hp = t > 0 ? slot[NIG * 256 + o] : h0v;
            const float dy = slot[(NIG + NIY) * 256 + o];
            float dhc = keep;
            if (k > 0) {
                if (t + 1 < plen) dhc += gates_sum(part, j, u);     // the previous step was active: W_hh^T . dgh of that step
                if (pok) a.dx[(dx_row0 + (t + 1) * dx_step) * H + u] = gates_sum(dpart, j, u);
            }
            const bool on = t < plen;
            const float d = on ? dy + dhc : 0.f;
            const float dn = d * (1.f - z);
            const float dzz = d * (hp - n);
            const float an = dn * (1.f - n * n);
            const float dar = an * hn * r * (1.f - r);
            const float daz = dzz * z * (1.f - z);
            const float dgn = an * r;
            keep = on ? d * z : dhc;
            dgs[(j * 4 + 0) * HS + u] = dar;
            dgs[(j * 4 + 1) * HS + u] = daz;
            dgs[(j * 4 + 2) * HS + u] = dgn;
            dgs[(j * 4 + 3) * HS + u] = an;
            dbs[0] += dar; dbs[1] += daz; dbs[2] += an; dbs[3] += dgn;
            if (pok && a.dg) {       // (no weight gradients wanted: nothing reads the gate gradients after this kernel)
                float *dg = a.dg + ((long)t * B + b0 + j) * 4 * H + u;
                dg[0] = dar; dg[H] = daz; dg[2 * H] = an; dg[3 * H] = dgn;
            }
        }
        FSTAMP(1);
        lds_barrier();   // A
        FSTAMP(2);
        if (wave < 8) product(is_dx ? 3 : 2, is_dx ? dpart : part);
        FSTAMP(3);
        lds_barrier();   // B
        FSTAMP(4);
    }
    FSTAMP_FLUSH
    lds_barrier();       // (matches the loader's last one)
    // ---- dh0, the dX row of step 0, and the per-sequence bias-gradient partial sums --------------------------
    if (pok) {
        const int j = pj, u = pu;
        float dh = keep;
        if (0 < plen) dh += gates_sum(part, j, u);
        if (a.dh0) a.dh0[sb] = dh;
        a.dx[dx_row0 * H + u] = gates_sum(dpart, j, u);
        float *dbp = a.db_part + (long)(b0 + j) * 4 * H + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dbp[q * H] = dbs[q];
    }
}

bool gru_mid96_bwd_supported(const BwdArgs &a) {
    if (switches().no_fuse || a.H != m96::H || a.In != m96::H || !a.dx || !a.w_ih) return false;
    if (a.dxmap.mode == 2) return false;                  // packed rows: the separate product handles them
    // the loader wave addresses the four gate planes through one buffer descriptor with 32-bit byte offsets
    return (long)a.T * a.B * m96::H * 16 < (1L << 31);
}

hipError_t launch_gru_mid96_bwd(const BwdArgs &a, hipStream_t s) {
    using namespace m96;
    dim3 grid((a.B + kSeq - 1) / kSeq), block(64 * kBwdWaves);
    const size_t lds = (size_t)(kSlots * NIB * 256 + kSeq * 4 * HS + 2 * kParts) * sizeof(float);
    TGRU_LAUNCH(gru_mid96_bwd_kernel, grid, block, lds, s, a);
    return hipGetLastError();
}

// The fused kernel covers H = In = 96 with input rows in (T,B) or (B,T) order (not packed rows), 31-bit byte
// offsets into x.
bool gru_mid96_fwd_supported(const FwdArgs &a) {
    if (switches().no_fuse || a.H != m96::H || a.In != m96::H || !a.x || !a.w_ih || !a.b_ih) return false;
    if (a.xmap.mode == 2 || (reinterpret_cast<uintptr_t>(a.x) & 15)) return false;
    return (long)a.T * a.B * m96::H * 4 < (1L << 31);
}

hipError_t launch_gru_mid96_fwd(const FwdArgs &a, hipStream_t s) {
    using namespace m96;
    dim3 grid((a.B + kSeq - 1) / kSeq), block(64 * kWaves);
    const size_t lds = (size_t)(2 * kSeq * HS + 2 * (kMain + kPart) + kSeq * H) * sizeof(float);
    TGRU_LAUNCH(gru_mid96_fwd_kernel, grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace tgru
