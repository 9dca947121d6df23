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
