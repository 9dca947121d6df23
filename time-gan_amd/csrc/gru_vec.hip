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
// Layout of one workgroup: one COMPUTE wave per layer (layers run as a wavefront: wave l works on
// time step tick - l) plus one LOADER wave.  In a compute wave, lane = (p, j): j = lane & 31 is a
// hidden unit (forward) or an input / hidden column (backward), p = lane >> 5 selects the half of
// the work:
//   forward   p = 0: rows j of W_ir, W_iz, W_in times the layer input x_t
//             p = 1: rows j of W_hr, W_hz, W_hn times h_{t-1}
//   backward  p = 0: column j of W_ih (dx_t[j]);  p = 1: column j of W_hh (dh_{t-1}[j])
// Weights stay in VGPRs for the whole kernel; the vectors they multiply are broadcast through LDS
// (all lanes of a half read the same 16 bytes); one v_permlane32_swap per value joins the two
// halves.
//
// The loader wave is the only one that reads global memory inside the time loop: it fetches the
// operands of a step several steps ahead into a register ring and hands them over through LDS.
// Compute waves only store.  A wave that has both loads and stores in flight cannot wait for "the
// loads issued N steps ago" (the two complete out of order, so the wait becomes vmcnt(0) and every
// step would pay a store round trip); with the roles split, neither kind of wave ever waits for a
// store.
//
// Backward also accumulates the weight gradients: dW += dg (x) input is a rank-1 update per step,
// issued as three v_mfma_f32_32x32x1 (two independent 32x32 blocks: block 0 = dW_ih from the lower
// half's operands, block 1 = dW_hh from the upper half's), accumulating in AGPRs off the critical
// path; each workgroup leaves one partial gradient, reduced by a second kernel in a fixed order.
#include "tgru_internal.h"

namespace tgru {

#ifdef TGRU_STAMPS   // diagnostic build only: per-wave cycle counts of the forward tick's phases
__device__ unsigned long long tgru_stamps_vec[16 * 8];
#define VSTAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0};
#define VSTAMP(k)                                                  \
    do {                                                           \
        __builtin_amdgcn_sched_barrier(0);                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += now_ - st_last; st_last = now_;               \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)
#define VSTAMP_FLUSH                                                         \
    if (blockIdx.x == 0 && lane == 0)                                        \
        for (int k_ = 0; k_ < 6; ++k_) tgru_stamps_vec[(l + VSTAMP_BASE) * 8 + k_] = st_acc[k_];
extern "C" int tgru_debug_stamps_vec(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_vec), sizeof(unsigned long long) * 16 * 8);
}
#else
#define VSTAMP_DECL
#define VSTAMP(k)
#define VSTAMP_FLUSH
#endif

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

// Waits for every outstanding global load of this wave.  Placed in front of a time loop so that the
// compiler does not put its own wait for the loop-invariant operands (weights) INSIDE the loop,
// where, from the second step on, it would wait for that step's stores instead.
__device__ __forceinline__ void wait_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Branch-free predication: every global access in the time loops is a raw buffer access; a lane
// (or a step) that must not touch memory uses an offset past the end of the buffer, for which loads
// return 0 and stores are dropped.  An absent tensor gets a descriptor with zero records.
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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_st2(__amdgpu_buffer_rsrc_t r, int off, float a, float b) {
    const u32x2 u = {__float_as_uint(a), __float_as_uint(b)};
    __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// Direct-to-LDS loads: lane i's element lands at lds + i * size, no VGPR is involved and the compiler
// tracks nothing: the issuing wave waits with an explicit vmcnt before the barrier that publishes
// the data.  Out-of-range lanes write 0.
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_load_b32(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 4, off, 0, 0, 0);
}
__device__ __forceinline__ void lds_load_b128(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, off, 0, 0, 0);
}
// s_waitcnt vmcnt(n) alone (expcnt and lgkmcnt fields left at "no wait")
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// lane offset + step offset; either may be kOOB (both < 2^31, so the unsigned sum stays out of range)
__device__ __forceinline__ int off_add(int lane_off, int step_off) { return (int)((unsigned)lane_off + (unsigned)step_off); }

// row of (t, b) in a caller tensor of one of the two padded layouts: t*st + b*sb (packed rows are
// not handled by these kernels: the library routes them to the MFMA whole-stack kernels)
struct ExtMap { int st, sb; };
__device__ __forceinline__ ExtMap ext_map_of(const RowMap &m) {
    ExtMap e;
    e.st = m.mode == 1 ? 1 : m.B;
    e.sb = m.mode == 1 ? m.T : 1;
    return e;
}
__device__ __forceinline__ int ext_row(const ExtMap &m, int t, int b) { return t * m.st + b * m.sb; }

constexpr int kVecMaxL = 4;   // layers (compute waves) per workgroup
constexpr int kFwdAhead = 8;  // time steps the layer-0 input is loaded ahead
constexpr int kFwdSlots = kFwdAhead + 1;
constexpr int kBwdAhead = 4;  // time steps the backward operands are loaded ahead
constexpr int kBwdSlots = kBwdAhead + 1;
// one backward operand slot (floats): gates[kVecMaxL][32][4] | y[kVecMaxL][32] | x0[32] | dy[32]
constexpr int kSlotG = 0, kSlotY = kVecMaxL * 128, kSlotX = kSlotY + kVecMaxL * 32, kSlotD = kSlotX + 32;
constexpr int kSlotFloats = kSlotD + 32;
constexpr int kBwdLoads = 2 * kVecMaxL + 2;   // direct-to-LDS loads per step

}  // namespace vec

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
#define VSTAMP_BASE 0
template <int KP, bool PAIR>
__global__ __launch_bounds__(320) void gru_vec_fwd_kernel(VecPair<SmallFwdArgs> ga) {
    using namespace vec;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hb[2][L][KP] | xb[kFwdSlots][32]
    // workgroups [0, nwg0) serve the first call of the pair, the others the second (a single call is
    // compiled apart, so that its argument reads stay at fixed kernel-argument offsets)
    const int set = PAIR && (int)blockIdx.x >= ga.nwg0;
    const SmallFwdArgs &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x, nwg = set ? ga.nwg1 : ga.nwg0;
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int l = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // L = the loader wave
    const int p = lane >> 5, j = lane & 31;
    float *xb = smem + 2 * L * KP;
    const int nticks = T + L - 1;

    if (l == L) {
        // ---- loader: x[t] of layer 0 straight into LDS slot t % kFwdSlots, kFwdAhead steps ahead ----
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, (long)T * B * a.In * 4);
        const ExtMap xm = ext_map_of(a.xmap);
        const int xstep = xm.st * a.In * 4;
        for (int b = wg; b < B; b += nwg) {
            const int len = a.lengths ? a.lengths[b] : T;
            const int xbase = lane < a.In ? (b * xm.sb * a.In + lane) * 4 : kOOB;
            auto issue = [&](int t) {   // lanes 0..31; lanes >= In and steps >= len write zeros
                if (lane < 32) lds_load_b32(rs_x, xb + (t % kFwdSlots) * 32, off_add(xbase, t < len ? t * xstep : kOOB));
            };
            for (int t = 0; t < kFwdAhead; ++t) issue(t);
            wait_vmcnt<kFwdAhead - 1>();
            lds_barrier();
            for (int k = 0; k < nticks; ++k) {
                issue(k + kFwdAhead);
                wait_vmcnt<kFwdAhead - 1>();     // x[k+1] has landed
                lds_barrier();
            }
            wait_vmcnt<0>();                     // the slots are reused by the next sequence
        }
        return;
    }

    const int In_l = (l == 0) ? a.In : H;
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
    const int tbh = (int)(plane * 4);
    const __amdgpu_buffer_rsrc_t rs_yint = make_rsrc(a.y_int[l], tbh);
    const __amdgpu_buffer_rsrc_t rs_yext = make_rsrc(top ? a.y_ext : nullptr, tbh);
    const __amdgpu_buffer_rsrc_t rs_gates = make_rsrc(a.gates[l], 4L * tbh);
    const __amdgpu_buffer_rsrc_t rs_hn = make_rsrc(a.h_n, (long)L * B * H * 4);
    const ExtMap ym = ext_map_of(a.ymap);
    // LDS addresses: this layer's state (written), and the vector this half multiplies
    float *hw = smem + l * KP;                                         // + parity * L * KP
    const bool from_x = p == 0 && l == 0;
    const float *vsrc = p ? smem + l * KP : smem + (l - 1) * KP;       // + parity * L * KP

    for (int b = wg; b < B; b += nwg) {
        const int len = a.lengths ? a.lengths[b] : T;
        float hreg = (unit && a.h0) ? a.h0[((long)l * B + b) * H + j] : 0.f;
        if (p == 0 && j < KP) {
            hw[j] = hreg;
            hw[L * KP + j] = hreg;
        }
        wait_loads();
        lds_barrier();

        // results of a step are stored at the top of the next one, in the shadow of its LDS reads
        float q_a = 0.f, q_b = 0.f, q_y = 0.f;
        int q_og = kOOB, q_oi = kOOB, q_oe = kOOB;
        bool q_last = false;
        auto flush = [&]() {
            buf_st2(rs_gates, q_og, q_a, q_b);        // [row][unit][r z | n gh_n]: each half stores a pair
            buf_st(rs_yint, q_oi, q_y);
            buf_st(rs_yext, q_oe, q_y);
            if (q_last) buf_st(rs_hn, (unit && p == 0) ? ((l * B + b) * H + j) * 4 : kOOB, q_y);
        };

        VSTAMP_DECL
        for (int k = 0; k < nticks; ++k) {
            const int par = k & 1;
            const int t = k - l;
            const bool work = t >= 0 && t < T;
            VSTAMP(5);
            if (work) {
                const bool active = t < len;
                const float *src = from_x ? xb + (t % kFwdSlots) * 32 : vsrc + par * L * KP;
                float4 v4[KP / 4];
#pragma unroll
                for (int q = 0; q < KP / 4; ++q) v4[q] = *reinterpret_cast<const float4 *>(src + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
                flush();
                __builtin_amdgcn_sched_barrier(0);
                VSTAMP(0);
                f32x2 acc0 = {0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
                for (int q = 0; q < KP / 4; ++q) {
                    const f32x2 va = {v4[q].x, v4[q].y}, vb = {v4[q].z, v4[q].w};
                    acc0 = pk_fma(w[0][2 * q], va, acc0); acc1 = pk_fma(w[1][2 * q], va, acc1);
                    acc2 = pk_fma(w[2][2 * q], va, acc2);
                    acc0 = pk_fma(w[0][2 * q + 1], vb, acc0); acc1 = pk_fma(w[1][2 * q + 1], vb, acc1);
                    acc2 = pk_fma(w[2][2 * q + 1], vb, acc2);
                }
                float gi_r, gh_r, gi_z, gh_z, gi_n, gh_n;
                VSTAMP(1);
                both_halves(acc0.x + acc0.y, gi_r, gh_r);
                both_halves(acc1.x + acc1.y, gi_z, gh_z);
                both_halves(acc2.x + acc2.y, gi_n, gh_n);
                const float r = fast_sigmoid(gi_r + gh_r + b_r);
                const float z = fast_sigmoid(gi_z + gh_z + b_z);
                const float ghn = gh_n + b_hn;
                const float n = fast_tanh(gi_n + b_in + r * ghn);
                const float hnew = n + z * (hreg - n);
                if (active && unit) hreg = hnew;
                if (p == 0 && j < KP) hw[(par ^ 1) * L * KP + j] = hreg;
                VSTAMP(2);
                // the two halves hold the same values: each stores half of them
                const int oi = ((t * B + b) * H + j) * 4;
                q_y = active ? hreg : 0.f;
                q_a = p ? n : r;
                q_b = p ? ghn : z;
                q_og = (unit && active) ? 4 * oi + 8 * p : kOOB;
                q_oi = (unit && p == 0) ? oi : kOOB;
                q_oe = (unit && p == 1) ? (ext_row(ym, t, b) * H + j) * 4 : kOOB;
                q_last = t == len - 1;
            } else {
                flush();
                q_og = q_oi = q_oe = kOOB;
                q_last = false;
            }
            VSTAMP(3);
            lds_barrier();
            VSTAMP(4);
        }
        flush();
        VSTAMP_FLUSH
    }
}

// ----------------------------------------------------------------------------------------------
// backward (dx, dh0 and per-workgroup partial weight / bias gradients)
// ----------------------------------------------------------------------------------------------
#undef VSTAMP_BASE
#define VSTAMP_BASE 8
template <int KP, bool PAIR>
__global__ __launch_bounds__(320) void gru_vec_bwd_kernel(VecPair<SmallBwdArgs> ga) {
    using namespace vec;
    const int set = PAIR && (int)blockIdx.x >= ga.nwg0;
    const SmallBwdArgs &a = ga.g[set];
    const int wg = set ? blockIdx.x - ga.nwg0 : blockIdx.x, nwg = set ? ga.nwg1 : ga.nwg0;
    // gv[L][4][KP] | dxb[2][L][KP] | ops[kBwdSlots][kSlotFloats]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int l = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // L = the loader wave
    const int p = lane >> 5, j = lane & 31;
    float *dxb = smem + L * 4 * KP;
    float *ops = dxb + 2 * L * KP;
    const long plane = (long)T * B * H;
    const int tbh = (int)(plane * 4);
    const int nticks = T + L - 1;

    if (l == L) {
        // ---- loader (lower half of the wave): operands of every layer, kBwdAhead steps ahead --------
        // At one tick layer q works on step t_q = T-1-tick+(L-1-q).  Per tick: the four saved gate
        // values of every layer (one 16-byte load per unit), the layer outputs y_q[t_{q+1}] (each is
        // both the input of layer q+1 and the previous state of layer q at this tick), the top
        // layer's previous state, the stack input x and the incoming gradient dy; all of them go
        // straight into LDS slot tick % kBwdSlots.  Straight-line code for kVecMaxL layers: a layer
        // the stack does not have gets descriptors with zero records (zeros land in LDS).
        const ExtMap dym = ext_map_of(a.dymap), xm = ext_map_of(a.xmap);
        __amdgpu_buffer_rsrc_t rs_g[kVecMaxL], rs_y[kVecMaxL];
#pragma unroll
        for (int q = 0; q < kVecMaxL; ++q) {
            rs_g[q] = make_rsrc(q < L ? a.gates[q] : nullptr, 4L * tbh);
            rs_y[q] = make_rsrc(q < L - 1 ? a.y_int[q] : nullptr, tbh);     // outputs feeding a layer above
        }
        const __amdgpu_buffer_rsrc_t rs_ytop = make_rsrc(a.y_int[L - 1], tbh);
        const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(a.dy_ext, tbh);
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, (long)T * B * a.In * 4);
        const int gstep = B * H * 16, ystep = B * H * 4, xstep = xm.st * a.In * 4, dstep = dym.st * H * 4;
        for (int b = wg; b < B; b += nwg) {
            const int len = a.lengths ? a.lengths[b] : T;
            const int gbase = j < H ? (b * H + j) * 16 : kOOB;
            const int ybase = j < H ? (b * H + j) * 4 : kOOB;
            const int xbase = j < a.In ? (b * xm.sb * a.In + j) * 4 : kOOB;
            const int dbase = j < H ? (b * dym.sb * H + j) * 4 : kOOB;
            auto issue = [&](int k) {
                if (p) return;
                float *slot = ops + (k % kBwdSlots) * kSlotFloats;
                const int t0 = T - 1 - k + (L - 1);          // step of layer 0; layer q works on t0 - q
#pragma unroll
                for (int q = 0; q < kVecMaxL; ++q) {
                    const int t = t0 - q;
                    const bool ok = t >= 0 && t < len;
                    lds_load_b128(rs_g[q], slot + kSlotG + q * 128, off_add(gbase, ok ? t * gstep : kOOB));
                    // y_{q-1}[t]: input of layer q and previous state of layer q-1 at this tick
                    if (q > 0) lds_load_b32(rs_y[q - 1], slot + kSlotY + (q - 1) * 32, off_add(ybase, ok ? t * ystep : kOOB));
                }
                const int tt = t0 - (L - 1);                  // step of the top layer
                const bool okt = tt >= 0 && tt < len;
                lds_load_b32(rs_ytop, slot + kSlotY + (kVecMaxL - 1) * 32, off_add(ybase, (okt && tt > 0) ? (tt - 1) * ystep : kOOB));
                lds_load_b32(rs_x, slot + kSlotX, off_add(xbase, (t0 >= 0 && t0 < len) ? t0 * xstep : kOOB));
                lds_load_b32(rs_dy, slot + kSlotD, off_add(dbase, okt ? tt * dstep : kOOB));
            };
            for (int k = 0; k < kBwdAhead; ++k) issue(k);
            wait_vmcnt<(kBwdAhead - 1) * kBwdLoads>();
            lds_barrier();
            for (int k = 0; k < nticks; ++k) {
                issue(k + kBwdAhead);
                wait_vmcnt<(kBwdAhead - 1) * kBwdLoads>();    // the operands of tick k+1 have landed
                lds_barrier();
            }
            wait_vmcnt<0>();                                  // the slots are reused by the next sequence
        }
        return;
    }

    const int In_l = (l == 0) ? a.In : H;
    float *gv = smem + l * 4 * KP;
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

    const __amdgpu_buffer_rsrc_t rs_dx = make_rsrc(l == 0 ? a.dx_ext : nullptr, (long)T * B * a.In * 4);
    const ExtMap dxm = ext_map_of(a.dxmap);

    for (int b = wg; b < B; b += nwg) {
        const int len = a.lengths ? a.lengths[b] : T;
        // the gradient of h_n enters as the initial carry: steps past the length pass it through
        const long sb = ((long)l * B + b) * H + j;
        float dhc = (a.dh_n && unit) ? a.dh_n[sb] : 0.f;
        const float h0v = (a.h0 && unit) ? a.h0[sb] : 0.f;
        wait_loads();
        lds_barrier();

        VSTAMP_DECL
        for (int k = 0; k < nticks; ++k) {
            const int par = k & 1;
            const int t = T - 1 - k + (L - 1 - l);
            VSTAMP(5);
            if (t >= 0 && t < T) {
                const bool active = t < len;
                const float *slot = ops + (k % kBwdSlots) * kSlotFloats;
                const float4 oa = *reinterpret_cast<const float4 *>(slot + kSlotG + (l * 32 + j) * 4);
                const float o_r = oa.x, o_z = oa.y, o_n = oa.z, o_hn = oa.w;
                const float o_hp = (t == 0) ? h0v : slot[kSlotY + (top ? kVecMaxL - 1 : l) * 32 + j];
                const float o_xin = slot[(l == 0 ? kSlotX : kSlotY + (l - 1) * 32) + j];
                float dyv = top ? slot[kSlotD + j] : 0.f;
                if (!top && j < KP) dyv += dxb[((par ^ 1) * L + (l + 1)) * KP + j];
                const bool on = active && unit;      // both halves run the gate arithmetic
                const float d = on ? dyv + dhc : 0.f;
                const float dn = d * (1.f - o_z);
                const float dzz = d * (o_hp - o_n);
                const float an = dn * (1.f - o_n * o_n);
                const float dar = an * o_hn * o_r * (1.f - o_r);
                const float daz = dzz * o_z * (1.f - o_z);
                const float dgn = an * o_r;
                const float keep = on ? d * o_z : dhc;
                if (j < KP) {   // each half publishes two of the four vectors
                    gv[(p ? 1 : 0) * KP + j] = p ? daz : dar;
                    gv[(p ? 3 : 2) * KP + j] = p ? dgn : an;
                }
                dbs[0] += dar; dbs[1] += daz; dbs[2] += an; dbs[3] += dgn;
                VSTAMP(0);
                // same wave wrote the vectors: LDS executes a wave's accesses in order
                VSTAMP(1);
                f32x2 acc0 = {0.f, 0.f}, acc1 = acc0, acc2 = acc0;
                const float *g2 = gv + (p ? 3 : 2) * KP;
                // rank-1 weight-gradient updates (rows from this half's dg, columns from its input):
                // one MFMA every other block of the matrix-vector product below, so that no MFMA has
                // to wait for the previous one and stall the instruction stream behind it
                const float s = p ? o_hp : o_xin;
                const float dg2 = p ? dgn : an;
#pragma unroll
                for (int q = 0; q < KP / 4; ++q) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(gv + 0 * KP + 4 * q);
                    const float4 v1 = *reinterpret_cast<const float4 *>(gv + 1 * KP + 4 * q);
                    const float4 v2 = *reinterpret_cast<const float4 *>(g2 + 4 * q);
                    acc0 = pk_fma(wT[0][2 * q], (f32x2){v0.x, v0.y}, acc0);
                    acc0 = pk_fma(wT[0][2 * q + 1], (f32x2){v0.z, v0.w}, acc0);
                    acc1 = pk_fma(wT[1][2 * q], (f32x2){v1.x, v1.y}, acc1);
                    acc1 = pk_fma(wT[1][2 * q + 1], (f32x2){v1.z, v1.w}, acc1);
                    acc2 = pk_fma(wT[2][2 * q], (f32x2){v2.x, v2.y}, acc2);
                    acc2 = pk_fma(wT[2][2 * q + 1], (f32x2){v2.z, v2.w}, acc2);
                    if (q == 0 || q == 2 || q == 4) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (q == 0) dWa[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(dar, s, dWa[0], 0, 0, 0);
                        if (q == 2) dWa[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(daz, s, dWa[1], 0, 0, 0);
                        if (q == 4) dWa[2] = __builtin_amdgcn_mfma_f32_32x32x1f32(dg2, s, dWa[2], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const f32x2 sum = acc0 + acc1 + acc2;
                VSTAMP(2);
                float dxv, dhv;
                both_halves(sum.x + sum.y, dxv, dhv);
                dhc = keep + (on ? dhv : 0.f);
                if (l > 0) {
                    if (p == 0 && j < KP) dxb[(par * L + l) * KP + j] = dxv;
                } else {
                    buf_st(rs_dx, (p == 0 && j < a.In) ? (ext_row(dxm, t, b) * a.In + j) * 4 : kOOB, dxv);
                }
            }
            VSTAMP(3);
            lds_barrier();
            VSTAMP(4);
        }
        VSTAMP_FLUSH
        if (a.dh0 && p == 0 && unit) a.dh0[sb] = dhc;
    }

    // ---- per-workgroup partial gradients: slab laid out like the gradients themselves -------------
    // result layout of the two-block 32x32x1 MFMA: register 16*blk + rr of lane (hh, jc) holds
    // D[blk][i][jc] with i = 8*(rr/4) + rr%4 + 4*hh
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    if (!a.slabs) return;        // no weight / bias gradients wanted by the caller: nothing will sum the partials
    long off = 0;
    for (int q = 0; q < l; ++q) off += 3L * H * ((q == 0 ? a.In : H) + H);
    float *slab = a.slabs + (long)wg * a.slab_stride + off;
    const __amdgpu_buffer_rsrc_t rs_slab = make_rsrc(slab, 3L * H * (In_l + H) * 4);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int nc = blk ? H : In_l;
        const int lane_off = j < nc ? (blk ? 3 * H * In_l + j : j) * 4 : kOOB;   // column j of dW_ih / dW_hh
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int i = 8 * (rr / 4) + (rr % 4) + 4 * p;                       // row inside the gate block
            const int row_off = i < H ? i * nc * 4 : kOOB;
#pragma unroll
            for (int g = 0; g < 3; ++g)
                buf_st(rs_slab, off_add(lane_off, row_off == kOOB ? kOOB : row_off + g * H * nc * 4), dWa[g][16 * blk + rr]);
        }
    }
    if (p == 0 && unit) {
        float *dbp = a.db_part + ((long)wg * L + l) * 4 * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) dbp[g * H + j] = dbs[g];
    }
}

// Workgroups of the two kernels (each loops over its share of the sequences).  The backward kernel
// fits one workgroup per CU; its count is also the number of partial weight gradients.
int gru_vec_workgroups(int B) { return B < 256 ? B : 256; }
static int fwd_workgroups(int B) { return B < 512 ? B : 512; }

// Up to 1024 sequences and 4 layers; byte offsets into every per-(t,b) buffer must fit 31 bits.
bool gru_vec_supported(int B, int T, int In, int H, int L) {
    if (H > 32 || In > 32 || L > vec::kVecMaxL || B > 1024) return false;
    const long widest = H > In ? H : In;
    return (long)T * B * widest * 16 < (1L << 31);
}

// The two calls of a pair share one launch: same layer count and the same padded width class, so the
// block shape and the LDS layout agree; batch size, sequence length and every pointer are per call.
bool gru_vec_pairable(int In0, int H0, int L0, int In1, int H1, int L1) {
    return L0 == L1 && ((H0 <= 24 && In0 <= 24) == (H1 <= 24 && In1 <= 24));
}

hipError_t launch_gru_vec_fwd(const SmallFwdArgs &a, const SmallFwdArgs *second, hipStream_t s) {
    const bool k24 = a.H <= 24 && a.In <= 24;
    const int KP = k24 ? 24 : 32;
    VecPair<SmallFwdArgs> ga{};
    ga.g[0] = a;
    ga.nwg0 = fwd_workgroups(a.B);
    if (second) {
        ga.g[1] = *second;
        ga.nwg1 = fwd_workgroups(second->B);
    }
    dim3 grid(ga.nwg0 + ga.nwg1), block(64 * (a.L + 1));
    const size_t lds = (size_t)(2 * a.L * KP + vec::kFwdSlots * 32) * sizeof(float);
    if (k24 && second) TGRU_LAUNCH((gru_vec_fwd_kernel<24, true>), grid, block, lds, s, ga);
    else if (k24) TGRU_LAUNCH((gru_vec_fwd_kernel<24, false>), grid, block, lds, s, ga);
    else if (second) TGRU_LAUNCH((gru_vec_fwd_kernel<32, true>), grid, block, lds, s, ga);
    else TGRU_LAUNCH((gru_vec_fwd_kernel<32, false>), grid, block, lds, s, ga);
    return hipGetLastError();
}

hipError_t launch_gru_vec_bwd(const SmallBwdArgs &a, const SmallBwdArgs *second, hipStream_t s) {
    const bool k24 = a.H <= 24 && a.In <= 24;
    const int KP = k24 ? 24 : 32;
    VecPair<SmallBwdArgs> ga{};
    ga.g[0] = a;
    ga.nwg0 = gru_vec_workgroups(a.B);
    if (second) {
        ga.g[1] = *second;
        ga.nwg1 = gru_vec_workgroups(second->B);
    }
    dim3 grid(ga.nwg0 + ga.nwg1), block(64 * (a.L + 1));
    const size_t lds = (size_t)((4 * a.L + 2 * a.L) * KP + vec::kBwdSlots * vec::kSlotFloats) * sizeof(float);
    if (k24 && second) TGRU_LAUNCH((gru_vec_bwd_kernel<24, true>), grid, block, lds, s, ga);
    else if (k24) TGRU_LAUNCH((gru_vec_bwd_kernel<24, false>), grid, block, lds, s, ga);
    else if (second) TGRU_LAUNCH((gru_vec_bwd_kernel<32, true>), grid, block, lds, s, ga);
    else TGRU_LAUNCH((gru_vec_bwd_kernel<32, false>), grid, block, lds, s, ga);
    return hipGetLastError();
}

}  // namespace tgru
