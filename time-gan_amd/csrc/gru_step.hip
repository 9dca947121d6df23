// Recurrent part of the GRU for large hidden states (H > 128, H a multiple of 32), on tiles of the
// (sequences x hidden units) plane -- 128 x 32 forward, 64 x 64 backward -- that cover 2048 sequences x 512
// units with one workgroup per compute unit.
//
// A workgroup that owned its sequences for all steps AND all units would need ALL of W_hh at every step (3 MB
// at H = 512, far beyond LDS) and stream it from the L2 once per step.  A tile needs 32 (x 3 gates) or 64 rows of
// the matrix and the whole previous state / gate gradients of its sequences, which the other unit tiles of the
// same row tile produce.  Kernel families, newest first (the launcher picks; each older one stays behind a
// switch and under the same tests):
//   * persistent, forward with the layer's input projection inside (loader waves, direct-to-LDS tiles);
//     persistent, backward with the W_hh tile resident in registers and LDS (H = 256 / 384 / 512);
//   * persistent with streamed tiles: the time loop inside the kernel, the workgroups of a row tile handing
//     each other a step's rows through write-through stores and an arrival counter;
//   * one launch per time step (no co-residency requirement).
//
// forward, step t:   gh = h_{t-1} . W_hh^T for the tile's r, z, n columns (v_mfma_f32_32x32x2_f32), then the gate
//                    nonlinearities, the state update and every store of the step in the epilogue -- the three
//                    gate values of a (sequence, unit) pair sit in the same lane and register of three accumulators.
// backward, step t:  dh_t = dy_t + carry + dgh_{t+1} . W_hh (K = 3H), then the gate gradients of step t in the
//                    epilogue: dg row t, the carry d*z for step t-1 and the running column sums for the bias
//                    gradients.
// Workgroups are numbered so that the tiles of one XCD (every eighth workgroup) share their rows of
// h / dg: the XCD's L2 then holds those rows plus W_hh, and a row tile's hand-off stays inside one L2.
#include "tgru_internal.h"
#include <stdlib.h>

namespace tgru {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRows = 128;    // sequences per workgroup (four waves of 32)
constexpr int kUnits = 32;    // hidden units per workgroup
constexpr int kFwdBK = 16;    // k-tile of the forward product
#ifndef TGRU_STEP_DEPTH
#define TGRU_STEP_DEPTH 4
#endif
constexpr int kDepth = TGRU_STEP_DEPTH;     // operand tiles in flight ahead of the product; even, so that the
                                            // LDS buffer of a tile is known at compile time (kt & 1 == d & 1)

struct StepFwd {
    FwdArgs f;
    const float *hprev;   // (B,H) state entering this step, or null for a zero state
    int t, MT, mtx;
};

struct StepBwd {
    BwdArgs b;
    float *carry;         // (B,H): d_{t+1} * z_{t+1}, updated in place
    int t;                // -1: only dh0 = carry + dgh_0 . W_hh
    int first;            // t == T-1: nothing carried yet, bias partial sums start here
    int MT, mtx;
};

// workgroup -> (row tile, unit tile); XCD x (workgroups x, x+8, ...) serves row tiles x, x+8, ...
__device__ __forceinline__ bool tile_of(int MT, int mtx, int &mt, int &nt) {
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
    mt = xcd + 8 * (k % mtx);
    nt = k / mtx;
    return mt < MT;
}

// row of the MFMA result held in register r of this lane (32x32 tile, lk = lane >> 5)
__device__ __forceinline__ int d_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(StepFwd s) {
    constexpr int BK = kFwdBK, BN = 3 * kUnits, LDA = kRows + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    const FwdArgs &a = s.f;
    const int H = a.H, B = a.B, T = a.T, t = s.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int m0 = mt * kRows, u0 = nt * kUnits;
    const int u = u0 + lrow;

    // Operands of the epilogue, requested before the product so that they have arrived by then.  Every
    // load is unconditional (rows past the batch are clamped, ended sequences read values nobody uses):
    // a per-lane condition in front of a load would turn sixteen loads in flight into sixteen round trips.
    int lenv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        lenv[r] = a.lengths ? a.lengths[mc] : T;
    }
    float gir[16], giz[16], gin[16], hp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        const float *gp = a.gi + ((long)t * B + mc) * 3 * H + u;
        gir[r] = gp[0];
        giz[r] = gp[H];
        gin[r] = gp[2 * H];
        hp[r] = s.hprev ? s.hprev[(long)mc * H + u] : 0.f;
    }
    unsigned valid = 0, on = 0, last = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = m0 + 32 * wave + d_row(r, lk) < B;
        valid |= (ok ? 1u : 0u) << r;
        on |= ((ok && t < lenv[r]) ? 1u : 0u) << r;
        last |= ((ok && t == lenv[r] - 1) ? 1u : 0u) << r;
    }
    const float b_r = a.b_hh[u], b_z = a.b_hh[H + u], b_n = a.b_hh[2 * H + u];

    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    if (s.hprev) {
        // tile slots (one float4 each): A kRows x BK/4, B 96 x BK/4 (rows of W_hh: gate*H + u0 + j)
        constexpr int KQ = BK / 4, PA = kRows * KQ / 256, PB = (BN * KQ + 255) / 256;
        // Element offsets of the slots at k0 = 0.  No load is conditional: rows past the batch are clamped
        // to the last sequence (their results are never stored), spare W_hh slots re-read a valid row.
        int a_off[PA], b_off[PB];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
            a_off[p] = min(m0 + row, B - 1) * H + kc;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int slot = min(tid + p * 256, BN * KQ - 1), row = slot / KQ, kc = (slot % KQ) << 2;
            b_off[p] = ((row >> 5) * H + u0 + (row & 31)) * H + kc;
        }
        struct Regs { float4 a[PA], b[PB]; };
        auto load_tiles = [&](int k0, Regs &g) {
#pragma unroll
            for (int p = 0; p < PA; ++p) g.a[p] = *reinterpret_cast<const float4 *>(s.hprev + a_off[p] + k0);
#pragma unroll
            for (int p = 0; p < PB; ++p) g.b[p] = *reinterpret_cast<const float4 *>(a.w_hh + b_off[p] + k0);
        };
        auto store_tiles = [&](int buf, const Regs &g) {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                As[buf][kc + 0][row] = g.a[p].x; As[buf][kc + 1][row] = g.a[p].y;
                As[buf][kc + 2][row] = g.a[p].z; As[buf][kc + 3][row] = g.a[p].w;
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                if (slot < BN * KQ) {
                    Bs[buf][kc + 0][row] = g.b[p].x; Bs[buf][kc + 1][row] = g.b[p].y;
                    Bs[buf][kc + 2][row] = g.b[p].z; Bs[buf][kc + 3][row] = g.b[p].w;
                }
            }
        };
        // The operands come from the far side of the L2 (it is written back and invalidated at every
        // kernel boundary), so tiles are requested kDepth products ahead in a register ring.
        const int nkt = H / BK;
        Regs ring[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) load_tiles(min(d, nkt - 1) * BK, ring[d]);
        store_tiles(0, ring[0]);
        load_tiles(min(kDepth, nkt - 1) * BK, ring[0]);
        __syncthreads();
        for (int kt0 = 0; kt0 < nkt; kt0 += kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const int kt = kt0 + d, cur = kt & 1;
                if (kt < nkt) {
                    auto product = [&](int kk) {
                        const float av = As[cur][kk + lk][32 * wave + lrow];
                        const float b0 = Bs[cur][kk + lk][lrow];
                        const float b1 = Bs[cur][kk + lk][32 + lrow];
                        const float b2 = Bs[cur][kk + lk][64 + lrow];
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, acc[2], 0, 0, 0);
                    };
                    // the next tile goes to the other LDS buffer in the MIDDLE of this tile's products: its
                    // LDS writes issue while the matrix pipe still has work queued
#pragma unroll
                    for (int kk = 0; kk < BK / 2; kk += 2) product(kk);
                    if (kt + 1 < nkt) {
                        store_tiles(cur ^ 1, ring[(d + 1) % kDepth]);
                        load_tiles(min(kt + 1 + kDepth, nkt - 1) * BK, ring[(d + 1) % kDepth]);
                    }
#pragma unroll
                    for (int kk = BK / 2; kk < BK; kk += 2) product(kk);
                    __syncthreads();
                }
            }
        }
    }

    const long plane = (long)T * B * H;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (!((valid >> r) & 1)) continue;
        const int m = m0 + 32 * wave + d_row(r, lk);
        const bool active = (on >> r) & 1;
        const long row = (long)t * B + m;
        const float ghn = acc[2][r] + b_n;
        const float rg = fast_sigmoid(gir[r] + acc[0][r] + b_r);
        const float zg = fast_sigmoid(giz[r] + acc[1][r] + b_z);
        const float ng = fast_tanh(gin[r] + rg * ghn);
        const float hnew = (1.f - zg) * ng + zg * hp[r];
        const float yv = active ? hnew : 0.f;
        if (a.y_int) a.y_int[row * H + u] = yv;
        if (a.y_ext) {
            const long er = map_tb(a.ymap, t, m);
            if (er >= 0) a.y_ext[er * H + u] = yv;
        }
        if (a.gates && active) {
            float *gp = a.gates + row * H + u;
            gp[0] = rg; gp[plane] = zg; gp[2 * plane] = ng; gp[3 * plane] = ghn;
        }
        if (a.h_n && ((last >> r) & 1)) a.h_n[(long)m * H + u] = yv;
    }
}

// ----------------------------------------------------------------------------------------------
// Persistent variants: the time loop INSIDE the kernel, same tiles.  A launch per step pays, at every
// step, the drain of the step's stores, the kernel boundary (with the write-back of the L2) and the
// first operand fetch from beyond the L2 -- about as long as the products themselves take.  But a row
// tile (128 sequences) at step t only depends on the SAME row tile at step t-1: the H / 32 workgroups
// that own its unit tiles.  So the workgroups stay resident for all steps and only the workgroups of one
// row tile wait for each other, through a counter in device memory:
//   * a workgroup stores its 128 x 32 tile of the new state with write-through (`sc1`) 16-byte stores,
//     every wave waits for its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and one lane
//     adds 1 to the row tile's counter (agent scope);
//   * before the products of step t one lane polls the counter (write-through loads) until all unit tiles
//     of step t-1 have arrived, the workgroup meets at a barrier, and the state tiles are fetched with
//     `sc1` loads (they bypass the CU's L1; each 128-byte line is written whole by one store instruction).
// Everything else a step needs from memory (its input projections, the saved gates) is requested BEFORE
// the wait; the workgroup's own tile of the previous state / carry stays in registers.
// All workgroups of a launch must be resident at once: the launcher gives a launch at most one workgroup
// per CU (a second one does not fit beside the first: registers) and splits larger batches into several
// launches.  Every wait is bounded: a workgroup that does not see its peers within `spin_limit` polls stops
// waiting, so the grid always drains (the results are then wrong: the call's error word is set, and the check
// kernel that ends the call turns it into NaN results and a count the host can read).
// ----------------------------------------------------------------------------------------------
#ifdef TGRU_STAMPS   // diagnostic build only: time (100 MHz ticks) one wave of workgroup 0 spends per phase
__device__ unsigned long long tgru_stamps_step[16];
#define PSTAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PSTAMP(k)                                                     \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += now_ - st_last; st_last = now_;                  \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define PSTAMP_FLUSH(base)                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0)                          \
        for (int k_ = 0; k_ < 8; ++k_) tgru_stamps_step[(base) + k_] = st_acc[k_];
extern "C" int tgru_debug_stamps_step(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_step), sizeof(unsigned long long) * 16);
}
#else
#define PSTAMP_DECL
#define PSTAMP(k)
#define PSTAMP_FLUSH(base)
#endif

#ifndef TGRU_PERSIST_DEPTH
#define TGRU_PERSIST_DEPTH 2
#endif
#ifndef TGRU_PERSIST_BK
#define TGRU_PERSIST_BK 64
#endif
constexpr int kPersistBK = TGRU_PERSIST_BK;                       // k-tile of the persistent kernels
constexpr int kPersistDepth = TGRU_PERSIST_DEPTH;    // operand tiles in flight ahead of the products (even)
constexpr int kAuxSc1 = 16;   // cache policy bits of the raw buffer builtins: sc1 (write-through / L1 bypass)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// descriptor of `bytes` bytes at p, or an empty one (loads give 0, stores are dropped) when p is null
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_opt(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
#ifndef TGRU_STREAM_AUX
#define TGRU_STREAM_AUX 0
#endif
// (cache policy of the once-touched streams -- input projections, saved gates, dy: 0 default, 2 non-temporal)
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, int byte_off, int sbase) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, sbase, TGRU_STREAM_AUX));
}
__device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t r, int byte_off, int sbase, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, sbase, TGRU_STREAM_AUX);
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kAuxSc1);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, float4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, kAuxSc1);
}

// The poll of one waiting lane: until `*cnt` has reached `want`, for at most `limit` polls.  A lane that gives up
// sets the call's error word; every 256th poll looks at that word, so that once one wait of a call has timed
// out no other wait of the call lasts longer than 256 polls (the results are wrong from then on anyway, and the
// grid must drain in bounded time).
__device__ __forceinline__ void poll_peers(unsigned *cnt, unsigned want, unsigned *err, int limit) {
    if (limit <= 0) {      // (test setting: behave as if the peers never came)
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (spins > limit ||
            ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// One lane of the workgroup waits until `*cnt` has reached `want`; the others wait at the barrier.
__device__ __forceinline__ void wait_for_peers(unsigned *cnt, unsigned want, unsigned *err, int limit) {
    if (threadIdx.x == 0) poll_peers(cnt, want, err, limit);
    __syncthreads();
}

// All stores of this workgroup issued so far are complete; then one lane tells the row tile's counter.
__device__ __forceinline__ void arrive(unsigned *cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PersistFwd {
    FwdArgs f;
    float *ybuf;          // (T*B, H) time-major buffer the states travel through (y_int, or a time-major y_ext)
    unsigned *sync;       // sync[1 + mt]: arrivals of row tile mt, zero at launch
    unsigned *err;        // error word of the library call this launch belongs to
    int mt0, MT, mtx;     // this launch serves row tiles [mt0, mt0 + MT)
    int spin_limit;       // polls after which a wait gives up
};

template <int BK>      // k-tile: 64, or 32 when H is not a multiple of 64
__global__ __launch_bounds__(256) void gru_persist_fwd_kernel(PersistFwd s) {
    // Operand tiles sit in LDS as the global rows do, k contiguous (rows of BK floats + 4 of padding), so a
    // tile is filled with one 16-byte write per 16-byte load and a lane fetches FOUR k-steps of an operand
    // with one 16-byte read.  The reduction index of an MFMA step may be permuted freely as long as both
    // operands agree: step j of chunk q stands for k = 8q + 4*(lane >> 5) + j for A and B alike.
    constexpr int BN = 3 * kUnits, LDK = BK + 4, LDH = kUnits + 4;
    // one LDS array: the operand tiles of the products, and -- once the products of a step are done -- the
    // new state tile on its way out
    __shared__ __attribute__((aligned(16))) float lds[2 * (kRows + BN) * LDK];
    static_assert(kRows * LDH <= 2 * (kRows + BN) * LDK, "state tile fits over the operand tiles");
    float (*As)[kRows][LDK] = reinterpret_cast<float (*)[kRows][LDK]>(lds);
    float (*Bs)[BN][LDK] = reinterpret_cast<float (*)[BN][LDK]>(lds + 2 * kRows * LDK);
    float (*Hs)[LDH] = reinterpret_cast<float (*)[LDH]>(lds);
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const FwdArgs &a = s.f;
    const int H = a.H, B = a.B, T = a.T, NT = H / kUnits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int m0 = mt * kRows, u0 = nt * kUnits;
    const int u = u0 + lrow;
    unsigned *cnt = s.sync + 1 + mt;
    const long plane = (long)T * B * H;
    const long step_bytes = (long)B * H * 4;

    int lenv[16];
    unsigned valid = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + d_row(r, lk);
        const int mc = min(m, B - 1);
        lenv[r] = a.lengths ? a.lengths[mc] : T;
        valid |= (m < B ? 1u : 0u) << r;
    }
    // byte offset of (sequence of register r, unit u) inside one step's (B, H) block; every per-step
    // operand is addressed as descriptor(step) + this offset, rows past the batch clamped to the last one
    int roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) roff[r] = (min(m0 + 32 * wave + d_row(r, lk), B - 1) * H + u) * 4;
    const float b_r = a.b_hh[u], b_z = a.b_hh[H + u], b_n = a.b_hh[2 * H + u];
    float hp[16];      // this workgroup's own tile of the state entering the step
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        hp[r] = a.h0 ? a.h0[(long)mc * H + u] : 0.f;
    }

    // tile slots (one float4 each): A kRows x BK/4, B 96 x BK/4 (rows of W_hh: gate*H + u0 + j); no load is
    // conditional: rows past the batch are clamped (their results are never stored), spare W_hh slots
    // re-read a valid row
    constexpr int KQ = BK / 4, PA = kRows * KQ / 256, PB = (BN * KQ + 255) / 256;
    int a_off[PA], b_off[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
        a_off[p] = (min(m0 + row, B - 1) * H + kc) * 4;       // bytes inside one step's (B, H) block
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int slot = min(tid + p * 256, BN * KQ - 1), row = slot / KQ, kc = (slot % KQ) << 2;
        b_off[p] = ((row >> 5) * H + u0 + (row & 31)) * H + kc;
    }
    struct Regs { float4 a[PA], b[PB]; };
    const int nkt = H / BK;
    PSTAMP_DECL

    for (int t = 0; t < T; ++t) {
        PSTAMP(7);
        // operands of the epilogue, requested before the wait and the products
        float gir[16], giz[16], gin[16];
        {
            const __amdgpu_buffer_rsrc_t rs_gi = rsrc_of(a.gi + (long)t * B * 3 * H, 3 * step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {      // row m of gi is 3H wide: 3 * (m * H) + u
                const int o = 3 * roff[r] - 8 * u;
                gir[r] = ld1(rs_gi, o, 0);
                giz[r] = ld1(rs_gi, o, H * 4);
                gin[r] = ld1(rs_gi, o, 2 * H * 4);
            }
        }
        f32x16 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

        const bool have_state = t > 0 || a.h0 != nullptr;
        PSTAMP(0);
        if (t > 0) wait_for_peers(cnt, (unsigned)(NT * t), s.err, s.spin_limit);
        PSTAMP(1);
        if (have_state) {
            const __amdgpu_buffer_rsrc_t rs_h =
                rsrc_of(t > 0 ? s.ybuf + (long)(t - 1) * B * H : a.h0, step_bytes);
            auto load_tiles = [&](int k0, Regs &g) {
#pragma unroll
                for (int p = 0; p < PA; ++p) g.a[p] = ld4_sc1(rs_h, a_off[p] + k0 * 4);
#pragma unroll
                for (int p = 0; p < PB; ++p) g.b[p] = *reinterpret_cast<const float4 *>(a.w_hh + b_off[p] + k0);
            };
            auto store_tiles = [&](int buf, const Regs &g) {
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                    *reinterpret_cast<float4 *>(&As[buf][row][kc]) = g.a[p];
                }
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                    if (slot < BN * KQ) *reinterpret_cast<float4 *>(&Bs[buf][row][kc]) = g.b[p];
                }
            };
            constexpr int kDepth = kPersistDepth;
            Regs ring[kDepth];
#pragma unroll
            for (int d = 0; d < kDepth; ++d) load_tiles(min(d, nkt - 1) * BK, ring[d]);
            store_tiles(0, ring[0]);
            load_tiles(min(kDepth, nkt - 1) * BK, ring[0]);
            __syncthreads();
            PSTAMP(2);
            for (int kt0 = 0; kt0 < nkt; kt0 += kDepth) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) {
                    const int kt = kt0 + d, cur = kt & 1;
                    if (kt < nkt) {
                        auto product = [&](int q) {      // chunk q: eight k-steps, four per lane half
                            const float4 av = *reinterpret_cast<const float4 *>(&As[cur][32 * wave + lrow][8 * q + 4 * lk]);
                            const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[cur][lrow][8 * q + 4 * lk]);
                            const float4 b1 = *reinterpret_cast<const float4 *>(&Bs[cur][32 + lrow][8 * q + 4 * lk]);
                            const float4 b2 = *reinterpret_cast<const float4 *>(&Bs[cur][64 + lrow][8 * q + 4 * lk]);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc[2], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc[2], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2.z, acc[2], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b2.w, acc[2], 0, 0, 0);
                        };
#pragma unroll
                        for (int q = 0; q < BK / 16; ++q) product(q);
                        if (kt + 1 < nkt) {
                            store_tiles(cur ^ 1, ring[(d + 1) % kDepth]);
                            load_tiles(min(kt + 1 + kDepth, nkt - 1) * BK, ring[(d + 1) % kDepth]);
                        }
#pragma unroll
                        for (int q = BK / 16; q < BK / 8; ++q) product(q);
                        __syncthreads();
                    }
                }
            }
        }

        PSTAMP(3);
        // gates and the new state of this tile; the state goes to LDS first so that it can leave as whole
        // 128-byte lines (one 16-byte store per lane)
        float rgv[16], zgv[16], ngv[16], ghv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool active = ((valid >> r) & 1) && t < lenv[r];
            const float ghn = acc[2][r] + b_n;
            const float rg = fast_sigmoid(gir[r] + acc[0][r] + b_r);
            const float zg = fast_sigmoid(giz[r] + acc[1][r] + b_z);
            const float ng = fast_tanh(gin[r] + rg * ghn);
            const float hnew = (1.f - zg) * ng + zg * hp[r];
            const float yv = active ? hnew : 0.f;
            rgv[r] = rg; zgv[r] = zg; ngv[r] = ng; ghv[r] = ghn;
            hp[r] = yv;
            Hs[32 * wave + d_row(r, lk)][lrow] = yv;
        }
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rs_y = rsrc_of(s.ybuf + (long)t * B * H, step_bytes);
#pragma unroll
            for (int p = 0; p < kRows * (kUnits / 4) / 256; ++p) {
                const int slot = tid + p * 256, row = slot >> 3, c4 = (slot & 7) << 2;
                const float4 v = *reinterpret_cast<const float4 *>(&Hs[row][c4]);
                // rows past the batch: an offset beyond the buffer, the store is dropped
                const int off = m0 + row < B ? ((m0 + row) * H + u0 + c4) * 4 : 0x7fffffff;
                st4_sc1(rs_y, off, v);
            }
        }
        PSTAMP(4);
        if (t + 1 < T) arrive(cnt);      // (also keeps Hs from being rewritten before every wave has read it)
        else __syncthreads();
        PSTAMP(5);
        // what no workgroup waits for: the caller's y (when it is not the exchange buffer), the saved
        // gates and h_n.  These stores drain while the next step's products run.
        if (a.y_ext && a.y_ext != s.ybuf) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + d_row(r, lk);
                const long er = ((valid >> r) & 1) ? map_tb(a.ymap, t, m) : -1L;
                if (er >= 0) a.y_ext[er * H + u] = hp[r];
            }
        }
        {   // (absent tensors have empty descriptors; an offset beyond the block drops a lane's store)
            float *gt = a.gates ? a.gates + (long)t * B * H : nullptr;
            const __amdgpu_buffer_rsrc_t rs_r = rsrc_opt(gt, step_bytes), rs_z = rsrc_opt(gt ? gt + plane : nullptr, step_bytes),
                                         rs_n = rsrc_opt(gt ? gt + 2 * plane : nullptr, step_bytes),
                                         rs_h = rsrc_opt(gt ? gt + 3 * plane : nullptr, step_bytes),
                                         rs_hn = rsrc_opt(a.h_n, step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (valid >> r) & 1;
                const int og = ok && t < lenv[r] ? roff[r] : 0x7fffffff;
                st1(rs_r, og, 0, rgv[r]); st1(rs_z, og, 0, zgv[r]); st1(rs_n, og, 0, ngv[r]); st1(rs_h, og, 0, ghv[r]);
                st1(rs_hn, ok && t == lenv[r] - 1 ? roff[r] : 0x7fffffff, 0, hp[r]);
            }
        }
        PSTAMP(6);
    }
    PSTAMP_FLUSH(0)
}

__global__ __launch_bounds__(256) void gru_step_bwd_kernel(StepBwd s) {
    constexpr int BK = 32, LDA = kRows + 4, LDB = kUnits + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    const BwdArgs &a = s.b;
    const int H = a.H, B = a.B, T = a.T, t = s.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int m0 = mt * kRows, u0 = nt * kUnits;
    const int u = u0 + lrow;
    const long plane = (long)T * B * H;

    // operands of the epilogue (gate gradients of step t), requested before the product; all loads are
    // unconditional, see the forward kernel
    const int tc = max(t, 0);
    int lenv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        lenv[r] = a.lengths ? a.lengths[mc] : T;
    }
    float gr[16], gz[16], gn[16], ghn[16], hp[16], dy[16], cin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        const long row = (long)tc * B + mc;
        const float *gp = a.gates + row * H + u;
        gr[r] = gp[0]; gz[r] = gp[plane]; gn[r] = gp[2 * plane]; ghn[r] = gp[3 * plane];
        cin[r] = s.first ? 0.f : s.carry[(long)mc * H + u];
        if (tc > 0) hp[r] = a.y_int[(row - B) * H + u];
        else hp[r] = a.h0 ? a.h0[(long)mc * H + u] : 0.f;
        float d = a.dy_int ? a.dy_int[row * H + u] : 0.f;
        if (a.dy_ext) {
            const long er = map_tb(a.dymap, tc, mc);
            const float e = a.dy_ext[max(er, 0L) * H + u];
            d += er >= 0 ? e : 0.f;
        }
        if (a.dh_n) {
            const float e = a.dh_n[(long)mc * H + u];
            d += tc == lenv[r] - 1 ? e : 0.f;
        }
        dy[r] = d;
    }
    unsigned valid = 0, on = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = m0 + 32 * wave + d_row(r, lk) < B;
        valid |= (ok ? 1u : 0u) << r;
        on |= ((ok && t >= 0 && t < lenv[r]) ? 1u : 0u) << r;
    }

    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    if (!s.first) {
        // A = (da_r | da_z | da_n*r) of step t+1: columns [0,2H) and [3H,4H) of the dg rows; B = W_hh (3H,H)
        const float *dgn = a.dg + (long)(t + 1) * B * 4 * H;
        constexpr int KQ = BK / 4, PA = kRows * KQ / 256;   // A: kRows x BK/4 float4 slots; B: BK x 8, one per thread
        static_assert(BK * 8 == 256, "one W_hh slot per thread");
        int a_off[PA];   // rows past the batch are clamped (their results are never stored): no conditional loads
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
            a_off[p] = min(m0 + row, B - 1) * 4 * H + kc;
        }
        const int b_k = tid >> 3, b_c = (tid & 7) << 2;
        const float *b_src = a.w_hh + (long)b_k * H + u0 + b_c;
        struct Regs { float4 a[PA], b; };
        auto load_tiles = [&](int k0, Regs &g) {
            const int kcol = k0 < 2 * H ? k0 : k0 + H;
#pragma unroll
            for (int p = 0; p < PA; ++p) g.a[p] = *reinterpret_cast<const float4 *>(dgn + a_off[p] + kcol);
            g.b = *reinterpret_cast<const float4 *>(b_src + (long)k0 * H);
        };
        auto store_tiles = [&](int buf, const Regs &g) {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                As[buf][kc + 0][row] = g.a[p].x; As[buf][kc + 1][row] = g.a[p].y;
                As[buf][kc + 2][row] = g.a[p].z; As[buf][kc + 3][row] = g.a[p].w;
            }
            *reinterpret_cast<float4 *>(&Bs[buf][b_k][b_c]) = g.b;
        };
        const int nkt = 3 * H / BK;
        Regs ring[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) load_tiles(min(d, nkt - 1) * BK, ring[d]);
        store_tiles(0, ring[0]);
        load_tiles(min(kDepth, nkt - 1) * BK, ring[0]);
        __syncthreads();
        for (int kt0 = 0; kt0 < nkt; kt0 += kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const int kt = kt0 + d, cur = kt & 1;
                if (kt < nkt) {
                    auto product = [&](int kk) {
                        const float a0 = As[cur][kk + lk][32 * wave + lrow], b0 = Bs[cur][kk + lk][lrow];
                        const float a1 = As[cur][kk + 2 + lk][32 * wave + lrow], b1 = Bs[cur][kk + 2 + lk][lrow];
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1], 0, 0, 0);
                    };
#pragma unroll
                    for (int kk = 0; kk < BK / 2; kk += 4) product(kk);
                    if (kt + 1 < nkt) {   // in the middle of the tile's products, see the forward kernel
                        store_tiles(cur ^ 1, ring[(d + 1) % kDepth]);
                        load_tiles(min(kt + 1 + kDepth, nkt - 1) * BK, ring[(d + 1) % kDepth]);
                    }
#pragma unroll
                    for (int kk = BK / 2; kk < BK; kk += 4) product(kk);
                    __syncthreads();
                }
            }
        }
    }

    if (t < 0) {   // gradient of the initial state
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!((valid >> r) & 1)) continue;
            const int m = m0 + 32 * wave + d_row(r, lk);
            a.dh0[(long)m * H + u] = cin[r] + acc[0][r] + acc[1][r];
        }
        return;
    }

    float sr = 0.f, sz = 0.f, sn = 0.f, sg = 0.f;   // column sums over this wave's 32 sequences
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (!((valid >> r) & 1)) continue;
        const int m = m0 + 32 * wave + d_row(r, lk);
        const bool active = (on >> r) & 1;
        const float dhc = cin[r] + acc[0][r] + acc[1][r];
        const float d = active ? dy[r] + dhc : 0.f;
        const float dn = d * (1.f - gz[r]);
        const float dzz = d * (hp[r] - gn[r]);
        const float an = active ? dn * (1.f - gn[r] * gn[r]) : 0.f;
        const float dar = active ? an * ghn[r] * gr[r] * (1.f - gr[r]) : 0.f;
        const float daz = active ? dzz * gz[r] * (1.f - gz[r]) : 0.f;
        const float dgn = active ? an * gr[r] : 0.f;
        s.carry[(long)m * H + u] = active ? d * gz[r] : dhc;
        float *dst = a.dg + ((long)t * B + m) * 4 * H + u;
        dst[0] = dar; dst[H] = daz; dst[2 * H] = an; dst[3 * H] = dgn;
        sr += dar; sz += daz; sn += an; sg += dgn;
    }
    sr += __shfl_xor(sr, 32); sz += __shfl_xor(sz, 32); sn += __shfl_xor(sn, 32); sg += __shfl_xor(sg, 32);
    if (lk == 0) {
        // one partial row per (row tile, wave); this workgroup is the only writer of its 32 columns
        float *p = a.db_part + (long)(mt * 4 + wave) * 4 * H + u;
        if (s.first) { p[0] = sr; p[H] = sz; p[2 * H] = sn; p[3 * H] = sg; }
        else { p[0] += sr; p[H] += sz; p[2 * H] += sn; p[3 * H] += sg; }
    }
}

struct PersistBwd {
    BwdArgs b;
    unsigned *sync;       // sync[1 + mt]: arrivals of row tile mt, zero at launch
    unsigned *err;        // error word of the library call this launch belongs to
    int mt0, MT, mtx;
    int spin_limit;       // polls after which a wait gives up
};

// Backward twin of gru_persist_fwd_kernel.  The workgroups of a row tile hand each other the gate gradients
// of step t+1 (rows of `dg`: every 128-byte line is written whole, write-through, by one store instruction);
// the carry d*z of the workgroup's own tile and the running bias sums stay in registers for all steps.
template <int BK>
__global__ __launch_bounds__(256) void gru_persist_bwd_kernel(PersistBwd s) {
    // A tiles as in the forward kernel: rows k-contiguous, 16-byte writes and 16-byte fragment reads, MFMA
    // step j of chunk q standing for k = 8q + 4*(lane >> 5) + j; the W_hh tile is k-major as in memory
    constexpr int LDK = BK + 4, LDB = kUnits + 4, LDH = kUnits + 4;
    // one LDS array: the operand tiles of the products, and -- once the products of a step are done -- the
    // four gate-gradient tiles on their way out
    constexpr int kTileFloats = 2 * kRows * LDK + 2 * BK * LDB, kOutFloats = 4 * kRows * LDH;
    __shared__ __attribute__((aligned(16))) float lds[kTileFloats > kOutFloats ? kTileFloats : kOutFloats];
    float (*As)[kRows][LDK] = reinterpret_cast<float (*)[kRows][LDK]>(lds);
    float (*Bs)[BK][LDB] = reinterpret_cast<float (*)[BK][LDB]>(lds + 2 * kRows * LDK);
    float (*Gs)[kRows][LDH] = reinterpret_cast<float (*)[kRows][LDH]>(lds);
    __shared__ int Ls[kRows];      // length of each sequence of the row tile (0 for rows past the batch)
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const BwdArgs &a = s.b;
    const int H = a.H, B = a.B, T = a.T, NT = H / kUnits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int m0 = mt * kRows, u0 = nt * kUnits;
    const int u = u0 + lrow;
    unsigned *cnt = s.sync + 1 + mt;
    const long plane = (long)T * B * H;
    const long step_bytes = (long)B * 4 * H * 4;      // one step's rows of dg

    if (tid < kRows) Ls[tid] = m0 + tid < B ? (a.lengths ? a.lengths[m0 + tid] : T) : 0;
    __syncthreads();
    // byte offset of (sequence of register r, unit u) inside one step's (B, H) block, rows past the batch
    // clamped to the last one; every per-step operand is addressed as descriptor(step) + this offset
    int roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) roff[r] = (min(m0 + 32 * wave + d_row(r, lk), B - 1) * H + u) * 4;
    const long bh_bytes = (long)B * H * 4;
    // the caller's dy: time-major (row t*B + b) or batch-major (row b*T + t); s.dy_rows = rows of a sequence
    const int dy_bt = a.dymap.mode == 1;
    float cin[16];          // d_{t+1} * z_{t+1} of this workgroup's own tile
#pragma unroll
    for (int r = 0; r < 16; ++r) cin[r] = 0.f;
    float sr = 0.f, sz = 0.f, sn = 0.f, sg = 0.f;   // column sums over this wave's 32 sequences, all steps

    constexpr int KQ = BK / 4, PA = kRows * KQ / 256, PB = BK * 8 / 256;   // float4 slots: A kRows x BK/4, B BK x 8
    static_assert(BK * 8 % 256 == 0, "whole W_hh slots per thread");
    int a_off[PA];   // rows past the batch are clamped (their results are never stored): no conditional loads
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
        a_off[p] = (min(m0 + row, B - 1) * 4 * H + kc) * 4;
    }
    const int b_k = tid >> 3, b_c = (tid & 7) << 2;      // slot p of this thread: k = b_k + 32 p
    const float *b_src = a.w_hh + (long)b_k * H + u0 + b_c;
    struct Regs { float4 a[PA], b[PB]; };
    const int nkt = 3 * H / BK;
    const int t_last = a.dh0 ? -1 : 0;
    PSTAMP_DECL

    for (int t = T - 1; t >= t_last; --t) {
        PSTAMP(7);
        // operands of the epilogue (gate gradients of step t), requested before the wait and the products
        const int tc = max(t, 0);
        // No load below is conditional: absent tensors have empty descriptors (their loads give 0), and
        // which tensor a descriptor points at is decided once per step, not per element.
        float gr[16], gz[16], gn[16], ghn[16], hp[16], dy[16];
        {
            const float *gt = a.gates + (long)tc * B * H;
            const __amdgpu_buffer_rsrc_t rs_r = rsrc_of(gt, bh_bytes), rs_z = rsrc_of(gt + plane, bh_bytes),
                                         rs_n = rsrc_of(gt + 2 * plane, bh_bytes), rs_h = rsrc_of(gt + 3 * plane, bh_bytes);
            const __amdgpu_buffer_rsrc_t rs_hp = rsrc_opt(tc > 0 ? a.y_int + (long)(tc - 1) * B * H : a.h0, bh_bytes);
            const __amdgpu_buffer_rsrc_t rs_di = rsrc_opt(a.dy_int ? a.dy_int + (long)tc * B * H : nullptr, bh_bytes);
            // batch-major dy: the block of step tc starts tc rows in and its sequences are T rows apart
            const __amdgpu_buffer_rsrc_t rs_de =
                rsrc_opt(a.dy_ext ? a.dy_ext + (dy_bt ? (long)tc * H : (long)tc * B * H) : nullptr,
                         dy_bt ? ((long)(B - 1) * T + 1) * H * 4 : bh_bytes);
            const __amdgpu_buffer_rsrc_t rs_dn = rsrc_opt(a.dh_n, bh_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gr[r] = ld1(rs_r, roff[r], 0); gz[r] = ld1(rs_z, roff[r], 0);
                gn[r] = ld1(rs_n, roff[r], 0); ghn[r] = ld1(rs_h, roff[r], 0);
                hp[r] = ld1(rs_hp, roff[r], 0);
                const int oe = dy_bt ? (roff[r] - 4 * u) * T + 4 * u : roff[r];
                const float e = ld1(rs_dn, roff[r], 0);
                dy[r] = ld1(rs_di, roff[r], 0) + ld1(rs_de, oe, 0) + (tc == Ls[32 * wave + d_row(r, lk)] - 1 ? e : 0.f);
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

        PSTAMP(0);
        if (t < T - 1) {
            wait_for_peers(cnt, (unsigned)(NT * (T - 1 - t)), s.err, s.spin_limit);
            PSTAMP(1);
            // A = (da_r | da_z | da_n*r) of step t+1: columns [0,2H) and [3H,4H) of the dg rows; B = W_hh (3H,H)
            const __amdgpu_buffer_rsrc_t rs_dg = rsrc_of(a.dg + (long)(t + 1) * B * 4 * H, step_bytes);
            auto load_tiles = [&](int k0, Regs &g) {
                const int kcol = k0 < 2 * H ? k0 : k0 + H;
#pragma unroll
                for (int p = 0; p < PA; ++p) g.a[p] = ld4_sc1(rs_dg, a_off[p] + kcol * 4);
#pragma unroll
                for (int p = 0; p < PB; ++p) g.b[p] = *reinterpret_cast<const float4 *>(b_src + (long)(k0 + 32 * p) * H);
            };
            auto store_tiles = [&](int buf, const Regs &g) {
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                    *reinterpret_cast<float4 *>(&As[buf][row][kc]) = g.a[p];
                }
#pragma unroll
                for (int p = 0; p < PB; ++p) *reinterpret_cast<float4 *>(&Bs[buf][b_k + 32 * p][b_c]) = g.b[p];
            };
            constexpr int kDepth = kPersistDepth;
            Regs ring[kDepth];
#pragma unroll
            for (int d = 0; d < kDepth; ++d) load_tiles(min(d, nkt - 1) * BK, ring[d]);
            store_tiles(0, ring[0]);
            load_tiles(min(kDepth, nkt - 1) * BK, ring[0]);
            __syncthreads();
            PSTAMP(2);
            for (int kt0 = 0; kt0 < nkt; kt0 += kDepth) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) {
                    const int kt = kt0 + d, cur = kt & 1;
                    if (kt < nkt) {
                        auto product = [&](int q) {      // chunk q: eight k-steps, four per lane half
                            const int k = 8 * q + 4 * lk;
                            const float4 av = *reinterpret_cast<const float4 *>(&As[cur][32 * wave + lrow][k]);
                            const float b0 = Bs[cur][k][lrow], b1 = Bs[cur][k + 1][lrow];
                            const float b2 = Bs[cur][k + 2][lrow], b3 = Bs[cur][k + 3][lrow];
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc[1], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc[1], 0, 0, 0);
                        };
#pragma unroll
                        for (int q = 0; q < BK / 16; ++q) product(q);
                        if (kt + 1 < nkt) {
                            store_tiles(cur ^ 1, ring[(d + 1) % kDepth]);
                            load_tiles(min(kt + 1 + kDepth, nkt - 1) * BK, ring[(d + 1) % kDepth]);
                        }
#pragma unroll
                        for (int q = BK / 16; q < BK / 8; ++q) product(q);
                        __syncthreads();
                    }
                }
            }
        }

        PSTAMP(3);
        if (t < 0) {   // gradient of the initial state
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + d_row(r, lk);
                if (m < B) a.dh0[(long)m * H + u] = cin[r] + acc[0][r] + acc[1][r];
            }
            break;
        }

#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + d_row(r, lk);
            const bool active = t < Ls[row];
            const float dhc = cin[r] + acc[0][r] + acc[1][r];
            const float d = active ? dy[r] + dhc : 0.f;
            const float dn = d * (1.f - gz[r]);
            const float dzz = d * (hp[r] - gn[r]);
            const float an = active ? dn * (1.f - gn[r] * gn[r]) : 0.f;
            const float dar = active ? an * ghn[r] * gr[r] * (1.f - gr[r]) : 0.f;
            const float daz = active ? dzz * gz[r] * (1.f - gz[r]) : 0.f;
            const float dgn = active ? an * gr[r] : 0.f;
            cin[r] = active ? d * gz[r] : dhc;
            Gs[0][row][lrow] = dar; Gs[1][row][lrow] = daz; Gs[2][row][lrow] = an; Gs[3][row][lrow] = dgn;
            sr += dar; sz += daz; sn += an; sg += dgn;      // (zero for ended sequences and rows past the batch)
        }
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(a.dg + (long)t * B * 4 * H, step_bytes);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < kRows * (kUnits / 4) / 256; ++p) {
                    const int slot = tid + p * 256, row = slot >> 3, c4 = (slot & 7) << 2;
                    const float4 v = *reinterpret_cast<const float4 *>(&Gs[q][row][c4]);
                    const int off = m0 + row < B ? ((m0 + row) * 4 * H + q * H + u0 + c4) * 4 : 0x7fffffff;
                    st4_sc1(rs_o, off, v);
                }
        }
        PSTAMP(4);
        if (t > t_last) arrive(cnt);     // (also keeps Gs from being rewritten before every wave has read it)
        PSTAMP(5);
    }
    PSTAMP_FLUSH(8)

    sr += __shfl_xor(sr, 32); sz += __shfl_xor(sz, 32); sn += __shfl_xor(sn, 32); sg += __shfl_xor(sg, 32);
    if (lk == 0) {
        // one partial row per (row tile, wave); this workgroup is the only writer of its 32 columns
        float *p = a.db_part + (long)(mt * 4 + wave) * 4 * H + u;
        p[0] = sr; p[H] = sz; p[2 * H] = sn; p[3 * H] = sg;
    }
}


// The same kernel on 64-sequence x 64-unit tiles (two row blocks x two unit blocks of waves).  A backward step
// is bound by the bytes a CU takes from the L2 (its operands are (rows + units) x 3H floats per workgroup and
// step): the square tile reads a fifth less than the 128 x 32 one, and a row tile's hand-off group is eight
// workgroups instead of sixteen.  Needs H to be a multiple of 64.
template <int BK>
__global__ __launch_bounds__(256) void gru_persist_bwd_sq_kernel(PersistBwd s) {
    constexpr int kRows = 64, kUnits = 64;       // (shadow the 128 x 32 geometry of the other kernels)
    // A tiles as in the forward kernel: rows k-contiguous, 16-byte writes and 16-byte fragment reads, MFMA
    // step j of chunk q standing for k = 8q + 4*(lane >> 5) + j; the W_hh tile is k-major as in memory
    constexpr int LDK = BK + 4, LDB = kUnits + 4, LDH = kUnits + 4;
    static_assert(BK % 16 == 0, "k-tile");
    // one LDS array: the operand tiles of the products, and -- once the products of a step are done -- the
    // four gate-gradient tiles on their way out
    constexpr int kTileFloats = 2 * kRows * LDK + 2 * BK * LDB, kOutFloats = 4 * kRows * LDH;
    __shared__ __attribute__((aligned(16))) float lds[kTileFloats > kOutFloats ? kTileFloats : kOutFloats];
    float (*As)[kRows][LDK] = reinterpret_cast<float (*)[kRows][LDK]>(lds);
    float (*Bs)[BK][LDB] = reinterpret_cast<float (*)[BK][LDB]>(lds + 2 * kRows * LDK);
    float (*Gs)[kRows][LDH] = reinterpret_cast<float (*)[kRows][LDH]>(lds);
    __shared__ int Ls[kRows];      // length of each sequence of the row tile (0 for rows past the batch)
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const BwdArgs &a = s.b;
    const int H = a.H, B = a.B, T = a.T, NT = H / kUnits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;      // this wave's row block and unit block of the tile
    const int m0 = mt * kRows, u0 = nt * kUnits;
    const int u = u0 + 32 * wc + lrow;
    unsigned *cnt = s.sync + 1 + mt;
    const long plane = (long)T * B * H;
    const long step_bytes = (long)B * 4 * H * 4;      // one step's rows of dg

    if (tid < kRows) Ls[tid] = m0 + tid < B ? (a.lengths ? a.lengths[m0 + tid] : T) : 0;
    __syncthreads();
    // byte offset of (sequence of register r, unit u) inside one step's (B, H) block, rows past the batch
    // clamped to the last one; every per-step operand is addressed as descriptor(step) + this offset
    int roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) roff[r] = (min(m0 + 32 * wr + d_row(r, lk), B - 1) * H + u) * 4;
    const long bh_bytes = (long)B * H * 4;
    // the caller's dy: time-major (row t*B + b) or batch-major (row b*T + t); s.dy_rows = rows of a sequence
    const int dy_bt = a.dymap.mode == 1;
    // d_{t+1} * z_{t+1} of this wave's part of the tile.  It starts as the gradient of the final state: steps past a
    // sequence's end pass the carry on unchanged (their gate gradients are zero), so it reaches the last valid
    // step, where the other kernels add dh_n to dy -- one load per step and element less.
    float cin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wr + d_row(r, lk), B - 1);
        cin[r] = a.dh_n ? a.dh_n[(long)mc * H + u] : 0.f;
    }
    const bool dy_ext = a.dy_ext != nullptr;      // (the launcher sends a call with both dy tensors to the 128 x 32 kernel)
    float sr = 0.f, sz = 0.f, sn = 0.f, sg = 0.f;   // column sums over this wave's 32 sequences, all steps

    constexpr int KQ = BK / 4, PA = kRows * KQ / 256, PB = BK * 16 / 256;   // float4 slots: A kRows x BK/4, B BK x 16
    static_assert(BK * 16 % 256 == 0 && kRows * KQ % 256 == 0, "whole slots per thread");
    int a_off[PA];   // rows past the batch are clamped (their results are never stored): no conditional loads
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
        a_off[p] = (min(m0 + row, B - 1) * 4 * H + kc) * 4;
    }
    const int b_k = tid >> 4, b_c = (tid & 15) << 2;     // slot p of this thread: k = b_k + 16 p
    const float *b_src = a.w_hh + (long)b_k * H + u0 + b_c;
    struct Regs { float4 a[PA], b[PB]; };
    const int nkt = 3 * H / BK;
    const int t_last = a.dh0 ? -1 : 0;
    

    for (int t = T - 1; t >= t_last; --t) {
        // operands of the epilogue (gate gradients of step t), requested before the wait and the products
        const int tc = max(t, 0);
        // No load below is conditional: absent tensors have empty descriptors (their loads give 0), and
        // which tensor a descriptor points at is decided once per step, not per element.
        float gr[16], gz[16], gn[16], ghn[16], hp[16], dy[16];
        {
            const float *gt = a.gates + (long)tc * B * H;
            const __amdgpu_buffer_rsrc_t rs_r = rsrc_of(gt, bh_bytes), rs_z = rsrc_of(gt + plane, bh_bytes),
                                         rs_n = rsrc_of(gt + 2 * plane, bh_bytes), rs_h = rsrc_of(gt + 3 * plane, bh_bytes);
            const __amdgpu_buffer_rsrc_t rs_hp = rsrc_opt(tc > 0 ? a.y_int + (long)(tc - 1) * B * H : a.h0, bh_bytes);
            // the gradient of this layer's output: the caller's dy (top layer; batch-major: the block of step tc
            // starts tc rows in and its sequences are T rows apart) or the layer above's input gradient
            const bool bt = dy_ext && dy_bt;
            const __amdgpu_buffer_rsrc_t rs_d =
                dy_ext ? rsrc_of(a.dy_ext + (bt ? (long)tc * H : (long)tc * B * H), bt ? ((long)(B - 1) * T + 1) * H * 4 : bh_bytes)
                       : rsrc_opt(a.dy_int ? a.dy_int + (long)tc * B * H : nullptr, bh_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gr[r] = ld1(rs_r, roff[r], 0); gz[r] = ld1(rs_z, roff[r], 0);
                gn[r] = ld1(rs_n, roff[r], 0); ghn[r] = ld1(rs_h, roff[r], 0);
                hp[r] = ld1(rs_hp, roff[r], 0);
                dy[r] = ld1(rs_d, bt ? (roff[r] - 4 * u) * T + 4 * u : roff[r], 0);
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        if (t < T - 1) {
            // A = (da_r | da_z | da_n*r) of step t+1: columns [0,2H) and [3H,4H) of the dg rows; B = W_hh (3H,H)
            const __amdgpu_buffer_rsrc_t rs_dg = rsrc_of(a.dg + (long)(t + 1) * B * 4 * H, step_bytes);
            auto load_a = [&](int k0, Regs &g) {
                const int kcol = k0 < 2 * H ? k0 : k0 + H;
#pragma unroll
                for (int p = 0; p < PA; ++p) g.a[p] = ld4_sc1(rs_dg, a_off[p] + kcol * 4);
            };
            auto load_b = [&](int k0, Regs &g) {
#pragma unroll
                for (int p = 0; p < PB; ++p) g.b[p] = *reinterpret_cast<const float4 *>(b_src + (long)(k0 + 16 * p) * H);
            };
            auto load_tiles = [&](int k0, Regs &g) { load_a(k0, g); load_b(k0, g); };
            auto store_tiles = [&](int buf, const Regs &g) {
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int slot = tid + p * 256, row = slot / KQ, kc = (slot % KQ) << 2;
                    *reinterpret_cast<float4 *>(&As[buf][row][kc]) = g.a[p];
                }
#pragma unroll
                for (int p = 0; p < PB; ++p) *reinterpret_cast<float4 *>(&Bs[buf][b_k + 16 * p][b_c]) = g.b[p];
            };
            constexpr int kDepth = kPersistDepth;
            Regs ring[kDepth];
            // the W_hh halves of the first tiles depend on nobody: they are requested before the wait for the peers
#pragma unroll
            for (int d = 0; d < kDepth; ++d) load_b(min(d, nkt - 1) * BK, ring[d]);
            wait_for_peers(cnt, (unsigned)(NT * (T - 1 - t)), s.err, s.spin_limit);
#pragma unroll
            for (int d = 0; d < kDepth; ++d) load_a(min(d, nkt - 1) * BK, ring[d]);
            store_tiles(0, ring[0]);
            load_tiles(min(kDepth, nkt - 1) * BK, ring[0]);
            __syncthreads();
            for (int kt0 = 0; kt0 < nkt; kt0 += kDepth) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) {
                    const int kt = kt0 + d, cur = kt & 1;
                    if (kt < nkt) {
                        auto product = [&](int q) {      // chunk q: eight k-steps, four per lane half
                            const int k = 8 * q + 4 * lk;
                            const float4 av = *reinterpret_cast<const float4 *>(&As[cur][32 * wr + lrow][k]);
                            const float b0 = Bs[cur][k][32 * wc + lrow], b1 = Bs[cur][k + 1][32 * wc + lrow];
                            const float b2 = Bs[cur][k + 2][32 * wc + lrow], b3 = Bs[cur][k + 3][32 * wc + lrow];
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc[1], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc[1], 0, 0, 0);
                        };
#pragma unroll
                        for (int q = 0; q < BK / 16; ++q) product(q);
                        if (kt + 1 < nkt) {
                            store_tiles(cur ^ 1, ring[(d + 1) % kDepth]);
                            load_tiles(min(kt + 1 + kDepth, nkt - 1) * BK, ring[(d + 1) % kDepth]);
                        }
#pragma unroll
                        for (int q = BK / 16; q < BK / 8; ++q) product(q);
                        __syncthreads();
                    }
                }
            }
        }
        if (t < 0) {   // gradient of the initial state
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wr + d_row(r, lk);
                if (m < B) a.dh0[(long)m * H + u] = cin[r] + acc[0][r] + acc[1][r];
            }
            break;
        }

#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wr + d_row(r, lk);
            const bool active = t < Ls[row];
            const float dhc = cin[r] + acc[0][r] + acc[1][r];
            const float d = active ? dy[r] + dhc : 0.f;
            const float dn = d * (1.f - gz[r]);
            const float dzz = d * (hp[r] - gn[r]);
            const float an = active ? dn * (1.f - gn[r] * gn[r]) : 0.f;
            const float dar = active ? an * ghn[r] * gr[r] * (1.f - gr[r]) : 0.f;
            const float daz = active ? dzz * gz[r] * (1.f - gz[r]) : 0.f;
            const float dgn = active ? an * gr[r] : 0.f;
            cin[r] = active ? d * gz[r] : dhc;
            Gs[0][row][32 * wc + lrow] = dar; Gs[1][row][32 * wc + lrow] = daz; Gs[2][row][32 * wc + lrow] = an; Gs[3][row][32 * wc + lrow] = dgn;
            sr += dar; sz += daz; sn += an; sg += dgn;      // (zero for ended sequences and rows past the batch)
        }
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(a.dg + (long)t * B * 4 * H, step_bytes);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < kRows * (kUnits / 4) / 256; ++p) {
                    const int slot = tid + p * 256, row = slot >> 4, c4 = (slot & 15) << 2;
                    const float4 v = *reinterpret_cast<const float4 *>(&Gs[q][row][c4]);
                    const int off = m0 + row < B ? ((m0 + row) * 4 * H + q * H + u0 + c4) * 4 : 0x7fffffff;
                    st4_sc1(rs_o, off, v);
                }
        }
        if (t > t_last) arrive(cnt);     // (also keeps Gs from being rewritten before every wave has read it)
    }
    

    sr += __shfl_xor(sr, 32); sz += __shfl_xor(sz, 32); sn += __shfl_xor(sn, 32); sg += __shfl_xor(sg, 32);
    if (lk == 0) {
        // one partial row per block of 32 sequences; this wave is the only writer of its 32 columns of it
        float *p = a.db_part + (long)(mt * 2 + wr) * 4 * H + u;
        p[0] = sr; p[H] = sz; p[2 * H] = sn; p[3 * H] = sg;
    }
}



// ----------------------------------------------------------------------------------------------
// Loader-wave variant of the persistent forward kernel.  Moving an operand tile from registers into LDS
// (ds_write_b128) is not covered by the products: the matrix pipe waits while the data crosses from the
// register file, and with one workgroup per CU no other wave fills the gap (one MFMA per 80 cycles where the
// pipe takes one per 64).  Here two extra waves fetch every operand tile with direct-to-LDS loads into a ring
// of slots, ahead of the four product waves, which only read LDS, multiply and do the gate arithmetic; all
// six waves meet at one barrier per k-tile.  The loader waves alone issue vector-memory loads inside the
// product loop, so their vmcnt is an exact count of tiles in flight.  A loader wave also does the waiting for
// the row tile's peers (it is the wave that loads what they hand over), so the product waves go from the
// gates of one step straight to the barrier of the next step's first tile.
//
// A direct-to-LDS load writes 64 consecutive 16-byte pieces, so tiles have no padding: a k-contiguous tile
// sits as [row][BK floats], piece c of row r at position (c + r * (BK/4) / 16) modulo BK/4 of its row, which
// makes the 16-byte fragment reads conflict-free; the rotation is applied on the global side (each lane
// fetches the piece that belongs at its LDS position).
// The backward kernels keep their register-staged tiles: built the same way (four loader waves, the saved gates
// streamed into the ring slots that fall free under the last two k-tiles), on 128 x 32 tiles as on 64 x 64 ones,
// they were no faster.  A backward k-tile is 32 - 40 KiB per 32 products against 28 KiB per 48 here, 0.8 - 1 MB
// per step and CU: with the products removed a step still took 26 us, with the tile loads removed 34 us (21 us
// of products) -- the loaders cannot deliver that many bytes under the products.
// ----------------------------------------------------------------------------------------------
namespace pd {
#ifndef TGRU_STEP_FWD_LOADERS
#define TGRU_STEP_FWD_LOADERS 2
#endif
#ifndef TGRU_STEP_FWD_BK
#define TGRU_STEP_FWD_BK 32
#endif
#ifndef TGRU_STEP_FWD_SLOTS
#define TGRU_STEP_FWD_SLOTS 4
#endif
// (k-tile 32 x four slots x two loader waves; 64 x two slots, 32 x three slots and four loader waves measured the
// same forward time within half a percent)
constexpr int kLoaders = TGRU_STEP_FWD_LOADERS;
constexpr int kFwdBK = TGRU_STEP_FWD_BK, kFwdSlots = TGRU_STEP_FWD_SLOTS;
constexpr int kThreads = 256 + 64 * kLoaders;
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// position (in 16-byte pieces) of piece c of row r inside its row of BK floats
template <int BK>
__device__ __forceinline__ int rot(int c, int r) { return (c + r * (BK / 4) / 16) & (BK / 4 - 1); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float *lds, int voff, int soff, int aux) {
    if (aux) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, voff, soff, 0, kAuxSc1);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, voff, soff, 0, 0);
}
// one lane of a loader wave waits until `*cnt` has reached `want` (bounded, see poll_peers)
__device__ __forceinline__ void loader_wait_for_peers(unsigned *cnt, unsigned want, unsigned *err, int limit, int lane) {
    if (lane == 0) poll_peers(cnt, want, err, limit);
}
}  // namespace pd

template <int BK>      // k-tile (32)
__global__ __launch_bounds__(pd::kThreads) void gru_persist_fwd_dma_kernel(PersistFwd s) {
    using namespace pd;
    constexpr int BN = 3 * kUnits, Q = BK / 4, LDH = kUnits + 4;
    constexpr int kSlots = kFwdSlots;                       // the tile being read, those in flight, one being refilled
    constexpr int kAhead = kSlots - 1, kFly = kAhead - 1;   // tiles issued ahead / still in flight when one is awaited
    constexpr int kATile = kRows * BK, kBTile = BN * BK, kSlot = kATile + kBTile;      // floats
    constexpr int kRowsPerInst = 256 / BK;                  // tile rows one load instruction covers (1 KiB)
    constexpr int kAPer = kATile / 256 / kLoaders, kBPer = kBTile / 256 / kLoaders;    // instructions per loader wave
    constexpr int kPerTile = kAPer + kBPer;
    static_assert(kATile % (256 * kLoaders) == 0 && kBTile % (256 * kLoaders) == 0 && kFly * kPerTile < 64, "tile split");
    __shared__ __attribute__((aligned(1024))) float lds[kSlots * kSlot + kRows * LDH];
    float (*Hs)[LDH] = reinterpret_cast<float (*)[LDH]>(lds + kSlots * kSlot);
    __shared__ int Ls[kRows];      // length of each sequence of the row tile (0 for rows past the batch)
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const FwdArgs &a = s.f;
    const int H = a.H, B = a.B, T = a.T, NT = H / kUnits;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = mt * kRows, u0 = nt * kUnits;
    unsigned *cnt = s.sync + 1 + mt;
    const long step_bytes = (long)B * H * 4;
    const int nkt = H / BK;
    if (tid < kRows) Ls[tid] = m0 + tid < B ? (a.lengths ? a.lengths[m0 + tid] : T) : 0;
    __syncthreads();

    if (wave >= 4) {
        // ---- loader waves ------------------------------------------------------------------------------
        const int l = wave - 4;
        int va[kAPer], vb[kBPer];
#pragma unroll
        for (int i = 0; i < kAPer; ++i) {
            const int r = (l * kAPer + i) * kRowsPerInst + lane / Q, p = lane % Q;
            const int c = (p - r * Q / 16) & (Q - 1);
            va[i] = (min(m0 + r, B - 1) * H + 4 * c) * 4;              // bytes inside one step's (B, H) block
        }
#pragma unroll
        for (int i = 0; i < kBPer; ++i) {
            const int r = (l * kBPer + i) * kRowsPerInst + lane / Q, p = lane % Q;
            const int c = (p - r * Q / 16) & (Q - 1);
            vb[i] = (((r >> 5) * H + u0 + (r & 31)) * H + 4 * c) * 4;   // rows of W_hh: gate * H + unit
        }
        const __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.w_hh, (long)3 * H * H * 4);
        for (int t = 0; t < T; ++t) {
            if (t > 0) loader_wait_for_peers(cnt, (unsigned)(NT * t), s.err, s.spin_limit, lane);
            if (t > 0 || a.h0 != nullptr) {
                const __amdgpu_buffer_rsrc_t rs_h = rsrc_of(t > 0 ? s.ybuf + (long)(t - 1) * B * H : a.h0, step_bytes);
                auto issue = [&](int kt) {
                    float *slot = lds + (kt % kSlots) * kSlot;
#pragma unroll
                    for (int i = 0; i < kAPer; ++i) dma16(rs_h, slot + (l * kAPer + i) * 256, va[i], kt * BK * 4, 1);
#pragma unroll
                    for (int i = 0; i < kBPer; ++i) dma16(rs_w, slot + kATile + (l * kBPer + i) * 256, vb[i], kt * BK * 4, 0);
                };
#pragma unroll
                for (int d = 0; d < kAhead; ++d)
                    if (d < nkt) issue(d);
                for (int kt = 0; kt < nkt; ++kt) {
                    const int fly = min(kFly, nkt - 1 - kt);         // younger tiles that may stay in flight
                    if (kFly >= 2 && fly == 2) wait_vmcnt<(kFly >= 2 ? 2 : 0) * kPerTile>();
                    else if (kFly >= 1 && fly >= 1) wait_vmcnt<(kFly >= 1 ? 1 : 0) * kPerTile>();
                    else wait_vmcnt<0>();
                    lds_barrier();                   // tile kt may be read; everybody is done with tile kt - 1
                    if (kt + kAhead < nkt) issue(kt + kAhead);
                }
            }
            lds_barrier();                           // (the product waves' state tile is in LDS)
            lds_barrier();                           // (their stores are done: arrival)
        }
        return;
    }

    // ---- product waves -----------------------------------------------------------------------------------
    // (A wave of this workgroup may use 256 registers; the accumulators stay in the vector half, because asking for
    // the accumulator half would leave 128 vector registers and spill.  Every fragment read here is a 16-byte read,
    // for which the two forms run alike.)
    const int lrow = lane & 31, lk = lane >> 5;
    const int u = u0 + lrow;
    const long plane = (long)T * B * H;
    unsigned valid = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) valid |= (m0 + 32 * wave + d_row(r, lk) < B ? 1u : 0u) << r;
    int roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) roff[r] = (min(m0 + 32 * wave + d_row(r, lk), B - 1) * H + u) * 4;
    const float b_r = a.b_hh[u], b_z = a.b_hh[H + u], b_n = a.b_hh[2 * H + u];
    float hp[16];      // this workgroup's own tile of the state entering the step
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        hp[r] = a.h0 ? a.h0[(long)mc * H + u] : 0.f;
    }
    // float offsets of this lane's fragment rows inside a slot (piece position added per chunk)
    const int ma = 32 * wave + lrow;
    const int fa = ma * BK, fb = kATile + lrow * BK;
    PSTAMP_DECL

    for (int t = 0; t < T; ++t) {
        PSTAMP(7);
        float gir[16], giz[16], gin[16];
        {
            const __amdgpu_buffer_rsrc_t rs_gi = rsrc_of(a.gi + (long)t * B * 3 * H, 3 * step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {      // row m of gi is 3H wide: 3 * (m * H) + u
                const int o = 3 * roff[r] - 8 * u;
                gir[r] = ld1(rs_gi, o, 0);
                giz[r] = ld1(rs_gi, o, H * 4);
                gin[r] = ld1(rs_gi, o, 2 * H * 4);
            }
        }
        f32x16 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

        PSTAMP(0);
        if (t > 0 || a.h0 != nullptr) {
            for (int kt = 0; kt < nkt; ++kt) {
                lds_barrier();
#ifdef TGRU_STAMPS
                if (kt == 0) PSTAMP(1);      // wait for the peers and the first tile
#endif
                const float *slot = lds + (kt % kSlots) * kSlot;
#pragma unroll
                for (int q = 0; q < BK / 8; ++q) {      // chunk q: eight k-steps, four per lane half
                    const int c = 2 * q + lk;
                    const float4 av = *reinterpret_cast<const float4 *>(slot + fa + 4 * rot<BK>(c, ma));
                    // (rows lrow, 32 + lrow, 64 + lrow of the W_hh tile share one rotation: 32 rows are a whole turn)
                    const float *bp = slot + fb + 4 * rot<BK>(c, lrow);
                    const float4 b0 = *reinterpret_cast<const float4 *>(bp);
                    const float4 b1 = *reinterpret_cast<const float4 *>(bp + 32 * BK);
                    const float4 b2 = *reinterpret_cast<const float4 *>(bp + 64 * BK);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc[2], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc[2], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2.z, acc[2], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b2.w, acc[2], 0, 0, 0);
                }
            }
        }

        // gates and the new state of this tile; the state goes to LDS first so that it can leave as whole
        // 128-byte lines (one 16-byte store per lane)
        PSTAMP(3);
        // The saved gates leave with the state: holding them in registers until after the arrival would cost
        // sixty-four registers of the 256 a wave of this workgroup may use.
        {
            float *gt = a.gates ? a.gates + (long)t * B * H : nullptr;      // (absent tensors: empty descriptors)
            const __amdgpu_buffer_rsrc_t rs_r = rsrc_opt(gt, step_bytes), rs_z = rsrc_opt(gt ? gt + plane : nullptr, step_bytes),
                                         rs_n = rsrc_opt(gt ? gt + 2 * plane : nullptr, step_bytes),
                                         rs_h = rsrc_opt(gt ? gt + 3 * plane : nullptr, step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int len = Ls[32 * wave + d_row(r, lk)];      // 0 for rows past the batch
                const bool active = t < len;
                const float ghn = acc[2][r] + b_n;
                const float rg = fast_sigmoid(gir[r] + acc[0][r] + b_r);
                const float zg = fast_sigmoid(giz[r] + acc[1][r] + b_z);
                const float ng = fast_tanh(gin[r] + rg * ghn);
                const float hnew = (1.f - zg) * ng + zg * hp[r];
                const float yv = active ? hnew : 0.f;
                hp[r] = yv;
                Hs[32 * wave + d_row(r, lk)][lrow] = yv;
                const int og = active ? roff[r] : 0x7fffffff;       // an offset beyond the block drops a lane's store
                st1(rs_r, og, 0, rg); st1(rs_z, og, 0, zg); st1(rs_n, og, 0, ng); st1(rs_h, og, 0, ghn);
            }
        }
        lds_barrier();
        {
            const __amdgpu_buffer_rsrc_t rs_y = rsrc_of(s.ybuf + (long)t * B * H, step_bytes);
#pragma unroll
            for (int p = 0; p < kRows * (kUnits / 4) / 256; ++p) {
                const int slot = tid + p * 256, row = slot >> 3, c4 = (slot & 7) << 2;
                const float4 v = *reinterpret_cast<const float4 *>(&Hs[row][c4]);
                // rows past the batch: an offset beyond the buffer, the store is dropped
                const int off = m0 + row < B ? ((m0 + row) * H + u0 + c4) * 4 : 0x7fffffff;
                st4_sc1(rs_y, off, v);
            }
        }
        PSTAMP(4);
        // every store of this workgroup issued so far is complete; then one lane tells the row tile's counter
        // (the barrier also keeps Hs from being rewritten before every wave has read it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (t + 1 < T && tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PSTAMP(5);
        // what no workgroup waits for: the caller's y (when it is not the exchange buffer), the saved
        // gates and h_n.  These stores drain while the next step's products run.
        if (a.y_ext && a.y_ext != s.ybuf) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + d_row(r, lk);
                const long er = ((valid >> r) & 1) ? map_tb(a.ymap, t, m) : -1L;
                if (er >= 0) a.y_ext[er * H + u] = hp[r];
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rs_hn = rsrc_opt(a.h_n, step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st1(rs_hn, t == Ls[32 * wave + d_row(r, lk)] - 1 ? roff[r] : 0x7fffffff, 0, hp[r]);
        }
        PSTAMP(6);
    }
    PSTAMP_FLUSH(0)
}



// ----------------------------------------------------------------------------------------------
// The loader-wave forward kernel WITH THE INPUT PROJECTION INSIDE.  Between the moment a workgroup has stored
// its tile of h_t and the moment it may start on step t+1 lie the drain of its stores, the arrival, its
// peers' arrivals and the first fetch of what they stored: about 9 us of the 31 a step takes, with the matrix
// pipe idle.  The input projection of step t+1, x_{t+1} . W_ih^T for the same tile, depends on none of that
// (the layer's input is complete before the launch), has the very shape of the recurrent product (128 rows x
// 96 columns, k-tiles of W_ih in place of W_hh, rows of x in place of rows of h) and the same cost.  So it
// runs right there: step t = recurrent product (r and z accumulate ON TOP of the projection's sums, the n gate
// keeps the two parts apart) -> gates, state tile, stores -> first half of the projection of step t+1 -> the
// stores have long drained: arrival -> second half of the projection, under which the peers' arrivals and
// the hand-off disappear.  The batched input-projection GEMM of the layer and the (T*B, 3H) buffer it wrote
// and the recurrent kernel read back are gone.  Needs the layer input time-major or batch-major with a width
// that is a multiple of the k-tile.
//
// Loader waves and product waves walk ONE sequence of tiles -- P(0), then for every step R(t) [absent at t = 0
// without an initial state], P(t+1) [absent after the last step] -- through the same ring; the two barriers of a
// step that are not tile barriers (state tile in LDS; arrival) sit after R(t) and in the middle of P(t+1).  A
// tile of R(t+1) may only be requested once this workgroup itself has arrived for step t (the loader polls for
// ALL workgroups of the row tile, its own included), so the loader's run-ahead stops at that barrier and is
// made up behind it.
// ----------------------------------------------------------------------------------------------
template <int BK>      // k-tile (32)
__global__ __launch_bounds__(pd::kThreads) void gru_persist_fwd_fused_kernel(PersistFwd s) {
    using namespace pd;
    constexpr int BN = 3 * kUnits, Q = BK / 4, LDH = kUnits + 4;
    constexpr int kSlots = kFwdSlots;
    constexpr int kAhead = kSlots - 1, kFly = kAhead - 1;
    constexpr int kATile = kRows * BK, kBTile = BN * BK, kSlot = kATile + kBTile;      // floats
    constexpr int kRowsPerInst = 256 / BK;
    constexpr int kAPer = kATile / 256 / kLoaders, kBPer = kBTile / 256 / kLoaders;
    constexpr int kPerTile = kAPer + kBPer;
    static_assert(kATile % (256 * kLoaders) == 0 && kBTile % (256 * kLoaders) == 0 && kFly * kPerTile < 64, "tile split");
    static_assert(kFly == 2, "the vmcnt switch below is written for two younger tiles in flight");
    __shared__ __attribute__((aligned(1024))) float lds[kSlots * kSlot + kRows * LDH];
    float (*Hs)[LDH] = reinterpret_cast<float (*)[LDH]>(lds + kSlots * kSlot);
    __shared__ int Ls[kRows];
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const FwdArgs &a = s.f;
    const int H = a.H, B = a.B, T = a.T, In = a.In, NT = H / kUnits;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = mt * kRows, u0 = nt * kUnits;
    unsigned *cnt = s.sync + 1 + mt;
    const long step_bytes = (long)B * H * 4;
    const int nR = H / BK, nP = In / BK, nP1 = nP / 2;      // tiles of a recurrent product / a projection (its first half)
    const bool h0 = a.h0 != nullptr;
    if (tid < kRows) Ls[tid] = m0 + tid < B ? (a.lengths ? a.lengths[m0 + tid] : T) : 0;
    __syncthreads();

    if (wave >= 4) {
        // ---- loader waves ------------------------------------------------------------------------------
        const int l = wave - 4;
        const int x_bt = a.xmap.mode == 1;                     // layer input batch-major: row b*T + t
        int va[kAPer], vx[kAPer], vb[kBPer], vi[kBPer];
#pragma unroll
        for (int i = 0; i < kAPer; ++i) {
            const int r = (l * kAPer + i) * kRowsPerInst + lane / Q, p = lane % Q;
            const int c = (p - r * Q / 16) & (Q - 1);
            const int mc = min(m0 + r, B - 1);
            va[i] = (mc * H + 4 * c) * 4;                                   // bytes inside one step's (B, H) block
            vx[i] = ((x_bt ? mc * T : mc) * In + 4 * c) * 4;                // ... inside the input rows of one step
        }
#pragma unroll
        for (int i = 0; i < kBPer; ++i) {
            const int r = (l * kBPer + i) * kRowsPerInst + lane / Q, p = lane % Q;
            const int c = (p - r * Q / 16) & (Q - 1);
            vb[i] = (((r >> 5) * H + u0 + (r & 31)) * H + 4 * c) * 4;       // rows of W_hh: gate * H + unit
            vi[i] = (((r >> 5) * H + u0 + (r & 31)) * In + 4 * c) * 4;      // rows of W_ih
        }
        const __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.w_hh, (long)3 * H * H * 4);
        const __amdgpu_buffer_rsrc_t rs_wi = rsrc_of(a.w_ih, (long)3 * H * In * 4);
        const long x_step = x_bt ? (long)In : (long)B * In;                 // floats from one step's input rows to the next
        const long x_bytes = x_bt ? ((long)(B - 1) * T + 1) * In * 4 : (long)B * In * 4;
        // the tile sequence as the requests walk it: (step, recurrent or projection, k-tile)
        int it_t = 0, it_kt = 0;
        bool it_rec = false;                                                 // starts with P(0)
        int issued = 0, consumed = 0, arrived = -1;                          // arrived: last step this workgroup arrived for
        bool more = true;
        auto issue_one = [&]() -> bool {       // request the next tile of the sequence, unless it has to wait
            if (!more) return false;
            if (it_rec && it_t - 1 > arrived) return false;                 // R(t) needs this workgroup's arrival for t-1
            float *slot = lds + (issued % kSlots) * kSlot;
            if (it_rec) {
                if (it_kt == 0 && it_t > 0) loader_wait_for_peers(cnt, (unsigned)(NT * it_t), s.err, s.spin_limit, lane);
                const __amdgpu_buffer_rsrc_t rs_h = rsrc_of(it_t > 0 ? s.ybuf + (long)(it_t - 1) * B * H : a.h0, step_bytes);
#pragma unroll
                for (int i = 0; i < kAPer; ++i) dma16(rs_h, slot + (l * kAPer + i) * 256, va[i], it_kt * BK * 4, 1);
#pragma unroll
                for (int i = 0; i < kBPer; ++i) dma16(rs_w, slot + kATile + (l * kBPer + i) * 256, vb[i], it_kt * BK * 4, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rs_x = rsrc_of(a.x + it_t * x_step, x_bytes);
#pragma unroll
                for (int i = 0; i < kAPer; ++i) dma16(rs_x, slot + (l * kAPer + i) * 256, vx[i], it_kt * BK * 4, 0);
#pragma unroll
                for (int i = 0; i < kBPer; ++i) dma16(rs_wi, slot + kATile + (l * kBPer + i) * 256, vi[i], it_kt * BK * 4, 0);
            }
            ++issued;
            // advance: P(t) -> R(t) [or straight to P(t+1) at t = 0 without an initial state] -> P(t+1) -> ...
            if (++it_kt == (it_rec ? nR : nP)) {
                it_kt = 0;
                if (it_rec) {                      // R(t) done: P(t+1) follows unless t was the last step
                    if (it_t + 1 < T) { it_rec = false; ++it_t; }
                    else more = false;
                } else if (it_t > 0 || h0) {       // P(t) done: R(t)
                    it_rec = true;
                } else if (T > 1) {                // P(0) without a state: gates of step 0 need no product, P(1) next
                    it_t = 1;
                } else {
                    more = false;
                }
            }
            return true;
        };
        auto consume = [&](int n) {             // the barriers of n tiles, each after its tile has landed
            for (int i = 0; i < n; ++i) {
                const int younger = issued - consumed - 1;
                if (younger >= 2) wait_vmcnt<2 * kPerTile>();
                else if (younger == 1) wait_vmcnt<kPerTile>();
                else wait_vmcnt<0>();
                lds_barrier();                   // tile `consumed` may be read; everybody is done with the one before
                ++consumed;
                while (issued < consumed + kAhead && issue_one()) {}
            }
        };
        while (issued < kAhead && issue_one()) {}
        consume(nP);                                                         // P(0)
        for (int t = 0; t < T; ++t) {
            if (t > 0 || h0) consume(nR);                                    // R(t)
            lds_barrier();                                                   // (the product waves' state tile is in LDS)
            if (t + 1 < T) consume(nP1);
            lds_barrier();                                                   // (their stores are done: arrival)
            arrived = t;
            while (issued < consumed + kAhead && issue_one()) {}             // what had to wait for the arrival
            if (t + 1 < T) consume(nP - nP1);
        }
        return;
    }

    // ---- product waves -----------------------------------------------------------------------------------
    const int lrow = lane & 31, lk = lane >> 5;
    const int u = u0 + lrow;
    const long plane = (long)T * B * H;
    unsigned valid = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) valid |= (m0 + 32 * wave + d_row(r, lk) < B ? 1u : 0u) << r;
    int roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) roff[r] = (min(m0 + 32 * wave + d_row(r, lk), B - 1) * H + u) * 4;
    // r and z see the sum of both biases; the n gate keeps b_in with the projection and b_hn with the recurrent part
    const float b_r = a.b_ih[u] + a.b_hh[u], b_z = a.b_ih[H + u] + a.b_hh[H + u];
    const float b_nx = a.b_ih[2 * H + u], b_nh = a.b_hh[2 * H + u];
    float hp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + 32 * wave + d_row(r, lk), B - 1);
        hp[r] = a.h0 ? a.h0[(long)mc * H + u] : 0.f;
    }
    const int ma = 32 * wave + lrow;
    const int fa = ma * BK, fb = kATile + lrow * BK;
    int g = 0;                                   // tiles consumed so far: the ring slot of the next one
    f32x16 acc_r, acc_z, acc_nx, acc_nh;
    // one k-tile of a product: r and z always, the n part into `acc_n`
    auto tile = [&](f32x16 &acc_n) {
        lds_barrier();
        const float *slot = lds + (g % kSlots) * kSlot;
        ++g;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int c = 2 * q + lk;
            const float4 av = *reinterpret_cast<const float4 *>(slot + fa + 4 * rot<BK>(c, ma));
            const float *bp = slot + fb + 4 * rot<BK>(c, lrow);
            const float4 b0 = *reinterpret_cast<const float4 *>(bp);
            const float4 b1 = *reinterpret_cast<const float4 *>(bp + 32 * BK);
            const float4 b2 = *reinterpret_cast<const float4 *>(bp + 64 * BK);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc_n, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc_n, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2.z, acc_n, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b2.w, acc_n, 0, 0, 0);
        }
    };
    auto clear_projection = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_nx[r] = 0.f; }
    };
    clear_projection();
    for (int kt = 0; kt < nP; ++kt) tile(acc_nx);                           // P(0)

    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_nh[r] = 0.f;
        if (t > 0 || h0)
            for (int kt = 0; kt < nR; ++kt) tile(acc_nh);                   // R(t), on top of P(t) for r and z

        float ynew[16];
        {
            float *gt = a.gates ? a.gates + (long)t * B * H : nullptr;      // (absent tensors: empty descriptors)
            const __amdgpu_buffer_rsrc_t rs_r = rsrc_opt(gt, step_bytes), rs_z = rsrc_opt(gt ? gt + plane : nullptr, step_bytes),
                                         rs_n = rsrc_opt(gt ? gt + 2 * plane : nullptr, step_bytes),
                                         rs_h = rsrc_opt(gt ? gt + 3 * plane : nullptr, step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int len = Ls[32 * wave + d_row(r, lk)];      // 0 for rows past the batch
                const bool active = t < len;
                const float ghn = acc_nh[r] + b_nh;
                const float rg = fast_sigmoid(acc_r[r] + b_r);
                const float zg = fast_sigmoid(acc_z[r] + b_z);
                const float ng = fast_tanh(acc_nx[r] + b_nx + rg * ghn);
                const float hnew = (1.f - zg) * ng + zg * hp[r];
                const float yv = active ? hnew : 0.f;
                hp[r] = yv;
                ynew[r] = yv;
                Hs[32 * wave + d_row(r, lk)][lrow] = yv;
                const int og = active ? roff[r] : 0x7fffffff;       // an offset beyond the block drops a lane's store
                st1(rs_r, og, 0, rg); st1(rs_z, og, 0, zg); st1(rs_n, og, 0, ng); st1(rs_h, og, 0, ghn);
            }
        }
        lds_barrier();
        {
            const __amdgpu_buffer_rsrc_t rs_y = rsrc_of(s.ybuf + (long)t * B * H, step_bytes);
#pragma unroll
            for (int p = 0; p < kRows * (kUnits / 4) / 256; ++p) {
                const int slot = tid + p * 256, row = slot >> 3, c4 = (slot & 7) << 2;
                const float4 v = *reinterpret_cast<const float4 *>(&Hs[row][c4]);
                const int off = m0 + row < B ? ((m0 + row) * H + u0 + c4) * 4 : 0x7fffffff;
                st4_sc1(rs_y, off, v);
            }
        }
        // the caller's y (when it is not the exchange buffer) and h_n: nobody waits for these
        if (a.y_ext && a.y_ext != s.ybuf) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + d_row(r, lk);
                const long er = ((valid >> r) & 1) ? map_tb(a.ymap, t, m) : -1L;
                if (er >= 0) a.y_ext[er * H + u] = ynew[r];
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rs_hn = rsrc_opt(a.h_n, step_bytes);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st1(rs_hn, t == Ls[32 * wave + d_row(r, lk)] - 1 ? roff[r] : 0x7fffffff, 0, ynew[r]);
        }
        const bool next = t + 1 < T;
        if (next) {
            clear_projection();
            for (int kt = 0; kt < nP1; ++kt) tile(acc_nx);                  // first half of P(t+1)
        }
        // every store of this workgroup issued so far is complete (they had half a projection to drain); then
        // one lane tells the row tile's counter (the barrier also keeps Hs from being rewritten too early)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (next && tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (next)
            for (int kt = nP1; kt < nP; ++kt) tile(acc_nx);                 // second half, covering the hand-off
    }
}


// ----------------------------------------------------------------------------------------------
// Backward recurrence with W_hh RESIDENT in the workgroup (H a multiple of 128).
//
// A step of the 64 x 64 kernel above is bound by the bytes its CU takes from the L2, 0.8 MB of operand tiles for
// 12.6 MFLOP, half of them the workgroup's 3H x 64 tile of W_hh -- the same tile at every step.  The tile is
// 3H x 64 floats (384 KiB at H = 512): more than LDS, less than LDS plus the register file of four waves that have
// a SIMD each (512 registers per lane).  So each wave keeps a QUARTER OF THE K RANGE of the tile for good -- the
// MFMA B operands of its 3H/4 values of k, most of them in registers, the rest in LDS -- and multiplies the whole
// 64-unit width with it:
//   * the waves of a workgroup split k, not the output: a wave needs columns [w 3H/4, (w+1) 3H/4) of the row tile's
//     dg rows and nothing else, so its operand tiles are private to it.  It fetches them itself with
//     direct-to-LDS loads into its own ring of slots and reads them back after a count of its own loads: no
//     barrier inside the product, no register staging, and only the dg rows (half the bytes) cross from the L2;
//   * the four partial 32 x 64 sums meet in LDS once per product; each wave sums -- in wave order -- the eight rows
//     it then owns for the gate arithmetic;
//   * the 64 sequences of a row tile are two independent chains of 32 (a sequence only ever needs its own row of
//     dg), each with its own arrival counter, worked alternately: while the workgroups of a row tile drain their
//     stores of one half, arrive and wait for each other, every one of them multiplies the other half.  With
//     W_hh resident the second product costs no second pass over the matrix.
// Needs every workgroup of the launch resident, like the other persistent kernels; every wait is bounded.
// ----------------------------------------------------------------------------------------------
namespace res {
#ifndef TGRU_RES_SLOTS
#define TGRU_RES_SLOTS 4
#endif
#ifndef TGRU_RES_KS
#define TGRU_RES_KS 16
#endif
#ifndef TGRU_RES_NCL512
#define TGRU_RES_NCL512 12
#endif
constexpr int kSlots = TGRU_RES_SLOTS;      // ring slots per wave: one being read, the others in flight
constexpr int KS = TGRU_RES_KS;             // k per slot: 32 rows x 16 k = 2 KiB = two load instructions of a wave (or 32 k: four)
constexpr int kNcl512 = TGRU_RES_NCL512;    // chunks of W_hh a wave keeps in LDS at H = 512 (what LDS has room for)
__device__ __forceinline__ void wait_loads(int n) {      // (n is a constant once the stage loop is unrolled)
    switch (n) {
    case 24: pd::wait_vmcnt<24>(); break;
    case 20: pd::wait_vmcnt<20>(); break;
    case 16: pd::wait_vmcnt<16>(); break;
    case 12: pd::wait_vmcnt<12>(); break;
    case 10: pd::wait_vmcnt<10>(); break;
    case 8: pd::wait_vmcnt<8>(); break;
    case 6: pd::wait_vmcnt<6>(); break;
    case 4: pd::wait_vmcnt<4>(); break;
    case 2: pd::wait_vmcnt<2>(); break;
    default: pd::wait_vmcnt<0>(); break;      // (0, or a count this switch does not know: wait for everything)
    }
}
// (write-through stores of 16 bytes: a 4-byte one costs six times as much per byte)
__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, int sbase, float4 f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, sbase, kAuxSc1);
}
__device__ __forceinline__ float4 ld4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, TGRU_STREAM_AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// A direct-to-LDS load the compiler does not know to be one.  A wave of this kernel both requests its operand
// tiles and reads them back, and the compiler makes every LDS read that follows a direct-to-LDS load it knows of
// wait for ALL vector-memory loads in flight (it cannot tell which LDS bytes the load writes): the ring would
// never hold more than the tile being waited for.  Written as an instruction the compiler does not look into,
// the load is ordered against the LDS reads by this kernel's own counts (wait_loads) and compiler barriers.
// The descriptor is the four words make_buffer_rsrc would build (raw buffer, 32-bit elements).
__device__ __forceinline__ u32x4 desc_words(const void *p, long bytes) {
    const unsigned long v = reinterpret_cast<unsigned long>(p);
    u32x4 d;
    d.x = (unsigned)v; d.y = (unsigned)(v >> 32) & 0xffffu; d.z = (unsigned)bytes; d.w = 0x00020000u;
    return d;
}
__device__ __forceinline__ void dma16_sc1(u32x4 rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
}  // namespace res

template <int H, int NCL>     // NCL: how many of a wave's 3H/32 chunks (eight k) of W_hh live in LDS, not registers
__global__ __launch_bounds__(256) void gru_persist_bwd_res_kernel(PersistBwd s) {
    using namespace pd;
    constexpr int kR = 64, kU = 64;
    constexpr int KW = 3 * H / 4, NC = KW / 8, NCR = NC - NCL;          // a wave's k range, in chunks of eight
    constexpr int KS = res::KS, NST = KW / KS, CPS = KS / 8, NS = res::kSlots;
    constexpr int kSlotF = 32 * KS, IPS = kSlotF / 256, Q = KS / 4;      // a slot in floats, its load instructions, pieces per row
    static_assert((KS == 16 || KS == 32) && IPS * (NS - 1) <= 12, "stage");
    static_assert(H % 128 == 0 && KW % KS == 0 && NCL >= 0 && NCL < NC && NS >= 2 && NS <= 6 && NST > NS, "shape");
    constexpr int kRingF = 4 * NS * kSlotF, kWlF = 4 * NCL * 2 * 256, kXF = 16 * 2 * 256;       // floats
    static_assert((kRingF + kWlF + kXF) * 4 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) float lds[kRingF + kWlF + kXF];
    int mt, nt;
    if (!tile_of(s.MT, s.mtx, mt, nt)) return;
    mt += s.mt0;
    const BwdArgs &a = s.b;
    const int B = a.B, T = a.T, NT = H / kU;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const int m0 = mt * kR, u0 = nt * kU;
    unsigned *cnt = s.sync + 1 + 2 * mt;           // [half]
    const long plane = (long)T * B * H;
    const long step_bytes = (long)B * 4 * H * 4, bh_bytes = (long)B * H * 4;
    float *ring = lds + wave * (NS * kSlotF);
    const unsigned ring_addr = (unsigned)(unsigned long)(lds_void *)ring;      // (LDS byte address, same in every lane)
    const float4 *Wl = reinterpret_cast<const float4 *>(lds + kRingF) + wave * (NCL * 2 * 64) + lane;
    float4 *X = reinterpret_cast<float4 *>(lds + kRingF + kWlF);        // [owner wave][source wave][unit block][lane]
    const float *Xf = lds + kRingF + kWlF;

    // ---- this wave's part of W_hh: MFMA step j of chunk c stands for k = w 3H/4 + 8c + 4 (lane >> 5) + j ----------
    float wreg[NCR][2][4];
    {
        const __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.w_hh, (long)3 * H * H * 4);
        const int wbase = ((wave * KW + 4 * lk) * H + u0 + lrow) * 4;
        float4 *wl = reinterpret_cast<float4 *>(lds + kRingF) + wave * (NCL * 2 * 64) + lane;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ld1(rs_w, wbase + ((8 * c + j) * H + 32 * ub) * 4, 0);
                if (c < NCR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wreg[c < NCR ? c : 0][ub][j] = v[j];
                } else {
                    wl[((c - NCR) * 2 + ub) * 64] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
    __syncthreads();

    // ---- what this thread owns once the partial sums have met: of each half, rows tid/16 and 16 + tid/16, units
    //      4 (tid % 16) .. + 3 of the tile (16-byte loads and stores, 256 contiguous bytes per 16 lanes) -----------
    const int cg = tid & 15, er = tid >> 4;
    int rowoff[2][2], len[2][2], va[2][IPS], xoff[2];
    unsigned inside = 0;      // bit 2h + i: the row is inside the batch
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * h + 16 * i + er;
            rowoff[h][i] = min(m, B - 1) * H * 4;                                // bytes inside one step's (B, H) block
            inside |= (m < B ? 1u : 0u) << (2 * h + i);
            len[h][i] = m < B ? (a.lengths ? a.lengths[m] : T) : 0;              // (0: a row past the batch is never active)
        }
#pragma unroll
        for (int i = 0; i < IPS; ++i) {      // load instruction i of a slot: 64 / Q rows of Q 16-byte pieces each
            const int r = (64 / Q) * i + lane / Q, p = lane % Q;
            const int c = (p - r * Q / 16) & (Q - 1);                            // the piece that belongs at LDS position p
            va[h][i] = (min(m0 + 32 * h + r, B - 1) * 4 * H + 4 * c) * 4;
        }
    }
    // where the partial sums of (row 16 i + er, unit 4 cg + q) lie in X: owner wave row / 8, unit block cg / 8,
    // writer lane 32 ((row / 4) & 1) + unit % 32, component row % 4 -- as a float index, without the source wave's
    // part and without q.  A writer lane's 16-byte slot is its lane number with the two low bits flipped by
    // bit 4 of the lane and by the unit block (xq below, seen from the reading side): the 64 lanes of one read
    // instruction -- eight unit groups x two unit blocks x four rows -- then touch 64 different banks (read
    // straight, they are 4-way conflicts: 0.5 us per slice).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + er;
        xoff[i] = ((((row >> 3) * 4) * 2 + (cg >> 3)) * 64 + 32 * ((row >> 2) & 1) + 4 * (cg & 7)) * 4 + (row & 3);
    }
    const int xq = ((cg >> 2) & 1) ^ (2 * (cg >> 3));      // unit q of this thread's group sits in slot q ^ xq
    const int ucol = (u0 + 4 * cg) * 4;
    const int dy_bt = a.dymap.mode == 1;
    const bool dy_ext = a.dy_ext != nullptr;      // (a call with both dy tensors goes to the 128 x 32 kernel)
    // d_{t+1} * z_{t+1}; starts as the gradient of the final state (steps past a sequence's end pass it on unchanged)
    float4 cin[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            cin[h][i] = a.dh_n ? *reinterpret_cast<const float4 *>(a.dh_n + (long)(rowoff[h][i] >> 2) + u0 + 4 * cg)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sums[4];      // column sums of the four gate gradients over this thread's rows, all steps: [gate].unit
#pragma unroll
    for (int q = 0; q < 4; ++q) sums[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int t_last = a.dh0 ? -1 : 0;
    PSTAMP_DECL

    // The work goes slice by slice -- (T-1, half 0), (T-1, half 1), (T-2, half 0), ... -- and the END of a slice
    // prepares the next one, between the moment its own stores are issued and the moment they have drained:
    // the next slice's peers are looked up (their arrival is a slice old: the counter was read under this
    // slice's products) and its first operand tiles are requested; then ONE counted wait covers the stores of this
    // slice (vector-memory operations complete in issue order), and the arrival for this slice follows.  The
    // operands of a slice's gate arithmetic are requested when its product ends and arrive while the partial sums
    // meet: held through the product they would cost 48 of the registers W_hh lives in.
    float4 gr[2], gz[2], gn[2], ghn[2], hp[2], dy[2];      // operands of the gate arithmetic of this slice
    auto load_operands = [&](int tt, auto hh) {
        constexpr int h = decltype(hh)::value;
        const int tc = max(tt, 0);
        const float *gt = a.gates + (long)tc * B * H;
        const __amdgpu_buffer_rsrc_t rs_r = rsrc_of(gt, bh_bytes), rs_z = rsrc_of(gt + plane, bh_bytes),
                                     rs_n = rsrc_of(gt + 2 * plane, bh_bytes), rs_h = rsrc_of(gt + 3 * plane, bh_bytes);
        const __amdgpu_buffer_rsrc_t rs_hp = rsrc_opt(tc > 0 ? a.y_int + (long)(tc - 1) * B * H : a.h0, bh_bytes);
        // the gradient of this layer's output: the caller's dy (top layer; batch-major: the block of step tc starts
        // tc rows in and its sequences are T rows apart) or the layer above's input gradient
        const bool bt = dy_ext && dy_bt;
        const __amdgpu_buffer_rsrc_t rs_d =
            dy_ext ? rsrc_of(a.dy_ext + (bt ? (long)tc * H : (long)tc * B * H), bt ? ((long)(B - 1) * T + 1) * H * 4 : bh_bytes)
                   : rsrc_opt(a.dy_int ? a.dy_int + (long)tc * B * H : nullptr, bh_bytes);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ro = rowoff[h][i], o = ro + ucol;
            gr[i] = res::ld4(rs_r, o); gz[i] = res::ld4(rs_z, o);
            gn[i] = res::ld4(rs_n, o); ghn[i] = res::ld4(rs_h, o);
            hp[i] = res::ld4(rs_hp, o);
            dy[i] = res::ld4(rs_d, bt ? ro * T + ucol : o);
        }
    };
    // request stage st of the product of (step tt, half h): A = (da_r | da_z | da_n*r) of step tt+1, columns [0,2H)
    // and [3H,4H) of the dg rows of the half
    auto issue = [&](int tt, auto hh, int st) {
        constexpr int h = decltype(hh)::value;
        const u32x4 rs_dg = res::desc_words(a.dg + (long)(tt + 1) * B * 4 * H, step_bytes);
        const int k0 = wave * KW + st * KS;
        const int kcol = k0 < 2 * H ? k0 : k0 + H;
        const unsigned dst = ring_addr + (unsigned)((st % NS) * kSlotF * 4);
#pragma unroll
        for (int i = 0; i < IPS; ++i) res::dma16_sc1(rs_dg, dst + i * 1024, va[h][i], kcol * 4);
    };
    // everything the slice (tt, half h) needs before its product can start; returns the number of vector-memory
    // operations it issued (all younger than the stores of the slice that ends)
    auto prepare = [&](int tt, auto hh, bool peers_in) -> int {
        constexpr int h = decltype(hh)::value;
        if (tt < T - 1) {
            if (lane == 0 && !peers_in) poll_peers(cnt + h, (unsigned)(NT * (T - 1 - tt)), s.err, s.spin_limit);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int d = 0; d < NS - 1; ++d) issue(tt, hh, d);
            return IPS * (NS - 1);
        }
        return 0;
    };
    const std::integral_constant<int, 0> h0c{};
    const std::integral_constant<int, 1> h1c{};
    prepare(T - 1, h0c, true);

    for (int t = T - 1; t >= t_last; --t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            PSTAMP(7);
            f32x16 acc[2];
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ub][r] = 0.f;
            float4 val[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) val[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            bool peers_in = false;      // the next slice's peers had all arrived when this slice's product was half done

            PSTAMP(0);
            if (t < T - 1) {
                unsigned seen = 0;
                // a chunk's operand reads (the dg fragment; the W_hh values of a chunk that lives in LDS) are issued
                // one chunk ahead of its products
                auto frag = [&](int c) {
                    const float *slot = ring + ((c / CPS) % NS) * kSlotF;
                    return *reinterpret_cast<const float4 *>(slot + lrow * KS + 4 * rot<KS>(2 * (c % CPS) + lk, lrow));
                };
                res::wait_loads(IPS * (NS - 2));
                asm volatile("" ::: "memory");
                float4 av = frag(0), wl0 = make_float4(0.f, 0.f, 0.f, 0.f), wl1 = wl0;
                if (NCR == 0) { wl0 = Wl[0]; wl1 = Wl[64]; }
#pragma unroll
                for (int c = 0; c < NC; ++c) {      // chunk: eight k, four per lane half
                    const int st = c / CPS, cc = c % CPS;
                    // The scheduling barriers keep this order: products of chunk c - 1 | request, count | reads of
                    // chunk c + 1 | products of chunk c.  The slot a request overwrites was read by stage st - 1, and
                    // the last of those reads was waited for by the products of chunk c - 1.
                    __builtin_amdgcn_sched_barrier(0);
                    if (cc == 0 && st == NST / 2)      // (one more load in flight: the counts below wait a little longer)
                        seen = __hip_atomic_load(cnt + (1 - h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cc == 0 && st + NS - 1 < NST) {
                        if (h == 0) issue(t, h0c, st + NS - 1);
                        else issue(t, h1c, st + NS - 1);
                    }
                    if (cc == CPS - 1 && st + 1 < NST) {
                        // stage st + 1 has landed when only the stages requested after it are still in flight
                        res::wait_loads(IPS * (NST - 2 - st < NS - 2 ? NST - 2 - st : NS - 2));
                        asm volatile("" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    float4 b0, b1;
                    if (c < NCR) {
                        const int cr = c < NCR ? c : 0;
                        b0 = make_float4(wreg[cr][0][0], wreg[cr][0][1], wreg[cr][0][2], wreg[cr][0][3]);
                        b1 = make_float4(wreg[cr][1][0], wreg[cr][1][1], wreg[cr][1][2], wreg[cr][1][3]);
                    } else {
                        b0 = wl0; b1 = wl1;
                    }
                    const float4 a4 = av;
                    if (c + 1 < NC) {
                        av = frag(c + 1);
                        if (c + 1 >= NCR) { wl0 = Wl[((c + 1 - NCR) * 2 + 0) * 64]; wl1 = Wl[((c + 1 - NCR) * 2 + 1) * 64]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the next chunk's reads stay ahead of this chunk's products)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
                }
                PSTAMP(2);
                if (h == 0) load_operands(t, h0c);
                else load_operands(t, h1c);
                // (looked at here, ahead of this slice's stores: the wait for the counter's value is over by now)
                peers_in = seen >= (unsigned)(NT * (T - 1 - (h == 0 ? t : t - 1)));
                asm volatile("" :: "v"(seen));
                // the four partial sums meet: group v of an accumulator (registers 4v .. 4v+3: rows 8v + 4 lk + j)
                // goes to wave v's part of X; a thread then adds the four sources of its own elements in wave order
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int ub = 0; ub < 2; ++ub)
                        X[((v * 4 + wave) * 2 + ub) * 64 + (lane ^ ((lane >> 4) & 1) ^ (2 * ub))] =
                            make_float4(acc[ub][4 * v], acc[ub][4 * v + 1], acc[ub][4 * v + 2], acc[ub][4 * v + 3]);
                __syncthreads();
                PSTAMP(6);
#pragma unroll
                for (int src = 0; src < 4; ++src)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float *p = Xf + xoff[i] + src * (2 * 64 * 4);
                        val[i].x += p[4 * xq]; val[i].y += p[4 * (1 ^ xq)]; val[i].z += p[4 * (2 ^ xq)]; val[i].w += p[4 * (3 ^ xq)];
                    }
                PSTAMP(3);
            } else {
                if (h == 0) load_operands(t, h0c);
                else load_operands(t, h1c);
            }

            if (t < 0) {   // gradient of the initial state
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + 32 * h + 16 * i + er;
                    const float4 c = cin[h][i], v = val[i];
                    if (m < B)
                        *reinterpret_cast<float4 *>(a.dh0 + (long)m * H + u0 + 4 * cg) = make_float4(c.x + v.x, c.y + v.y, c.z + v.z, c.w + v.w);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(a.dg + (long)t * B * 4 * H, step_bytes);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool active = t < len[h][i];
                    float o_r[4], o_z[4], o_a[4], o_g[4], c_new[4];
                    const float c_[4] = {cin[h][i].x, cin[h][i].y, cin[h][i].z, cin[h][i].w};
                    const float v_[4] = {val[i].x, val[i].y, val[i].z, val[i].w};
                    const float r_[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w}, z_[4] = {gz[i].x, gz[i].y, gz[i].z, gz[i].w};
                    const float n_[4] = {gn[i].x, gn[i].y, gn[i].z, gn[i].w}, g_[4] = {ghn[i].x, ghn[i].y, ghn[i].z, ghn[i].w};
                    const float p_[4] = {hp[i].x, hp[i].y, hp[i].z, hp[i].w}, d_[4] = {dy[i].x, dy[i].y, dy[i].z, dy[i].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dhc = c_[q] + v_[q];
                        const float d = active ? d_[q] + dhc : 0.f;
                        const float dn = d * (1.f - z_[q]);
                        const float dzz = d * (p_[q] - n_[q]);
                        const float an = active ? dn * (1.f - n_[q] * n_[q]) : 0.f;
                        o_r[q] = active ? an * g_[q] * r_[q] * (1.f - r_[q]) : 0.f;
                        o_z[q] = active ? dzz * z_[q] * (1.f - z_[q]) : 0.f;
                        o_a[q] = an;
                        o_g[q] = active ? an * r_[q] : 0.f;
                        c_new[q] = active ? d * z_[q] : dhc;
                    }
                    cin[h][i] = make_float4(c_new[0], c_new[1], c_new[2], c_new[3]);
                    const float4 vr = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]), vz = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
                    const float4 vn = make_float4(o_a[0], o_a[1], o_a[2], o_a[3]), vg = make_float4(o_g[0], o_g[1], o_g[2], o_g[3]);
                    // (dg rows are 4H wide; a row past the batch: an offset beyond the block, the stores are dropped)
                    const int o = ((inside >> (2 * h + i)) & 1u) ? 4 * rowoff[h][i] + ucol : 0x7fffffff;
                    res::st4_sc1(rs_o, o, 0, vr); res::st4_sc1(rs_o, o, H * 4, vz);
                    res::st4_sc1(rs_o, o, 2 * H * 4, vn); res::st4_sc1(rs_o, o, 3 * H * 4, vg);
                    sums[0].x += vr.x; sums[0].y += vr.y; sums[0].z += vr.z; sums[0].w += vr.w;
                    sums[1].x += vz.x; sums[1].y += vz.y; sums[1].z += vz.z; sums[1].w += vz.w;
                    sums[2].x += vn.x; sums[2].y += vn.y; sums[2].z += vn.z; sums[2].w += vn.w;
                    sums[3].x += vg.x; sums[3].y += vg.y; sums[3].z += vg.z; sums[3].w += vg.w;
                }
            }
            PSTAMP(4);
            // the next slice: the other half of this step, or half 0 of the step before
            int younger = 0;
            if (h == 0) younger = prepare(t, h1c, peers_in);
            else if (t > t_last) younger = prepare(t - 1, h0c, peers_in);
            PSTAMP(1);
            // this slice's stores have landed when only the requests that followed them are in flight: arrival.  The
            // barrier also keeps X from being rewritten before every wave has read it.
            res::wait_loads(younger);
            asm volatile("" ::: "memory");
            lds_barrier();
            if (t > t_last && tid == 0) __hip_atomic_fetch_add(cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PSTAMP(5);
        }
    }
    PSTAMP_FLUSH(8)

    // bias partial sums: one row per block of 32 sequences is expected; this tile's two blocks give their sum in the
    // first row and zeros in the second.  The lanes that share a unit group by shuffle, the waves through LDS in
    // wave order.
    float *Xw = lds + kRingF + kWlF;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = sums[q];
        v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
        v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
        if (lane < 16) *reinterpret_cast<float4 *>(Xw + ((wave * 4 + q) * 16 + cg) * 4) = v;
    }
    __syncthreads();
    if (tid < 64) {      // unit tid of the tile: the four gates
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += Xw[(w * 4 + q) * 64 + tid];
            float *p = a.db_part + (long)(mt * 2) * 4 * H + q * H + u0 + tid;
            p[0] = v;
            p[4 * H] = 0.f;
        }
    }
}

}  // namespace

// (32-bit element offsets inside one step's rows: B * 4H must stay below 2^31)
bool gru_step_supported(int B, int H) {
    return H > 128 && H <= 2048 && (H & 31) == 0 && (long)B * 4 * H < (1L << 31);
}
int gru_step_bias_rows(int B) { return 4 * ((B + kRows - 1) / kRows); }

// Compute units of the current device (looked up once per process; the kernels are gfx950-only and a
// process drives GPUs of one kind).  The persistent kernels need every workgroup of a launch resident.
static int cu_count() {
    static std::atomic<int> n{0};
    int v = n.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = -1;
        n.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Row tiles one persistent launch may serve when a row tile is `nt` workgroups.  Workgroup ids go round the
// eight XCDs and tile_of() gives XCD x the row tiles x, x + 8, ...: an XCD receives ceil(MT / 8) * nt
// workgroups of a launch, one per compute unit, and ALL of them must be resident at once (the workgroups of a
// row tile wait for each other), so the bound is per XCD -- an eighth of the chip's compute units -- and not
// the chip's total.  0: a row tile does not fit on one XCD, the per-step kernels are used.
static int persist_row_tiles(int nt) {
    const int per_xcd = cu_count() / 8;
    return per_xcd >= nt ? 8 * (per_xcd / nt) : 0;
}

// (one word per arrival counter of the finest tiling, the backward kernels' blocks of 32 sequences; word 0 is unused)
size_t gru_step_sync_words(int B) { return 1 + 4 * (size_t)((B + kRows - 1) / kRows); }

// (the persistent kernels address one step's rows through 32-bit buffer offsets: the largest block, a step's
// rows of dg, must stay below 2 GiB)
static bool persist_ok(int B, int H, const unsigned *sync) {
    return sync && !switches().no_persist && persist_row_tiles(H / kUnits) > 0 && (long)B * 4 * H * 4 < (1L << 31);
}
// The forward kernel with the input projection inside: layer input time-major or batch-major (32-bit byte offsets
// inside one step's rows), a width that is a multiple of the k-tile, everything 16-byte aligned.
bool gru_step_fwd_fuses(const FwdArgs &a) {
    if (switches().no_step_fuse || switches().no_step_dma || !a.x || !a.w_ih || !a.b_ih) return false;
    if (!gru_step_persistent(a.B, a.H) || a.H % pd::kFwdBK != 0 || a.In % pd::kFwdBK != 0) return false;
    if (a.xmap.mode != 0 && a.xmap.mode != 1) return false;
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.w_ih)) & 15) return false;
    const long x_bytes = a.xmap.mode == 1 ? ((long)(a.B - 1) * a.T + 1) * a.In * 4 : (long)a.B * a.In * 4;
    return x_bytes < (1L << 31) && (long)3 * a.H * a.In * 4 < (1L << 31);
}

bool gru_step_persistent(int B, int H) {
    return gru_step_supported(B, H) && !switches().no_persist && persist_row_tiles(H / kUnits) > 0 &&
           (long)B * 4 * H * 4 < (1L << 31);
}

// ----------------------------------------------------------------------------------------------
// A wait that timed out must not pass unseen.  ONE call of the library has one error word in its workspace: it
// is zeroed when the call starts (gru_step_begin), set by any wait of any layer that gives up, and looked at by
// a one-wave kernel at the end of the call (gru_step_check), which then overwrites the head of every result of
// the call with NaN and adds one to a counter that lives in the library's code object (one copy per GPU) until
// gru_step_timeouts() reads it.
// ----------------------------------------------------------------------------------------------
static __device__ unsigned g_timeouts;     // host code uses timeouts_ptr(), never this name

static unsigned *timeouts_ptr() {
    constexpr int kMaxDev = 64;
    static std::atomic<unsigned *> ptr[kMaxDev];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        (void)hipGetLastError();
        return nullptr;
    }
    unsigned *p = ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        void *sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_timeouts)) != hipSuccess || !sym) {
            (void)hipGetLastError();
            return nullptr;
        }
        p = static_cast<unsigned *>(sym);
        ptr[dev].store(p, std::memory_order_release);
    }
    return p;
}

struct PoisonList {
    float *p[8];
    int n[8];
    int count;
};

__global__ __launch_bounds__(64) void persist_check_kernel(const unsigned *err, unsigned *timeouts, PoisonList l) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (threadIdx.x == 0) atomicAdd(timeouts, 1u);
    const float nan = __uint_as_float(0x7fc00000u);
    for (int k = 0; k < l.count; ++k)
        for (int i = threadIdx.x; i < l.n[k]; i += 64) l.p[k][i] = nan;
}

hipError_t gru_step_begin(unsigned *err, hipStream_t s) { return hipMemsetAsync(err, 0, sizeof(unsigned), s); }

hipError_t gru_step_check(const unsigned *err, float *const *outs, const int *ns, int count, hipStream_t s) {
    unsigned *tp = timeouts_ptr();
    if (!tp) return hipErrorInvalidValue;
    PoisonList l{};
    for (int k = 0; k < count && l.count < 8; ++k)
        if (outs[k] && ns[k] > 0) { l.p[l.count] = outs[k]; l.n[l.count] = ns[k]; ++l.count; }
    TGRU_LAUNCH(persist_check_kernel, dim3(1), dim3(64), 0, s, err, tp, l);
    return hipGetLastError();
}

long gru_step_timeouts(int clear) {
    unsigned *tp = timeouts_ptr();
    if (!tp) return -1;
    unsigned v = 0;
    if (hipMemcpy(&v, tp, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (clear && v != 0 && hipMemset(tp, 0, sizeof(v)) != hipSuccess) return -1;
    return (long)v;
}

// zero the arrival counters of a layer's launches
static hipError_t zero_arrivals(unsigned *sync, int B, hipStream_t s) {
    return hipMemsetAsync(sync, 0, gru_step_sync_words(B) * sizeof(unsigned), s);
}

hipError_t launch_gru_step_fwd(const FwdArgs &a, unsigned *sync, unsigned *err, hipStream_t s) {
    // the state of step t-1 is read back from the time-major output of this layer
    float *ysrc = a.y_int ? a.y_int : (a.ymap.mode == 0 ? a.y_ext : nullptr);
    if (!gru_step_supported(a.B, a.H) || !ysrc) return hipErrorInvalidValue;
    if (err && persist_ok(a.B, a.H, sync)) {
        const int NT = a.H / kUnits, MT = (a.B + kRows - 1) / kRows;
        const int per_launch = persist_row_tiles(NT);       // row tiles whose workgroups are resident together
        hipError_t e = zero_arrivals(sync, a.B, s);
        if (e != hipSuccess) return e;
        for (int mt0 = 0; mt0 < MT; mt0 += per_launch) {
            PersistFwd k;
            k.f = a;
            k.ybuf = ysrc;
            k.sync = sync;
            k.err = err;
            k.mt0 = mt0;
            k.MT = min(per_launch, MT - mt0);
            k.mtx = (k.MT + 7) / 8;
            k.spin_limit = switches().spin_limit;
            if (gru_step_fwd_fuses(a)) TGRU_LAUNCH(gru_persist_fwd_fused_kernel<pd::kFwdBK>, dim3(8 * k.mtx * NT), dim3(pd::kThreads), 0, s, k);
            else if (!switches().no_step_dma && a.H % pd::kFwdBK == 0) TGRU_LAUNCH(gru_persist_fwd_dma_kernel<pd::kFwdBK>, dim3(8 * k.mtx * NT), dim3(pd::kThreads), 0, s, k);
            else if (a.H % kPersistBK == 0) TGRU_LAUNCH(gru_persist_fwd_kernel<kPersistBK>, dim3(8 * k.mtx * NT), dim3(256), 0, s, k);
            else TGRU_LAUNCH(gru_persist_fwd_kernel<32>, dim3(8 * k.mtx * NT), dim3(256), 0, s, k);
        }
        return hipGetLastError();
    }
    StepFwd k;
    k.f = a;
    k.MT = (a.B + kRows - 1) / kRows;
    k.mtx = (k.MT + 7) / 8;
    const dim3 grid(8 * k.mtx * (a.H / kUnits));
    for (int t = 0; t < a.T; ++t) {
        k.t = t;
        k.hprev = t ? ysrc + (size_t)(t - 1) * a.B * a.H : a.h0;
        TGRU_LAUNCH(gru_step_fwd_kernel, grid, dim3(256), 0, s, k);
    }
    return hipGetLastError();
}

hipError_t launch_gru_step_bwd(const BwdArgs &a, float *carry, unsigned *err, hipStream_t s) {
    if (!gru_step_supported(a.B, a.H) || !carry || !a.y_int) return hipErrorInvalidValue;
    // the persistent kernel keeps the carry in registers: the carry buffer (B x H floats) holds its sync words
    unsigned *sync = reinterpret_cast<unsigned *>(carry);
    // the caller's dy is read through one descriptor per step: time-major or batch-major rows only, and a
    // batch-major tensor must stay below 2 GiB
    const bool dy_ok = !a.dy_ext || a.dymap.mode == 0 ||
                       (a.dymap.mode == 1 && (long)a.B * a.T * a.H * 4 < (1L << 31));
    if (err && (size_t)a.B * a.H >= gru_step_sync_words(a.B) && dy_ok && persist_ok(a.B, a.H, sync)) {
        const int NT = a.H / kUnits, MT = (a.B + kRows - 1) / kRows;
        const int per_launch = persist_row_tiles(NT);
        hipError_t e = zero_arrivals(sync, a.B, s);
        if (e != hipSuccess) return e;
        if (a.H % 64 == 0 && !switches().step_bwd_wide && !(a.dy_int && a.dy_ext)) {
            // 64-sequence x 64-unit tiles; an even number of row tiles, so that every block of 32 sequences the
            // bias partial sums are indexed by gets written (a tile past the batch writes zeros)
            const int NT2 = a.H / 64, MT2 = 2 * MT, per2 = persist_row_tiles(NT2);
            for (int mt0 = 0; mt0 < MT2; mt0 += per2) {
                PersistBwd k;
                k.b = a;
                k.sync = sync;
                k.err = err;
                k.mt0 = mt0;
                k.MT = min(per2, MT2 - mt0);
                k.mtx = (k.MT + 7) / 8;
                k.spin_limit = switches().spin_limit;
                const bool resident = !switches().no_step_res;      // W_hh tile resident: H a multiple of 128 up to 512
                if (resident && a.H == 512) TGRU_LAUNCH((gru_persist_bwd_res_kernel<512, res::kNcl512>), dim3(8 * k.mtx * NT2), dim3(256), 0, s, k);
                else if (resident && a.H == 384) TGRU_LAUNCH((gru_persist_bwd_res_kernel<384, 0>), dim3(8 * k.mtx * NT2), dim3(256), 0, s, k);
                else if (resident && a.H == 256) TGRU_LAUNCH((gru_persist_bwd_res_kernel<256, 0>), dim3(8 * k.mtx * NT2), dim3(256), 0, s, k);
                else TGRU_LAUNCH(gru_persist_bwd_sq_kernel<64>, dim3(8 * k.mtx * NT2), dim3(256), 0, s, k);
            }
            return hipGetLastError();
        }
        for (int mt0 = 0; mt0 < MT; mt0 += per_launch) {
            PersistBwd k;
            k.b = a;
            k.sync = sync;
            k.err = err;
            k.mt0 = mt0;
            k.MT = min(per_launch, MT - mt0);
            k.mtx = (k.MT + 7) / 8;
            k.spin_limit = switches().spin_limit;
            if (a.H % kPersistBK == 0) TGRU_LAUNCH(gru_persist_bwd_kernel<kPersistBK>, dim3(8 * k.mtx * NT), dim3(256), 0, s, k);
            else TGRU_LAUNCH(gru_persist_bwd_kernel<32>, dim3(8 * k.mtx * NT), dim3(256), 0, s, k);
        }
        return hipGetLastError();
    }
    StepBwd k;
    k.b = a;
    k.carry = carry;
    k.MT = (a.B + kRows - 1) / kRows;
    k.mtx = (k.MT + 7) / 8;
    const dim3 grid(8 * k.mtx * (a.H / kUnits));
    for (int t = a.T - 1; t >= (a.dh0 ? -1 : 0); --t) {
        k.t = t;
        k.first = t == a.T - 1;
        TGRU_LAUNCH(gru_step_bwd_kernel, grid, dim3(256), 0, s, k);
    }
    return hipGetLastError();
}

}  // namespace tgru
