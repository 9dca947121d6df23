// Recurrent part of the GRU for large hidden states (H > 128, H a multiple of 32): ONE LAUNCH PER
// TIME STEP, the whole chip working on that step.
//
// Why not a persistent kernel here: with the time loop inside the kernel a workgroup must own its
// sequences for all steps, so it needs ALL of W_hh at every step (3 MB at H = 512, far beyond LDS):
// each workgroup streams the matrix from L2 once per step and that traffic, not the MFMA pipe, sets
// the step time (about 80 us at H = 512 whatever the batch).  A launch per step lets a workgroup own
// a 128-sequence x 32-unit tile instead: it reads 32 (x 3 gates) rows of W_hh and 128 rows of
// h_{t-1}, 12.6 MFLOP for 0.45 MB, and 256 such tiles cover 2048 sequences x 512 units in one wave of
// workgroups.  The price is a kernel boundary per step (a few microseconds against about 30).
//
// forward, step t:   gh = h_{t-1} . W_hh^T for the tile's r, z, n columns (v_mfma_f32_32x32x2_f32,
//                    two LDS tile buffers fed from a register ring that holds the operand tiles of the
//                    next kDepth products), then the gate nonlinearities, the state
//                    update and every store of the step in the epilogue -- the three gate values of a
//                    (sequence, unit) pair sit in the same lane and register of three accumulators.
// backward, step t:  dh_t = dy_t + carry + dgh_{t+1} . W_hh (same tiling, K = 3H), then the gate
//                    gradients of step t in the epilogue: dg row t, the carry d*z for step t-1 and
//                    the running column sums for the bias gradients.
// Workgroups are numbered so that the tiles of one XCD (every eighth workgroup) share their rows of
// h / dg: the XCD's L2 then holds those rows plus W_hh.
#include "tgru_internal.h"

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

}  // namespace

// (32-bit element offsets inside one step's rows: B * 4H must stay below 2^31)
bool gru_step_supported(int B, int H) {
    return H > 128 && H <= 2048 && (H & 31) == 0 && (long)B * 4 * H < (1L << 31);
}
int gru_step_bias_rows(int B) { return 4 * ((B + kRows - 1) / kRows); }

hipError_t launch_gru_step_fwd(const FwdArgs &a, hipStream_t s) {
    // the state of step t-1 is read back from the time-major output of this layer
    const float *ysrc = a.y_int ? a.y_int : (a.ymap.mode == 0 ? a.y_ext : nullptr);
    if (!gru_step_supported(a.B, a.H) || !ysrc) return hipErrorInvalidValue;
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

hipError_t launch_gru_step_bwd(const BwdArgs &a, float *carry, hipStream_t s) {
    if (!gru_step_supported(a.B, a.H) || !carry || !a.y_int) return hipErrorInvalidValue;
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
