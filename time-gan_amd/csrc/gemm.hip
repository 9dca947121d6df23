// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-tiled, register-staged double buffer.
//
// Used for every batched (non-recurrent) product of the GRU:
//   gi   = X . W_ih^T + b_ih         (a_kc, b_kc)      rows = all (b,t) at once
//   dX   = dGI . W_ih                (a_kc, !b_kc)
//   dW_ih = dGI^T . X, dW_hh = dGH^T . H_prev   (!a_kc, !b_kc, split over the row dimension)
#include "tgru_internal.h"
#include <stdlib.h>

namespace tgru {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int NTHREADS = 256;

// Diagnostic builds only (scratch/gemm_knockout.sh): a bit mask that removes one part of the kernel at a
// time, so that the time of what remains can be measured.  1: no operand loads inside the loop, 2: no
// LDS refill and no barrier inside the loop, 4: no result stores, 8: no LDS
// refill (barrier kept), 16: no barrier (refill kept); loader-wave kernel: 32: no loads after the
// first two tiles, 64: a quarter of the products.  The results are then wrong.
#ifndef TGRU_GEMM_KO
#define TGRU_GEMM_KO 0
#endif

// FAST: every operand load is an unconditional 16-byte load (the launcher has checked alignment,
// K and kchunk multiples of BK, no packed row maps, M / N multiples of 4 where they are contiguous);
// rows and columns past the edge are clamped onto valid ones and their results are never stored.
// A per-lane condition in front of a load costs a branch and serialises the loads of a tile.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool FAST>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs g) {
    // LDS tiles keep the orientation the operand has in memory, so a tile is always filled with one
    // 16-byte write per 16-byte load: a k-contiguous operand as [row][k] (rows of BK floats + 4 of padding),
    // the other kind as [k][row].  The reduction index of an MFMA step may be permuted freely as long as
    // both operands agree: step j of chunk q stands for k = 8q + 4*(lane >> 5) + j.  A k-contiguous
    // operand then supplies FOUR steps with one 16-byte fragment read (conflict-free: 80-byte rows).
    constexpr int LDK = BK + 4;
    constexpr int LDA_S = BM + 4;
    constexpr int LDB_S = BN + 4;
    constexpr int A_FLOATS = A_KC ? BM * LDK : BK * LDA_S;
    constexpr int B_FLOATS = B_KC ? BN * LDK : BK * LDB_S;
    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_FLOATS];
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad wave tiling");
    constexpr int SLOTS_A = BM * BK / 4;   // float4 slots per tile
    constexpr int SLOTS_B = BN * BK / 4;
    constexpr int PA = (SLOTS_A + NTHREADS - 1) / NTHREADS;
    constexpr int PB = (SLOTS_B + NTHREADS - 1) / NTHREADS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup -> (row tile, column tile, k slab).  Consecutive workgroup ids go round the eight XCDs,
    // each with its own L2, so the ids are dealt out in GROUPS that share operands: without split-K a
    // group is one row tile with all its column tiles (the A rows are fetched from HBM by one XCD, once),
    // with split-K a group is one k slab with all its tiles (each row of both operands likewise).  Group
    // gi runs on XCD gi % 8, its members on consecutive ids of that XCD.
    const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
    const int gsize = g.nslabs > 1 ? MT * NT : NT;
    const int ngroups = g.nslabs > 1 ? g.nslabs : MT;
    const int wid = blockIdx.x, grp = ((wid >> 3) / gsize) * 8 + (wid & 7), mem = (wid >> 3) % gsize;
    if (grp >= ngroups) return;
    const int tile_m = g.nslabs > 1 ? mem % MT : grp;
    const int tile_n = g.nslabs > 1 ? mem / MT : mem;
    const int slab = g.nslabs > 1 ? grp : 0;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int kbeg = slab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    const bool a_vec = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const bool b_vec = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.Bm) & 15) == 0);

    // per-thread slot coordinates (constant over the k loop)
    long a_base[PA];   // A_KC: element offset of (row, k=0) or -1; !A_KC: m index or -1
    int a_r[PA], a_c[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int slot = tid + p * NTHREADS;
        if (A_KC) {
            a_r[p] = slot >> 2;           // tile row (m)
            a_c[p] = (slot & 3) << 2;     // k offset inside the tile
            const int m = m0 + a_r[p];
            long r = -1;
            if (slot < SLOTS_A && m < g.M) r = map_row(g.amap, m);
            a_base[p] = (r < 0) ? -1 : r * g.lda;
        } else {
            a_r[p] = slot / (BM / 4);           // tile k row
            a_c[p] = (slot % (BM / 4)) << 2;    // m offset inside the tile
            a_base[p] = (slot < SLOTS_A) ? (long)(m0 + a_c[p]) : -1;
        }
    }
    long b_base[PB];
    int b_r[PB], b_c[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int slot = tid + p * NTHREADS;
        if (B_KC) {
            b_r[p] = slot >> 2;           // tile col (n)
            b_c[p] = (slot & 3) << 2;
            const int n = n0 + b_r[p];
            b_base[p] = (slot < SLOTS_B && n < g.N) ? (long)n * g.ldb : -1;
        } else {
            b_r[p] = slot / (BN / 4);
            b_c[p] = (slot % (BN / 4)) << 2;
            b_base[p] = (slot < SLOTS_B) ? (long)(n0 + b_c[p]) : -1;
        }
    }

    long a_fast[PA], b_fast[PB];   // FAST: element offset of the slot at k = 0
    if (FAST) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int slot = min(tid + p * NTHREADS, SLOTS_A - 1);
            if (A_KC) a_fast[p] = map_row(g.amap, min(m0 + (slot >> 2), g.M - 1)) * g.lda + ((slot & 3) << 2);
            else a_fast[p] = (long)(slot / (BM / 4)) * g.lda + min(m0 + ((slot % (BM / 4)) << 2), g.M - 4);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int slot = min(tid + p * NTHREADS, SLOTS_B - 1);
            if (B_KC) b_fast[p] = (long)min(n0 + (slot >> 2), g.N - 1) * g.ldb + ((slot & 3) << 2);
            else b_fast[p] = (long)(slot / (BN / 4)) * g.ldb + min(n0 + ((slot % (BN / 4)) << 2), g.N - 4);
        }
    }

    float4 ra[PA], rb[PB];

    auto load_fast = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const float4 v = *reinterpret_cast<const float4 *>(g.A + a_fast[p] + (A_KC ? (long)k0 : (long)k0 * g.lda));
            ra[p] = make_float4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const float4 v = *reinterpret_cast<const float4 *>(g.Bm + b_fast[p] + (B_KC ? (long)k0 : (long)k0 * g.ldb));
            rb[p] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    auto load_guarded = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_base[p] >= 0) {
                if (A_KC) {
                    const int k = k0 + a_c[p];
                    const float *src = g.A + a_base[p] + k;
                    if (a_vec && k + 3 < kend) {
                        v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        if (k + 0 < kend) v.x = src[0];
                        if (k + 1 < kend) v.y = src[1];
                        if (k + 2 < kend) v.z = src[2];
                        if (k + 3 < kend) v.w = src[3];
                    }
                } else {
                    const int k = k0 + a_r[p];
                    if (k < kend) {
                        const long m = a_base[p];
                        const float *src = g.A + (long)k * g.lda + m;
                        if (a_vec && m + 3 < g.M) {
                            v = *reinterpret_cast<const float4 *>(src);
                        } else {
                            if (m + 0 < g.M) v.x = src[0];
                            if (m + 1 < g.M) v.y = src[1];
                            if (m + 2 < g.M) v.z = src[2];
                            if (m + 3 < g.M) v.w = src[3];
                        }
                    }
                }
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_base[p] >= 0) {
                if (B_KC) {
                    const int k = k0 + b_c[p];
                    const float *src = g.Bm + b_base[p] + k;
                    if (b_vec && k + 3 < kend) {
                        v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        if (k + 0 < kend) v.x = src[0];
                        if (k + 1 < kend) v.y = src[1];
                        if (k + 2 < kend) v.z = src[2];
                        if (k + 3 < kend) v.w = src[3];
                    }
                } else {
                    const int k = k0 + b_r[p];
                    const long krow = (k < kend) ? map_row(g.bmap, k) : -1L;
                    if (krow >= 0) {
                        const long n = b_base[p];
                        const float *src = g.Bm + krow * g.ldb + n;
                        if (b_vec && n + 3 < g.N) {
                            v = *reinterpret_cast<const float4 *>(src);
                        } else {
                            if (n + 0 < g.N) v.x = src[0];
                            if (n + 1 < g.N) v.y = src[1];
                            if (n + 2 < g.N) v.z = src[2];
                            if (n + 3 < g.N) v.w = src[3];
                        }
                    }
                }
            }
            rb[p] = v;
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            if (tid + p * NTHREADS < SLOTS_A) {
                if (A_KC) *reinterpret_cast<float4 *>(&As[buf][a_r[p] * LDK + a_c[p]]) = ra[p];
                else *reinterpret_cast<float4 *>(&As[buf][a_r[p] * LDA_S + a_c[p]]) = ra[p];
            }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            if (tid + p * NTHREADS < SLOTS_B) {
                if (B_KC) *reinterpret_cast<float4 *>(&Bs[buf][b_r[p] * LDK + b_c[p]]) = rb[p];
                else *reinterpret_cast<float4 *>(&Bs[buf][b_r[p] * LDB_S + b_c[p]]) = rb[p];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (kend - kbeg + BK - 1) / BK;
    const int lrow = lane & 31;
    const int lk = lane >> 5;
    int cur = 0;
    auto load_tile = [&](int kt) {
        if (FAST) load_fast(kbeg + kt * BK); else load_guarded(kbeg + kt * BK);
    };
    auto products = [&]() {
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {      // chunk q: eight k-steps, four per lane half
            const int k = 8 * q + 4 * lk;
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = wm * WTM + i * 32 + lrow;
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4 *>(&As[cur][m * LDK + k]);
                    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = As[cur][(k + e) * LDA_S + m];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn * WTN + j * 32 + lrow;
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4 *>(&Bs[cur][n * LDK + k]);
                    b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bs[cur][(k + e) * LDB_S + n];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };
    {
        if (nkt > 0) {
            load_tile(0);
            store_tiles(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            if (!(TGRU_GEMM_KO & 1) && kt + 1 < nkt) load_tile(kt + 1);
            products();
            if (!(TGRU_GEMM_KO & 2)) {
                if (!(TGRU_GEMM_KO & 8) && kt + 1 < nkt) store_tiles(cur ^ 1);
                if (!(TGRU_GEMM_KO & 16)) __syncthreads();
                cur ^= 1;
            }
        }
    }

    // epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *Cz = g.C + (long)slab * g.slab_stride;
    float bias[TN];   // read once: inside the store loop the compiler must assume that C aliases it
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + lrow;
        bias[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (g.cmap.mode == 0 && !g.accumulate && m0 + BM <= g.M && n0 + BN <= g.N) {
        // whole tile inside C, rows in order: one base address per lane, everything else constant offsets
        float *base = Cz + (long)(m0 + wm * WTM + 4 * lk) * g.ldc + (n0 + wn * WTN + lrow);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *row = base + (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if ((TGRU_GEMM_KO & 4) && acc[i][j][r] != 12345.678f) continue;
                    row[j * 32] = acc[i][j][r] + bias[j];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m >= g.M) continue;
            const long crow = map_row(g.cmap, m);
            if (crow < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + lrow;
                if (n < g.N) {
                    float v = acc[i][j][r] + bias[j];
                    if ((TGRU_GEMM_KO & 4) && v != 12345.678f) continue;
                    float *dst = Cz + crow * g.ldc + n;
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Loader-wave variant of the 128 x 128 kernel for products that qualify for unconditional 16-byte loads.
// Moving a tile from registers into LDS (ds_write_b128) is not covered by the products of the other waves:
// with the operand loads removed, the LDS writes alone cost the kernel a tenth of its time.  Here two extra
// waves fetch every tile with direct-to-LDS loads (no registers, no LDS write instructions) into a ring of
// slots, ahead of the four product waves, which only read LDS and multiply; the six waves meet at one barrier
// per k-tile.  The loader waves alone issue vector-memory operations inside the loop, so their vmcnt is an
// exact count of tiles in flight, and the product waves' LDS reads wait for nothing but LDS.
// Measured at the stress shapes (input projection / split weight gradient, TFLOP/s, one loader wave): k-tile 16
// with three slots 131 / 138, with four 135 / 138, k-tile 32 with three slots 133 / 140, with two 137 / 142
// (half the barriers per product, and two workgroups per CU of 64 KiB each still cover each other's barriers);
// two loader waves instead of one: + 3 % / + 1 %, four like two.
//
// A direct-to-LDS load writes 64 consecutive 16-byte pieces, so a tile has no padding.  A k-contiguous operand
// sits as [row][k-tile floats] = Q 16-byte pieces per row; piece c of row r is kept at position
// (c + r * Q / 16) modulo Q of its row, which makes the 16-byte fragment reads conflict-free (a ds_read_b128 is
// served in groups of sixteen lanes; the rows of a group that share banks then differ in their rotation); the
// rotation is applied on the global side, each lane fetching the piece that belongs at its LDS position.  The
// other kind sits as [k][128 floats] and is read 4 bytes at a time, conflict-free as it stands.
// ---------------------------------------------------------------------------------------------------------
namespace dma {
constexpr int BM = 128, BN = 128;
#ifndef TGRU_GEMM_DMA_BK
#define TGRU_GEMM_DMA_BK 32
#endif
#ifndef TGRU_GEMM_DMA_SLOTS
#define TGRU_GEMM_DMA_SLOTS 2
#endif
constexpr int DK = TGRU_GEMM_DMA_BK;                  // k-tile of this kernel
constexpr int kSlots = TGRU_GEMM_DMA_SLOTS;           // ring slots: the tile being read + kSlots - 1 ahead
constexpr int kAhead = kSlots - 1;
constexpr int kTileFloats = 128 * DK;                 // one operand tile
constexpr int kSlotFloats = 2 * kTileFloats;
constexpr int kLoadsPerTile = kTileFloats * 4 / 1024; // load instructions per operand tile (1 KiB each)
constexpr int kQ = DK / 4;                            // 16-byte pieces per row of a k-contiguous tile
#ifndef TGRU_GEMM_DMA_LOADERS
#define TGRU_GEMM_DMA_LOADERS 2
#endif
constexpr int kLoaders = TGRU_GEMM_DMA_LOADERS;       // loader waves; each fetches its share of every tile
constexpr int kPer = kLoadsPerTile / kLoaders;        // load instructions per loader wave and operand tile
constexpr int kThreads = 256 + 64 * kLoaders;
static_assert(kLoadsPerTile % kLoaders == 0 && (kAhead - 1) * 2 * kPer < 64, "vmcnt is a 6-bit count");
// position of piece c of row r inside its row
__device__ __forceinline__ int rot(int c, int r) { return (c + r * kQ / 16) & (kQ - 1); }
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
}  // namespace dma

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(dma::kThreads) void gemm_dma_kernel(GemmArgs g) {
    using namespace dma;
    __shared__ __attribute__((aligned(1024))) float smem[kSlots * kSlotFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (row tile, column tile, k slab): as in gemm_kernel
    const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
    const int gsize = g.nslabs > 1 ? MT * NT : NT;
    const int ngroups = g.nslabs > 1 ? g.nslabs : MT;
    const int wid = blockIdx.x, grp = ((wid >> 3) / gsize) * 8 + (wid & 7), mem = (wid >> 3) % gsize;
    if (grp >= ngroups) return;
    const int tile_m = g.nslabs > 1 ? mem % MT : grp;
    const int tile_n = g.nslabs > 1 ? mem / MT : mem;
    const int slab = g.nslabs > 1 ? grp : 0;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = slab * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nkt = (kend - kbeg) / DK;

    if (wave >= 4) {
        const int l = wave - 4;
        // ---- loader ------------------------------------------------------------------------------------
        // Buffer offsets are 32-bit: each operand's descriptor starts at the first byte this workgroup may
        // touch (the launcher has checked that the extent from there fits).
        const long a_row0 = A_KC ? (g.amap.mode == 0 ? (long)m0 : 0L) : (long)kbeg;
        const long b_row0 = B_KC ? (long)n0 : (long)kbeg;
        const __amdgpu_buffer_rsrc_t rs_a =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A + a_row0 * g.lda), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.Bm + b_row0 * g.ldb), 0, -1, 0x00020000);
        int va[kPer], vb[kPer];
#pragma unroll
        for (int ii = 0; ii < kPer; ++ii) {
            const int i = l * kPer + ii;
            // k-contiguous: an instruction covers 64 / kQ rows; lane -> (row, position), fetches piece c
            const int r = i * (64 / kQ) + lane / kQ, c = ((lane % kQ) - r * kQ / 16) & (kQ - 1);
            if (A_KC) {
                const long row = map_row(g.amap, min(m0 + r, g.M - 1)) - a_row0;
                va[ii] = (int)(row * g.lda * 4) + c * 16;
            } else {
                va[ii] = (int)((2 * i + (lane >> 5)) * g.lda * 4) + min(m0 + 4 * (lane & 31), g.M - 4) * 4;
            }
            if (B_KC) {
                vb[ii] = (int)((long)(min(n0 + r, g.N - 1) - n0) * g.ldb * 4) + c * 16;
            } else {
                vb[ii] = (int)((2 * i + (lane >> 5)) * g.ldb * 4) + min(n0 + 4 * (lane & 31), g.N - 4) * 4;
            }
        }
        auto issue = [&](int kt) {
            float *slot = smem + (kt % kSlots) * kSlotFloats;
            const int sa = A_KC ? (kbeg + kt * DK) * 4 : (int)((long)kt * DK * g.lda * 4);
            const int sb = B_KC ? (kbeg + kt * DK) * 4 : (int)((long)kt * DK * g.ldb * 4);
#pragma unroll
            for (int i = 0; i < kPer; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void *)(slot + (l * kPer + i) * 256), 16, va[i], sa, 0, 0);
#pragma unroll
            for (int i = 0; i < kPer; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void *)(slot + kTileFloats + (l * kPer + i) * 256), 16, vb[i], sb, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < kAhead; ++d)
            if (d < nkt) issue(d);
        for (int kt = 0; kt < nkt; ++kt) {
            // tile kt has landed (up to kAhead - 1 younger tiles may stay in flight)
            if (kAhead > 1 && kt + kAhead - 1 < nkt) wait_vmcnt<(kAhead - 1) * 2 * kPer>(); else wait_vmcnt<0>();
            lds_barrier();                       // the product waves may read it, and are done with tile kt - 1
            if (!(TGRU_GEMM_KO & 32) && kt + kAhead < nkt) issue(kt + kAhead);     // into the slot tile kt - 1 was in
        }
        return;
    }

    // ---- product waves: 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles ---------------------------------------------
    // A workgroup of five waves caps a wave at 256 registers, and the compiler then keeps the accumulators in
    // the vector half of the register file; an operand with the accumulator-register constraint makes it use
    // the accumulator half for them, as it does in the 256-thread kernel.
    { float z = 0.f; asm volatile("" ::"a"(z)); }
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // float offsets of this lane's fragments inside an operand tile (chunk q adds to them)
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + lrow, n = wn * 64 + i * 32 + lrow;
        fa[i] = A_KC ? m * DK : m;
        fb[i] = B_KC ? n * DK : n;
    }
    for (int kt = 0; kt < nkt; ++kt) {
        lds_barrier();
        const float *ta = smem + (kt % kSlots) * kSlotFloats, *tb = ta + kTileFloats;
#pragma unroll
        for (int q = 0; q < DK / 8; ++q) {      // chunk q: eight k-steps, four per lane half
            float a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (A_KC) {
                    const int m = wm * 64 + i * 32 + lrow;
                    const float4 v = *reinterpret_cast<const float4 *>(ta + fa[i] + 4 * rot(2 * q + lk, m));
                    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = ta[(8 * q + 4 * lk + e) * BM + fa[i]];
                }
                if (B_KC) {
                    const int n = wn * 64 + i * 32 + lrow;
                    const float4 v = *reinterpret_cast<const float4 *>(tb + fb[i] + 4 * rot(2 * q + lk, n));
                    b[i][0] = v.x; b[i][1] = v.y; b[i][2] = v.z; b[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[i][e] = tb[(8 * q + 4 * lk + e) * BN + fb[i]];
                }
            }
#pragma unroll
            for (int e = 0; e < ((TGRU_GEMM_KO & 64) ? 1 : 4); ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *Cz = g.C + (long)slab * g.slab_stride;
    float bias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + lrow;
        bias[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (g.cmap.mode == 0 && !g.accumulate && m0 + BM <= g.M && n0 + BN <= g.N) {
        // whole tile inside C, rows in order: one base address per lane, everything else constant offsets
        float *base = Cz + (long)(m0 + wm * 64 + 4 * lk) * g.ldc + (n0 + wn * 64 + lrow);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *row = base + (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc;
#pragma unroll
                for (int j = 0; j < 2; ++j) row[j * 32] = acc[i][j][r] + bias[j];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m >= g.M) continue;
            const long crow = map_row(g.cmap, m);
            if (crow < 0) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + lrow;
                if (n < g.N) {
                    float v = acc[i][j][r] + bias[j];
                    float *dst = Cz + crow * g.ldc + n;
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// may every operand load of this product be an unconditional 16-byte load?
static bool fast_loads(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs) {
    if (switches().gemm_guarded) return false;   // tests: keep the guarded loads covered
    if (!aligned16(g.A) || !aligned16(g.Bm) || (g.lda & 3) || (g.ldb & 3)) return false;
    if (g.K < BK || (g.K % BK) || (nslabs > 1 && (g.kchunk % BK))) return false;
    if (nslabs == 1 && g.kchunk < g.K) return false;
    if ((a_kc && g.amap.mode == 2) || (!b_kc && g.bmap.mode != 0)) return false;   // B rows: identity map only
    if (!a_kc && ((g.M & 3) || g.M < 4)) return false;
    if (!b_kc && ((g.N & 3) || g.N < 4)) return false;
    return true;
}

// The loader-wave kernel addresses its operands through buffer descriptors with 32-bit byte offsets: from the
// first row a workgroup may touch, everything it loads must lie within 2 GiB.
static bool dma_loads(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs) {
    if (switches().gemm_no_dma) return false;
    const long lim = 1L << 31;
    const long kext = nslabs > 1 ? g.kchunk : g.K;
    const long a_ext = a_kc ? (g.amap.mode == 0 ? (long)dma::BM : (long)g.M) * g.lda * 4 : kext * g.lda * 4;
    const long b_ext = b_kc ? (long)dma::BN * g.ldb * 4 : kext * g.ldb * 4;
    if ((g.K % dma::DK) || (nslabs > 1 && (g.kchunk % dma::DK))) return false;
    return a_ext < lim && b_ext < lim && (long)g.K * 4 < lim;
}

static hipError_t launch_dma(const GemmArgs &g0, bool a_kc, bool b_kc, int nslabs, hipStream_t s) {
    GemmArgs g = g0;
    g.nslabs = nslabs;
    const long MT = (g.M + dma::BM - 1) / dma::BM, NT = (g.N + dma::BN - 1) / dma::BN;
    const long gsize = nslabs > 1 ? MT * NT : NT, ngroups = nslabs > 1 ? nslabs : MT;
    dim3 grid((unsigned)(8 * ((ngroups + 7) / 8) * gsize));
    dim3 block(dma::kThreads);
    if (a_kc && b_kc) TGRU_LAUNCH((gemm_dma_kernel<true, true>), grid, block, 0, s, g);
    else if (a_kc && !b_kc) TGRU_LAUNCH((gemm_dma_kernel<true, false>), grid, block, 0, s, g);
    else if (!a_kc && !b_kc) TGRU_LAUNCH((gemm_dma_kernel<false, false>), grid, block, 0, s, g);
    else TGRU_LAUNCH((gemm_dma_kernel<false, true>), grid, block, 0, s, g);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, bool FAST>
static hipError_t launch_cfg(const GemmArgs &g0, bool a_kc, bool b_kc, int nslabs, hipStream_t s) {
    GemmArgs g = g0;
    g.nslabs = nslabs;
    const long MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
    const long gsize = nslabs > 1 ? MT * NT : NT, ngroups = nslabs > 1 ? nslabs : MT;
    dim3 grid((unsigned)(8 * ((ngroups + 7) / 8) * gsize));
    dim3 block(NTHREADS);
    if (a_kc && b_kc)
        TGRU_LAUNCH((gemm_kernel<BM, BN, WM, WN, true, true, FAST>), grid, block, 0, s, g);
    else if (a_kc && !b_kc)
        TGRU_LAUNCH((gemm_kernel<BM, BN, WM, WN, true, false, FAST>), grid, block, 0, s, g);
    else if (!a_kc && !b_kc)
        TGRU_LAUNCH((gemm_kernel<BM, BN, WM, WN, false, false, FAST>), grid, block, 0, s, g);
    else
        TGRU_LAUNCH((gemm_kernel<BM, BN, WM, WN, false, true, FAST>), grid, block, 0, s, g);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (!switches().no_rows && rows_gemm_supported(g, a_kc, b_kc, nslabs)) return launch_rows_gemm(g, b_kc, s);
    // narrow outputs get the 128x32 tile, everything else 128x128
    const int waste128 = ((g.N + 127) / 128) * 128 - g.N;
    const int waste32 = ((g.N + 31) / 32) * 32 - g.N;
    const bool narrow = (g.N <= 96) || (waste128 * 4 > g.N && waste32 * 4 <= g.N);
    const bool fast = fast_loads(g, a_kc, b_kc, nslabs);
    if (narrow) return fast ? launch_cfg<128, 32, 4, 1, true>(g, a_kc, b_kc, nslabs, s)
                            : launch_cfg<128, 32, 4, 1, false>(g, a_kc, b_kc, nslabs, s);
    if (fast && dma_loads(g, a_kc, b_kc, nslabs)) return launch_dma(g, a_kc, b_kc, nslabs, s);
    return fast ? launch_cfg<128, 128, 2, 2, true>(g, a_kc, b_kc, nslabs, s)
                : launch_cfg<128, 128, 2, 2, false>(g, a_kc, b_kc, nslabs, s);
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *slabs, long stride, int nslabs,
                                                          float *out, long n, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // four interleaved partial sums (fixed order): four loads in flight instead of one
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int z = 0;
    for (; z + 3 < nslabs; z += 4) {
        v0 += slabs[(long)z * stride + i];
        v1 += slabs[(long)(z + 1) * stride + i];
        v2 += slabs[(long)(z + 2) * stride + i];
        v3 += slabs[(long)(z + 3) * stride + i];
    }
    for (; z < nslabs; ++z) v0 += slabs[(long)z * stride + i];
    float v = (v0 + v1) + (v2 + v3);
    if (accumulate) v += out[i];
    out[i] = v;
}

hipError_t launch_slab_reduce(const float *slabs, long stride, int nslabs, float *out, long n,
                              int accumulate, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256);
    TGRU_LAUNCH(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, stride, nslabs, out,
                       n, accumulate);
    return hipGetLastError();
}

}  // namespace tgru
