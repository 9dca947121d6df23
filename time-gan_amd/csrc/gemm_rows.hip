// Row-streaming fp32 GEMM for the two batched products of a GRU layer with H = In = 96 (the energy config):
//      gi = X . W_ih^T + b_ih      (M x 96) . (96 x 288)        K = 96,  N = 288, W k-contiguous
//      dX = dgi . W_ih             (M x 288) . (288 x 96)       K = 288, N = 96,  W n-contiguous
// M is tens of thousands, K and N are tiny: a tile kernel spends its time in prologues and epilogues (three
// k-steps per 128 x 32 tile).  Here the small matrix never leaves the register file: a workgroup owns a
// run of rows, twelve waves = 2 row halves x 6 column groups hold the W fragments of their columns for all K
// (72 VGPRs), and the rows stream past through an LDS ring in stages of 32, fetched two stages ahead by a
// thirteenth wave with direct-to-LDS loads (it alone loads, so its load counter is exact; the product waves
// only store).  v_mfma_f32_16x16x4_f32; a lane's 16-byte A fragment feeds four k-steps (the reduction index
// of step j of piece s is k = 16 s + 4 (lane >> 4) + j for both operands).
//
// LDS layout of a stage: row r is Q = K/4 16-byte chunks; chunk q sits at position (q + r) % Q of its row.
// Rows are a multiple of 32 banks long, so without the rotation the sixteen rows of a fragment read would all
// hit the same banks; direct-to-LDS loads write 64 consecutive chunks per instruction, so the rotation is done
// on the global side (each lane fetches the chunk that belongs at its LDS position).
#include "tgru_internal.h"

namespace tgru {
namespace rowsg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kStage = 32;         // rows per stage
constexpr int kSlots = 3;          // LDS ring: the stage being read and two in flight
constexpr int kWaves = 12;         // product waves; wave 12 loads
constexpr int kOOB = 0x7ffffff0;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_load_b128(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, off, 0, 0, 0);
}

struct Args {
    const float *A; int lda;       // M rows, k-contiguous
    int a_bt, B, T;                // a_bt: row m = (t, b) of the product is row b*T + t of A (B % 32 == 0)
    const float *W; int ldw;       // B_KC: (N, K) rows of K; else (K, N) rows of N
    const float *bias;             // N or null
    float *C; int ldc;
    RowMap cmap;                   // identity or batch-major
    int M, rows_per_wg;
};

}  // namespace rowsg

template <int K, int NT, bool B_KC>
__global__ __launch_bounds__(64 * (rowsg::kWaves + 1)) void rows_gemm_kernel(rowsg::Args a) {
    using namespace rowsg;
    constexpr int Q = K / 4;                   // 16-byte chunks per row
    constexpr int NI = kStage * Q / 64;        // load instructions per stage
    constexpr int KS = K / 16;                 // fragment pieces per row
    static_assert(Q >= kStage / 2 && kStage * Q % 64 == 0 && K % 16 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];        // kSlots x [kStage][K]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m_begin = blockIdx.x * a.rows_per_wg, m_end = min(a.M, m_begin + a.rows_per_wg);
    if (m_begin >= m_end) return;
    const int nstages = (m_end - m_begin + kStage - 1) / kStage;

    if (wave == kWaves) {
        // ---- loader ---------------------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.A, (long)a.M * a.lda * 4);
        const int row_bytes = (a.a_bt ? a.T : 1) * a.lda * 4;            // distance between consecutive rows of a stage
        int rel[NI], rrow[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = i * 64 + lane;                                 // LDS chunk of the stage
            const int r = c / Q, p = c - r * Q;
            int q = p - r % Q;                                           // the chunk that belongs there
            if (q < 0) q += Q;
            rrow[i] = r;
            rel[i] = r * row_bytes + q * 16;
        }
        auto issue = [&](int st) {
            const int m0 = m_begin + st * kStage;
            int base;
            if (a.a_bt) { const int t = m0 / a.B, b0 = m0 - t * a.B; base = (b0 * a.T + t) * a.lda * 4; }
            else base = m0 * a.lda * 4;
            const int left = m_end - m0;                                 // rows of the stage that exist (may be <= 0)
            float *slot = smem + (st % kSlots) * kStage * K;
#pragma unroll
            for (int i = 0; i < NI; ++i) lds_load_b128(rs, slot + i * 256, rrow[i] < left ? base + rel[i] : kOOB);
        };
        issue(0);
        issue(1);
        for (int st = 0; st < nstages; ++st) {
            wait_vmcnt<NI>();                  // stage st has landed (st + 1 may be in flight)
            lds_barrier();                     // the product waves may read it, and are done with stage st - 1
            issue(st + 2);                     // into the slot stage st - 1 was in
        }
        wait_vmcnt<0>();
        return;
    }

    // ---- product waves: rows rg*16 .. +15 of every stage, columns (cg*NT + c)*16 .. +15 ---------------
    const int rg = wave / 6, cg = wave - rg * 6;
    const int l16 = lane & 15, kk = lane >> 4;
    float wreg[NT][K / 4], bias[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = (cg * NT + c) * 16 + l16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (B_KC) {
                const float4 v = *reinterpret_cast<const float4 *>(a.W + (long)n * a.ldw + 16 * s + 4 * kk);
                wreg[c][4 * s] = v.x; wreg[c][4 * s + 1] = v.y; wreg[c][4 * s + 2] = v.z; wreg[c][4 * s + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wreg[c][4 * s + j] = a.W[(long)(16 * s + 4 * kk + j) * a.ldw + n];
            }
        }
        bias[c] = a.bias ? a.bias[n] : 0.f;
    }
    const int R = rg * 16 + l16;               // this lane's row of the stage (A operand)
    const int p0 = (kk + R) % Q;               // position of chunk kk of row R; piece s is 4 s further round the row

    for (int st = 0; st < nstages; ++st) {
        lds_barrier();
        const float *row = smem + (st % kSlots) * kStage * K + R * K;
        f32x4 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int p = p0 + 4 * s;
            if (p >= Q) p -= Q;
            const float4 av = *reinterpret_cast<const float4 *>(row + 4 * p);
            const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[j], wreg[c][4 * s + j], acc[c], 0, 0, 0);
        }
        // register j of lane (l16, kk) is row 4 kk + j, column l16 of the 16 x 16 tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m_begin + st * kStage + rg * 16 + 4 * kk + j;
            if (m >= m_end) continue;
            const long crow = map_row(a.cmap, m);
            float *dst = a.C + crow * a.ldc + cg * NT * 16 + l16;
#pragma unroll
            for (int c = 0; c < NT; ++c) dst[c * 16] = acc[c][j] + bias[c];
        }
    }
}

// The shapes the row-streaming kernel covers (everything else goes to the tile kernel).
bool rows_gemm_supported(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs) {
    if (!a_kc || nslabs != 1 || g.accumulate || g.kchunk < g.K) return false;
    const bool gi = b_kc && g.K == 96 && g.N == 288 && g.ldb == 96;
    const bool dx = !b_kc && g.K == 288 && g.N == 96 && g.ldb == 96 && g.bmap.mode == 0;
    if (!gi && !dx) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.Bm) & 15) || (g.lda & 3)) return false;
    if (g.amap.mode == 2 || g.cmap.mode == 2) return false;                          // packed rows
    if (g.amap.mode == 1 && (g.amap.B % rowsg::kStage)) return false;                // a stage must share its time step
    if ((long)g.M * g.lda * 4 >= (1L << 31)) return false;                           // 31-bit byte offsets
    return true;
}

hipError_t launch_rows_gemm(const GemmArgs &g, bool b_kc, hipStream_t s) {
    rowsg::Args a{};
    a.A = g.A; a.lda = (int)g.lda; a.a_bt = g.amap.mode == 1; a.B = g.amap.B; a.T = g.amap.T;
    a.W = g.Bm; a.ldw = (int)g.ldb; a.bias = g.bias; a.C = g.C; a.ldc = (int)g.ldc; a.cmap = g.cmap; a.M = g.M;
    const int per = (g.M + 255) / 256;
    a.rows_per_wg = (per + rowsg::kStage - 1) / rowsg::kStage * rowsg::kStage;
    const int nwg = (g.M + a.rows_per_wg - 1) / a.rows_per_wg;
    const size_t lds = (size_t)rowsg::kSlots * rowsg::kStage * g.K * sizeof(float);
    dim3 block(64 * (rowsg::kWaves + 1));
    if (b_kc) {
        TGRU_LAUNCH((rows_gemm_kernel<96, 3, true>), dim3(nwg), block, lds, s, a);
    } else {
        // 108 KB of dynamic LDS: above the default cap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rows_gemm_kernel<288, 1, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        TGRU_LAUNCH((rows_gemm_kernel<288, 1, false>), dim3(nwg), block, lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace tgru
