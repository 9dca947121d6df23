// Weight gradients of one layer for mid-size hidden states (32 < H <= 128), both matrices in one launch:
//      dW_ih (3H x In) = dgi^T . X          dW_hh (3H x H) = dgh^T . H_prev
// contracted over all K = T*B rows.  dgi = dg[:, 0:3H], dgh = dg[:, 0:2H] | dg[:, 3H:4H].
//
// The rows are cut into chunks and every chunk into SIX jobs, one workgroup each: (gate g, matrix m) computes
// rows [gH, (g+1)H) of dW_ih (m = 0) or dW_hh (m = 1).  A job reads ONE H-wide plane of dg (da_r, da_z, and
// da_n for dW_ih / da_n*r for dW_hh) and the chunk's rows of X or H_prev, streamed through LDS in stages of
// 32 rows with direct-to-LDS loads into a ring of three slots (two stages in flight while one is read, one
// barrier per stage); wave (i, q) holds the 32 x 32 output tile (row block i of the gate, column block q) and
// reads the next stage's fragments into registers while the matrix pipe works on the current ones.  Each chunk leaves one partial [dW_ih | dW_hh]; a second kernel sums the
// partials in a fixed order.  (One workgroup per chunk computing all of both matrices left 256 partials of
// 3H x (In + H) floats: at H = 96 more bytes written and re-read than the inputs hold.)
#include "tgru_internal.h"

namespace tgru {
namespace midw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kRows = 32;          // rows per stage
constexpr int kJobs = 6;           // (gate, matrix) jobs per chunk of rows
constexpr int kSlots = 3;          // LDS ring: the stage being read and two in flight
constexpr int kMaxLoads = 8;       // 64-lane load instructions one wave issues per stage, at most (H = In = 32)
constexpr int kOOB = 0x7ffffff0;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// all but this wave's n youngest loads (one stage's worth) have landed
__device__ __forceinline__ void wait_all_but(int n) {
    switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    default: wait_vmcnt<8>(); break;
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_load_b128(__amdgpu_buffer_rsrc_t r, float *lds, int off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, off, 0, 0, 0);
}

}  // namespace midw

struct MidWgradArgs {
    const float *dg;               // (K, 4H) time-major
    const float *xin; int In;      // (K, In) rows; x_bt: caller tensor is (B,T,In) and row (t,b) is at b*T+t
    int x_bt;
    const float *y;                // (K, H) this layer's output, time-major: H_prev row k = y row k - B
    const float *h0;               // (B, H) or null: H_prev rows k < B
    float *slabs; long slab_stride;
    int rows_per_chunk;            // multiple of kRows
    int njobs;                     // workgroups with work: 6 per chunk that owns rows
    int B, T, H;
};

// HC: H = In = HC known at compile time (LDS offsets become immediates, the edge masks fold away), 0: any shape
#ifdef TGRU_STAMPS   // diagnostic build only: shader cycles of wave 0 of workgroup 0: prologue, stage loop, epilogue
__device__ unsigned long long tgru_stamps_wgrad[4];
extern "C" int tgru_debug_stamps_wgrad(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgru_stamps_wgrad), sizeof(unsigned long long) * 4);
}
#define WSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                       if (blockIdx.x == 0 && threadIdx.x == 0) tgru_stamps_wgrad[k] = n_ - wst_; wst_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define WSTAMP_DECL unsigned long long wst_ = __builtin_amdgcn_s_memtime();
#else
#define WSTAMP(k)
#define WSTAMP_DECL
#endif

// The layers of a stack as one launch: blockIdx.y picks the layer (the layers' weight gradients depend on nothing a
// later recurrence produces, so they can all wait for the last one: one launch and one reduction per stack instead
// of one per layer).
struct MidWgradBatch { MidWgradArgs l[kMidWgradMaxLayers]; };

template <int HC>
__global__ __launch_bounds__(576) void gru_mid_wgrad_kernel(MidWgradBatch batch) {
    using namespace midw;
    const MidWgradArgs &a = batch.l[blockIdx.y];
    WSTAMP_DECL
    extern __shared__ __attribute__((aligned(16))) float smem[];   // kSlots x ( as[kRows][H] | bs[kRows][W] )
    const int H = HC ? HC : a.H, In = HC ? HC : a.In, B = a.B;
    const int K = a.T * B;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // Workgroup ids go round-robin over the 8 XCDs; the six jobs of a chunk read the same rows, so they are
    // given ids that land on one XCD (one L2): XCD x owns jobs [x * per_xcd, (x + 1) * per_xcd)
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= a.njobs) return;
    const int chunk = lin / kJobs, job = lin - chunk * kJobs;
    const int gate = job >> 1, hh = job & 1;           // which H rows of which matrix
    const int W = hh ? H : In;                          // width of the B operand (X or H_prev rows)
    const int acol = hh && gate == 2 ? 3 * H : gate * H;   // first column of this job's plane of dg
    const int nq = (W + 31) / 32;                       // column blocks; waves are (row block, column block)
    const int stage_floats = kRows * (H + W);
    const int kbeg = chunk * a.rows_per_chunk, kend = min(K, kbeg + a.rows_per_chunk);
    const __amdgpu_buffer_rsrc_t rs_dg = make_rsrc(a.dg, (long)K * 4 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.xin, (long)K * In * 4);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.y, (long)K * H * 4);
    const __amdgpu_buffer_rsrc_t rs_h0 = make_rsrc(a.h0, (long)B * H * 4);

    // A stage = 16-byte chunks flattened as | bs, dealt round-robin to the waves as 64-lane instructions; an
    // instruction never straddles the two parts (H and In are multiples of 16: both parts are whole
    // instructions).  Every wave issues the same number of loads for every stage -- rows past the chunk's end
    // are loaded out of range (zeros), and a wave whose share is one short loads out of range into a spare
    // kilobyte behind the ring -- so "all but my youngest n loads" is "all but the youngest stage".
    const int ca = kRows * H / 4, cb = kRows * W / 4;
    const int ninstr = (ca + cb) / 64;
    const int per_wave = HC ? kRows * 2 * HC / 256 / ((HC / 32) * (HC / 32)) + 1 : (ninstr + nwaves - 1) / nwaves;
    constexpr int kLoads = HC ? kRows * 2 * HC / 256 / ((HC / 32) * (HC / 32)) + 1 : kMaxLoads;
    float *const spare = smem + kSlots * kRows * (H + (H > In ? H : In));
    // per-lane byte offset of each load relative to the stage's first row, and the row it belongs to (the
    // divisions are done once, not per stage)
    int rel[kLoads], rrow[kLoads];
    const int brow_bytes = (!hh && a.x_bt ? a.T : 1) * W * 4;           // distance between consecutive rows of B
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
        const int i = wave + j * nwaves;
        const int c = i * 64 + lane;
        if (i >= ninstr) {
            rrow[j] = 1 << 20; rel[j] = 0;                                 // never in range
        } else if (c < ca) {
            const int r = c / (H / 4), o = c - r * (H / 4);
            rrow[j] = r; rel[j] = r * 4 * H * 4 + (acol + 4 * o) * 4;
        } else {
            const int cc = c - ca, r = cc / (W / 4), o = cc - r * (W / 4);
            rrow[j] = r; rel[j] = r * brow_bytes + 16 * o;
        }
    }
    auto issue = [&](int k0, float *buf) {
        // B is a multiple of kRows: the rows of a stage share their time step, and are all initial-state rows or none
        const int base_a = k0 * 4 * H * 4;
        int base_b;
        __amdgpu_buffer_rsrc_t rs_b;
        if (!hh) {
            rs_b = rs_x;
            if (a.x_bt) { const int t = k0 / B, b0 = k0 - t * B; base_b = (b0 * a.T + t) * In * 4; }
            else base_b = k0 * In * 4;
        } else if (k0 < B) { rs_b = rs_h0; base_b = k0 * H * 4; }
        else { rs_b = rs_y; base_b = (k0 - B) * H * 4; }
        const int left = kend - k0;                                      // rows of the stage that exist (may be <= 0)
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            if (!HC && j >= per_wave) break;
            const int i = wave + j * nwaves;
            float *dst = i < ninstr ? buf + i * 256 : spare;
            if (i * 64 < ca) lds_load_b128(rs_dg, dst, rrow[j] < left ? base_a + rel[j] : kOOB);
            else lds_load_b128(rs_b, dst, rrow[j] < left ? base_b + rel[j] : kOOB);
        }
    };

    // ---- this wave's output tile ----------------------------------------------------------------------
    const int i32 = lane & 31, kk = lane >> 5;
    const int wi = wave / nq, wq = wave - wi * nq;       // row block of the gate, column block
    const int m = wi * 32 + i32;                         // output row inside the gate's H rows
    const int n = wq * 32 + i32;                         // output column
    // Lanes of a tile's rows >= H or columns >= W read some valid element instead: row m of the A operand only
    // reaches row m of the product (column n of B only column n), and those are never stored.
    const int aoff = m < H ? m : 0, boff = kRows * H + (n < W ? n : 0);
    f32x16 acc, acc9;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc9[r] = 0.f;
    // H = In = 96 has nine tiles: eight waves (two per SIMD: nine would leave one SIMD with half again the
    // others' products) own one each and share the ninth, each taking two of a stage's sixteen row pairs of it.
    constexpr bool kShared = HC == 96;
    constexpr int kFrag = kRows / 2 + (kShared ? 2 : 0);

    // Stage s lives in ring slot s % 3.  The fragments of stage s + 1 are read from LDS into registers while
    // the matrix pipe works on the fragments of stage s read one turn earlier: a wave's LDS reads and address
    // arithmetic hide behind its own products instead of alternating with them between barriers.
    const int nstages = (kend - kbeg + kRows - 1) / kRows;
    float fa0[kFrag], fb0[kFrag], fa1[kFrag], fb1[kFrag];
    const int r9 = (4 * wave + kk) * H + 64 + i32;       // shared tile: rows 64.., columns 64.., row pairs 2*wave, 2*wave + 1
    auto read_frags = [&](int slot, float (&av)[kFrag], float (&bv)[kFrag]) {
        const float *st = smem + slot * stage_floats;
#pragma unroll
        for (int s2 = 0; s2 < kRows / 2; ++s2) {
            const int r = 2 * s2 + kk;
            av[s2] = st[r * H + aoff];
            bv[s2] = st[boff + r * W];
        }
        if constexpr (kShared) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                av[kRows / 2 + e] = st[r9 + 2 * e * HC];
                bv[kRows / 2 + e] = st[kRows * HC + r9 + 2 * e * HC];
            }
        }
    };
    int slot = 0;                                        // ring slot of the stage whose products are due
    auto turn = [&](int sti, float (&av)[kFrag], float (&bv)[kFrag], float (&nav)[kFrag], float (&nbv)[kFrag]) {
        if (HC) wait_vmcnt<kLoads>(); else wait_all_but(per_wave);      // my loads of stage sti + 1 have landed (sti + 2 may be in flight)
        lds_barrier();                                   // so have everyone's, and everyone holds stage sti in registers
        issue(kbeg + (sti + 3) * kRows, smem + slot * stage_floats);    // into the slot stage sti was in
        slot = slot == kSlots - 1 ? 0 : slot + 1;
        read_frags(slot, nav, nbv);
        __builtin_amdgcn_sched_barrier(0);               // the reads are issued before the products, not sunk to their use
#pragma unroll
        for (int s2 = 0; s2 < kRows / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
        if constexpr (kShared) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                acc9 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kRows / 2 + e], bv[kRows / 2 + e], acc9, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(kbeg, smem);
    issue(kbeg + kRows, smem + stage_floats);
    if (HC) wait_vmcnt<kLoads>(); else wait_all_but(per_wave);
    lds_barrier();                                       // stage 0 has landed
    issue(kbeg + 2 * kRows, smem + 2 * stage_floats);
    read_frags(0, fa0, fb0);
    WSTAMP(0);
    for (int sti = 0; sti < nstages; sti += 2) {
        turn(sti, fa0, fb0, fa1, fb1);
        if (sti + 1 < nstages) turn(sti + 1, fa1, fb1, fa0, fb0);
    }
    wait_vmcnt<0>();                                     // the loads past the end still target this workgroup's LDS
    mfma_drain();
    WSTAMP(1);

    // ---- this job's rows of the chunk's partial [dW_ih | dW_hh]: register r of lane (kk, j) is row
    // 8*(r/4) + r%4 + 4*kk of the tile
    float *dst = a.slabs + (long)chunk * a.slab_stride + (hh ? (long)3 * H * In : 0) + (long)gate * H * W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wi * 32 + 8 * (r / 4) + (r % 4) + 4 * kk;
        if (row < H && n < W) dst[(long)row * W + n] = acc[r];
    }
    if constexpr (kShared) {
        // the eight partial sums of the shared tile, added in wave order
        lds_barrier();                                   // every wave is past its last fragment read: the ring is free
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[(wave * 16 + r) * 64 + lane] = acc9[r];
        lds_barrier();
        for (int e = threadIdx.x; e < 16 * 64; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += smem[w * 1024 + e];
            const int r = e >> 6, l = e & 63;
            const int row = 64 + 8 * (r / 4) + (r % 4) + 4 * (l >> 5), col = 64 + (l & 31);
            dst[(long)row * HC + col] = v;
        }
    }
    WSTAMP(2);
}

// Workgroups [0, wblocks): out_a[i] / out_b[i] (=|+=) sum over slabs: slab = [a (na) | b (nb)]; 32 elements x 8
// slab groups per workgroup.  Workgroups from wblocks on: the bias gradients, column sums of the per-sequence
// partial sums db_part (rows x 4H: da_r | da_z | da_n | da_n*r): db_ih = columns 0 .. 3H-1, db_hh = columns
// 0 .. 2H-1 and 3H .. 4H-1; 8 columns x 32 row groups per workgroup.  (One launch for both: each alone is a
// few microseconds of launch latency around almost no work.)
struct MidReduceArgs {
    const float *slabs; long stride; int nslabs;
    float *out_a; long na; float *out_b; long nb;
    const float *db_part; int db_rows; float *db_ih, *db_hh;
};
struct MidReduceBatch { MidReduceArgs l[kMidWgradMaxLayers]; };

__global__ __launch_bounds__(256) void mid_wgrad_reduce_kernel(MidReduceBatch batch, int accumulate, int wblocks, int H) {
    const MidReduceArgs &q = batch.l[blockIdx.y];
    const float *slabs = q.slabs, *db_part = q.db_part;
    const long stride = q.stride, na = q.na, nb = q.nb;
    const int nslabs = q.nslabs, db_rows = q.db_rows;
    float *out_a = q.out_a, *out_b = q.out_b, *db_ih = q.db_ih, *db_hh = q.db_hh;
    if ((int)blockIdx.x >= wblocks && !db_part) return;      // (a layer without bias partials in a batch that has some)
    __shared__ float part[256];
    const bool bias = (int)blockIdx.x >= wblocks;
    // weights: 32 elements x 8 slab groups; biases: 8 columns x 32 row groups (rows are many, columns few)
    const int ne = bias ? 8 : 32, ng = 256 / ne;
    const int e = threadIdx.x % ne, grp = threadIdx.x / ne;
    const long idx = (long)(bias ? blockIdx.x - wblocks : blockIdx.x) * ne + e;
    const long n = bias ? 4L * H : na + nb;
    const float *src = bias ? db_part : slabs;
    const long step = bias ? 4L * H : stride;
    const int rows = bias ? db_rows : nslabs;
    const int per = (rows + ng - 1) / ng;
    const int c0 = grp * per, c1 = min(rows, c0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (idx < n) {
        int c = c0;
        for (; c + 3 < c1; c += 4) {
            v0 += src[(long)c * step + idx];
            v1 += src[(long)(c + 1) * step + idx];
            v2 += src[(long)(c + 2) * step + idx];
            v3 += src[(long)(c + 3) * step + idx];
        }
        for (; c < c1; ++c) v0 += src[(long)c * step + idx];
    }
    part[grp * ne + e] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (grp != 0 || idx >= n) return;
    float v = 0.f;
    for (int g = 0; g < ng; ++g) v += part[g * ne + e];
    if (!bias) {
        float *o = idx < na ? out_a + idx : out_b + (idx - na);
        *o = accumulate ? *o + v : v;
    } else if (idx < 3 * H) {
        db_ih[idx] = accumulate ? db_ih[idx] + v : v;
        if (idx < 2 * H) db_hh[idx] = accumulate ? db_hh[idx] + v : v;
    } else {
        const long j = idx - H;                    // column 2H + (idx - 3H)
        db_hh[j] = accumulate ? db_hh[j] + v : v;
    }
}

// Shapes the kernel covers: the parts of a stage must fill whole 64-lane load instructions
// (H and In multiples of 16), a stage must not mix initial-state rows with later ones (B multiple of
// 16), byte offsets must fit 31 bits.
bool gru_mid_wgrad_supported(int B, int T, int In, int H) {
    if (!gru_mid_supported(H) || H > 96 || In > 96 || (H & 15) || (In & 15) || B % midw::kRows) return false;   // at most 3 x 3 waves
    const long widest = 4L * H > In ? 4L * H : In;
    return (long)T * B * widest * 4 < (1L << 31);
}
// chunks of rows (= partial results); six workgroups per chunk, about one workgroup per CU in all
int gru_mid_wgrad_workgroups(int B, int T) {
    const int stages = (B * T + midw::kRows - 1) / midw::kRows;
    const int chunks = 256 / midw::kJobs;
    return stages < chunks ? stages : chunks;
}
long gru_mid_wgrad_slab_floats(int In, int H) { return 3L * H * (In + H); }

hipError_t launch_gru_mid_wgrad_batch(const MidWgradJob *jobs, int njobs_, int B, int T, int H, int accumulate,
                                      hipStream_t s) {
    if (njobs_ < 1 || njobs_ > kMidWgradMaxLayers) return hipErrorInvalidValue;
    const int In = jobs[0].In;
    for (int k = 1; k < njobs_; ++k)
        if (jobs[k].In != In || (jobs[k].db_part != nullptr) != (jobs[0].db_part != nullptr)) return hipErrorInvalidValue;
    MidWgradBatch wb{};
    MidReduceBatch rb{};
    const int nchunks = gru_mid_wgrad_workgroups(B, T);
    const int K = B * T;
    const int rows_per_chunk = ((K + nchunks - 1) / nchunks + midw::kRows - 1) / midw::kRows * midw::kRows;
    const int used = (K + rows_per_chunk - 1) / rows_per_chunk;     // chunks that own rows
    const long na = 3L * H * In, nb = 3L * H * H;
    for (int k = 0; k < njobs_; ++k) {
        MidWgradArgs &a = wb.l[k];
        const MidWgradJob &j = jobs[k];
        a.dg = j.dg; a.xin = j.xin; a.In = In; a.x_bt = j.x_bt; a.y = j.y; a.h0 = j.h0; a.slabs = j.slabs;
        a.slab_stride = gru_mid_wgrad_slab_floats(In, H);
        a.B = B; a.T = T; a.H = H;
        a.rows_per_chunk = rows_per_chunk;
        a.njobs = used * midw::kJobs;
        MidReduceArgs &q = rb.l[k];
        q.slabs = j.slabs; q.stride = a.slab_stride; q.nslabs = used; q.out_a = j.dw_ih; q.na = na; q.out_b = j.dw_hh;
        q.nb = nb; q.db_part = j.db_part; q.db_rows = j.db_rows; q.db_ih = j.db_ih; q.db_hh = j.db_hh;
    }
    const int widest = In > H ? In : H;
    const bool nine = H == 96 && In == 96;              // eight waves share the ninth tile
    const int nwaves = nine ? 8 : ((H + 31) / 32) * ((widest + 31) / 32);
    const size_t lds = (size_t)midw::kSlots * midw::kRows * (H + widest) * sizeof(float) + 1024;   // ring + spare
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_mid_wgrad_kernel<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_mid_wgrad_kernel<96>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((used * midw::kJobs + 7) / 8 * 8, njobs_);
    if (nine) TGRU_LAUNCH(gru_mid_wgrad_kernel<96>, grid, dim3(64 * nwaves), lds, s, wb);
    else TGRU_LAUNCH(gru_mid_wgrad_kernel<0>, grid, dim3(64 * nwaves), lds, s, wb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int wblocks = (int)((na + nb + 31) / 32), bblocks = jobs[0].db_part ? (4 * H + 7) / 8 : 0;
    TGRU_LAUNCH(mid_wgrad_reduce_kernel, dim3(wblocks + bblocks, njobs_), dim3(256), 0, s, rb, accumulate, wblocks, H);
    return hipGetLastError();
}

hipError_t launch_gru_mid_wgrad(const float *dg, const float *xin, int In, int x_bt, const float *y,
                                const float *h0, float *slabs, float *dw_ih, float *dw_hh, int B, int T, int H,
                                int accumulate, const float *db_part, int db_rows, float *db_ih, float *db_hh,
                                hipStream_t s) {
    const MidWgradJob j{dg, xin, In, x_bt, y, h0, slabs, dw_ih, dw_hh, db_part, db_rows, db_ih, db_hh};
    return launch_gru_mid_wgrad_batch(&j, 1, B, T, H, accumulate, s);
}

}  // namespace tgru
