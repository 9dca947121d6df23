// Internal declarations shared by the tgru HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

struct tgru_head_args;
struct tgru_reduce_job;

namespace tgru {

// Process-wide count of the kernel launches the library has issued (diagnostics only: a caller reads it
// before and after a step to learn how many launches the step cost).
extern std::atomic<unsigned long long> g_launches;
#define TGRU_LAUNCH(...)                                                 \
    do {                                                                 \
        ::tgru::g_launches.fetch_add(1, std::memory_order_relaxed);      \
        hipLaunchKernelGGL(__VA_ARGS__);                                 \
    } while (0)

// Kernel-family switches from the environment (TGRU_* variables), read once per process; read() reads
// them again.
struct Switches {
    bool force_generic, no_vec, no_mid, no_step, no_persist, no_h24, no_rows, no_fuse, gemm_guarded, gemm_no_dma, no_step_dma, no_step_fuse, no_step_res, step_bwd_wide, no_wide_head;
    int spin_limit;     // polls after which a wait of the persistent H > 128 kernels gives up (TGRU_SPIN_LIMIT)
    void read();
    Switches() { read(); }
};
Switches &switches();

// ----------------------------------------------------------------------------------------------
// Row addressing.  Inside the library every per-(b,t) buffer is time-major: row = t*B + b.
// A RowMap turns such an internal row into the row of a caller tensor in its layout, or -1
// when that (b,t) has no row (a packed sequence that has already ended).
// ----------------------------------------------------------------------------------------------
struct RowMap {
    int mode;          // 0 identity (T,B,F), 1 batch-major (B,T,F), 2 packed rows
    int B, T;
    const int *toff;   // packed only: int32[T+1] prefix sum of batch sizes
};

__device__ __forceinline__ long map_row(const RowMap &rm, int m) {
    if (rm.mode == 0) return m;
    const int t = m / rm.B;
    const int b = m - t * rm.B;
    if (rm.mode == 1) return (long)b * rm.T + t;
    const int o = rm.toff[t];
    return (b < rm.toff[t + 1] - o) ? (long)(o + b) : -1L;
}

// same mapping from (t, b) directly (no division)
__device__ __forceinline__ long map_tb(const RowMap &rm, int t, int b) {
    if (rm.mode == 0) return (long)t * rm.B + b;
    if (rm.mode == 1) return (long)b * rm.T + t;
    const int o = rm.toff[t];
    return (b < rm.toff[t + 1] - o) ? (long)(o + b) : -1L;
}

// Gate nonlinearities on the hardware exp2 / rcp units (about 1 ulp each).
__device__ __forceinline__ float fast_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
__device__ __forceinline__ float fast_tanh(float v) {
    const float a = fabsf(v);
    const float e = __builtin_amdgcn_exp2f(-2.88539008177793f * a);
    const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
    const float a2 = a * a;   // odd Taylor polynomial below 0.25, where 1 - e cancels
    const float small = a * (1.f + a2 * (-0.333333333f + a2 * (0.133333333f + a2 * (-0.0539682540f + a2 * 0.0218694885f))));
    return copysignf(a < 0.25f ? small : big, v);
}

// ----------------------------------------------------------------------------------------------
// fp32 MFMA GEMM:  C(m,n) [+]= sum_k A(m,k) * B(k,n)  (+ bias[n])
//   a_kc: A element (m,k) is at A[arow(m)*lda + k]  (k contiguous), else at A[k*lda + m]
//   b_kc: B element (k,n) is at B[n*ldb + k]        (k contiguous), else at B[k*ldb + n]
// split-K: gridDim.z slabs, slab z covers k in [z*kchunk, (z+1)*kchunk) and writes to
// C + z*slab_stride (plain store); the caller reduces the slabs.
// ----------------------------------------------------------------------------------------------
struct GemmArgs {
    const float *A; long lda;
    const float *Bm; long ldb;
    float *C; long ldc;
    const float *bias;
    int M, N, K;
    int kchunk; long slab_stride;
    int nslabs;         // set by launch_gemm
    int accumulate;     // only without split-K: C += result
    RowMap amap;        // rows of A (a_kc only)
    RowMap bmap;        // rows of B (!b_kc only): reduction index k -> caller row
    RowMap cmap;        // rows of C
};

hipError_t launch_gemm(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs, hipStream_t s);
// row-streaming kernel for the H = In = 96 input projection and its transpose (launch_gemm dispatches to it)
bool rows_gemm_supported(const GemmArgs &g, bool a_kc, bool b_kc, int nslabs);
hipError_t launch_rows_gemm(const GemmArgs &g, bool b_kc, hipStream_t s);

// split-K plan of a weight-gradient product C (M x N) over K rows: number of slabs, rows per slab
int gemm_split_for(int M, int N, long K, int *kchunk);

// out[i] (=|+=) sum_z slabs[z*stride + i], i < n
hipError_t launch_slab_reduce(const float *slabs, long stride, int nslabs, float *out, long n,
                              int accumulate, hipStream_t s);

// ----------------------------------------------------------------------------------------------
// Recurrent kernels (one launch per layer).
// ----------------------------------------------------------------------------------------------
struct FwdArgs {
    const float *gi;        // (T*B, 3H) input projection incl. b_ih, time-major
    const float *wp;        // packed W_hh fragments, see pack_whh_kernel
    const float *w_hh;      // W_hh as the caller holds it (3H, H): kernels of gru_mid.hip
    const float *b_hh;      // (3H)
    const float *h0;        // (B,H) or null
    float *y_int;           // (T*B, H) time-major layer output
    float *y_ext;           // top layer only: caller's y in its layout, or null
    RowMap ymap;
    float *gates;           // 4 planes (r,z,n,hn) of (T*B, H), or null (inference)
    float *h_n;             // (B,H) or null
    const int *lengths;     // int32[B] or null
    int B, T, H;
    // the layer's input and input weights, for the kernel that computes gi itself (gru_mid96.hip); the
    // other kernels read gi only
    const float *x; RowMap xmap; int In;
    const float *w_ih, *b_ih;
};
hipError_t launch_gru_fwd(const FwdArgs &a, hipStream_t s);

struct BwdArgs {
    const float *dy_int;    // (T*B, H) time-major gradient of this layer's output, or null
    const float *dy_ext;    // top layer: caller's dy in its layout, or null
    RowMap dymap;
    const float *dh_n;      // (B,H) or null
    const float *wtp;       // packed W_hh^T fragments, see pack_whh_t_kernel
    const float *w_hh;      // W_hh as the caller holds it (3H, H): kernels of gru_mid.hip
    const float *y_int;     // (T*B, H) this layer's output (h_t), time-major
    const float *h0;        // (B,H) or null
    const float *gates;     // 4 planes (r,z,n,hn)
    float *dg;              // (T*B, 4H): da_r | da_z | da_n | da_n*r
    float *dh0;             // (B,H) or null
    float *db_part;         // (nblocks, 4H) per-workgroup column sums of dg
    const int *lengths;
    int B, T, H;
    // the layer's input weights and the gradient of its input, for the kernel that forms dX itself
    // (gru_mid96.hip); the other kernels leave dX to a product over dg
    const float *w_ih; float *dx; RowMap dxmap; int In;
};
hipError_t launch_gru_bwd(const BwdArgs &a, hipStream_t s);
int gru_batch_blocks(int B);

// Packed weight fragment sizes (floats) and pack launches.
size_t packed_whh_floats(int H);
hipError_t launch_pack_whh(const float *w_hh, float *wp, float *wtp, int H, hipStream_t s);

// db_ih / db_hh from the per-workgroup partial sums.
hipError_t launch_bias_grad(const float *db_part, int nblocks, int H, float *db_ih, float *db_hh,
                            int accumulate, hipStream_t s);

// ----------------------------------------------------------------------------------------------
// Whole-stack kernels for small hidden sizes (gru_small.hip).
// ----------------------------------------------------------------------------------------------
constexpr int kSmallMaxH = 32;
constexpr int kSmallMaxIn = 32;
constexpr int kSmallMaxL = 8;

struct SmallFwdArgs {
    const float *x; RowMap xmap; int In;
    const float *w_ih[kSmallMaxL], *w_hh[kSmallMaxL], *b_ih[kSmallMaxL], *b_hh[kSmallMaxL];
    const float *h0;               // (L,B,H) or null
    float *y_ext; RowMap ymap;     // caller's y
    float *y_int[kSmallMaxL];      // per layer (T*B,H) time-major, or null
    float *gates[kSmallMaxL];      // per layer 4 planes, or null
    float *h_n;                    // (L,B,H) or null
    const int *lengths;
    int B, T, H, L;
};
hipError_t launch_gru_small_fwd(const SmallFwdArgs &a, hipStream_t s);

struct SmallBwdArgs {
    const float *dy_ext; RowMap dymap;
    const float *dh_n;             // (L,B,H) or null
    const float *w_ih[kSmallMaxL], *w_hh[kSmallMaxL];
    const float *y_int[kSmallMaxL];
    const float *gates[kSmallMaxL];
    const float *h0;
    float *dg[kSmallMaxL];         // per layer (T*B, 4H)
    float *dx_ext; RowMap dxmap; int In;
    float *dh0;                    // (L,B,H) or null
    float *db_part;                // (nblk, L, 4H)
    const int *lengths;
    int B, T, H, L;
    // one-sequence-per-workgroup kernels only (gru_vec.hip): layer-0 input and the per-workgroup
    // partial weight gradients
    const float *x; RowMap xmap;
    float *slabs; long slab_stride;
};
hipError_t launch_gru_small_bwd(const SmallBwdArgs &a, hipStream_t s);

struct SmallWgradArgs {
    const float *dg[kSmallMaxL];
    const float *x; RowMap xmap; int In;
    const float *y_int[kSmallMaxL];
    const float *h0;
    const float *db_part; int nblk;
    float *dw_ih[kSmallMaxL], *dw_hh[kSmallMaxL], *db_ih[kSmallMaxL], *db_hh[kSmallMaxL];
    float *slabs; long slab_stride; int nchunks;   // per-chunk partial gradients (workspace)
    int B, T, H, L, accumulate;
};
hipError_t launch_gru_small_wgrad(const SmallWgradArgs &a, hipStream_t s);
hipError_t launch_gru_small_wgrad_reduce(const SmallWgradArgs &a, const SmallWgradArgs *second,
                                         hipStream_t s);   // stage 2 alone, for one call or a pair
long small_wgrad_slab_floats(int In, int H, int L);
int small_wgrad_chunks(int B, int T);

// 16 < H <= 24, In <= 24 specialisation of the whole-stack kernels (gru_h24.hip)
hipError_t launch_gru_h24_fwd(const SmallFwdArgs &a, hipStream_t s);
hipError_t launch_gru_h24_bwd(const SmallBwdArgs &a, hipStream_t s);

// 32 < H <= 128, four sequences per workgroup on 4x4x1 MFMAs (gru_mid.hip)
bool gru_mid_supported(int H);
hipError_t launch_gru_mid_fwd(const FwdArgs &a, hipStream_t s);
// H = In = 96: the recurrent kernel with the input projection inside (no gi in memory)
bool gru_mid96_fwd_supported(const FwdArgs &a);
hipError_t launch_gru_mid96_fwd(const FwdArgs &a, hipStream_t s);
// H = In = 96: the backward recurrent kernel with dX = dgi . W_ih inside
bool gru_mid96_bwd_supported(const BwdArgs &a);
hipError_t launch_gru_mid96_bwd(const BwdArgs &a, hipStream_t s);
hipError_t launch_gru_mid_bwd(const BwdArgs &a, hipStream_t s);   // db_part: one row of 4H per SEQUENCE

// both weight gradients of a mid-H layer in one pass over the rows (gru_mid_wgrad.hip)
bool gru_mid_wgrad_supported(int B, int T, int In, int H);
int gru_mid_wgrad_workgroups(int B, int T);
long gru_mid_wgrad_slab_floats(int In, int H);
// The same for up to kMidWgradMaxLayers layers of one stack in ONE launch (+ one reduction): every job with the same
// In, all with bias partials or none.  Each layer needs its own dg, slabs and bias partials until then.
constexpr int kMidWgradMaxLayers = 4;
struct MidWgradJob {
    const float *dg, *xin; int In, x_bt; const float *y, *h0; float *slabs, *dw_ih, *dw_hh;
    const float *db_part; int db_rows; float *db_ih, *db_hh;
};
hipError_t launch_gru_mid_wgrad_batch(const MidWgradJob *jobs, int njobs, int B, int T, int H, int accumulate,
                                      hipStream_t s);
hipError_t launch_gru_mid_wgrad(const float *dg, const float *xin, int In, int x_bt, const float *y,
                                const float *h0, float *slabs, float *dw_ih, float *dw_hh, int B, int T, int H,
                                int accumulate, const float *db_part, int db_rows, float *db_ih, float *db_hh,
                                hipStream_t s);   // db_part (or null): the bias gradients are summed in the same launch

// H > 128, H % 32 == 0: one launch per time step, 128 sequences x 32 units per workgroup (gru_step.hip).
// Forward reads the state of step t-1 back from a.y_int (or from a.y_ext when that is time-major);
// backward needs a (B,H) carry buffer and leaves gru_step_bias_rows(B) rows in a.db_part.
bool gru_step_supported(int B, int H);
int gru_step_bias_rows(int B);
// `sync`: gru_step_sync_words(B) 32-bit words of device memory for the persistent variants (time loop inside
// the kernel, the workgroups of a row tile wait for each other through counters); null, or TGRU_NO_PERSIST=1,
// selects one launch per step.
size_t gru_step_sync_words(int B);
// `err`: the call's error word (see gru_step_begin); null selects one launch per step as well.
hipError_t launch_gru_step_fwd(const FwdArgs &a, unsigned *sync, unsigned *err, hipStream_t s);
hipError_t launch_gru_step_bwd(const BwdArgs &a, float *carry, unsigned *err, hipStream_t s);
// Will a call of these dimensions run on the persistent kernels (with the current switches and device)?
bool gru_step_persistent(int B, int H);
// Will launch_gru_step_fwd compute this layer's input projection itself (a.x, a.xmap, a.In, a.w_ih, a.b_ih set)?
// Then a.gi is not read and the caller skips the projection GEMM.
bool gru_step_fwd_fuses(const FwdArgs &a);
// One library call has one error word (a 32-bit word of its workspace): a wait of a persistent kernel that gives
// up sets it.  gru_step_begin zeroes it before the first layer; gru_step_check, after the last layer, launches
// one wave that -- if the word is set -- fills the first ns[k] floats of every outs[k] (null entries skipped, at
// most 8) with NaN and counts the call in a per-device counter; gru_step_timeouts returns that counter
// (synchronous copy from the device; -1 on failure) and zeroes it on request.
constexpr size_t kStepStatusFloats = 64;      // workspace set aside for the error word
hipError_t gru_step_begin(unsigned *err, hipStream_t s);
hipError_t gru_step_check(const unsigned *err, float *const *outs, const int *ns, int count, hipStream_t s);
long gru_step_timeouts(int clear);

// H, In <= 32, one sequence per workgroup on the vector ALU (gru_vec.hip); the backward kernel also
// leaves per-workgroup partial weight / bias gradients in a.slabs / a.db_part for the reduce stage
// Two independent calls in one launch: workgroups [0, nwg0) serve g[0], the others g[1] (nwg1 may be 0).
template <typename A>
struct VecPair {
    A g[2];
    int nwg0, nwg1;
};
bool gru_vec_pairable(int In0, int H0, int L0, int In1, int H1, int L1);
hipError_t launch_gru_vec_fwd(const SmallFwdArgs &a, const SmallFwdArgs *second, hipStream_t s);
hipError_t launch_gru_vec_bwd(const SmallBwdArgs &a, const SmallBwdArgs *second, hipStream_t s);
int gru_vec_workgroups(int B);
bool gru_vec_supported(int B, int T, int In, int H, int L);

// Fused heads beyond head.hip's widths (head_wide.hip): weights in LDS, weight gradients as split-K products.
// The argument blocks are the C ABI's (include/tgru.h), checked by the caller.
bool head_wide_supported(int din, int dmid, int dout, bool deep);
size_t head_wide_workspace_floats(int M, int din, int dmid, int dout);
int head_wide_forward(const ::tgru_head_args &c, hipStream_t s);
int head_wide_backward(const ::tgru_head_args &c, hipStream_t s);   // incl. the reductions unless deferred
int head_wide_reduce_jobs(const ::tgru_head_args &c, ::tgru_reduce_job *jobs, int max_jobs);

}  // namespace tgru
