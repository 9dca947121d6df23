// C ABI of the tgru library: argument checks, buffer carving and the per-layer launch sequence.
#include "../../include/tgru.h"
#include "tgru_internal.h"
#include <stdlib.h>
#include <string.h>

using namespace tgru;

namespace tgru {
std::atomic<unsigned long long> g_launches{0};

void Switches::read() {
    auto on = [](const char *name) {
        const char *e = getenv(name);
        return e && e[0] == '1';
    };
    force_generic = on("TGRU_FORCE_GENERIC");
    no_vec = on("TGRU_NO_VEC");
    no_mid = on("TGRU_NO_MID");
    no_step = on("TGRU_NO_STEP");
    no_persist = on("TGRU_NO_PERSIST");
    no_step_dma = on("TGRU_NO_STEP_DMA");
    no_step_fuse = on("TGRU_NO_STEP_FUSE");
    no_step_res = on("TGRU_NO_STEP_RES");
    step_bwd_wide = on("TGRU_STEP_BWD_WIDE");
    no_h24 = on("TGRU_NO_H24");
    no_rows = on("TGRU_NO_ROWS");
    no_fuse = on("TGRU_NO_FUSE");
    no_wide_head = on("TGRU_NO_WIDE_HEAD");
    gemm_guarded = getenv("TGRU_GEMM_GUARDED") != nullptr;
    gemm_no_dma = getenv("TGRU_GEMM_NO_DMA") != nullptr;
    // about ten seconds of polling; tests set a tiny limit to see a timeout reported
    spin_limit = 1 << 24;
    if (const char *e = getenv("TGRU_SPIN_LIMIT")) {
        const long v = strtol(e, nullptr, 10);      // 0: every wait gives up at once (tests of the report path)
        if (e[0] != 0 && v >= 0 && v <= (1L << 30)) spin_limit = (int)v;
    }
}
Switches &switches() {
    static Switches s;
    return s;
}
int gemm_split_for(int M, int N, long K, int *kchunk) {
    const int bn = (N <= 96) ? 32 : 128;
    const long tiles = (long)((M + 127) / 128) * ((N + bn - 1) / bn);
    long want = 768 / tiles;
    if (want < 1) want = 1;
    long maxs = K / 128;
    if (maxs < 1) maxs = 1;
    long ns = want < maxs ? want : maxs;
    long kc = (K + ns - 1) / ns;
    kc = (kc + 31) / 32 * 32;      // whole k-tiles of either GEMM kernel (16 / 32)
    if (kc < 32) kc = 32;
    ns = (K + kc - 1) / kc;
    if (ns < 1) ns = 1;
    *kchunk = (int)kc;
    return (int)ns;
}
}  // namespace tgru

namespace {

constexpr int kMaxHidden = 512;

inline size_t align64(size_t floats) { return (floats + 63) & ~(size_t)63; }

struct Dims {
    int B, T, In, H, L, layout;
    size_t TB() const { return (size_t)T * B; }
    size_t TBH() const { return (size_t)T * B * H; }
};

int check_dims(int B, int T, int In, int H, int L, int layout) {
    if (B <= 0 || T <= 0 || In <= 0 || H <= 0 || L <= 0) return TGRU_E_DIMS;
    if (H > kMaxHidden) return TGRU_E_DIMS;
    if ((long long)B * T >= (1LL << 31) / 4) return TGRU_E_DIMS;
    if (layout != TGRU_LAYOUT_TBF && layout != TGRU_LAYOUT_BTF && layout != TGRU_LAYOUT_PACKED)
        return TGRU_E_LAYOUT;
    return TGRU_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Kernel-family switches from the environment, read ONCE per process (tgru_reload_switches() reads them
// again; the tests use it to cover every family on the same fixtures):
//   TGRU_FORCE_GENERIC=1  per-layer kernels instead of the whole-stack kernels of gru_small.hip
//   TGRU_NO_VEC=1         MFMA whole-stack kernels instead of one sequence per workgroup (gru_vec.hip)
//   TGRU_NO_MID=1         16-sequence kernels of gru_recurrent.hip instead of gru_mid.hip (32 < H <= 128)
//   TGRU_NO_STEP=1        persistent 16-sequence kernels instead of one launch per step (gru_step.hip)
//   TGRU_NO_PERSIST=1     H > 128: one launch per time step instead of the persistent kernels (gru_step.hip)
//   TGRU_NO_STEP_DMA=1    H > 128: persistent kernels with register-staged operand tiles instead of loader waves
//   TGRU_NO_STEP_FUSE=1   H > 128: the layer's input projection as a batched GEMM instead of inside the forward kernel
//   TGRU_NO_STEP_RES=1    H = 256 / 384 / 512: backward recurrence with streamed W_hh tiles (64 x 64) instead of resident ones
//   TGRU_STEP_BWD_WIDE=1  H > 128: backward persistent kernel on 128-sequence x 32-unit tiles instead of 64 x 64
//   TGRU_SPIN_LIMIT=n     H > 128: polls after which a wait of a persistent kernel gives up (default 2^24)
//   TGRU_NO_H24=1         generic small-H MFMA kernels at 16 < H <= 24 (gru_small.hip instead of gru_h24.hip)
//   TGRU_NO_WIDE_HEAD=1   heads beyond 32 / 64 wide are not taken by the fused kernels of head_wide.hip
//   TGRU_GEMM_GUARDED=1   bounds-checked operand loads in gemm_kernel
//   TGRU_GEMM_NO_DMA=1    128 x 128 products on the register-staged kernel instead of the loader-wave kernel
bool use_small(int In, int H, int L) {
    if (H > kSmallMaxH || In > kSmallMaxIn || L > kSmallMaxL) return false;
    return !switches().force_generic;
}

// Small stacks with up to 1024 sequences in a padded layout run one sequence per workgroup on the
// vector ALU (gru_vec.hip).
bool use_vec(int B, int T, int In, int H, int L, int layout) {
    if (layout == TGRU_LAYOUT_PACKED) return false;
    if (!use_small(In, H, L) || !gru_vec_supported(B, T, In, H, L)) return false;
    return !switches().no_vec;
}

// 32 < H <= 128: recurrent kernels with four sequences per workgroup (gru_mid.hip).
bool use_mid(int H) { return gru_mid_supported(H) && !switches().no_mid; }

// H > 128 (multiples of 32): one launch per time step (gru_step.hip).
bool use_step(int B, int H) { return gru_step_supported(B, H) && !switches().no_step; }

// The kernels are built for gfx950 only.  The architecture of each device is looked up once
// (0 = not looked up yet, 1 = gfx950, -1 = anything else, or no device at all).
int device_ok() {
    constexpr int kMaxDev = 64;
    static std::atomic<signed char> known[kMaxDev];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        (void)hipGetLastError();
        return TGRU_E_NOGPU;
    }
    signed char k = known[dev].load(std::memory_order_relaxed);
    if (k == 0) {
        hipDeviceProp_t prop;
        k = -1;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) k = 1;
        else (void)hipGetLastError();
        known[dev].store(k, std::memory_order_relaxed);
    }
    return k == 1 ? TGRU_OK : TGRU_E_NOGPU;
}

// reserve: [L x 4 gate planes][(L-1) intermediate outputs][top output if layout != TBF]
size_t reserve_floats(const Dims &d) {
    size_t n = (size_t)d.L * 4 * align64(d.TBH());
    n += (size_t)(d.L - 1) * align64(d.TBH());
    if (d.layout != TGRU_LAYOUT_TBF) n += align64(d.TBH());
    return n;
}
float *reserve_gates(const Dims &d, float *r, int l) { return r + (size_t)l * 4 * align64(d.TBH()); }
float *reserve_y(const Dims &d, float *r, int l) {   // l < L-1, or l == L-1 when layout != TBF
    return r + (size_t)d.L * 4 * align64(d.TBH()) + (size_t)l * align64(d.TBH());
}

// forward workspace: gi | packed W_hh | two ping-pong layer outputs (inference without reserve)
size_t fwd_ws_floats(const Dims &d) {
    return align64(d.TB() * 3 * d.H) + align64(packed_whh_floats(d.H)) + 2 * align64(d.TBH()) + kStepStatusFloats;
}
// (the error word of the persistent H > 128 kernels sits behind the ping-pong buffers)
float *fwd_ws_status(const Dims &d, float *ws) {
    return ws + align64(d.TB() * 3 * d.H) + align64(packed_whh_floats(d.H)) + 2 * align64(d.TBH());
}


// backward workspace: dg | 2 x dy ping-pong | packed W_hh^T | bias partials | split-K slabs
struct BwdWs { size_t dg, dyb, wtp, dbp, slabs, carry, status, total; int max_slabs;
               size_t dg_stride, dbp_stride, slabs_stride; };      // per layer on the mid-H path (0: one buffer for all layers)
BwdWs bwd_ws(const Dims &d) {
    BwdWs w;
    size_t off = 0;
    // the whole-stack path keeps one dg buffer per layer (sized for it regardless of the env switch)
    const bool small_shape = d.H <= kSmallMaxH && d.In <= kSmallMaxIn && d.L <= kSmallMaxL;
    // the mid-H path keeps dg, the bias partials and the weight-gradient slabs of every layer until the end of the
    // call: the layers' weight gradients then go as one launch
    const bool mid_layers = !small_shape && gru_mid_supported(d.H) && d.L <= kMidWgradMaxLayers;
    w.dg_stride = mid_layers ? align64(d.TB() * 4 * d.H) : 0;
    w.dg = off; off += (small_shape || mid_layers ? (size_t)d.L : 1) * align64(d.TB() * 4 * d.H);
    w.dyb = off; off += 2 * align64(d.TB() * (size_t)(d.H > d.In ? d.H : d.In));
    w.wtp = off; off += align64(packed_whh_floats(d.H));
    size_t nblk = (size_t)gru_batch_blocks(d.B);
    const bool vec_shape = gru_vec_supported(d.B, d.T, d.In, d.H, d.L);
    if (vec_shape && (size_t)gru_vec_workgroups(d.B) > nblk) nblk = (size_t)gru_vec_workgroups(d.B);
    if (gru_mid_supported(d.H)) nblk = (size_t)d.B;      // those kernels leave one bias partial per sequence
    if (gru_step_supported(d.B, d.H) && (size_t)gru_step_bias_rows(d.B) > nblk) nblk = (size_t)gru_step_bias_rows(d.B);
    w.dbp_stride = mid_layers ? align64(nblk * 4 * d.H) : 0;
    w.dbp = off; off += mid_layers ? (size_t)d.L * w.dbp_stride : align64(nblk * 4 * d.H * (small_shape ? d.L : 1));
    int kc;
    const int widest = d.In > d.H ? d.In : d.H;
    int ns = gemm_split_for(3 * d.H, widest, (long)d.TB(), &kc);
    int ns2 = gemm_split_for(3 * d.H, d.H, (long)d.TB(), &kc);
    int ns3 = gemm_split_for(3 * d.H, d.In, (long)d.TB(), &kc);
    w.max_slabs = ns > ns2 ? ns : ns2;
    if (ns3 > w.max_slabs) w.max_slabs = ns3;
    size_t slab_floats = (size_t)w.max_slabs * 3 * d.H * widest;
    if (small_shape) {
        const size_t need = (size_t)small_wgrad_chunks(d.B, d.T) * small_wgrad_slab_floats(d.In, d.H, d.L);
        if (need > slab_floats) slab_floats = need;
    }
    if (vec_shape) {
        const size_t need = (size_t)gru_vec_workgroups(d.B) * small_wgrad_slab_floats(d.In, d.H, d.L);
        if (need > slab_floats) slab_floats = need;
    }
    if (gru_mid_supported(d.H)) {
        const int widest_in = d.In > d.H ? d.In : d.H;
        const size_t need = (size_t)gru_mid_wgrad_workgroups(d.B, d.T) * gru_mid_wgrad_slab_floats(widest_in, d.H);
        if (need > slab_floats) slab_floats = need;
    }
    w.slabs_stride = mid_layers ? align64(slab_floats) : 0;
    w.slabs = off; off += (mid_layers ? (size_t)d.L : 1) * align64(slab_floats);
    w.carry = off; off += gru_step_supported(d.B, d.H) ? align64((size_t)d.B * d.H) : 0;   // per-step kernels only
    w.status = off; off += kStepStatusFloats;      // error word of the persistent H > 128 kernels
    w.total = off;
    return w;
}

RowMap ext_map(const Dims &d, const int32_t *toff) {
    RowMap m;
    m.mode = d.layout; m.B = d.B; m.T = d.T; m.toff = toff;
    return m;
}
RowMap id_map(const Dims &d) {
    RowMap m;
    m.mode = 0; m.B = d.B; m.T = d.T; m.toff = nullptr;
    return m;
}

// Scratch memory of tgru_forward_static: an array inside the library's code object, one copy per GPU at
// its own address.  `scratch` itself is only a host-side handle: its device address is looked up once per
// device with hipGetSymbolAddress and every use goes through that pointer.
constexpr size_t kStaticFloats = TGRU_STATIC_WORKSPACE_BYTES / sizeof(float);
static __device__ __align__(256) float scratch[kStaticFloats];

int static_workspace(float **out) {
    constexpr int kMaxDev = 64;
    static std::atomic<float *> ptr[kMaxDev];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        (void)hipGetLastError();
        return TGRU_E_NOGPU;
    }
    float *p = ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        // first call on this device (which is current): resolve this GPU's copy of the array
        void *sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym) {
            (void)hipGetLastError();
            return TGRU_E_LAUNCH;
        }
        p = static_cast<float *>(sym);
        ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return TGRU_OK;
}

#define TGRU_HIP(expr)                                   \
    do {                                                 \
        if ((expr) != hipSuccess) return TGRU_E_LAUNCH;  \
    } while (0)

}  // namespace

extern "C" {

int tgru_version(void) { return 200; }

unsigned long long tgru_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

void tgru_reload_switches(void) { switches().read(); }

long tgru_persist_timeouts(int clear) {
    if (device_ok() != TGRU_OK) return -1;
    return gru_step_timeouts(clear);
}

const char *tgru_error_string(int code) {
    switch (code) {
    case TGRU_OK: return "ok";
    case TGRU_E_DIMS: return "bad dimensions (all must be > 0, hidden_size <= 512, B*T < 2^29)";
    case TGRU_E_NULL: return "a required pointer is NULL";
    case TGRU_E_LAYOUT: return "unknown layout, or packed layout without lengths/toff";
    case TGRU_E_ALIGN: return "a buffer is not 16-byte aligned";
    case TGRU_E_LAUNCH: return "a kernel launch failed";
    case TGRU_E_NOGPU: return "the current device is not a gfx950 GPU";
    case TGRU_E_TOOBIG: return "the call needs more scratch memory than tgru_forward_static holds "
                               "(TGRU_STATIC_WORKSPACE_BYTES); use tgru_forward with a workspace";
    default: return "unknown error";
    }
}

size_t tgru_reserve_bytes(int B, int T, int In, int H, int L, int layout) {
    if (check_dims(B, T, In, H, L, layout) != TGRU_OK) return 0;
    const Dims d{B, T, In, H, L, layout};
    return reserve_floats(d) * sizeof(float);
}

size_t tgru_forward_workspace_bytes(int B, int T, int In, int H, int L) {
    if (check_dims(B, T, In, H, L, TGRU_LAYOUT_TBF) != TGRU_OK) return 0;
    const Dims d{B, T, In, H, L, TGRU_LAYOUT_TBF};
    return fwd_ws_floats(d) * sizeof(float);
}

size_t tgru_backward_workspace_bytes(int B, int T, int In, int H, int L) {
    if (check_dims(B, T, In, H, L, TGRU_LAYOUT_TBF) != TGRU_OK) return 0;
    const Dims d{B, T, In, H, L, TGRU_LAYOUT_TBF};
    return bwd_ws(d).total * sizeof(float);
}

int tgru_input_projection(const float *a, const float *w, const float *bias, float *out, int M, int N,
                          int K, tgru_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TGRU_E_DIMS;
    if (!a || !w || !out) return TGRU_E_NULL;
    GemmArgs g{};
    g.A = a; g.lda = K; g.Bm = w; g.ldb = K; g.C = out; g.ldc = N; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.slab_stride = 0; g.accumulate = 0;
    g.amap = RowMap{0, 1, 1, nullptr};
    g.bmap = g.amap;
    g.cmap = g.amap;
    TGRU_HIP(launch_gemm(g, true, true, 1, (hipStream_t)stream));
    return TGRU_OK;
}

int tgru_linear_input_grad(const float *dy, const float *w, float *dx, int rows, int out_f, int in_f,
                           tgru_stream_t stream) {
    if (rows <= 0 || out_f <= 0 || in_f <= 0) return TGRU_E_DIMS;
    if (!dy || !w || !dx) return TGRU_E_NULL;
    // dx (rows, in_f) = dy (rows, out_f) . w (out_f, in_f): the reduction runs along dy's contiguous index and
    // down w's rows, the shape of the dX product of tgru_backward
    GemmArgs g{};
    g.A = dy; g.lda = out_f; g.Bm = w; g.ldb = in_f; g.C = dx; g.ldc = in_f; g.bias = nullptr;
    g.M = rows; g.N = in_f; g.K = out_f; g.kchunk = out_f; g.slab_stride = 0; g.accumulate = 0;
    g.amap = RowMap{0, 1, 1, nullptr};
    g.bmap = g.amap;
    g.cmap = g.amap;
    TGRU_HIP(launch_gemm(g, true, false, 1, (hipStream_t)stream));
    return TGRU_OK;
}

size_t tgru_linear_weight_grad_workspace_bytes(int rows, int out_f, int in_f) {
    if (rows <= 0 || out_f <= 0 || in_f <= 0) return 0;
    int kc;
    const int ns = gemm_split_for(out_f, in_f, rows, &kc);
    return (size_t)ns * out_f * in_f * sizeof(float);
}

int tgru_linear_weight_grad(const float *dy, const float *x, float *dw, float *workspace, int rows, int out_f,
                            int in_f, int accumulate, tgru_stream_t stream) {
    if (rows <= 0 || out_f <= 0 || in_f <= 0) return TGRU_E_DIMS;
    if (!dy || !x || !dw || !workspace) return TGRU_E_NULL;
    if (accumulate != 0 && accumulate != 1) return TGRU_E_DIMS;
    int kc;
    const int ns = gemm_split_for(out_f, in_f, rows, &kc);
    GemmArgs g{};
    g.A = dy; g.lda = out_f; g.Bm = x; g.ldb = in_f; g.C = workspace; g.ldc = in_f; g.bias = nullptr;
    g.M = out_f; g.N = in_f; g.K = rows; g.kchunk = kc; g.slab_stride = (long)out_f * in_f; g.accumulate = 0;
    g.amap = RowMap{0, 1, 1, nullptr};
    g.bmap = g.amap;
    g.cmap = g.amap;
    TGRU_HIP(launch_gemm(g, false, false, ns, (hipStream_t)stream));
    TGRU_HIP(launch_slab_reduce(workspace, g.slab_stride, ns, dw, (long)out_f * in_f, accumulate, (hipStream_t)stream));
    return TGRU_OK;
}

static int check_forward(const tgru_forward_args &c) {
    int rc = check_dims(c.B, c.T, c.In, c.H, c.L, c.layout);
    if (rc != TGRU_OK) return rc;
    if (!c.x || !c.y || !c.workspace || !c.w_ih || !c.w_hh || !c.b_ih || !c.b_hh) return TGRU_E_NULL;
    if (c.layout == TGRU_LAYOUT_PACKED && (!c.lengths || !c.toff)) return TGRU_E_LAYOUT;
    if (!aligned16(c.x) || !aligned16(c.y) || !aligned16(c.workspace) || (c.reserve && !aligned16(c.reserve)) ||
        (c.h_n && !aligned16(c.h_n)))
        return TGRU_E_ALIGN;
    for (int l = 0; l < c.L; ++l) {
        if (!c.w_ih[l] || !c.w_hh[l] || !c.b_ih[l] || !c.b_hh[l]) return TGRU_E_NULL;
        if (!aligned16(c.w_ih[l]) || !aligned16(c.w_hh[l]) || !aligned16(c.b_ih[l]) || !aligned16(c.b_hh[l]))
            return TGRU_E_ALIGN;
    }
    if (c.h0 && !aligned16(c.h0)) return TGRU_E_ALIGN;
    return device_ok();
}

// arguments of the whole-stack forward kernels for one call
static void fill_small_fwd(const tgru_forward_args &c, SmallFwdArgs &f) {
    const Dims d{c.B, c.T, c.In, c.H, c.L, c.layout};
    f.x = c.x; f.xmap = ext_map(d, c.toff); f.In = c.In;
    for (int l = 0; l < c.L; ++l) {
        f.w_ih[l] = c.w_ih[l]; f.w_hh[l] = c.w_hh[l]; f.b_ih[l] = c.b_ih[l]; f.b_hh[l] = c.b_hh[l];
        const bool top = l == c.L - 1;
        const bool need_int = !top || c.layout != TGRU_LAYOUT_TBF;
        f.y_int[l] = (c.reserve && need_int) ? reserve_y(d, c.reserve, l) : nullptr;
        f.gates[l] = c.reserve ? reserve_gates(d, c.reserve, l) : nullptr;
    }
    f.h0 = c.h0; f.y_ext = c.y; f.ymap = ext_map(d, c.toff); f.h_n = c.h_n; f.lengths = c.lengths;
    f.B = c.B; f.T = c.T; f.H = c.H; f.L = c.L;
}

int tgru_forward(const float *x, const float *h0, const int32_t *lengths, const int32_t *toff,
                 const float *const *w_ih, const float *const *w_hh, const float *const *b_ih,
                 const float *const *b_hh, float *y, float *h_n, float *reserve, float *workspace,
                 int B, int T, int In, int H, int L, int layout, tgru_stream_t stream) {
    const tgru_forward_args c{x, h0, lengths, toff, w_ih, w_hh, b_ih, b_hh, y, h_n, reserve, workspace,
                              B, T, In, H, L, layout};
    int rc = check_forward(c);
    if (rc != TGRU_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const Dims d{B, T, In, H, L, layout};
    float *gi = workspace;
    float *wp = gi + align64(d.TB() * 3 * H);
    float *pp = wp + align64(packed_whh_floats(H));

    if (use_small(In, H, L)) {
        SmallFwdArgs f{};
        fill_small_fwd(c, f);
        if (use_vec(B, T, In, H, L, layout)) TGRU_HIP(launch_gru_vec_fwd(f, nullptr, s));
        else TGRU_HIP(launch_gru_small_fwd(f, s));
        return TGRU_OK;
    }

    // persistent H > 128 kernels: the call's error word starts at zero
    unsigned *err = use_step(B, H) && gru_step_persistent(B, H) ? reinterpret_cast<unsigned *>(fwd_ws_status(d, workspace)) : nullptr;
    if (err) TGRU_HIP(gru_step_begin(err, s));
    const float *in = x;
    for (int l = 0; l < L; ++l) {
        const int In_l = l == 0 ? In : H;
        const bool top = l == L - 1;
        const bool mid = use_mid(H), stepk = use_step(B, H);
        FwdArgs f{};
        f.gi = gi; f.wp = wp; f.w_hh = w_hh[l]; f.b_hh = b_hh[l];
        f.x = in; f.xmap = l == 0 ? ext_map(d, toff) : id_map(d); f.In = In_l; f.w_ih = w_ih[l]; f.b_ih = b_ih[l];
        f.B = B; f.T = T; f.H = H;
        // kernels that compute gi themselves: the H = 96 kernel, and the persistent H > 128 kernel where it can
        const bool step_fused = stepk && err && gru_step_fwd_fuses(f);
        const bool fused = (mid && !stepk && gru_mid96_fwd_supported(f)) || step_fused;
        if (!fused) {
            GemmArgs g{};
            g.A = in; g.lda = In_l; g.Bm = w_ih[l]; g.ldb = In_l; g.C = gi; g.ldc = 3 * H; g.bias = b_ih[l];
            g.M = (int)d.TB(); g.N = 3 * H; g.K = In_l; g.kchunk = In_l; g.slab_stride = 0; g.accumulate = 0;
            g.amap = f.xmap;
            g.bmap = id_map(d);
            g.cmap = id_map(d);
            TGRU_HIP(launch_gemm(g, true, true, 1, s));
        }
        if (!mid && !stepk) TGRU_HIP(launch_pack_whh(w_hh[l], wp, nullptr, H, s));

        f.h0 = h0 ? h0 + (size_t)l * B * H : nullptr;
        const bool need_int = !top || layout != TGRU_LAYOUT_TBF;
        f.y_int = nullptr;
        if (need_int) f.y_int = reserve ? reserve_y(d, reserve, l) : pp + (size_t)(l & 1) * align64(d.TBH());
        if (top && !reserve) f.y_int = nullptr;   // inference: nobody reads the top layer's copy
        // the per-step kernels read h_{t-1} back from a time-major output
        if (stepk && !f.y_int && layout != TGRU_LAYOUT_TBF) f.y_int = pp + (size_t)(l & 1) * align64(d.TBH());
        f.y_ext = top ? y : nullptr;
        f.ymap = ext_map(d, toff);
        f.gates = reserve ? reserve_gates(d, reserve, l) : nullptr;
        f.h_n = h_n ? h_n + (size_t)l * B * H : nullptr;
        f.lengths = lengths;
        f.B = B; f.T = T; f.H = H;
        // (the packed-W_hh region of the workspace is free on this path: it holds the sync words)
        if (stepk) TGRU_HIP(launch_gru_step_fwd(f, reinterpret_cast<unsigned *>(wp), err, s));
        else if (fused && !step_fused) TGRU_HIP(launch_gru_mid96_fwd(f, s));
        else if (mid) TGRU_HIP(launch_gru_mid_fwd(f, s));
        else TGRU_HIP(launch_gru_fwd(f, s));
        in = f.y_int;
    }
    if (err) {     // a wait that gave up: NaN at the head of y and h_n, one more in tgru_persist_timeouts()
        float *outs[2] = {y, h_n};
        const int ns[2] = {H, H};
        TGRU_HIP(gru_step_check(err, outs, ns, 2, s));
    }
    return TGRU_OK;
}

int tgru_forward_static(const float *x, const float *h0, const int32_t *lengths, const int32_t *toff,
                        const float *const *w_ih, const float *const *w_hh, const float *const *b_ih,
                        const float *const *b_hh, float *y, float *h_n, float *reserve, int B, int T, int In,
                        int H, int L, int layout, tgru_stream_t stream) {
    // the argument checks of tgru_forward first (with a stand-in for the workspace pointer), then the size
    float *const standin = reinterpret_cast<float *>(uintptr_t(256));
    const tgru_forward_args c{x, h0, lengths, toff, w_ih, w_hh, b_ih, b_hh, y, h_n, reserve, standin,
                              B, T, In, H, L, layout};
    int rc = check_forward(c);
    if (rc != TGRU_OK) return rc;
    const Dims d{B, T, In, H, L, layout};
    if (fwd_ws_floats(d) > kStaticFloats) return TGRU_E_TOOBIG;
    float *ws = nullptr;
    rc = static_workspace(&ws);
    if (rc != TGRU_OK) return rc;
    // nothing in the forward workspace has to start at zero: every float is written before it is read
    return tgru_forward(x, h0, lengths, toff, w_ih, w_hh, b_ih, b_hh, y, h_n, reserve, ws, B, T, In, H, L,
                        layout, stream);
}

static bool pair_on_vec(int B0, int T0, int In0, int H0, int L0, int lay0, int B1, int T1, int In1, int H1, int L1,
                        int lay1) {
    return use_vec(B0, T0, In0, H0, L0, lay0) && use_vec(B1, T1, In1, H1, L1, lay1) &&
           gru_vec_pairable(In0, H0, L0, In1, H1, L1);
}

int tgru_forward_pair(const tgru_forward_args *a, const tgru_forward_args *b, tgru_stream_t stream) {
    if (!a || !b) return TGRU_E_NULL;
    int rc = check_forward(*a);
    if (rc == TGRU_OK) rc = check_forward(*b);
    if (rc != TGRU_OK) return rc;
    if (!pair_on_vec(a->B, a->T, a->In, a->H, a->L, a->layout, b->B, b->T, b->In, b->H, b->L, b->layout)) {
        rc = tgru_forward(a->x, a->h0, a->lengths, a->toff, a->w_ih, a->w_hh, a->b_ih, a->b_hh, a->y, a->h_n,
                          a->reserve, a->workspace, a->B, a->T, a->In, a->H, a->L, a->layout, stream);
        if (rc != TGRU_OK) return rc;
        return tgru_forward(b->x, b->h0, b->lengths, b->toff, b->w_ih, b->w_hh, b->b_ih, b->b_hh, b->y, b->h_n,
                            b->reserve, b->workspace, b->B, b->T, b->In, b->H, b->L, b->layout, stream);
    }
    SmallFwdArgs fa{}, fb{};
    fill_small_fwd(*a, fa);
    fill_small_fwd(*b, fb);
    TGRU_HIP(launch_gru_vec_fwd(fa, &fb, (hipStream_t)stream));
    return TGRU_OK;
}

static int check_backward(const tgru_backward_args &c) {
    int rc = check_dims(c.B, c.T, c.In, c.H, c.L, c.layout);
    if (rc != TGRU_OK) return rc;
    if (!c.x || !c.y || !c.reserve || !c.workspace || !c.w_ih || !c.w_hh) return TGRU_E_NULL;
    // all four gradient arrays NULL: the caller updates none of these parameters, only dx / dh0 are formed
    const bool want_w = c.dw_ih || c.dw_hh || c.db_ih || c.db_hh;
    if (want_w && (!c.dw_ih || !c.dw_hh || !c.db_ih || !c.db_hh)) return TGRU_E_NULL;
    if (c.layout == TGRU_LAYOUT_PACKED && (!c.lengths || !c.toff)) return TGRU_E_LAYOUT;
    if (!aligned16(c.x) || !aligned16(c.y) || !aligned16(c.workspace) || !aligned16(c.reserve) ||
        (c.dy && !aligned16(c.dy)) || (c.dx && !aligned16(c.dx)))
        return TGRU_E_ALIGN;
    for (int l = 0; l < c.L; ++l) {
        if (!c.w_ih[l] || !c.w_hh[l]) return TGRU_E_NULL;
        if (want_w && (!c.dw_ih[l] || !c.dw_hh[l] || !c.db_ih[l] || !c.db_hh[l])) return TGRU_E_NULL;
        if (!aligned16(c.w_ih[l]) || !aligned16(c.w_hh[l])) return TGRU_E_ALIGN;
    }
    if ((c.h0 && !aligned16(c.h0)) || (c.dh_n && !aligned16(c.dh_n)) || (c.dh0 && !aligned16(c.dh0)))
        return TGRU_E_ALIGN;
    return device_ok();
}

// arguments of the whole-stack backward kernels (b) and of the weight-gradient stage (g) for one call;
// `vec`: the one-sequence-per-workgroup kernels, which leave one partial gradient per workgroup
static void fill_small_bwd(const tgru_backward_args &c, bool vec, SmallBwdArgs &b, SmallWgradArgs &g) {
    const Dims d{c.B, c.T, c.In, c.H, c.L, c.layout};
    const BwdWs w = bwd_ws(d);
    float *dg = c.workspace + w.dg;
    float *dbp = c.workspace + w.dbp;
    float *slabs = c.workspace + w.slabs;
    float *res = const_cast<float *>(c.reserve);
    const size_t dg_stride = align64(d.TB() * 4 * c.H);
    for (int l = 0; l < c.L; ++l) {
        const bool top = l == c.L - 1;
        const float *y_l = (top && c.layout == TGRU_LAYOUT_TBF) ? c.y : reserve_y(d, res, l);
        b.w_ih[l] = c.w_ih[l]; b.w_hh[l] = c.w_hh[l];
        b.y_int[l] = y_l; b.gates[l] = reserve_gates(d, res, l);
        b.dg[l] = dg + (size_t)l * dg_stride;
        g.dg[l] = b.dg[l]; g.y_int[l] = y_l;
        if (c.dw_ih) {
            g.dw_ih[l] = c.dw_ih[l]; g.dw_hh[l] = c.dw_hh[l]; g.db_ih[l] = c.db_ih[l]; g.db_hh[l] = c.db_hh[l];
        }
    }
    b.dy_ext = c.dy; b.dymap = ext_map(d, c.toff); b.dh_n = c.dh_n; b.h0 = c.h0;
    b.dx_ext = c.dx; b.dxmap = ext_map(d, c.toff); b.In = c.In; b.dh0 = c.dh0; b.db_part = dbp;
    b.lengths = c.lengths; b.B = c.B; b.T = c.T; b.H = c.H; b.L = c.L;
    g.x = c.x; g.xmap = ext_map(d, c.toff); g.In = c.In; g.h0 = c.h0; g.db_part = dbp;
    g.B = c.B; g.T = c.T; g.H = c.H; g.L = c.L; g.accumulate = c.accumulate == TGRU_DEFER_REDUCE ? 0 : c.accumulate;
    g.slabs = slabs; g.slab_stride = small_wgrad_slab_floats(c.In, c.H, c.L);
    if (vec) {
        b.x = c.x; b.xmap = ext_map(d, c.toff); b.slabs = c.dw_ih ? slabs : nullptr; b.slab_stride = g.slab_stride;
        g.nblk = gru_vec_workgroups(c.B); g.nchunks = g.nblk;
    } else {
        g.nblk = gru_batch_blocks(c.B); g.nchunks = small_wgrad_chunks(c.B, c.T);
    }
}

int tgru_backward(const float *dy, const float *dh_n, const float *x, const float *h0, const float *y,
                  const int32_t *lengths, const int32_t *toff, const float *const *w_ih,
                  const float *const *w_hh, const float *const *b_ih, const float *const *b_hh,
                  const float *reserve, float *workspace, float *dx, float *dh0, float *const *dw_ih,
                  float *const *dw_hh, float *const *db_ih, float *const *db_hh, int B, int T, int In,
                  int H, int L, int layout, int accumulate, tgru_stream_t stream) {
    const tgru_backward_args c{dy, dh_n, x, h0, y, lengths, toff, w_ih, w_hh, b_ih, b_hh, reserve, workspace,
                               dx, dh0, dw_ih, dw_hh, db_ih, db_hh, B, T, In, H, L, layout, accumulate};
    int rc = check_backward(c);
    if (rc != TGRU_OK) return rc;
    const bool defer = accumulate == TGRU_DEFER_REDUCE;   // honoured by the one-sequence-per-workgroup path only
    if (defer) accumulate = 0;
    const bool want_w = dw_ih != nullptr;      // false: no weight / bias gradients (their stages are not launched)
    hipStream_t s = (hipStream_t)stream;
    const Dims d{B, T, In, H, L, layout};
    const BwdWs w = bwd_ws(d);
    float *wtp = workspace + w.wtp;
    const size_t dyb_stride = align64(d.TB() * (size_t)(H > In ? H : In));
    float *res = const_cast<float *>(reserve);
    const int nblk = gru_batch_blocks(B);
    const int TB = (int)d.TB();

    if (use_small(In, H, L)) {
        SmallBwdArgs b{};
        SmallWgradArgs g{};
        const bool vec = use_vec(B, T, In, H, L, layout);
        fill_small_bwd(c, vec, b, g);
        if (vec) {
            // the backward kernel accumulates the weight gradients itself: one partial per workgroup
            TGRU_HIP(launch_gru_vec_bwd(b, nullptr, s));
            if (!defer && want_w) TGRU_HIP(launch_gru_small_wgrad_reduce(g, nullptr, s));
            return TGRU_OK;
        }
        TGRU_HIP(launch_gru_small_bwd(b, s));
        if (want_w) TGRU_HIP(launch_gru_small_wgrad(g, s));
        return TGRU_OK;
    }

    // persistent H > 128 kernels: the call's error word starts at zero
    unsigned *err = use_step(B, H) && gru_step_persistent(B, H) ? reinterpret_cast<unsigned *>(workspace + w.status) : nullptr;
    if (err) TGRU_HIP(gru_step_begin(err, s));
    MidWgradJob wjobs[kMidWgradMaxLayers];      // mid-H layers whose weight gradients wait for the end of the call
    int nwjobs = 0;
    for (int l = L - 1; l >= 0; --l) {
        const int In_l = l == 0 ? In : H;
        // (per-layer buffers on the mid-H path, the shared ones elsewhere)
        float *dg = workspace + w.dg + (size_t)l * w.dg_stride;
        float *dbp = workspace + w.dbp + (size_t)l * w.dbp_stride;
        float *slabs = workspace + w.slabs + (size_t)l * w.slabs_stride;
        const bool top = l == L - 1;
        const float *y_l = (top && layout == TGRU_LAYOUT_TBF) ? y : reserve_y(d, res, l);
        const float *in = l == 0 ? x : reserve_y(d, res, l - 1);
        // the four-sequence kernels read the caller's dy directly only in the padded layouts
        const bool mid = use_mid(H) && !(top && layout == TGRU_LAYOUT_PACKED);
        const bool stepk = use_step(B, H);
        if (!mid && !stepk) TGRU_HIP(launch_pack_whh(w_hh[l], nullptr, wtp, H, s));

        BwdArgs b{};
        b.dy_int = top ? nullptr : workspace + w.dyb + (size_t)((l + 1) & 1) * dyb_stride;
        b.dy_ext = top ? dy : nullptr;
        b.dymap = ext_map(d, toff);
        b.dh_n = dh_n ? dh_n + (size_t)l * B * H : nullptr;
        b.wtp = wtp; b.w_hh = w_hh[l]; b.y_int = y_l;
        b.h0 = h0 ? h0 + (size_t)l * B * H : nullptr;
        b.gates = reserve_gates(d, res, l);
        b.dg = dg;
        b.dh0 = dh0 ? dh0 + (size_t)l * B * H : nullptr;
        b.db_part = dbp;
        b.lengths = lengths;
        b.B = B; b.T = T; b.H = H;
        // the gradient of the layer's input: the layer below's dy (or the caller's dx)
        float *dx_l = l == 0 ? dx : workspace + w.dyb + (size_t)(l & 1) * dyb_stride;
        b.w_ih = w_ih[l]; b.In = In_l; b.dx = dx_l; b.dxmap = l == 0 ? ext_map(d, toff) : id_map(d);
        const bool fused_dx = mid && !stepk && gru_mid96_bwd_supported(b);    // that kernel forms dX itself
        if (fused_dx && !want_w) b.dg = nullptr;      // neither dX nor a weight-gradient stage will read dg
        if (stepk) TGRU_HIP(launch_gru_step_bwd(b, workspace + w.carry, err, s));
        else if (fused_dx) TGRU_HIP(launch_gru_mid96_bwd(b, s));
        else if (mid) TGRU_HIP(launch_gru_mid_bwd(b, s));
        else TGRU_HIP(launch_gru_bwd(b, s));
        const bool mid_wgrad = mid && gru_mid_wgrad_supported(B, T, In_l, H) &&
                               !(l == 0 && layout == TGRU_LAYOUT_PACKED);
        const int db_rows = stepk ? gru_step_bias_rows(B) : mid ? B : nblk;
        if (!want_w) {
            // nothing: dg only feeds dX below
        } else if (mid_wgrad) {       // the weight-gradient reduction sums the bias partials in the same launch
            const MidWgradJob job{dg, in, In_l, l == 0 && layout == TGRU_LAYOUT_BTF, y_l,
                                  h0 ? h0 + (size_t)l * B * H : nullptr, slabs, dw_ih[l], dw_hh[l],
                                  dbp, db_rows, db_ih[l], db_hh[l]};
            // with buffers of its own the layer joins the stack's one launch after the loop (same input width only)
            if (w.dg_stride && (nwjobs == 0 || wjobs[0].In == In_l)) wjobs[nwjobs++] = job;
            else TGRU_HIP(launch_gru_mid_wgrad_batch(&job, 1, B, T, H, accumulate, s));
        } else {
            TGRU_HIP(launch_bias_grad(dbp, db_rows, H, db_ih[l], db_hh[l], accumulate, s));
        }
        // dW_ih = dgi^T . in   (rows of dg / in are the reduction dimension)
        if (want_w && !mid_wgrad) {
            int kc;
            const int ns = gemm_split_for(3 * H, In_l, TB, &kc);
            GemmArgs g{};
            g.A = dg; g.lda = 4 * H; g.Bm = in; g.ldb = In_l; g.C = slabs; g.ldc = In_l; g.bias = nullptr;
            g.M = 3 * H; g.N = In_l; g.K = TB; g.kchunk = kc; g.slab_stride = (long)3 * H * In_l;
            g.accumulate = 0;
            g.amap = id_map(d); g.cmap = id_map(d);
            g.bmap = l == 0 ? ext_map(d, toff) : id_map(d);
            TGRU_HIP(launch_gemm(g, false, false, ns, s));
            TGRU_HIP(launch_slab_reduce(slabs, g.slab_stride, ns, dw_ih[l], (long)3 * H * In_l, accumulate, s));
        }
        // dW_hh = dgh^T . h_prev : dg rows of step t pair with this layer's output rows of step t-1
        if (want_w && !mid_wgrad) {
            int kc;
            const int K = TB - B;
            const int ns = gemm_split_for(3 * H, H, K > 0 ? K : 1, &kc);
            GemmArgs g{};
            g.lda = 4 * H; g.Bm = y_l; g.ldb = H; g.ldc = H; g.bias = nullptr;
            g.N = H; g.K = K; g.kchunk = kc; g.slab_stride = (long)3 * H * H; g.accumulate = 0;
            g.amap = id_map(d); g.bmap = id_map(d); g.cmap = id_map(d);
            g.A = dg + (size_t)B * 4 * H; g.M = 2 * H; g.C = slabs;                 // r, z rows
            TGRU_HIP(launch_gemm(g, false, false, ns, s));
            g.A = dg + (size_t)B * 4 * H + 3 * H; g.M = H; g.C = slabs + (size_t)2 * H * H;   // n rows: da_n * r
            TGRU_HIP(launch_gemm(g, false, false, ns, s));
            TGRU_HIP(launch_slab_reduce(slabs, g.slab_stride, ns, dw_hh[l], (long)3 * H * H, accumulate, s));
            if (h0) {   // step 0 pairs with the initial state
                g.Bm = h0 + (size_t)l * B * H; g.K = B; g.kchunk = (B + 15) / 16 * 16; g.slab_stride = 0;
                g.accumulate = 1;
                g.A = dg; g.M = 2 * H; g.C = dw_hh[l];
                TGRU_HIP(launch_gemm(g, false, false, 1, s));
                g.A = dg + 3 * H; g.M = H; g.C = dw_hh[l] + (size_t)2 * H * H;
                TGRU_HIP(launch_gemm(g, false, false, 1, s));
            }
        }
        // dX = dgi . W_ih  -> gradient of the layer below (or the caller's dx)
        if ((l > 0 || dx) && !fused_dx) {
            GemmArgs g{};
            g.A = dg; g.lda = 4 * H; g.Bm = w_ih[l]; g.ldb = In_l; g.bias = nullptr;
            g.C = dx_l;
            g.ldc = In_l; g.M = TB; g.N = In_l; g.K = 3 * H; g.kchunk = 3 * H; g.slab_stride = 0;
            g.accumulate = 0;
            g.amap = id_map(d); g.bmap = id_map(d);
            g.cmap = l == 0 ? ext_map(d, toff) : id_map(d);
            TGRU_HIP(launch_gemm(g, true, false, 1, s));
        }
    }
    if (nwjobs) TGRU_HIP(launch_gru_mid_wgrad_batch(wjobs, nwjobs, B, T, H, accumulate, s));
    if (err) {     // a wait that gave up: NaN at the head of every gradient of the call
        float *outs[8] = {dx, dh0, want_w ? dw_ih[0] : nullptr, want_w ? dw_hh[0] : nullptr,
                          want_w ? db_ih[0] : nullptr, want_w ? db_hh[0] : nullptr,
                          want_w ? dw_ih[L - 1] : nullptr, want_w ? dw_hh[L - 1] : nullptr};
        const int ns[8] = {In, H, In, H, H, H, H, H};
        TGRU_HIP(gru_step_check(err, outs, ns, 8, s));
    }
    return TGRU_OK;
}

int tgru_backward_pair(const tgru_backward_args *a, const tgru_backward_args *b, tgru_stream_t stream) {
    if (!a || !b) return TGRU_E_NULL;
    int rc = check_backward(*a);
    if (rc == TGRU_OK) rc = check_backward(*b);
    if (rc != TGRU_OK) return rc;
    if (!pair_on_vec(a->B, a->T, a->In, a->H, a->L, a->layout, b->B, b->T, b->In, b->H, b->L, b->layout)) {
        const tgru_backward_args *both[2] = {a, b};
        for (const tgru_backward_args *c : both) {
            rc = tgru_backward(c->dy, c->dh_n, c->x, c->h0, c->y, c->lengths, c->toff, c->w_ih, c->w_hh, c->b_ih,
                               c->b_hh, c->reserve, c->workspace, c->dx, c->dh0, c->dw_ih, c->dw_hh, c->db_ih,
                               c->db_hh, c->B, c->T, c->In, c->H, c->L, c->layout, c->accumulate, stream);
            if (rc != TGRU_OK) return rc;
        }
        return TGRU_OK;
    }
    SmallBwdArgs ba{}, bb{};
    SmallWgradArgs ga{}, gb{};
    fill_small_bwd(*a, true, ba, ga);
    fill_small_bwd(*b, true, bb, gb);
    hipStream_t s = (hipStream_t)stream;
    TGRU_HIP(launch_gru_vec_bwd(ba, &bb, s));
    // a call whose reduction is deferred, or that wants no weight gradients, takes no part in the reduce launch
    const bool da = a->accumulate == TGRU_DEFER_REDUCE || !a->dw_ih, db = b->accumulate == TGRU_DEFER_REDUCE || !b->dw_ih;
    if (!da && !db) TGRU_HIP(launch_gru_small_wgrad_reduce(ga, &gb, s));
    else if (!da) TGRU_HIP(launch_gru_small_wgrad_reduce(ga, nullptr, s));
    else if (!db) TGRU_HIP(launch_gru_small_wgrad_reduce(gb, nullptr, s));
    return TGRU_OK;
}

int tgru_backward_reduce_jobs(const tgru_backward_args *c, tgru_reduce_job *jobs, int max_jobs) {
    if (!c || !jobs) return TGRU_E_NULL;
    int rc = check_backward(*c);
    if (rc != TGRU_OK) return rc;
    if (c->accumulate != TGRU_DEFER_REDUCE || !c->dw_ih || !use_vec(c->B, c->T, c->In, c->H, c->L, c->layout)) return 0;
    if (max_jobs < 5 * c->L) return TGRU_E_DIMS;
    SmallBwdArgs b{};
    SmallWgradArgs g{};
    fill_small_bwd(*c, true, b, g);
    const int H = c->H, L = c->L;
    int n = 0;
    long off = 0;
    for (int l = 0; l < L; ++l) {
        const int n_ih = 3 * H * (l == 0 ? c->In : H), n_hh = 3 * H * H;
        jobs[n++] = tgru_reduce_job{g.slabs + off, g.slab_stride, g.nchunks, n_ih, g.dw_ih[l]};
        jobs[n++] = tgru_reduce_job{g.slabs + off + n_ih, g.slab_stride, g.nchunks, n_hh, g.dw_hh[l]};
        off += n_ih + n_hh;
        // bias partials: one row (r | z | n | n*r) per workgroup and layer
        const float *bp = g.db_part + (long)l * 4 * H;
        const int64_t bs = (int64_t)L * 4 * H;
        jobs[n++] = tgru_reduce_job{bp, bs, g.nblk, 3 * H, g.db_ih[l]};
        jobs[n++] = tgru_reduce_job{bp, bs, g.nblk, 2 * H, g.db_hh[l]};
        jobs[n++] = tgru_reduce_job{bp + 3 * H, bs, g.nblk, H, g.db_hh[l] + 2 * H};
    }
    return n;
}

}  // extern "C"
