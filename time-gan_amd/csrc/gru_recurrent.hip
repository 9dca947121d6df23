// Recurrent part of the GRU on gfx950: one launch per layer, the time loop runs inside the kernel.
//
// Work split: one workgroup owns 16 sequences (the N dimension of v_mfma_f32_16x16x4_f32); its
// waves split the hidden units in tiles of 16.  Per step the forward kernel computes
//     gh^T (3H x 16) = W_hh (3H x H) . h_{t-1}^T (H x 16)
// with W_hh as the MFMA A operand (pre-packed into fragment order, resident in LDS when it fits,
// streamed from L2 otherwise) and h as the B operand.  The D fragment of hidden tile q
// (lane = 16*g + b holds hidden 16q+4g..16q+4g+3 of sequence b) is exactly the B fragment of
// k-block q of the next step, so h moves between steps as one lane-linear 16-byte LDS store and
// load per lane, with no transposition and no bank conflicts.  The gate sigmoid / tanh, the
// hidden update and the stores of y and of the saved activations are fused behind the MFMAs.
//
// The backward kernel walks time in reverse with the same ownership: the elementwise gate
// gradients of a step are written to LDS in fragment order and multiplied by W_hh^T to give the
// hidden-state gradient carried to step t-1.  Everything that is not serial in t (dX, dW_ih,
// dW_hh) is left to batched GEMMs over the dg buffer this kernel writes.
#include "tgru_internal.h"

namespace tgru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// see gru_small.hip: pads the MFMA-result hazard across control flow
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

__device__ __forceinline__ void lds_barrier() {
    // one asm statement with a memory clobber: the compiler must not move LDS accesses across it
    // (the s_barrier builtin alone is not a compiler-level memory barrier)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__host__ __device__ inline int hidden_tiles(int H) { return (H + 15) / 16; }

size_t packed_whh_floats(int H) {
    const size_t n = hidden_tiles(H);
    return 3 * n * n * 256;
}

// wp [gate][tile][q][lane][4]:  W_hh[gate*H + 16*tile + (lane&15)][16*q + 4*(lane>>4) + c]
// wtp[tile][gate*nHT + q][lane][4]:  W_hh[gate*H + 16*q + 4*(lane>>4) + c][16*tile + (lane&15)]
__global__ __launch_bounds__(256) void pack_whh_kernel(const float *__restrict__ w, float *wp, float *wtp,
                                                       int H) {
    const int nHT = hidden_tiles(H);
    const long total = 3L * nHT * nHT * 256;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = idx & 3;
        const int lane = (idx >> 2) & 63;
        long rest = idx >> 8;
        const int i = lane & 15, gq = lane >> 4;
        if (wp) {
            const int q = rest % nHT;
            const int tile = (rest / nHT) % nHT;
            const int gate = rest / ((long)nHT * nHT);
            const int row = 16 * tile + i, col = 16 * q + 4 * gq + c;
            wp[idx] = (row < H && col < H) ? w[(long)(gate * H + row) * H + col] : 0.f;
        }
        if (wtp) {
            const int qq = rest % (3 * nHT);
            const int tile = rest / (3 * nHT);
            const int gate = qq / nHT, q = qq % nHT;
            const int krow = 16 * q + 4 * gq + c, out = 16 * tile + i;
            wtp[idx] = (krow < H && out < H) ? w[(long)(gate * H + krow) * H + out] : 0.f;
        }
    }
}

hipError_t launch_pack_whh(const float *w_hh, float *wp, float *wtp, int H, hipStream_t s) {
    const long total = (long)packed_whh_floats(H);
    const int blocks = (int)min((total + 255) / 256, (long)2048);
    TGRU_LAUNCH(pack_whh_kernel, dim3(blocks), dim3(256), 0, s, w_hh, wp, wtp, H);
    return hipGetLastError();
}

int gru_batch_blocks(int B) { return (B + 15) / 16; }

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int TPW, bool WLDS>
__global__ __launch_bounds__(512) void gru_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T;
    const int nHT = hidden_tiles(H);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: tile offsets stay in SGPRs
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const bool vec = (H & 3) == 0;

    // LDS carve: h fragments (2 buffers of nHT*256 floats), then the packed weights
    float *lds_h = smem;
    float *lds_w = smem + 2 * nHT * 256;
    if (WLDS) {
        const long total = 3L * nHT * nHT * 256;
        for (long i = threadIdx.x * 4L; i < total; i += blockDim.x * 4L)
            *reinterpret_cast<float4 *>(lds_w + i) = *reinterpret_cast<const float4 *>(a.wp + i);
    }
    const float *wsrc = WLDS ? lds_w : a.wp;

    // per-lane state: hidden values of the owned tiles, recurrent biases
    float hreg[TPW][4];
    // recurrent biases: registers when one or two tiles per wave, re-read (cache hits) per step otherwise
    constexpr bool kBiasRegs = TPW <= 2;
    float bh[kBiasRegs ? TPW : 1][3][4];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave * TPW + i;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 16 * tile + 4 * gq + c;
            const bool ok = tile < nHT && j < H;
            hreg[i][c] = (ok && bvalid && a.h0) ? a.h0[(long)bglob * H + j] : 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (kBiasRegs) bh[kBiasRegs ? i : 0][g][c] = ok ? a.b_hh[g * H + j] : 0.f;
        }
        if (tile < nHT)
            *reinterpret_cast<float4 *>(lds_h + (tile * 64 + lane) * 4) =
                make_float4(hreg[i][0], hreg[i][1], hreg[i][2], hreg[i][3]);
    }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const bool active = t < len;
        const long row = (long)t * B + bglob;
        f32x4 acc[TPW][3];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const float *hcur = lds_h + cur * nHT * 256;
        // one k-block: the three gate fragments of every owned tile times the h fragment
        auto load_w = [&](int q, float4 (&wf)[TPW][3]) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = min(wave * TPW + i, nHT - 1);
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    wf[i][g] = *reinterpret_cast<const float4 *>(
                        wsrc + ((((long)g * nHT + tile) * nHT + q) * 64 + lane) * 4);
            }
        };
        auto block = [&](int q, const float4 (&wf)[TPW][3]) {
            const float4 hf = *reinterpret_cast<const float4 *>(hcur + (q * 64 + lane) * 4);
            const float hv[4] = {hf.x, hf.y, hf.z, hf.w};
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (wave * TPW + i < nHT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const float wv = c == 0 ? wf[i][g].x : c == 1 ? wf[i][g].y : c == 2 ? wf[i][g].z : wf[i][g].w;
                            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, hv[c], acc[i][g], 0, 0, 0);
                        }
                    }
                }
            }
        };
        if (WLDS) {
            for (int q = 0; q < nHT; ++q) {
                float4 wf[TPW][3];
                load_w(q, wf);
                block(q, wf);
            }
        } else {
            // weights streamed from L2: the fragments of k-block q+1 are in flight while q multiplies
            float4 w0[TPW][3], w1[TPW][3];
            load_w(0, w0);
            for (int q = 0; q < nHT; q += 2) {
                if (q + 1 < nHT) load_w(q + 1, w1);
                block(q, w0);
                if (q + 2 < nHT) load_w(q + 2, w0);
                if (q + 1 < nHT) block(q + 1, w1);
            }
        }

        // input projection of this step for the owned hidden units (after the k loop: held across it, these 12 registers per tile spill the streamed-weight variant)
        float gi[TPW][3][4];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave * TPW + i;
            const int j0 = 16 * tile + 4 * gq;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (active && tile < nHT && vec && j0 < H) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.gi + row * 3 * H + g * H + j0);
                    gi[i][g][0] = v.x; gi[i][g][1] = v.y; gi[i][g][2] = v.z; gi[i][g][3] = v.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        gi[i][g][c] = (active && tile < nHT && j0 + c < H)
                                          ? a.gi[row * 3 * H + g * H + j0 + c] : 0.f;
                }
            }
        }

        mfma_drain();
        float *hnxt = lds_h + (cur ^ 1) * nHT * 256;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave * TPW + i;
            if (tile >= nHT) continue;
            const int j0 = 16 * tile + 4 * gq;
            float yv[4], rv[4], zv[4], nv[4], hnv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool jok = j0 + c < H;
                const float b_r = kBiasRegs ? bh[kBiasRegs ? i : 0][0][c] : (jok ? a.b_hh[j0 + c] : 0.f);
                const float b_z = kBiasRegs ? bh[kBiasRegs ? i : 0][1][c] : (jok ? a.b_hh[H + j0 + c] : 0.f);
                const float b_n = kBiasRegs ? bh[kBiasRegs ? i : 0][2][c] : (jok ? a.b_hh[2 * H + j0 + c] : 0.f);
                const float ghn = acc[i][2][c] + b_n;
                const float r = fast_sigmoid(gi[i][0][c] + acc[i][0][c] + b_r);
                const float z = fast_sigmoid(gi[i][1][c] + acc[i][1][c] + b_z);
                const float n = fast_tanh(gi[i][2][c] + r * ghn);
                const float hnew = (1.f - z) * n + z * hreg[i][c];
                rv[c] = r; zv[c] = z; nv[c] = n; hnv[c] = ghn;
                if (active && j0 + c < H) hreg[i][c] = hnew;
                yv[c] = active ? hreg[i][c] : 0.f;
            }
            *reinterpret_cast<float4 *>(hnxt + (tile * 64 + lane) * 4) =
                make_float4(hreg[i][0], hreg[i][1], hreg[i][2], hreg[i][3]);
            if (bvalid && j0 < H) {
                long erow = -1;
                if (a.y_ext) erow = map_row(a.ymap, (int)row);
                if (vec) {
                    const float4 y4 = make_float4(yv[0], yv[1], yv[2], yv[3]);
                    if (a.y_int) *reinterpret_cast<float4 *>(a.y_int + row * H + j0) = y4;
                    if (erow >= 0) *reinterpret_cast<float4 *>(a.y_ext + erow * H + j0) = y4;
                    if (a.gates && active) {
                        const long plane = (long)T * B * H;
                        float *gp = a.gates + row * H + j0;
                        *reinterpret_cast<float4 *>(gp) = make_float4(rv[0], rv[1], rv[2], rv[3]);
                        *reinterpret_cast<float4 *>(gp + plane) = make_float4(zv[0], zv[1], zv[2], zv[3]);
                        *reinterpret_cast<float4 *>(gp + 2 * plane) = make_float4(nv[0], nv[1], nv[2], nv[3]);
                        *reinterpret_cast<float4 *>(gp + 3 * plane) = make_float4(hnv[0], hnv[1], hnv[2], hnv[3]);
                    }
                    if (a.h_n && t == len - 1)
                        *reinterpret_cast<float4 *>(a.h_n + (long)bglob * H + j0) = y4;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (j0 + c >= H) break;
                        if (a.y_int) a.y_int[row * H + j0 + c] = yv[c];
                        if (erow >= 0) a.y_ext[erow * H + j0 + c] = yv[c];
                        if (a.gates && active) {
                            const long plane = (long)T * B * H;
                            float *gp = a.gates + row * H + j0 + c;
                            gp[0] = rv[c]; gp[plane] = zv[c]; gp[2 * plane] = nv[c]; gp[3 * plane] = hnv[c];
                        }
                        if (a.h_n && t == len - 1) a.h_n[(long)bglob * H + j0 + c] = yv[c];
                    }
                }
            }
        }
        lds_barrier();
        cur ^= 1;
    }
}

struct RecCfg { int nw; int tpw; bool wlds; size_t lds; };

static bool pick_cfg(int H, bool bwd, RecCfg *c) {
    const int nHT = hidden_tiles(H);
    int tpw = 1;
    while (tpw < 8 && (nHT + tpw - 1) / tpw > 8) tpw *= 2;
    int nw = (nHT + tpw - 1) / tpw;
    if (nw > 16) return false;   // H > 2048
    c->nw = nw; c->tpw = tpw;
    const size_t wbytes = packed_whh_floats(H) * 4;
    // forward: two h buffers; backward: one dgh buffer of 3*nHT fragments
    const size_t xbytes = bwd ? (size_t)3 * nHT * 1024 : (size_t)2 * nHT * 1024;
    c->wlds = wbytes + xbytes <= 160 * 1024;
    c->lds = xbytes + (c->wlds ? wbytes : 0);
    return true;
}

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

#define TGRU_DISPATCH(KERNEL, cfg, grid, args, s)                                                   \
    do {                                                                                            \
        hipError_t e_ = hipSuccess;                                                                 \
        dim3 block_((cfg).nw * 64);                                                                 \
        switch ((cfg).tpw * 2 + ((cfg).wlds ? 1 : 0)) {                                             \
        case 3: e_ = set_lds(KERNEL<1, true>, (cfg).lds); if (e_ == hipSuccess) TGRU_LAUNCH((KERNEL<1, true>), grid, block_, (cfg).lds, s, args); break;   \
        case 2: e_ = set_lds(KERNEL<1, false>, (cfg).lds); if (e_ == hipSuccess) TGRU_LAUNCH((KERNEL<1, false>), grid, block_, (cfg).lds, s, args); break; \
        case 5: e_ = set_lds(KERNEL<2, true>, (cfg).lds); if (e_ == hipSuccess) TGRU_LAUNCH((KERNEL<2, true>), grid, block_, (cfg).lds, s, args); break;   \
        case 4: e_ = set_lds(KERNEL<2, false>, (cfg).lds); if (e_ == hipSuccess) TGRU_LAUNCH((KERNEL<2, false>), grid, block_, (cfg).lds, s, args); break; \
        case 8: e_ = set_lds(KERNEL<4, false>, (cfg).lds); if (e_ == hipSuccess) TGRU_LAUNCH((KERNEL<4, false>), grid, block_, (cfg).lds, s, args); break; \
        default: e_ = hipErrorInvalidValue;                                                         \
        }                                                                                           \
        if (e_ != hipSuccess) return e_;                                                            \
    } while (0)

hipError_t launch_gru_fwd(const FwdArgs &a, hipStream_t s) {
    RecCfg cfg;
    if (!pick_cfg(a.H, false, &cfg)) return hipErrorInvalidValue;
    dim3 grid(gru_batch_blocks(a.B));
    TGRU_DISPATCH(gru_fwd_kernel, cfg, grid, a, s);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// backward (BPTT)
// ----------------------------------------------------------------------------------------------
template <int TPW, bool WLDS>
__global__ __launch_bounds__(512) void gru_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T;
    const int nHT = hidden_tiles(H);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: tile offsets stay in SGPRs
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const bool vec = (H & 3) == 0;
    const long plane = (long)T * B * H;

    float *lds_g = smem;                       // dgh fragments: 3*nHT blocks of 256 floats
    float *lds_w = smem + 3 * nHT * 256;
    if (WLDS) {
        const long total = 3L * nHT * nHT * 256;
        for (long i = threadIdx.x * 4L; i < total; i += blockDim.x * 4L)
            *reinterpret_cast<float4 *>(lds_w + i) = *reinterpret_cast<const float4 *>(a.wtp + i);
    }
    const float *wsrc = WLDS ? lds_w : a.wtp;

    float dhc[TPW][4];        // gradient carried into h_t from step t+1
    float dbs[TPW][4][4];     // running column sums of dg for the bias gradients
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dhc[i][c] = 0.f;   // dh_n enters at each sequence's own last step, see load_in
#pragma unroll
            for (int g = 0; g < 4; ++g) dbs[i][g][c] = 0.f;
        }

    // everything the elementwise part of one step needs for one owned tile
    struct StepIn { float4 r, z, n, hn, hp, dy; };
    auto ld = [&](const float *p, int nv) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) { if (nv > 0) v = *reinterpret_cast<const float4 *>(p); }
        else {
            if (nv > 0) v.x = p[0];
            if (nv > 1) v.y = p[1];
            if (nv > 2) v.z = p[2];
            if (nv > 3) v.w = p[3];
        }
        return v;
    };
    auto load_in = [&](int t, int i, StepIn &s) {
        s.r = s.z = s.n = s.hn = s.hp = s.dy = make_float4(0.f, 0.f, 0.f, 0.f);
        const int tile = wave * TPW + i;
        const int j0 = 16 * tile + 4 * gq;
        const int jn = min(4, max(0, H - j0));
        if (tile < nHT && t >= 0 && t < len && jn > 0) {
            const long row = (long)t * B + bglob;
            const float *gp = a.gates + row * H + j0;
            s.r = ld(gp, jn); s.z = ld(gp + plane, jn); s.n = ld(gp + 2 * plane, jn); s.hn = ld(gp + 3 * plane, jn);
            if (t > 0) s.hp = ld(a.y_int + (row - B) * H + j0, jn);
            else if (a.h0) s.hp = ld(a.h0 + (long)bglob * H + j0, jn);
            if (a.dy_int) s.dy = ld(a.dy_int + row * H + j0, jn);
            if (a.dy_ext) {
                const long er = map_tb(a.dymap, t, bglob);
                if (er >= 0) { const float4 e = ld(a.dy_ext + er * H + j0, jn); s.dy.x += e.x; s.dy.y += e.y; s.dy.z += e.z; s.dy.w += e.w; }
            }
            if (a.dh_n && t == len - 1) {
                const float4 e = ld(a.dh_n + (long)bglob * H + j0, jn);
                s.dy.x += e.x; s.dy.y += e.y; s.dy.z += e.z; s.dy.w += e.w;
            }
        }
    };
    // with one tile per wave the step is short, so its inputs are loaded one step ahead
    StepIn nxt;
    if (TPW == 1) load_in(T - 1, 0, nxt);
    lds_barrier();

    for (int t = T - 1; t >= 0; --t) {
        const bool active = t < len;
        const long row = (long)t * B + bglob;
        float dz_keep[TPW][4];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave * TPW + i;
            if (tile >= nHT) continue;
            const int j0 = 16 * tile + 4 * gq;
            StepIn in;
            if (TPW == 1) in = nxt; else load_in(t, i, in);
            const float rr[4] = {in.r.x, in.r.y, in.r.z, in.r.w}, zz[4] = {in.z.x, in.z.y, in.z.z, in.z.w};
            const float nn[4] = {in.n.x, in.n.y, in.n.z, in.n.w}, hh[4] = {in.hn.x, in.hn.y, in.hn.z, in.hn.w};
            const float hq[4] = {in.hp.x, in.hp.y, in.hp.z, in.hp.w}, dyv[4] = {in.dy.x, in.dy.y, in.dy.z, in.dy.w};
            float dar[4], daz[4], dan[4], dgn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool on = active && j0 + c < H;
                const float d = on ? dyv[c] + dhc[i][c] : 0.f;
                const float dn = d * (1.f - zz[c]);
                const float dzz = d * (hq[c] - nn[c]);
                const float an = dn * (1.f - nn[c] * nn[c]);
                dan[c] = on ? an : 0.f;
                dar[c] = on ? an * hh[c] * rr[c] * (1.f - rr[c]) : 0.f;
                daz[c] = on ? dzz * zz[c] * (1.f - zz[c]) : 0.f;
                dgn[c] = on ? an * rr[c] : 0.f;
                dz_keep[i][c] = active ? d * zz[c] : dhc[i][c];
                dbs[i][0][c] += dar[c]; dbs[i][1][c] += daz[c]; dbs[i][2][c] += dan[c]; dbs[i][3][c] += dgn[c];
            }
            // fragments for the W_hh^T product: k-block (gate, tile)
            *reinterpret_cast<float4 *>(lds_g + ((0 * nHT + tile) * 64 + lane) * 4) = make_float4(dar[0], dar[1], dar[2], dar[3]);
            *reinterpret_cast<float4 *>(lds_g + ((1 * nHT + tile) * 64 + lane) * 4) = make_float4(daz[0], daz[1], daz[2], daz[3]);
            *reinterpret_cast<float4 *>(lds_g + ((2 * nHT + tile) * 64 + lane) * 4) = make_float4(dgn[0], dgn[1], dgn[2], dgn[3]);
            if (bvalid && j0 < H) {
                float *dst = a.dg + row * 4 * H + j0;
                if (vec) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(dar[0], dar[1], dar[2], dar[3]);
                    *reinterpret_cast<float4 *>(dst + H) = make_float4(daz[0], daz[1], daz[2], daz[3]);
                    *reinterpret_cast<float4 *>(dst + 2 * H) = make_float4(dan[0], dan[1], dan[2], dan[3]);
                    *reinterpret_cast<float4 *>(dst + 3 * H) = make_float4(dgn[0], dgn[1], dgn[2], dgn[3]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (j0 + c >= H) break;
                        dst[c] = dar[c]; dst[H + c] = daz[c]; dst[2 * H + c] = dan[c]; dst[3 * H + c] = dgn[c];
                    }
                }
            }
        }
        if (TPW == 1) load_in(t - 1, 0, nxt);
        lds_barrier();

        // dh_{t-1} += W_hh^T . dgh_t  (two accumulator chains per tile to cover MFMA latency)
        f32x4 acc[TPW][2];
#pragma unroll
        for (int i = 0; i < TPW; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
        const int nq = 3 * nHT;
        auto load_wt = [&](int qq, float4 (&wf)[TPW]) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = min(wave * TPW + i, nHT - 1);
                wf[i] = *reinterpret_cast<const float4 *>(wsrc + (((long)tile * 3 * nHT + qq) * 64 + lane) * 4);
            }
        };
        auto block_t = [&](int qq, const float4 (&wf)[TPW]) {
            const float4 gf = *reinterpret_cast<const float4 *>(lds_g + (qq * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (wave * TPW + i < nHT) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].x, gf.x, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].y, gf.y, acc[i][1], 0, 0, 0);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].z, gf.z, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].w, gf.w, acc[i][1], 0, 0, 0);
                }
            }
        };
        if (WLDS) {
            for (int qq = 0; qq < nq; ++qq) {
                float4 wf[TPW];
                load_wt(qq, wf);
                block_t(qq, wf);
            }
        } else {
            // weights streamed from L2: the fragments of the next kDepth k-blocks are in flight
            constexpr int kDepth = 4;
            float4 ring[kDepth][TPW];
#pragma unroll
            for (int d = 0; d < kDepth; ++d) load_wt(min(d, nq - 1), ring[d]);
            for (int q0 = 0; q0 < nq; q0 += kDepth) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) {
                    const int qq = q0 + d;
                    if (qq < nq) {
                        block_t(qq, ring[d]);
                        load_wt(min(qq + kDepth, nq - 1), ring[d]);
                    }
                }
            }
        }
        mfma_drain();
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) dhc[i][c] = dz_keep[i][c] + (active ? acc[i][0][c] + acc[i][1][c] : 0.f);
        lds_barrier();
    }

    // dh0 and the per-workgroup bias partial sums (sum over the 16 sequences of this workgroup)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave * TPW + i;
        if (tile >= nHT) continue;
        const int j0 = 16 * tile + 4 * gq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (a.dh0 && bvalid && j0 + c < H) a.dh0[(long)bglob * H + j0 + c] = dhc[i][c];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = dbs[i][g][c];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                if (b == 0 && j0 + c < H) a.db_part[(long)blockIdx.x * 4 * H + g * H + j0 + c] = v;
            }
        }
    }
}

hipError_t launch_gru_bwd(const BwdArgs &a, hipStream_t s) {
    RecCfg cfg;
    if (!pick_cfg(a.H, true, &cfg)) return hipErrorInvalidValue;
    dim3 grid(gru_batch_blocks(a.B));
    TGRU_DISPATCH(gru_bwd_kernel, cfg, grid, a, s);
    return hipGetLastError();
}

// db_ih = sums of (da_r | da_z | da_n), db_hh = sums of (da_r | da_z | da_n*r)
// db_ih / db_hh = column sums of the per-workgroup (or per-sequence) partial sums part[nblocks][4H]
// (columns: da_r | da_z | da_n | da_n*r).  Workgroup = 32 columns x 32 row groups; a group sums a
// contiguous range of rows with four interleaved accumulators, the groups combine in fixed order.
constexpr int kBiasGroups = 32;
__global__ __launch_bounds__(32 * kBiasGroups) void bias_grad_kernel(const float *part, int nblocks, int H,
                                                                     float *db_ih, float *db_hh, int accumulate) {
    __shared__ float red[kBiasGroups][32];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    const int per = (nblocks + kBiasGroups - 1) / kBiasGroups;
    const int k0 = grp * per, k1 = min(nblocks, k0 + per);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (i < 4 * H) {
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            v0 += part[(long)k * 4 * H + i];
            v1 += part[(long)(k + 1) * 4 * H + i];
            v2 += part[(long)(k + 2) * 4 * H + i];
            v3 += part[(long)(k + 3) * 4 * H + i];
        }
        for (; k < k1; ++k) v0 += part[(long)k * 4 * H + i];
    }
    red[grp][c] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (grp != 0 || i >= 4 * H) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < kBiasGroups; ++q) v += red[q][c];
    if (i < 3 * H) {
        db_ih[i] = accumulate ? db_ih[i] + v : v;
        if (i < 2 * H) db_hh[i] = accumulate ? db_hh[i] + v : v;
    } else {
        const int j = i - H;   // column 2H + (i - 3H)
        db_hh[j] = accumulate ? db_hh[j] + v : v;
    }
}

hipError_t launch_bias_grad(const float *db_part, int nblocks, int H, float *db_ih, float *db_hh,
                            int accumulate, hipStream_t s) {
    TGRU_LAUNCH(bias_grad_kernel, dim3((4 * H + 31) / 32), dim3(32 * kBiasGroups), 0, s, db_part, nblocks, H,
                       db_ih, db_hh, accumulate);
    return hipGetLastError();
}

}  // namespace tgru
