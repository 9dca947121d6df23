// Whole-stack GRU kernels for small hidden sizes (H <= 32, In <= 32, L <= 8) on gfx950.
//
// At these sizes a stack is latency-bound: every step is a 16 x 3H x 2H product per workgroup and
// there are T*L of them in sequence.  These kernels therefore run the WHOLE stack in one launch:
//
//   * one workgroup owns 16 sequences, one wave owns (layer, hidden tile of 16);
//   * layers run as a wavefront: at tick k layer l works on step t = k - l, so the critical
//     path is T + L - 1 ticks instead of T * L steps;
//   * W_ih and W_hh fragments stay in VGPRs for the whole kernel (MFMA A operands), the hidden
//     state and the layer-to-layer activations move between waves as MFMA B fragments through
//     LDS (the D fragment of a hidden tile is the B fragment of the matching k-block, so one
//     lane-linear 16-byte store / load per lane, no transposition);
//   * the input projection of every layer is fused into the step (no gi buffer at all).
//
// The backward kernel mirrors this in reverse time / reverse layer order and also produces dx;
// gru_small_wgrad_kernel then turns the dg buffers into all weight and bias gradients of all
// layers in a single launch (deterministic: fixed reduction order, no atomics).
#include "tgru_internal.h"

namespace tgru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm_(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// 4 consecutive floats p[0..3], element c valid iff c < nvalid; vector load when possible
__device__ __forceinline__ float4 load4(const float *p, int nvalid, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const float4 *>(p);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}
__device__ __forceinline__ void store4(float *p, float4 v, int nvalid, bool vec_ok) {
    if (nvalid >= 4 && vec_ok) { *reinterpret_cast<float4 *>(p) = v; return; }
    if (nvalid > 0) p[0] = v.x;
    if (nvalid > 1) p[1] = v.y;
    if (nvalid > 2) p[2] = v.z;
    if (nvalid > 3) p[3] = v.w;
}

#define MFMA4(acc, A, Bf)                                                          \
    do {                                                                           \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (Bf).x, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (Bf).y, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (Bf).z, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (Bf).w, acc, 0, 0, 0);   \
    } while (0)

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void gru_small_fwd_kernel(SmallFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // hbuf[2][L][NT][256]
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = wave / NT, tile = wave % NT;
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const int nQin = (In_l + 15) / 16;
    const bool hvec = (H & 3) == 0;
    const bool ivec = (In_l & 3) == 0;
    const int j0 = 16 * tile + 4 * gq;            // first hidden unit of this lane
    const int jn = min(4, max(0, H - j0));        // how many of its 4 units exist

    // weight fragments: lane (i = lane&15, gq) holds W[g*H + 16*tile + i][16*q + 4*gq + c]
    float4 wih[3][2], whh[3][NT];
    {
        const int r = 16 * tile + (lane & 15);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c0 = 16 * q + 4 * gq;
                wih[g][q] = (r < H && q < nQin)
                                ? load4(a.w_ih[l] + (long)(g * H + r) * In_l + c0, In_l - c0, ivec)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int c0 = 16 * q + 4 * gq;
                whh[g][q] = (r < H) ? load4(a.w_hh[l] + (long)(g * H + r) * H + c0, H - c0, hvec)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    float brz[2][4], bin[4], bhn[4], hreg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool ok = c < jn;
        brz[0][c] = ok ? a.b_ih[l][j0 + c] + a.b_hh[l][j0 + c] : 0.f;
        brz[1][c] = ok ? a.b_ih[l][H + j0 + c] + a.b_hh[l][H + j0 + c] : 0.f;
        bin[c] = ok ? a.b_ih[l][2 * H + j0 + c] : 0.f;
        bhn[c] = ok ? a.b_hh[l][2 * H + j0 + c] : 0.f;
        hreg[c] = (ok && bvalid && a.h0) ? a.h0[((long)l * B + bglob) * H + j0 + c] : 0.f;
    }
    // both parities start from h0 so that step 0 of every layer reads its initial state
    {
        const float4 h4 = make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
        *reinterpret_cast<float4 *>(smem + (((0 * L + l) * NT + tile) * 64 + lane) * 4) = h4;
        *reinterpret_cast<float4 *>(smem + (((1 * L + l) * NT + tile) * 64 + lane) * 4) = h4;
    }
    lds_barrier();

    const long plane = (long)T * B * H;
    // layer 0 reads x straight from global memory as B fragments, one step ahead
    float4 xn[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    auto load_x = [&](int t) {
        if (t < T && t < len) {
            const long er = map_row(a.xmap, t * B + bglob);
            if (er >= 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c0 = 16 * q + 4 * gq;
                    if (q < nQin) xn[q] = load4(a.x + er * a.In + c0, a.In - c0, ivec && ((a.In & 3) == 0));
                }
            }
        }
    };
    if (l == 0) load_x(0);

    int p = 0;
    const int nticks = T + L - 1;
    for (int tick = 0; tick < nticks; ++tick) {
        const int t = tick - l;
        if (t >= 0 && t < T) {
            const bool active = t < len;
            const float *hb = smem + (long)p * L * NT * 256;
            float4 xin[2], hp[NT];
            if (l == 0) {
                xin[0] = xn[0]; xin[1] = xn[1];
                load_x(t + 1);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    xin[q] = (q < NT) ? *reinterpret_cast<const float4 *>(hb + (((l - 1) * NT + q) * 64 + lane) * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q)
                hp[q] = *reinterpret_cast<const float4 *>(hb + ((l * NT + q) * 64 + lane) * 4);

            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ain = ar, ahn = ar;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < nQin) {
                    // interleave the three chains so dependent MFMAs are not back to back
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[0][q].x, xin[q].x, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[1][q].x, xin[q].x, az, 0, 0, 0);
                    ain = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[2][q].x, xin[q].x, ain, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[0][q].y, xin[q].y, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[1][q].y, xin[q].y, az, 0, 0, 0);
                    ain = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[2][q].y, xin[q].y, ain, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[0][q].z, xin[q].z, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[1][q].z, xin[q].z, az, 0, 0, 0);
                    ain = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[2][q].z, xin[q].z, ain, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[0][q].w, xin[q].w, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[1][q].w, xin[q].w, az, 0, 0, 0);
                    ain = __builtin_amdgcn_mfma_f32_16x16x4f32(wih[2][q].w, xin[q].w, ain, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[0][q].x, hp[q].x, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[1][q].x, hp[q].x, az, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[2][q].x, hp[q].x, ahn, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[0][q].y, hp[q].y, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[1][q].y, hp[q].y, az, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[2][q].y, hp[q].y, ahn, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[0][q].z, hp[q].z, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[1][q].z, hp[q].z, az, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[2][q].z, hp[q].z, ahn, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[0][q].w, hp[q].w, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[1][q].w, hp[q].w, az, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(whh[2][q].w, hp[q].w, ahn, 0, 0, 0);
            }

            float rv[4], zv[4], nv[4], hnv[4], yv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ghn = ahn[c] + bhn[c];
                const float r = sigm_(ar[c] + brz[0][c]);
                const float z = sigm_(az[c] + brz[1][c]);
                const float n = tanhf(ain[c] + bin[c] + r * ghn);
                const float hnew = (1.f - z) * n + z * hreg[c];
                rv[c] = r; zv[c] = z; nv[c] = n; hnv[c] = ghn;
                if (active && c < jn) hreg[c] = hnew;
                yv[c] = active ? hreg[c] : 0.f;
            }
            float *hw = smem + (long)(p ^ 1) * L * NT * 256;
            *reinterpret_cast<float4 *>(hw + ((l * NT + tile) * 64 + lane) * 4) =
                make_float4(hreg[0], hreg[1], hreg[2], hreg[3]);
            if (bvalid && jn > 0) {
                const long row = (long)t * B + bglob;
                const float4 y4 = make_float4(yv[0], yv[1], yv[2], yv[3]);
                if (a.y_int[l]) store4(a.y_int[l] + row * H + j0, y4, jn, hvec);
                if (l == L - 1 && a.y_ext) {
                    const long er = map_row(a.ymap, (int)row);
                    if (er >= 0) store4(a.y_ext + er * H + j0, y4, jn, hvec);
                }
                if (a.gates[l] && active) {
                    float *gp = a.gates[l] + row * H + j0;
                    store4(gp, make_float4(rv[0], rv[1], rv[2], rv[3]), jn, hvec);
                    store4(gp + plane, make_float4(zv[0], zv[1], zv[2], zv[3]), jn, hvec);
                    store4(gp + 2 * plane, make_float4(nv[0], nv[1], nv[2], nv[3]), jn, hvec);
                    store4(gp + 3 * plane, make_float4(hnv[0], hnv[1], hnv[2], hnv[3]), jn, hvec);
                }
                if (a.h_n && t == len - 1) store4(a.h_n + ((long)l * B + bglob) * H + j0, y4, jn, hvec);
            }
        }
        lds_barrier();
        p ^= 1;
    }
}

hipError_t launch_gru_small_fwd(const SmallFwdArgs &a, hipStream_t s) {
    const int NT = (a.H + 15) / 16;
    dim3 grid(gru_batch_blocks(a.B)), block(64 * NT * a.L);
    const size_t lds = (size_t)2 * a.L * NT * 1024;
    const bool big = block.x > 512;
    if (NT == 1 && !big) hipLaunchKernelGGL((gru_small_fwd_kernel<1, 512>), grid, block, lds, s, a);
    else if (NT == 1) hipLaunchKernelGGL((gru_small_fwd_kernel<1, 1024>), grid, block, lds, s, a);
    else if (!big) hipLaunchKernelGGL((gru_small_fwd_kernel<2, 512>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gru_small_fwd_kernel<2, 1024>), grid, block, lds, s, a);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void gru_small_bwd_kernel(SmallBwdArgs a) {
    // gbuf[L][4][NT][256] : da_r | da_z | da_n | da_n*r fragments, dxbuf[2][L][NT][256]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = wave / NT, tile = wave % NT;
    const int b = lane & 15, gq = lane >> 4;
    const int bglob = blockIdx.x * 16 + b;
    const bool bvalid = bglob < B;
    const int len = bvalid ? (a.lengths ? a.lengths[bglob] : T) : 0;
    const int In_l = (l == 0) ? a.In : H;
    const int nDx = (In_l + 15) / 16;             // dx tiles of this layer
    const bool two_dx = nDx > NT;                 // layer 0 with more input tiles than hidden tiles
    const bool hvec = (H & 3) == 0;
    const int j0 = 16 * tile + 4 * gq;
    const int jn = min(4, max(0, H - j0));
    const long plane = (long)T * B * H;
    float *gbuf = smem;
    float *dxbuf = smem + (long)L * 4 * NT * 256;

    // A fragments of the transposed weights.
    //   whhT[g][q]: lane (i, gq), c -> W_hh[G*H + 16q + 4gq + c][16*tile + i], G = r, z, n rows
    //   wihT[d][g][q]: lane (i, gq), c -> W_ih[g*H + 16q + 4gq + c][16*(tile + d*NT) + i]
    float4 whhT[3][NT], wihT[2][3][NT];
    {
        const int i = lane & 15;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kr = 16 * q + 4 * gq + c, oc = 16 * tile + i;
                    v[c] = (kr < H && oc < H) ? a.w_hh[l][(long)(g * H + kr) * H + oc] : 0.f;
                }
                whhT[g][q] = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    float u[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int kr = 16 * q + 4 * gq + c, oc = 16 * (tile + d * NT) + i;
                        u[c] = (kr < H && oc < In_l && (tile + d * NT) < nDx)
                                   ? a.w_ih[l][(long)(g * H + kr) * In_l + oc] : 0.f;
                    }
                    wihT[d][g][q] = make_float4(u[0], u[1], u[2], u[3]);
                }
            }
    }

    float dhc[4] = {0.f, 0.f, 0.f, 0.f};
    float dbs[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) dbs[g][c] = 0.f;

    // saved activations of the step this wave handles next, loaded one tick ahead
    float4 pr, pz, pn, phn, php, pdy;
    auto prefetch = [&](int t) {
        pr = pz = pn = phn = php = pdy = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < len && jn > 0) {
            const long row = (long)t * B + bglob;
            const float *gp = a.gates[l] + row * H + j0;
            pr = load4(gp, jn, hvec);
            pz = load4(gp + plane, jn, hvec);
            pn = load4(gp + 2 * plane, jn, hvec);
            phn = load4(gp + 3 * plane, jn, hvec);
            if (t > 0) php = load4(a.y_int[l] + (row - B) * H + j0, jn, hvec);
            else if (a.h0) php = load4(a.h0 + ((long)l * B + bglob) * H + j0, jn, hvec);
            if (l == L - 1 && a.dy_ext) {
                const long er = map_row(a.dymap, (int)row);
                if (er >= 0) pdy = load4(a.dy_ext + er * H + j0, jn, hvec);
            }
            if (a.dh_n && t == len - 1) {
                const float4 e = load4(a.dh_n + ((long)l * B + bglob) * H + j0, jn, hvec);
                pdy.x += e.x; pdy.y += e.y; pdy.z += e.z; pdy.w += e.w;
            }
        }
    };
    prefetch(T - 1);   // every layer starts at t = T-1

    int p = 0;
    const int nticks = T + L - 1;
    for (int tick = 0; tick < nticks; ++tick) {
        const int t = T - 1 - tick + (L - 1 - l);
        const bool work = t >= 0 && t < T;
        const bool active = work && t < len;
        float keep[4] = {dhc[0], dhc[1], dhc[2], dhc[3]};
        float *gb = gbuf;
        if (work) {
            float dyv[4] = {pdy.x, pdy.y, pdy.z, pdy.w};
            if (l < L - 1) {   // gradient from the layer above, produced at the previous tick
                const float4 u = *reinterpret_cast<const float4 *>(
                    dxbuf + (long)(p ^ 1) * L * NT * 256 + (((l + 1) * NT + tile) * 64 + lane) * 4);
                dyv[0] += u.x; dyv[1] += u.y; dyv[2] += u.z; dyv[3] += u.w;
            }
            const float rr[4] = {pr.x, pr.y, pr.z, pr.w}, zz[4] = {pz.x, pz.y, pz.z, pz.w};
            const float nn[4] = {pn.x, pn.y, pn.z, pn.w}, hh[4] = {phn.x, phn.y, phn.z, phn.w};
            const float hq[4] = {php.x, php.y, php.z, php.w};
            float dar[4], daz[4], dan[4], dgn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool on = active && c < jn;
                const float d = on ? dyv[c] + dhc[c] : 0.f;
                const float dn = d * (1.f - zz[c]);
                const float dzz = d * (hq[c] - nn[c]);
                const float an = dn * (1.f - nn[c] * nn[c]);
                dan[c] = on ? an : 0.f;
                dar[c] = on ? an * hh[c] * rr[c] * (1.f - rr[c]) : 0.f;
                daz[c] = on ? dzz * zz[c] * (1.f - zz[c]) : 0.f;
                dgn[c] = on ? an * rr[c] : 0.f;
                keep[c] = active ? d * zz[c] : dhc[c];
                dbs[0][c] += dar[c]; dbs[1][c] += daz[c]; dbs[2][c] += dan[c]; dbs[3][c] += dgn[c];
            }
            const float4 f0 = make_float4(dar[0], dar[1], dar[2], dar[3]);
            const float4 f1 = make_float4(daz[0], daz[1], daz[2], daz[3]);
            const float4 f2 = make_float4(dan[0], dan[1], dan[2], dan[3]);
            const float4 f3 = make_float4(dgn[0], dgn[1], dgn[2], dgn[3]);
            *reinterpret_cast<float4 *>(gb + (((l * 4 + 0) * NT + tile) * 64 + lane) * 4) = f0;
            *reinterpret_cast<float4 *>(gb + (((l * 4 + 1) * NT + tile) * 64 + lane) * 4) = f1;
            *reinterpret_cast<float4 *>(gb + (((l * 4 + 2) * NT + tile) * 64 + lane) * 4) = f2;
            *reinterpret_cast<float4 *>(gb + (((l * 4 + 3) * NT + tile) * 64 + lane) * 4) = f3;
            if (bvalid && jn > 0) {
                float *dst = a.dg[l] + ((long)t * B + bglob) * 4 * H + j0;
                store4(dst, f0, jn, hvec);
                store4(dst + H, f1, jn, hvec);
                store4(dst + 2 * H, f2, jn, hvec);
                store4(dst + 3 * H, f3, jn, hvec);
            }
            prefetch(t - 1);
        }
        lds_barrier();
        if (work) {
            f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, ax0 = ah0, ax1 = ah0;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 fr = *reinterpret_cast<const float4 *>(gb + (((l * 4 + 0) * NT + q) * 64 + lane) * 4);
                const float4 fz = *reinterpret_cast<const float4 *>(gb + (((l * 4 + 1) * NT + q) * 64 + lane) * 4);
                const float4 fn = *reinterpret_cast<const float4 *>(gb + (((l * 4 + 2) * NT + q) * 64 + lane) * 4);
                const float4 fg = *reinterpret_cast<const float4 *>(gb + (((l * 4 + 3) * NT + q) * 64 + lane) * 4);
                MFMA4(ah0, whhT[0][q], fr);
                MFMA4(ax0, wihT[0][0][q], fr);
                MFMA4(ah1, whhT[1][q], fz);
                MFMA4(ax0, wihT[0][1][q], fz);
                MFMA4(ah0, whhT[2][q], fg);
                MFMA4(ax0, wihT[0][2][q], fn);
                if (two_dx) {
                    MFMA4(ax1, wihT[1][0][q], fr);
                    MFMA4(ax1, wihT[1][1][q], fz);
                    MFMA4(ax1, wihT[1][2][q], fn);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) dhc[c] = keep[c] + (active ? ah0[c] + ah1[c] : 0.f);
            if (l > 0) {
                *reinterpret_cast<float4 *>(dxbuf + (long)p * L * NT * 256 + ((l * NT + tile) * 64 + lane) * 4) =
                    make_float4(ax0[0], ax0[1], ax0[2], ax0[3]);
            } else if (a.dx_ext && bvalid) {
                const long er = map_row(a.dxmap, t * B + bglob);
                if (er >= 0) {
                    const bool xvec = (a.In & 3) == 0;
                    const int c0 = 16 * tile + 4 * gq;
                    if (c0 < a.In)
                        store4(a.dx_ext + er * a.In + c0, make_float4(ax0[0], ax0[1], ax0[2], ax0[3]),
                               min(4, a.In - c0), xvec);
                    const int c1 = 16 * (tile + NT) + 4 * gq;
                    if (tile + NT < nDx && c1 < a.In)
                        store4(a.dx_ext + er * a.In + c1, make_float4(ax1[0], ax1[1], ax1[2], ax1[3]),
                               min(4, a.In - c1), xvec);
                }
            }
        }
        lds_barrier();
        p ^= 1;
    }

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a.dh0 && bvalid && c < jn) a.dh0[((long)l * B + bglob) * H + j0 + c] = dhc[c];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = dbs[g][c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (b == 0 && c < jn) a.db_part[((long)blockIdx.x * L + l) * 4 * H + g * H + j0 + c] = v;
        }
    }
}

hipError_t launch_gru_small_bwd(const SmallBwdArgs &a, hipStream_t s) {
    const int NT = (a.H + 15) / 16;
    dim3 grid(gru_batch_blocks(a.B)), block(64 * NT * a.L);
    const size_t lds = ((size_t)a.L * 4 * NT + (size_t)2 * a.L * NT) * 1024;
    const bool big = block.x > 512;
    const void *fn = NT == 1 ? (big ? (const void *)gru_small_bwd_kernel<1, 1024> : (const void *)gru_small_bwd_kernel<1, 512>)
                             : (big ? (const void *)gru_small_bwd_kernel<2, 1024> : (const void *)gru_small_bwd_kernel<2, 512>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (NT == 1 && !big) hipLaunchKernelGGL((gru_small_bwd_kernel<1, 512>), grid, block, lds, s, a);
    else if (NT == 1) hipLaunchKernelGGL((gru_small_bwd_kernel<1, 1024>), grid, block, lds, s, a);
    else if (!big) hipLaunchKernelGGL((gru_small_bwd_kernel<2, 512>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gru_small_bwd_kernel<2, 1024>), grid, block, lds, s, a);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// weight gradients of all layers in one launch
// ----------------------------------------------------------------------------------------------
// One workgroup per 16x16 output tile of dW_ih / dW_hh (plus one per layer for the biases); its 8
// waves split the T*B reduction rows, partial tiles are summed through LDS in a fixed order.
//   dW_ih[m][n] = sum_k dg[k][m] * xin[k][n]                     m < 3H
//   dW_hh[m][n] = sum_k dg[k][cm(m)] * h_{t-1}[k][n]             cm(m) = m < 2H ? m : m + H
__global__ __launch_bounds__(512) void gru_small_wgrad_kernel(SmallWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float red[8][64][4];
    const int H = a.H, B = a.B, T = a.T, L = a.L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Mt = (3 * H + 15) / 16, NtH = (H + 15) / 16;
    // decode the job
    int job = blockIdx.x, l = 0, kind = -1, mt = 0, nt = 0;
    for (l = 0; l < L; ++l) {
        const int In_l = (l == 0) ? a.In : H;
        const int n_ih = Mt * ((In_l + 15) / 16), n_hh = Mt * NtH;
        if (job < n_ih) { kind = 0; mt = job / ((In_l + 15) / 16); nt = job % ((In_l + 15) / 16); break; }
        job -= n_ih;
        if (job < n_hh) { kind = 1; mt = job / NtH; nt = job % NtH; break; }
        job -= n_hh;
        if (job < 1) { kind = 2; break; }
        job -= 1;
    }
    if (kind < 0) return;
    if (kind == 2) {   // biases from the per-workgroup partial sums of the backward kernel
        for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) {
            float v = 0.f;
            for (int k = 0; k < a.nblk; ++k) v += a.db_part[((long)k * L + l) * 4 * H + i];
            if (i < 3 * H) {
                a.db_ih[l][i] = a.accumulate ? a.db_ih[l][i] + v : v;
                if (i < 2 * H) a.db_hh[l][i] = a.accumulate ? a.db_hh[l][i] + v : v;
            } else {
                a.db_hh[l][i - H] = a.accumulate ? a.db_hh[l][i - H] + v : v;
            }
        }
        return;
    }
    const int In_l = (l == 0) ? a.In : H;
    const int N = kind == 0 ? In_l : H;
    const int K = T * B;
    const int i = lane & 15, gq = lane >> 4;
    const int m = 16 * mt + i;
    const int acol = (kind == 1 && m >= 2 * H) ? m + H : m;
    const bool mok = m < 3 * H;
    const int n = 16 * nt + i;
    const bool nok = n < N;
    const float *dg = a.dg[l];
    const float *src = kind == 0 ? (l == 0 ? a.x : a.y_int[l - 1]) : a.y_int[l];
    const int kchunk = ((K + 7) / 8 + 3) / 4 * 4;
    const int kbeg = wave * kchunk, kend = min(K, kbeg + kchunk);

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = k0 + 4 * s4 + gq;
            av[s4] = 0.f; bv[s4] = 0.f;
            if (k < kend) {
                if (mok) av[s4] = dg[(long)k * 4 * H + acol];
                if (nok) {
                    if (kind == 0) {
                        const long er = (l == 0) ? map_row(a.xmap, k) : (long)k;
                        if (er >= 0) bv[s4] = src[er * N + n];
                    } else {
                        if (k >= B) bv[s4] = src[(long)(k - B) * H + n];
                        else if (a.h0) bv[s4] = a.h0[((long)l * B + k) * H + n];
                    }
                }
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4], acc, 0, 0, 0);
    }
    *reinterpret_cast<float4 *>(&red[wave][lane][0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave == 0) {
        float4 s = *reinterpret_cast<const float4 *>(&red[0][lane][0]);
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(&red[w][lane][0]);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        // D: col = lane&15 (n), row = 4*(lane>>4) + reg (m)
        float *out = kind == 0 ? a.dw_ih[l] : a.dw_hh[l];
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = 16 * mt + 4 * gq + r;
            if (mm < 3 * H && nok) {
                float *dst = out + (long)mm * N + n;
                *dst = a.accumulate ? *dst + sv[r] : sv[r];
            }
        }
    }
}

hipError_t launch_gru_small_wgrad(const SmallWgradArgs &a, hipStream_t s) {
    const int Mt = (3 * a.H + 15) / 16, NtH = (a.H + 15) / 16;
    int jobs = 0;
    for (int l = 0; l < a.L; ++l) {
        const int In_l = l == 0 ? a.In : a.H;
        jobs += Mt * ((In_l + 15) / 16) + Mt * NtH + 1;
    }
    hipLaunchKernelGGL(gru_small_wgrad_kernel, dim3(jobs), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace tgru
