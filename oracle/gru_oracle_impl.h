/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the multi-layer GRU forward pass and
 * backprop-through-time that the reference runs through torch.nn.GRU.
 *
 * This file is included twice by gru_oracle.c, once with REAL=float and once with
 * REAL=double.  It is the checker for the HIP kernels; nothing in the product package
 * imports, links or calls it.
 *
 * What it restates (the arithmetic lives in PyTorch, a dependency of the reference that is
 * not vendored under the reference tree; the reference pins no version, the oracle is
 * pinned against torch 2.10.0 CPU through tests/golden):
 *   - cell equations: torch/nn/modules/rnn.py:1191-1194
 *         r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
 *         z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
 *         n = tanh   (W_in x + b_in + r * (W_hn h + b_hn))
 *         h' = (1 - z) * n + z * h
 *   - h_0 = 0 because no caller passes hx (rnn.py:1362-1370); call sites
 *     outdated_files/TimeGANv1.py:61,139,220,299,376.
 *   - gate row blocks of weight_ih / weight_hh / biases are ordered r, z, n.
 *   - pack_padded_sequence / pad_packed_sequence semantics at
 *     outdated_files/TimeGANv1.py:56-65: sequence b only runs for t < lengths[b]; output rows
 *     with t >= lengths[b] are zero; h_n[l][b] is the state at t = lengths[b]-1; gradients
 *     into padded rows are zero.
 *
 * Layout: x (B,T,In) batch-major, y (B,T,H) batch-major, h_n (L,B,H).
 * params / grads: per layer l the concatenation
 *     weight_ih (3H x In_l) | weight_hh (3H x H) | bias_ih (3H) | bias_hh (3H)
 * with In_0 = In and In_l = H for l > 0 (torch's _flat_weights order).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

static REAL FN(sigm)(REAL v) { return (REAL)1 / ((REAL)1 + EXPF(-v)); }

static size_t FN(layer_param_count)(int In_l, int H) {
    return (size_t)3 * H * In_l + (size_t)3 * H * H + (size_t)6 * H;
}

/*
 * Forward + optional backward.  dy == NULL -> forward only (dx / grads untouched).
 * dhn may be NULL (treated as zeros).  lengths may be NULL (all sequences have length T).
 * Returns 0, or a negative code: -1 bad dims, -2 a length is <= 0 or > T,
 * -3 lengths not sorted in decreasing order (enforce_sorted=True), -4 out of memory.
 */
int FN(gru_oracle_run)(const REAL *x, const int64_t *lengths, const REAL *params,
                       REAL *y, REAL *h_n,
                       const REAL *dy, const REAL *dhn, REAL *dx, REAL *grads,
                       int B, int T, int In, int H, int L) {
    if (B <= 0 || T <= 0 || In <= 0 || H <= 0 || L <= 0) return -1;
    if (lengths) {
        for (int b = 0; b < B; ++b) {
            if (lengths[b] <= 0 || lengths[b] > T) return -2;
            if (b > 0 && lengths[b] > lengths[b - 1]) return -3;
        }
    }
    const size_t BT = (size_t)B * T;
    /* per layer saved tensors: input of the layer, r, z, n, hn (= W_hn h + b_hn), output */
    REAL *act = (REAL *)calloc((size_t)(L + 1) * BT * (size_t)(In > H ? In : H), sizeof(REAL));
    REAL *sr = (REAL *)calloc((size_t)L * BT * H, sizeof(REAL));
    REAL *sz = (REAL *)calloc((size_t)L * BT * H, sizeof(REAL));
    REAL *sn = (REAL *)calloc((size_t)L * BT * H, sizeof(REAL));
    REAL *shn = (REAL *)calloc((size_t)L * BT * H, sizeof(REAL));
    REAL *gi = (REAL *)malloc((size_t)3 * H * sizeof(REAL));
    REAL *gh = (REAL *)malloc((size_t)3 * H * sizeof(REAL));
    if (!act || !sr || !sz || !sn || !shn || !gi || !gh) {
        free(act); free(sr); free(sz); free(sn); free(shn); free(gi); free(gh);
        return -4;
    }
    const size_t act_stride = BT * (size_t)(In > H ? In : H);
    /* layer 0 input */
    for (size_t i = 0; i < BT * (size_t)In; ++i) act[i] = x[i];

    const REAL *p = params;
    for (int l = 0; l < L; ++l) {
        const int In_l = (l == 0) ? In : H;
        const REAL *w_ih = p;
        const REAL *w_hh = w_ih + (size_t)3 * H * In_l;
        const REAL *b_ih = w_hh + (size_t)3 * H * H;
        const REAL *b_hh = b_ih + (size_t)3 * H;
        const REAL *in = act + (size_t)l * act_stride;
        REAL *out = act + (size_t)(l + 1) * act_stride;
        for (int b = 0; b < B; ++b) {
            const int len = lengths ? (int)lengths[b] : T;
            for (int t = 0; t < len; ++t) {
                const REAL *xt = in + ((size_t)b * T + t) * In_l;
                const REAL *hp = (t == 0) ? NULL : out + ((size_t)b * T + t - 1) * H;
                const size_t o = (((size_t)l * B + b) * T + t) * H;
                for (int g = 0; g < 3 * H; ++g) {
                    REAL a = b_ih[g];
                    for (int k = 0; k < In_l; ++k) a += w_ih[(size_t)g * In_l + k] * xt[k];
                    gi[g] = a;
                    REAL c = b_hh[g];
                    if (hp)
                        for (int k = 0; k < H; ++k) c += w_hh[(size_t)g * H + k] * hp[k];
                    gh[g] = c;
                }
                REAL *ht = out + ((size_t)b * T + t) * H;
                for (int j = 0; j < H; ++j) {
                    const REAL r = FN(sigm)(gi[j] + gh[j]);
                    const REAL z = FN(sigm)(gi[H + j] + gh[H + j]);
                    const REAL n = TANHF(gi[2 * H + j] + r * gh[2 * H + j]);
                    const REAL hprev = hp ? hp[j] : (REAL)0;
                    ht[j] = ((REAL)1 - z) * n + z * hprev;
                    sr[o + j] = r; sz[o + j] = z; sn[o + j] = n; shn[o + j] = gh[2 * H + j];
                }
            }
            if (h_n) {
                const REAL *last = out + ((size_t)b * T + len - 1) * H;
                for (int j = 0; j < H; ++j) h_n[((size_t)l * B + b) * H + j] = last[j];
            }
        }
        p += FN(layer_param_count)(In_l, H);
    }
    {
        const REAL *top = act + (size_t)L * act_stride;
        for (size_t i = 0; i < BT * (size_t)H; ++i) y[i] = top[i];
    }

    if (dy) {
        size_t total = 0;
        for (int l = 0; l < L; ++l) total += FN(layer_param_count)(l == 0 ? In : H, H);
        for (size_t i = 0; i < total; ++i) grads[i] = (REAL)0;
        /* gradient flowing into the output sequence of the current layer */
        REAL *dout = (REAL *)calloc(BT * (size_t)H, sizeof(REAL));
        REAL *din = (REAL *)calloc(BT * (size_t)(In > H ? In : H), sizeof(REAL));
        REAL *dh = (REAL *)malloc((size_t)H * sizeof(REAL));
        REAL *dgi = (REAL *)malloc((size_t)3 * H * sizeof(REAL));
        REAL *dgh = (REAL *)malloc((size_t)3 * H * sizeof(REAL));
        if (!dout || !din || !dh || !dgi || !dgh) {
            free(dout); free(din); free(dh); free(dgi); free(dgh);
            free(act); free(sr); free(sz); free(sn); free(shn); free(gi); free(gh);
            return -4;
        }
        for (size_t i = 0; i < BT * (size_t)H; ++i) dout[i] = dy[i];
        size_t poff = total;
        for (int l = L - 1; l >= 0; --l) {
            const int In_l = (l == 0) ? In : H;
            poff -= FN(layer_param_count)(In_l, H);
            const REAL *w_ih = params + poff;
            const REAL *w_hh = w_ih + (size_t)3 * H * In_l;
            REAL *gw_ih = grads + poff;
            REAL *gw_hh = gw_ih + (size_t)3 * H * In_l;
            REAL *gb_ih = gw_hh + (size_t)3 * H * H;
            REAL *gb_hh = gb_ih + (size_t)3 * H;
            const REAL *in = act + (size_t)l * act_stride;
            const REAL *out = act + (size_t)(l + 1) * act_stride;
            for (size_t i = 0; i < BT * (size_t)In_l; ++i) din[i] = (REAL)0;
            for (int b = 0; b < B; ++b) {
                const int len = lengths ? (int)lengths[b] : T;
                for (int j = 0; j < H; ++j)
                    dh[j] = dhn ? dhn[((size_t)l * B + b) * H + j] : (REAL)0;
                for (int t = len - 1; t >= 0; --t) {
                    const size_t o = (((size_t)l * B + b) * T + t) * H;
                    const REAL *xt = in + ((size_t)b * T + t) * In_l;
                    const REAL *hp = (t == 0) ? NULL : out + ((size_t)b * T + t - 1) * H;
                    const REAL *dyt = dout + ((size_t)b * T + t) * H;
                    for (int j = 0; j < H; ++j) {
                        const REAL r = sr[o + j], z = sz[o + j], n = sn[o + j], hn = shn[o + j];
                        const REAL hprev = hp ? hp[j] : (REAL)0;
                        const REAL d = dyt[j] + dh[j];
                        const REAL dn = d * ((REAL)1 - z);
                        const REAL dz = d * (hprev - n);
                        const REAL dan = dn * ((REAL)1 - n * n);
                        const REAL dar = dan * hn * r * ((REAL)1 - r);
                        const REAL daz = dz * z * ((REAL)1 - z);
                        dgi[j] = dar; dgi[H + j] = daz; dgi[2 * H + j] = dan;
                        dgh[j] = dar; dgh[H + j] = daz; dgh[2 * H + j] = dan * r;
                        dh[j] = d * z; /* carry through the z * h_prev term */
                    }
                    REAL *dxt = din + ((size_t)b * T + t) * In_l;
                    for (int g = 0; g < 3 * H; ++g) {
                        const REAL a = dgi[g], c = dgh[g];
                        for (int k = 0; k < In_l; ++k) {
                            dxt[k] += a * w_ih[(size_t)g * In_l + k];
                            gw_ih[(size_t)g * In_l + k] += a * xt[k];
                        }
                        for (int k = 0; k < H; ++k) {
                            dh[k] += c * w_hh[(size_t)g * H + k];
                            if (hp) gw_hh[(size_t)g * H + k] += c * hp[k];
                        }
                        gb_ih[g] += a;
                        gb_hh[g] += c;
                    }
                }
            }
            if (l > 0)
                for (size_t i = 0; i < BT * (size_t)H; ++i) dout[i] = din[i];
            else
                for (size_t i = 0; i < BT * (size_t)In; ++i) dx[i] = din[i];
        }
        free(dout); free(din); free(dh); free(dgi); free(dgh);
    }
    free(act); free(sr); free(sz); free(sn); free(shn); free(gi); free(gh);
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
