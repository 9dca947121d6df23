/*
 * tgru -- C ABI of the MI355X (gfx950) multi-layer GRU forward / BPTT library.
 *
 * The library sits behind the `torch.nn.GRU` module interface exactly as the reference exercises
 * it: nn.GRU(input_size, hidden_size, num_layers, batch_first=True), bias, unidirectional,
 * dropout 0 (outdated_files/TimeGANv1.py:35-38,115-118,194-197,272-275,351-354; scorer stack
 * metrics/GeneralRNN.py:41-44), called on a PackedSequence and followed by pad_packed_sequence
 * (outdated_files/TimeGANv1.py:56-65).  There is no FFI in the reference: the operator boundary
 * is `_VF.gru(data, batch_sizes, hx, flat_weights, ...)` inside torch/nn/modules/rnn.py:1427-1437,
 * and these entry points are what a binding for that operator would call.
 *
 * Conventions
 *   - all tensors are fp32, dense, in device memory; the caller owns every buffer; the library
 *     never allocates, frees or keeps a pointer after return; every launch goes to `stream`;
 *     no entry point synchronises with the host.
 *   - w_ih / w_hh / b_ih / b_hh (and their gradient twins) are HOST arrays of L DEVICE pointers,
 *     one per layer, with torch's shapes: weight_ih_l{k} (3H, In_k), weight_hh_l{k} (3H, H),
 *     bias_*_l{k} (3H); gate row blocks ordered r, z, n; In_0 = In, In_k = H for k > 0.
 *   - layout of x / y / dy / dx:
 *       TGRU_LAYOUT_TBF    (T, B, F) time-major (the library's native order),
 *       TGRU_LAYOUT_BTF    (B, T, F) batch-major (batch_first=True),
 *       TGRU_LAYOUT_PACKED rows packed like PackedSequence.data: row = toff[t] + b.
 *   - lengths: device int32[B], sequence b runs for t < lengths[b]; must be non-increasing
 *     (enforce_sorted=True) and in [1, T]; NULL means every sequence has length T.  Rows with
 *     t >= lengths[b] of a TBF / BTF output are written as 0 (pad_packed_sequence with
 *     padding_value 0), dx rows there are 0, h_n[l][b] is the state at t = lengths[b]-1.
 *   - toff: device int32[T+1], prefix sum of PackedSequence.batch_sizes; required for
 *     TGRU_LAYOUT_PACKED (together with lengths), ignored otherwise.
 *   - return value: TGRU_OK or a negative TGRU_E_* code; tgru_error_string() names it.
 */
#ifndef TGRU_H
#define TGRU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tgru_stream_t; /* a hipStream_t */

enum {
    TGRU_LAYOUT_TBF = 0,
    TGRU_LAYOUT_BTF = 1,
    TGRU_LAYOUT_PACKED = 2
};

enum {
    TGRU_OK = 0,
    TGRU_E_DIMS = -1,      /* a dimension is <= 0 or too large for 32-bit row indexing */
    TGRU_E_NULL = -2,      /* a required pointer is NULL */
    TGRU_E_LAYOUT = -3,    /* unknown layout, or PACKED without lengths / toff */
    TGRU_E_ALIGN = -4,     /* a buffer is not 16-byte aligned */
    TGRU_E_LAUNCH = -5,    /* a kernel launch failed; hipGetLastError() has the cause */
    TGRU_E_NOGPU = -6,     /* the current device is not a gfx950 GPU */
    TGRU_E_TOOBIG = -7     /* tgru_forward_static: the call needs more than TGRU_STATIC_WORKSPACE_BYTES */
};

/* Library version (major*10000 + minor*100 + patch).  No reference counterpart. */
int tgru_version(void);

/*
 * Number of kernel launches the library has issued in this process so far (all entry points, all
 * streams).  Read it before and after a step to learn the step's launch count.  No reference counterpart.
 */
unsigned long long tgru_launch_count(void);

/*
 * The library reads its kernel-family switches (environment variables TGRU_FORCE_GENERIC, TGRU_NO_VEC,
 * TGRU_NO_MID, TGRU_NO_STEP, TGRU_NO_WIDE_HEAD, ...; "1" selects the alternative kernels; the full list is in
 * the comment above use_small() in csrc/tgru_api.hip) once per process; this
 * call reads them again.  No reference counterpart.
 */
void tgru_reload_switches(void);

/*
 * Hidden sizes above 128 (multiples of 32) run on persistent kernels whose workgroups hand each other the
 * state of a time step through memory and therefore must all be resident on the GPU at once.  The library sizes
 * every launch for an otherwise idle GPU (per XCD: an eighth of the compute units); a kernel of another stream
 * or process that holds compute units for seconds can still keep a workgroup from starting.  Every wait is
 * bounded (TGRU_SPIN_LIMIT polls, default 2^24, several seconds), so such a call ends, but with wrong values.
 * It is never silent: the call's last kernel then writes NaN over the first row of every result of the call
 * (y and h_n; dx, dh0 and the first row of the weight / bias gradients) and adds one to a counter on that
 * GPU.  This function returns that counter for the current device -- the number of tgru_forward /
 * tgru_backward calls whose waits gave up since it was last cleared -- and zeroes it when `clear` is non-zero.
 * It copies from the device (a host synchronisation): synchronise the streams whose calls are to be counted
 * first, and do not call it under graph capture.  -1: no gfx950 device, or the copy failed.
 * TGRU_NO_PERSIST=1 selects one launch per time step instead (no residency requirement).
 * No reference counterpart.
 */
long tgru_persist_timeouts(int clear);

/* Text for a TGRU_E_* code.  No reference counterpart. */
const char *tgru_error_string(int code);

/*
 * Bytes of state the forward pass must save for tgru_backward (the analogue of the `reserve`
 * that torch's fused RNN paths keep between aten::gru forward and its autograd backward):
 * per layer the gate activations r, z, n and the candidate's hidden pre-activation
 * W_hn h + b_hn, plus each layer's output sequence in time-major order.
 */
size_t tgru_reserve_bytes(int B, int T, int In, int H, int L, int layout);

/* Bytes of scratch memory tgru_forward / tgru_backward need in `workspace`. */
size_t tgru_forward_workspace_bytes(int B, int T, int In, int H, int L);
size_t tgru_backward_workspace_bytes(int B, int T, int In, int H, int L);

/*
 * Multi-layer GRU forward.  Replaces nn.GRU.forward -> _VF.gru at
 * outdated_files/TimeGANv1.py:61,139,220,299,376 together with the pack / pad calls around it
 * (:56-59, :62-65), whose row reordering is folded into the kernels' addressing.
 *
 *   x        (rows, In) in `layout`
 *   h0       (L, B, H) initial state, or NULL for zeros (the reference never passes hx)
 *   y        (rows, H) in `layout`: the top layer's h_t for every step
 *   h_n      (L, B, H) final state per layer, or NULL if not wanted
 *   reserve  tgru_reserve_bytes() bytes, or NULL for inference (then tgru_backward cannot follow)
 *   workspace tgru_forward_workspace_bytes() bytes
 */
int tgru_forward(const float *x, const float *h0,
                 const int32_t *lengths, const int32_t *toff,
                 const float *const *w_ih, const float *const *w_hh,
                 const float *const *b_ih, const float *const *b_hh,
                 float *y, float *h_n, float *reserve, float *workspace,
                 int B, int T, int In, int H, int L, int layout,
                 tgru_stream_t stream);

/*
 * tgru_forward for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops (generation / scoring), harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * Same arguments and results as tgru_forward, but no `workspace`: the scratch memory (input projections,
 * packed W_hh, intermediate layer outputs) is a fixed array of TGRU_STATIC_WORKSPACE_BYTES inside the
 * library, one copy on every GPU the library runs on, held until the process ends.  The call allocates and
 * frees nothing.  A call that needs more (tgru_forward_workspace_bytes(B, T, In, H, L) >
 * TGRU_STATIC_WORKSPACE_BYTES) returns TGRU_E_TOOBIG and launches nothing.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads) race on it; calls on one stream are ordered and safe.  Nothing in the array has to be cleared
 * between calls (every float is written before it is read within a call).  The first call on a device
 * looks up the array's address there; make it outside graph capture.
 */
#define TGRU_STATIC_WORKSPACE_BYTES (64u << 20)
int tgru_forward_static(const float *x, const float *h0,
                        const int32_t *lengths, const int32_t *toff,
                        const float *const *w_ih, const float *const *w_hh,
                        const float *const *b_ih, const float *const *b_hh,
                        float *y, float *h_n, float *reserve,
                        int B, int T, int In, int H, int L, int layout,
                        tgru_stream_t stream);

/*
 * Backprop-through-time of tgru_forward.  Replaces the autograd graph torch records under
 * aten::gru and replays in loss.backward() (outdated_files/TimeGANv1.py:604,679,772,785,807).
 *
 *   dy       gradient of y in `layout`, or NULL for zeros
 *   dh_n     gradient of h_n (L, B, H), or NULL for zeros
 *   x, h0, y, reserve, weights: exactly what tgru_forward read and wrote
 *   dx       (rows, In) in `layout`, or NULL when the input needs no gradient
 *   dh0      (L, B, H), or NULL
 *   dw_ih / dw_hh / db_ih / db_hh: host arrays of L device pointers; every buffer is
 *            overwritten (accumulate == 0) or added to (accumulate != 0).  All four arrays NULL: no
 *            weight / bias gradients are formed, only dx / dh0 (a network the step does not update, whose
 *            gradients the reference zeroes unseen: outdated_files/TimeGANv1.py:765-769,780-782,796-799)
 *   workspace tgru_backward_workspace_bytes() bytes
 */
int tgru_backward(const float *dy, const float *dh_n,
                  const float *x, const float *h0, const float *y,
                  const int32_t *lengths, const int32_t *toff,
                  const float *const *w_ih, const float *const *w_hh,
                  const float *const *b_ih, const float *const *b_hh,
                  const float *reserve, float *workspace,
                  float *dx, float *dh0,
                  float *const *dw_ih, float *const *dw_hh,
                  float *const *db_ih, float *const *db_hh,
                  int B, int T, int In, int H, int L, int layout, int accumulate,
                  tgru_stream_t stream);

/*
 * The batched input projection on its own:  out (M, N) = a (M, K) . w (N, K)^T + bias (N),
 * fp32 MFMA GEMM.  This is the `x . W_ih^T + b_ih` term of the cell equations
 * (torch/nn/modules/rnn.py:1191-1194) over all rows at once; exported so that callers and
 * tests can reach the kernel directly.  bias may be NULL.
 */
int tgru_input_projection(const float *a, const float *w, const float *bias, float *out,
                          int M, int N, int K, tgru_stream_t stream);

/*
 * LayerNorm over the last dimension, rows of `width` floats (a multiple of 4, at most 1024): the
 * nn.LayerNorm(dim_hidden) stages of the heads (outdated_files/TimeGANv1.py:45,67) at hidden sizes the
 * fused head kernels do not cover.  forward: y = (x - mean) * rstd * gamma + beta, stats[2r] = mean,
 * stats[2r + 1] = rstd (biased variance + eps, torch.nn.functional.layer_norm).  backward: dx (NULL: not
 * wanted), dgamma / dbeta (NULL: not wanted; accumulate 0 / 1) from x, dy and the saved stats; `workspace`
 * holds one partial column sum per workgroup, added in a fixed order.  gamma / beta may be NULL (1 / 0).
 * The _act_ forms take `relu` (0 / 1): with 1 they are LayerNorm followed by ReLU in one pass -- the
 * `nn.LayerNorm, nn.ReLU` pair inside every feature MLP (outdated_files/TimeGANv1.py:45-46,122-123,201-202,
 * 280-281) -- forward y = max(LayerNorm(x), 0); backward takes dy with respect to THAT y, recomputes the
 * pre-activation from x, the stats, gamma and beta and lets dy pass where it was positive.
 */
int tgru_layer_norm_supported(int width);
size_t tgru_layer_norm_backward_workspace_bytes(long rows, int width);
int tgru_layer_norm_forward(const float *x, const float *gamma, const float *beta, float *y, float *stats,
                            long rows, int width, float eps, tgru_stream_t stream);
int tgru_layer_norm_backward(const float *x, const float *dy, const float *stats, const float *gamma, float *dx,
                             float *dgamma, float *dbeta, float *workspace, long rows, int width, int accumulate,
                             tgru_stream_t stream);
int tgru_layer_norm_act_forward(const float *x, const float *gamma, const float *beta, float *y, float *stats,
                                long rows, int width, float eps, int relu, tgru_stream_t stream);
int tgru_layer_norm_act_backward(const float *x, const float *dy, const float *stats, const float *gamma,
                                 const float *beta, float *dx, float *dgamma, float *dbeta, float *workspace,
                                 long rows, int width, int relu, int accumulate, tgru_stream_t stream);

/*
 * out (width) (=|+=) the sum of the rows of a (rows, width): the bias gradient autograd forms for a Linear
 * layer (outdated_files/TimeGANv1.py:39-50) from the gradient of its output.  width as for
 * tgru_layer_norm_supported; one partial per workgroup in `workspace`, added in a fixed order; accumulate 0 / 1.
 */
size_t tgru_column_sum_workspace_bytes(long rows, int width);
int tgru_column_sum(const float *a, float *out, float *workspace, long rows, int width, int accumulate,
                    tgru_stream_t stream);

/*
 * Input gradient of a Linear layer:  dx (rows, in_f) = dy (rows, out_f) . w (out_f, in_f), fp32 MFMA GEMM;
 * the product autograd forms for the Linear layers around the GRU (outdated_files/TimeGANv1.py:39-50,67-68)
 * at widths the fused head kernels do not cover.  (The layer's forward is tgru_input_projection.)
 */
int tgru_linear_input_grad(const float *dy, const float *w, float *dx, int rows, int out_f, int in_f,
                           tgru_stream_t stream);

/*
 * Weight gradient of a Linear layer over many rows:  dw (out_f, in_f) (=|+=) dy (rows, out_f)^T . x (rows, in_f),
 * the term autograd forms for the Linear layers around the GRU (outdated_files/TimeGANv1.py:39-50,67-68)
 * when the row count is that of a whole batch of sequences (B*T rows against 96 x 96 outputs: a product
 * that has to be split over the ROWS to fill the GPU).  fp32 MFMA GEMM split over the rows, partial
 * results in `workspace` (tgru_linear_weight_grad_workspace_bytes), summed in a fixed order.
 * accumulate: 0 or 1.
 */
size_t tgru_linear_weight_grad_workspace_bytes(int rows, int out_f, int in_f);
int tgru_linear_weight_grad(const float *dy, const float *x, float *dw, float *workspace,
                            int rows, int out_f, int in_f, int accumulate, tgru_stream_t stream);

/*
 * Two independent calls evaluated together, e.g. the embedder and the generator of one training step
 * (outdated_files/TimeGANv1.py:517-518), or the recovery and supervisor networks applied to the same
 * latent sequence (:436-437).  The structs carry exactly the arguments of tgru_forward / tgru_backward
 * and the results are those of the two separate calls; when both calls run on the
 * one-sequence-per-workgroup kernels (H, In <= 32, L <= 4, B <= 1024, a padded layout) and have the same
 * number of layers they share ONE launch -- the two stacks then run side by side on the GPU instead of
 * one after the other.  No reference counterpart (PyTorch issues the calls one by one).
 */
typedef struct tgru_forward_args {
    const float *x, *h0;
    const int32_t *lengths, *toff;
    const float *const *w_ih, *const *w_hh, *const *b_ih, *const *b_hh;
    float *y, *h_n, *reserve, *workspace;
    int B, T, In, H, L, layout;
} tgru_forward_args;
typedef struct tgru_backward_args {
    const float *dy, *dh_n, *x, *h0, *y;
    const int32_t *lengths, *toff;
    const float *const *w_ih, *const *w_hh, *const *b_ih, *const *b_hh;
    const float *reserve;
    float *workspace, *dx, *dh0;
    float *const *dw_ih, *const *dw_hh, *const *db_ih, *const *db_hh;
    int B, T, In, H, L, layout, accumulate;
} tgru_backward_args;
int tgru_forward_pair(const tgru_forward_args *a, const tgru_forward_args *b, tgru_stream_t stream);
int tgru_backward_pair(const tgru_backward_args *a, const tgru_backward_args *b, tgru_stream_t stream);

/*
 * The dense "head" the reference puts before / after every GRU stack, fused over M independent rows:
 *
 *      y = [LayerNorm(din)] -> Linear(din -> dmid) -> [LayerNorm(dmid) -> ReLU -> Linear(dmid -> dout)]
 *
 * It replaces these torch module chains (outdated_files/TimeGANv1.py): the feature MLP in front of
 * the embedder / generator GRU (:45-50,55 and :282-287,294), `norm` + `emb_linear` / `g_linear`
 * behind it (:67-68, :305-306), `norm` + the 4-stage `mlp` of recovery and supervisor (:145-148,
 * :226-227) and `norm` + Linear(H,1) of the discriminator (:382-383).  LayerNorm is torch's
 * (biased variance, eps inside the square root).
 *   ln0_w / ln0_b NULL: no input LayerNorm.   w2 NULL: shallow head, y = Linear1 output, dout == dmid.
 *   x (M, din), y (M, dout) row-major; weights in torch layout (out_features, in_features).
 *   Supported (tgru_head_supported): din, dout <= 32 and dmid <= 64 with the weights in registers; beyond
 *   that din, dout <= 128 and dmid <= 192 with both matrices in LDS (at most 150 KB for the padded pair),
 *   the Linear weight gradients then formed as products split over the rows.  TGRU_E_DIMS otherwise;
 *   callers keep the torch modules then.
 *
 * tgru_head_backward recomputes the forward from x, writes dx (may be NULL) and the parameter
 * gradients into ONE flat buffer `dparams` of tgru_head_grad_floats() floats laid out as
 *      dW1 (dmid*din) | db1 (dmid) | dln0_w (din) | dln0_b (din) | dln1_w (dmid) | dln1_b (dmid) |
 *      dW2 (dout*dmid) | db2 (dout)
 * (entries of absent stages are written as zeros unless accumulate == 1; dparams NULL: no parameter
 * gradients are formed, only dx -- for a network whose parameters the step does not update).  `workspace` needs
 * tgru_head_backward_workspace_bytes() bytes (16-byte aligned for the LDS-resident widths: TGRU_E_ALIGN
 * otherwise).  accumulate != 0 adds to dparams.
 */
int tgru_head_supported(int din, int dmid, int dout, int deep);   /* deep: Linear2 present */
size_t tgru_head_grad_floats(int din, int dmid, int dout);
size_t tgru_head_backward_workspace_bytes(int M, int din, int dmid, int dout);
int tgru_head_forward(const float *x, float *y, int M, int din, int dmid, int dout,
                      const float *ln0_w, const float *ln0_b, const float *w1, const float *b1,
                      const float *ln1_w, const float *ln1_b, const float *w2, const float *b2,
                      float eps, tgru_stream_t stream);
int tgru_head_backward(const float *x, const float *dy, float *dx, int M, int din, int dmid, int dout,
                       const float *ln0_w, const float *ln0_b, const float *w1, const float *b1,
                       const float *ln1_w, const float *ln1_b, const float *w2, const float *b2,
                       float eps, float *dparams, float *workspace, int accumulate,
                       tgru_stream_t stream);

/*
 * Two independent heads in one launch (the output stages of two networks that were evaluated as a
 * pair, see tgru_forward_pair); the fields are the arguments of tgru_head_forward / tgru_head_backward,
 * the results those of the two separate calls.  No reference counterpart.
 */
typedef struct tgru_head_args {
    const float *x;
    float *y;                /* forward only */
    const float *dy;         /* backward only */
    float *dx;               /* backward only, may be NULL */
    int M, din, dmid, dout;
    const float *ln0_w, *ln0_b, *w1, *b1, *ln1_w, *ln1_b, *w2, *b2;
    float eps;
    float *dparams, *workspace;   /* backward only */
    int accumulate;
} tgru_head_args;
int tgru_head_forward_pair(const tgru_head_args *a, const tgru_head_args *b, tgru_stream_t stream);
int tgru_head_backward_pair(const tgru_head_args *a, const tgru_head_args *b, tgru_stream_t stream);

/*
 * Deferred reductions.  Every backward kernel of the small-stack path (and of the heads) leaves one
 * partial parameter-gradient vector per workgroup in `workspace`, and a short second kernel sums them.
 * With accumulate == TGRU_DEFER_REDUCE that second kernel is NOT launched: the caller asks
 * tgru_backward_reduce_jobs / tgru_head_reduce_jobs for the reductions that are still owed (flat jobs
 * dst[i] = sum_{k < count} src[k * stride + i], i < n), gathers the jobs of as many calls as it likes
 * and runs them in one launch with tgru_reduce_jobs -- before anything reads the gradients and while
 * the workspaces are still alive.  Calls that do not run on that path ignore the flag (their
 * gradients are complete on return) and report zero jobs.  No reference counterpart.
 */
#define TGRU_DEFER_REDUCE 2
#define TGRU_MAX_REDUCE_JOBS 96      /* jobs per launch; tgru_reduce_jobs splits longer lists */
typedef struct tgru_reduce_job {
    const float *src;
    int64_t stride;
    int32_t count, n;
    float *dst;
} tgru_reduce_job;
/* both return the number of jobs written (0: nothing deferred) or a negative TGRU_E_* code */
int tgru_backward_reduce_jobs(const tgru_backward_args *a, tgru_reduce_job *jobs, int max_jobs);
int tgru_head_reduce_jobs(const tgru_head_args *a, tgru_reduce_job *jobs, int max_jobs);
int tgru_reduce_jobs(const tgru_reduce_job *jobs, int n_jobs, tgru_stream_t stream);

/*
 * torch.optim.Adam (fused, no weight decay, no amsgrad) for up to TGRU_MAX_ADAM_JOBS small tensors (at
 * most TGRU_MAX_ADAM_ELEMS elements each) in one launch: step += 1; m += (1-b1)(g-m);
 * v = b2 v + (1-b2) g^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).  `step` points to
 * the tensor's own step counter (a device float, as torch keeps it for fused / capturable Adam); no two
 * jobs of a launch may share a counter.  `skip` (device float or NULL): non-zero makes the launch a
 * no-op, counters included -- the on-device form of the reference's
 * `if d_loss > threshold: d_opt.step()` (outdated_files/TimeGANv1.py:805-808).
 * Replaces torch's _foreach_add_ + _fused_adam_ launches.
 */
#define TGRU_MAX_ADAM_JOBS 64
#define TGRU_MAX_ADAM_ELEMS 65536
typedef struct tgru_adam_job {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq, *step;
    int32_t n, reserved;
} tgru_adam_job;
int tgru_adam_jobs(const tgru_adam_job *jobs, int n_jobs, float lr, float beta1, float beta2, float eps,
                   const float *skip, tgru_stream_t stream);

/*
 * The loss expressions of the three training phases, fused (no reference FFI; they replace the
 * torch compositions at outdated_files/TimeGANv1.py:410-553):
 *   TGRU_LOSS_EMBED_SIDE     in = {x_tilde, x}                    out[0] = 10 sqrt(mse)            (:420-421)
 *   TGRU_LOSS_SUPERVISOR     in = {h_sup, h}                      out[0] = mse(h_sup[:, :-1], h[:, 1:])  (:470)
 *   TGRU_LOSS_EMBED_MAIN     in = {x_tilde, x, h_sup, h}          out = e_loss, e_loss0, e_loss_t0 (:444-451)
 *   TGRU_LOSS_DISCRIMINATOR  in = {y_real, y_fake, y_fake_e}      out[0] = d_loss                  (:497-503)
 *   TGRU_LOSS_GENERATOR      in = {y_fake, y_fake_e, h_sup, h, x_hat, x}   out[0] = g_loss         (:533-553)
 * x* are (B, T, F), h* are (B, T, H), y* are (B, T, 1), all dense fp32.  `in` / `din` / `dout` are HOST
 * arrays of device pointers (6 / 6 / 3 entries; unused entries and unwanted gradients NULL).
 *
 * Forward = tgru_loss_partial(pass 0) then tgru_loss_final, through one buffer `buf` of
 * layout[3] floats (tgru_loss_layout); the outputs are the 3 floats at buf[layout[2]], and `buf` is
 * what tgru_loss_backward needs.  Data-parallel ranks holding shards of one batch instead run
 * partial(pass 1), sum buf[0 .. layout[0]) over ranks, then (generator only) partial(pass 2) and
 * sum buf[layout[0] .. layout[0]+layout[1]) over ranks, then final: every rank then holds the loss
 * of the whole batch and tgru_loss_backward gives its gradient w.r.t. the local rows.
 * tgru_loss_backward writes din[i] = sum_j dout[j][0] * d out[j] / d in[i] for every non-NULL din[i].
 */
enum {
    TGRU_LOSS_EMBED_SIDE = 0,
    TGRU_LOSS_SUPERVISOR = 1,
    TGRU_LOSS_EMBED_MAIN = 2,
    TGRU_LOSS_DISCRIMINATOR = 3,
    TGRU_LOSS_GENERATOR = 4
};
int tgru_loss_layout(int kind, int T, int F, int64_t layout[4]);
int tgru_loss_partial(int kind, int pass, const float *const *in, int B, int T, int H, int F, float *buf,
                      tgru_stream_t stream);
int tgru_loss_final(int kind, int B, int T, int H, int F, float gamma, float *buf, tgru_stream_t stream);
int tgru_loss_backward(int kind, const float *const *in, float *const *din, const float *const *dout,
                       int B, int T, int H, int F, float gamma, const float *buf, tgru_stream_t stream);

/*
 * Uniform noise in [0, 1) with 24 random bits per float, the distribution of the `torch.rand_like(x)` the joint
 * phase draws (outdated_files/TimeGANv1.py:762,793), from a counter-based generator: out[i] is element
 * first + i of noise stream `stream` under `seed`, a pure function of those three numbers (Philox4x32-10:
 * counter = {element / 4, stream}, key = seed, word element % 4, (word >> 8) * 2^-24).  A data-parallel rank that
 * owns rows [lo, hi) of a global batch therefore draws exactly its rows (first = lo * row width), and all ranks
 * together draw the tensor one process would draw.  Replaces the draw only; the reference's own generator
 * (torch's, whose streams depend on the device and the launch shape) is not reproduced.
 */
int tgru_uniform(float *out, long n, unsigned long long seed, unsigned long long stream,
                 unsigned long long first, tgru_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* TGRU_H */
