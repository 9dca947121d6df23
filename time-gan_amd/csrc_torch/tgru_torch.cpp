// Thin torch extension over the C ABI of libtgru.so (include/tgru.h): the differentiable multi-layer GRU as a
// C++ autograd function, for eager callers.  It takes tensors, reads the current HIP stream of their device
// (torch's current stream on ROCm: getCurrentHIPStreamMasqueradingAsCUDA), allocates outputs / workspace / reserve through torch's caching allocator and
// calls tgru_forward / tgru_backward -- the calls nn.GRU.forward -> _VF.gru makes in the reference
// (torch/nn/modules/rnn.py:1427-1437 as reached from outdated_files/TimeGANv1.py:61,139,220,299,376) and the
// backward autograd replays for it.  The backward pass runs in autograd's own thread without entering Python.
//
// Same semantics as timegan_amd.functional._GRUFunction (the ctypes binding, which stays for the paired and
// deferred-reduction calls of the trainer and for the tests): weights flat in torch's order, lengths / toff
// int32 device tensors or undefined, layout TGRU_LAYOUT_*; an unused output's gradient arrives undefined and is
// passed as NULL; no weight wanting a gradient -> NULL gradient arrays (only dx / dh0 are formed).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <vector>

#include "tgru.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char *what) {
    TORCH_CHECK(rc == TGRU_OK, what, " failed: ", tgru_error_string(rc), " (", rc, ")");
}

// dense fp32 on the GPU, storage starting on a 16-byte boundary (the library requires it of every buffer)
Tensor dense16(const Tensor &t) {
    Tensor c = t.contiguous();
    return (reinterpret_cast<uintptr_t>(c.data_ptr()) & 15) ? c.clone() : c;
}

void require_gpu(const Tensor &t, const char *what) {
    TORCH_CHECK(t.is_cuda(), what, " is on ", t.device(),
                ": the tgru GRU only runs on a gfx950 GPU and has no CPU fallback");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, what, " must be float32, got ", t.scalar_type());
}

const float *fptr(const Tensor &t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
const int32_t *iptr(const Tensor &t) { return t.defined() ? t.data_ptr<int32_t>() : nullptr; }

struct Tables {      // host arrays of per-layer device pointers, torch's flat order split by kind
    std::vector<const float *> w_ih, w_hh, b_ih, b_hh;
    explicit Tables(const std::vector<Tensor> &w) {
        const size_t L = w.size() / 4;
        for (size_t l = 0; l < L; ++l) {
            w_ih.push_back(w[4 * l].data_ptr<float>());
            w_hh.push_back(w[4 * l + 1].data_ptr<float>());
            b_ih.push_back(w[4 * l + 2].data_ptr<float>());
            b_hh.push_back(w[4 * l + 3].data_ptr<float>());
        }
    }
};

class GRUFunction : public torch::autograd::Function<GRUFunction> {
public:
    // inputs: x, h0 (may be undefined), lengths, toff (may be undefined), then the 4 L weights
    static variable_list forward(AutogradContext *ctx, Tensor x, const c10::optional<Tensor> &h0_, const c10::optional<Tensor> &lengths_,
                                 const c10::optional<Tensor> &toff_, int64_t B, int64_t T, int64_t layout, bool grad_mode,
                                 at::TensorList weight_list) {
        // (an absent optional tensor is no input of the autograd node; a present one is, gradient wanted or not)
        Tensor h0 = h0_.value_or(Tensor()), lengths = lengths_.value_or(Tensor()), toff = toff_.value_or(Tensor());
        std::vector<Tensor> weights = weight_list.vec();
        TORCH_CHECK(!weights.empty() && weights.size() % 4 == 0, "expected weight_ih, weight_hh, bias_ih, bias_hh per layer");
        require_gpu(x, "input");
        const int L = (int)(weights.size() / 4);
        const int H = (int)weights[1].size(1), In = (int)weights[0].size(1);
        if (layout == TGRU_LAYOUT_TBF) {
            TORCH_CHECK(x.dim() == 3 && x.size(0) == T && x.size(1) == B && x.size(2) == In, "x is not (T,B,In)");
        } else if (layout == TGRU_LAYOUT_BTF) {
            TORCH_CHECK(x.dim() == 3 && x.size(0) == B && x.size(1) == T && x.size(2) == In, "x is not (B,T,In)");
        } else {
            TORCH_CHECK(x.dim() == 2 && x.size(1) == In, "packed x is not (rows,In)");
        }
        bool need_grad = x.requires_grad() || (h0.defined() && h0.requires_grad());
        x = dense16(x);
        for (auto &w : weights) {
            require_gpu(w, "weight");
            need_grad = need_grad || w.requires_grad();
            w = dense16(w);
        }
        if (h0.defined()) h0 = dense16(h0);
        need_grad = need_grad && grad_mode;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
        const auto f32 = x.options();
        auto shape = x.sizes().vec();
        shape.back() = H;
        Tensor y = torch::empty(shape, f32), h_n = torch::empty({L, B, H}, f32);
        Tensor ws = torch::empty({(int64_t)(tgru_forward_workspace_bytes(B, T, In, H, L) / 4)}, f32);
        Tensor reserve;
        if (need_grad) reserve = torch::empty({(int64_t)(tgru_reserve_bytes(B, T, In, H, L, (int)layout) / 4)}, f32);
        const Tables tb(weights);
        check(tgru_forward(fptr(x), fptr(h0), iptr(lengths), iptr(toff), tb.w_ih.data(), tb.w_hh.data(), tb.b_ih.data(),
                           tb.b_hh.data(), y.data_ptr<float>(), h_n.data_ptr<float>(),
                           reserve.defined() ? reserve.data_ptr<float>() : nullptr, ws.data_ptr<float>(), (int)B, (int)T,
                           In, H, L, (int)layout, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream()),
              "tgru_forward");
        ctx->saved_data["dims"] = std::vector<int64_t>{B, T, In, H, L, layout};
        ctx->saved_data["has_h0"] = h0.defined();
        ctx->saved_data["has_lengths"] = lengths.defined();
        ctx->saved_data["has_toff"] = toff.defined();
        ctx->set_materialize_grads(false);      // an unused h_n (or y) arrives in backward undefined, not as zeros
        variable_list saved{x, y};
        if (h0.defined()) saved.push_back(h0);
        if (lengths.defined()) saved.push_back(lengths);
        if (toff.defined()) saved.push_back(toff);
        if (reserve.defined()) saved.push_back(reserve);
        saved.insert(saved.end(), weights.begin(), weights.end());
        ctx->save_for_backward(saved);
        return {y, h_n};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const int B = (int)dims[0], T = (int)dims[1], In = (int)dims[2], H = (int)dims[3], L = (int)dims[4], layout = (int)dims[5];
        const auto saved = ctx->get_saved_variables();
        size_t k = 0;
        Tensor x = saved[k++], y = saved[k++], h0, lengths, toff, reserve;
        if (ctx->saved_data["has_h0"].toBool()) h0 = saved[k++];
        if (ctx->saved_data["has_lengths"].toBool()) lengths = saved[k++];
        if (ctx->saved_data["has_toff"].toBool()) toff = saved[k++];
        TORCH_CHECK(saved.size() == k + 1 + 4 * (size_t)L, "backward of a GRU call whose forward ran without gradients");
        reserve = saved[k++];
        std::vector<Tensor> weights(saved.begin() + k, saved.end());
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
        const auto f32 = x.options();
        Tensor dy = grads[0].defined() ? dense16(grads[0]) : torch::zeros_like(y);      // undefined: only h_n was used
        Tensor dh_n = grads[1].defined() ? dense16(grads[1]) : Tensor();
        Tensor ws = torch::empty({(int64_t)(tgru_backward_workspace_bytes(B, T, In, H, L) / 4)}, f32);
        // needs_input_grad counts the tensor inputs that were present: x, then h0 / lengths / toff where given, then the
        // weights; the returned list has one entry per ARGUMENT of forward(), the four scalars included
        const int w0 = 1 + (h0.defined() ? 1 : 0) + (lengths.defined() ? 1 : 0) + (toff.defined() ? 1 : 0);
        Tensor dx = ctx->needs_input_grad(0) ? torch::empty_like(x) : Tensor();
        Tensor dh0 = (h0.defined() && ctx->needs_input_grad(1)) ? torch::empty_like(h0) : Tensor();
        bool want_w = false;
        for (int i = 0; i < 4 * L; ++i) want_w = want_w || ctx->needs_input_grad(w0 + i);
        // one flat buffer for the weight gradients, every view starting on a 16-byte boundary
        std::vector<Tensor> gw(4 * L);
        std::vector<float *> dw_ih, dw_hh, db_ih, db_hh;
        if (want_w) {
            int64_t n = 0;
            std::vector<int64_t> off;
            for (const auto &w : weights) { off.push_back(n); n += (w.numel() + 3) / 4 * 4; }
            Tensor flat = torch::empty({n}, f32);
            for (int i = 0; i < 4 * L; ++i) gw[i] = flat.narrow(0, off[i], weights[i].numel()).view(weights[i].sizes());
            for (int l = 0; l < L; ++l) {
                dw_ih.push_back(gw[4 * l].data_ptr<float>());
                dw_hh.push_back(gw[4 * l + 1].data_ptr<float>());
                db_ih.push_back(gw[4 * l + 2].data_ptr<float>());
                db_hh.push_back(gw[4 * l + 3].data_ptr<float>());
            }
        }
        const Tables tb(weights);
        check(tgru_backward(dy.data_ptr<float>(), fptr(dh_n), fptr(x), fptr(h0), fptr(y), iptr(lengths), iptr(toff),
                            tb.w_ih.data(), tb.w_hh.data(), tb.b_ih.data(), tb.b_hh.data(), reserve.data_ptr<float>(),
                            ws.data_ptr<float>(), dx.defined() ? dx.data_ptr<float>() : nullptr,
                            dh0.defined() ? dh0.data_ptr<float>() : nullptr, want_w ? dw_ih.data() : nullptr,
                            want_w ? dw_hh.data() : nullptr, want_w ? db_ih.data() : nullptr, want_w ? db_hh.data() : nullptr,
                            B, T, In, H, L, layout, 0, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream()),
              "tgru_backward");
        variable_list out{dx, dh0, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
        for (int i = 0; i < 4 * L; ++i) out.push_back(ctx->needs_input_grad(w0 + i) ? gw[i] : Tensor());
        return out;
    }
};

std::vector<Tensor> gru_apply(const Tensor &x, const c10::optional<Tensor> &h0, const c10::optional<Tensor> &lengths,
                              const c10::optional<Tensor> &toff, int64_t B, int64_t T, int64_t layout,
                              const std::vector<Tensor> &weights) {
    auto r = GRUFunction::apply(x, h0, lengths, toff, B, T, layout, torch::GradMode::is_enabled(), at::TensorList(weights));
    return {r[0], r[1]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd binding of the tgru GRU (include/tgru.h) for eager callers";
    m.def("gru_apply", &gru_apply, "differentiable multi-layer GRU: (x, h0, lengths, toff, B, T, layout, weights) -> (y, h_n)",
          py::arg("x"), py::arg("h0"), py::arg("lengths"), py::arg("toff"), py::arg("B"), py::arg("T"), py::arg("layout"),
          py::arg("weights"));
}
