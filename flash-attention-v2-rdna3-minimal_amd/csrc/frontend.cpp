// frontend.cpp — the reference's pybind module `flash_attn_wmma` (rocwmma_fattn/host.cpp:60-64:
// forward(q, k, v, Br, Bc, causal, scale, permute_NH) -> [O_fwd, q_pad, k_pad, v_pad, O, L]) as a compiled front end
// over the C-ABI of libfa2_gfx950.so.  It does exactly what rocwmma_fattn/FlashAttn.py::_FlashAttnWmma.forward does in
// Python (dtype switch of host.cpp:30-45, D padded to a multiple of 8 only, O/L allocated on q's device with the
// reference's N-padded shapes, launch on torch's current stream) — about 6 us of host time per call instead of 11, which
// only matters for back-to-back tiny calls (an SDXL cross-attention layer).  Optional: when this module is not built the
// Python implementation is used; results are identical (tests/test_parity_gpu.py::test_compiled_front_end_matches_python).
// `backward` (unmasked calls) likewise: the Python backward costs ~50 us of host time per call, which SD-size training shapes are bound by.
//
// Host-only C++ (g++): no device code here, the kernels live in libfa2_gfx950.so.
#include <mutex>
#include <cstdlib>
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm presents its devices as "cuda": the masquerading guard / stream
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "fa2_gfx950.h"

namespace {

bool strides_ok(const at::Tensor& t) {
    const auto s = t.strides();
    return s[3] == 1 && ((s[0] | s[1] | s[2]) & 7) == 0 && (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15u) == 0;
}

at::Tensor kernel_ready(const at::Tensor& t) { return strides_ok(t) ? t : t.contiguous(); }

// Workspace sizes are a function of (dtype, shape, device, options): asked once per shape and option epoch instead of on every call
// (an SDXL UNet step makes 140 attention calls over 6 shapes).  Per thread: no lock on the call path.
struct WsKey {
    int which, dtype, dev; int64_t b, h, n, nkv, d;
    bool operator==(const WsKey& o) const { return which == o.which && dtype == o.dtype && dev == o.dev && b == o.b && h == o.h && n == o.n && nkv == o.nkv && d == o.d; }
};
size_t cached_ws_bytes(int which, int dtype, int dev, int64_t b, int64_t h, int64_t n, int64_t nkv, int64_t d) {
    struct Entry { WsKey key; size_t bytes; };
    thread_local std::vector<Entry> cache;
    thread_local int epoch = -1;
    const int now = fa2_get_option("epoch");
    if (now != epoch) { cache.clear(); epoch = now; }
    const WsKey key{which, dtype, dev, b, h, n, nkv, d};
    for (const Entry& e : cache)
        if (e.key == key) return e.bytes;
    const size_t bytes = which == 0 ? fa2_fwd_workspace_bytes(dtype, (int)b, (int)h, (int)n, (int)nkv, (int)d, 0)
                                    : fa2_bwd_workspace_bytes(dtype, (int)b, (int)h, (int)n, (int)nkv, (int)d, 0);
    if (cache.size() >= 64) cache.clear();
    cache.push_back({key, bytes});
    return bytes;
}

// The split's scratch: ONE block per (device, stream), reused by every later call on that stream and grown when a call needs more (kernels of one
// stream run in order: the merge of call n has read the tiles before the parts of call n + 1 write them).  The operator's footprint is then its outputs
// plus at most 64 MiB per stream in use, not a fresh block per call in flight (the reference records peak memory on every run, bench_with_sdpa.py:34).
// While the stream is being captured into a graph the block comes from the caching allocator (the capture's private pool keeps it alive for the
// replays).  FA2_WS_POOL=0: per-call allocation.  Forward (caller's thread) and backward (the autograd engine's device thread) share the pool: a mutex.
struct WsSlot { int dev; hipStream_t stream; at::Tensor ws; };
std::mutex g_ws_mu;
std::vector<WsSlot> g_ws_pool;

int64_t workspace_pool_bytes() {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    int64_t n = 0;
    for (const WsSlot& s : g_ws_pool) n += s.ws.numel();
    return n;
}

constexpr size_t kWsKeep = (size_t)4 << 20;      // a block up to this size stays whatever the next call needs

// a call on `stream` that needs no scratch: a pooled block above kWsKeep is released (the caching allocator keeps it; the next call that splits takes
// it back) — the operator's footprint follows the calls being made, not the largest one ever made (bench_with_sdpa.py:34 records the peak per run)
void workspace_unused(int dev, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (size_t i = 0; i < g_ws_pool.size(); ++i)
        if (g_ws_pool[i].dev == dev && g_ws_pool[i].stream == stream) {
            if ((size_t)g_ws_pool[i].ws.numel() > kWsKeep) g_ws_pool.erase(g_ws_pool.begin() + i);
            return;
        }
}

at::Tensor workspace(size_t bytes, const at::Tensor& like, hipStream_t stream) {
    static const bool pool_on = [] { const char* e = std::getenv("FA2_WS_POOL"); return !(e && e[0] == '0'); }();
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!pool_on || (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone))
        return at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
    using Slot = WsSlot;
    std::vector<Slot>& pool = g_ws_pool;
    const int dev = like.device().index();
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (size_t i = 0; i < pool.size(); ++i)
        if (pool[i].dev == dev && pool[i].stream == stream) {
            // grown when a call needs more, let go when a call needs less than half of a block above kWsKeep (round 6: the pool was a high-water mark)
            const size_t have = (size_t)pool[i].ws.numel();
            if (have < bytes || (have > kWsKeep && have > 2 * bytes)) {
                pool[i].ws = at::Tensor();
                pool[i].ws = at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
            }
            Slot s = pool[i];
            pool.erase(pool.begin() + i);
            pool.push_back(s);                      // most recently used last
            return s.ws;
        }
    if (pool.size() >= 8) pool.erase(pool.begin());  // least recently used stream
    pool.push_back({dev, stream, at::empty({(int64_t)bytes}, like.options().dtype(at::kByte))});
    return pool.back().ws;
}

std::vector<at::Tensor> forward(at::Tensor q, at::Tensor k, at::Tensor v, int64_t Br, int64_t Bc, int64_t flags, double scale,
                                bool permute_NH) {
    // flags: the reference's `causal` (0 / 1; a Python bool converts), or the C-ABI's call flags FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE
    const bool causal = (flags & FA2_FLAG_CAUSAL) != 0;
    (void)Bc;
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "fa2: q, k, v must be 4-D ([B,H,N,D] or [B,N,H,D] with BNHD_fmt)");
    TORCH_CHECK(q.is_cuda() && k.is_cuda() && v.is_cuda(), "fa2: q, k, v must be on a ROCm device (no CPU path in this operator)");
    TORCH_CHECK(k.device() == q.device() && v.device() == q.device(), "fa2: q, k, v must be on the same device");
    int dtype_code;
    if (q.scalar_type() == at::kHalf) {
        TORCH_CHECK(k.scalar_type() == at::kHalf && v.scalar_type() == at::kHalf, "fa2: q, k, v must share one dtype");
        dtype_code = FA2_DTYPE_F16;
    } else {
        dtype_code = FA2_DTYPE_BF16;     // host.cpp:42-45: everything else runs (and returns) as bf16
        if (q.scalar_type() != at::kBFloat16 || k.scalar_type() != at::kBFloat16 || v.scalar_type() != at::kBFloat16) {
            q = q.to(at::kBFloat16);
            k = k.to(at::kBFloat16);
            v = v.to(at::kBFloat16);
        }
    }
    const int n_ax = permute_NH ? 1 : 2, h_ax = permute_NH ? 2 : 1;
    const int64_t b = q.size(0), h = q.size(h_ax), n = q.size(n_ax), d = q.size(3), n_kv = k.size(n_ax);
    TORCH_CHECK(k.size(0) == b && k.size(h_ax) == h && k.size(3) == d && v.sizes() == k.sizes(), "fa2: inconsistent q/k/v shapes");
    TORCH_CHECK(fa2_padded_head_dim((int)(d + ((8 - d % 8) % 8))) > 0, "fa2: head dim ", d, " is larger than the largest gfx950 kernel");
    const int64_t d_pad = (8 - d % 8) % 8, d_kernel = d + d_pad;
    at::Tensor qp = q, kp = k, vp = v;
    if (d_pad) {
        qp = at::constant_pad_nd(q, {0, d_pad});
        kp = at::constant_pad_nd(k, {0, d_pad});
        vp = at::constant_pad_nd(v, {0, d_pad});
    }
    qp = kernel_ready(qp);
    kp = kernel_ready(kp);
    vp = kernel_ready(vp);

    const int64_t nq_pad = (Br - n % Br) % Br;
    at::Tensor O, L;
    const auto f32 = q.options().dtype(at::kFloat);
    if (nq_pad) {
        std::vector<int64_t> oshape = permute_NH ? std::vector<int64_t>{b, n + nq_pad, h, d_kernel} : std::vector<int64_t>{b, h, n + nq_pad, d_kernel};
        O = at::empty(oshape, qp.options());
        O.narrow(n_ax, n, nq_pad).zero_();
        L = at::empty({b, h, n + nq_pad}, f32);
        L.narrow(2, n, nq_pad).zero_();
    } else {
        O = at::empty_like(qp);
        if (!strides_ok(O)) O = at::empty(qp.sizes(), qp.options());
        L = at::empty({b, h, n}, f32);
    }
    auto s3 = [&](const at::Tensor& t, int64_t* out) {
        out[0] = t.stride(0);
        out[1] = t.stride(h_ax);
        out[2] = t.stride(n_ax);
    };
    int64_t qs[3], ks[3], vs[3], os[3], ls[2] = {h * (n + nq_pad), n + nq_pad};
    s3(qp, qs);
    s3(kp, ks);
    s3(vp, vs);
    s3(O, os);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
    const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q.device().index()).stream();
    // scratch for the KV-split of a partly filled last round of workgroups (fa2_fwd_ws): the per-stream block above
    at::Tensor ws;
    const size_t ws_bytes = causal ? 0 : cached_ws_bytes(0, dtype_code, q.device().index(), b, h, n, n_kv, d_kernel);
    if (ws_bytes) ws = workspace(ws_bytes, q, stream);
    else workspace_unused(q.device().index(), stream);
    const int rc = fa2_fwd_ws(dtype_code, qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), O.data_ptr(), L.data_ptr<float>(), (int)b, (int)h,
                              (int)n, (int)n_kv, (int)d_kernel, qs, ks, vs, os, ls, (float)scale, (int)(flags & 3),
                              ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, (void*)stream);
    TORCH_CHECK(rc == 0, "fa2 call failed (", rc, "): ", fa2_error_string(rc));
    at::Tensor O_fwd = O;
    if (nq_pad) O_fwd = O_fwd.narrow(n_ax, 0, n);
    if (d_pad) O_fwd = O_fwd.narrow(3, 0, d);
    return {O_fwd, qp, kp, vp, O, L};
}

// backward(Q, K, V, O, dO, L, act_n, act_nkv, act_d, Br, Bc, causal, scale, permute_NH) -> [dQ, dK, dV] of the reference's module
// (rocwmma_fattn/host.cpp:47-58, kernel_fp16.cu:878-1028), what _FlashAttnWmma.backward does in Python for unmasked calls: outputs and the delta
// workspace from torch's allocator, the split's scratch when fa2_bwd_workspace_bytes asks for it, results sliced to the actual sizes.
std::vector<at::Tensor> backward(at::Tensor Q, at::Tensor K, at::Tensor V, at::Tensor O, at::Tensor dO, at::Tensor L, int64_t act_n, int64_t act_nkv,
                                 int64_t act_d, int64_t Br, int64_t Bc, bool causal, double scale, bool permute_NH) {
    (void)Br;
    (void)Bc;
    TORCH_CHECK(Q.is_cuda() && dO.is_cuda(), "fa2: tensors must be on a ROCm device (no CPU path in this operator)");
    const int n_ax = permute_NH ? 1 : 2, h_ax = permute_NH ? 2 : 1;
    const int64_t b = Q.size(0), h = Q.size(h_ax), dk = Q.size(3);
    const int dtype_code = Q.scalar_type() == at::kHalf ? FA2_DTYPE_F16 : FA2_DTYPE_BF16;
    if (dO.scalar_type() != Q.scalar_type()) dO = dO.to(Q.scalar_type());          // host.cpp:47-58 dispatches on dO's dtype; Q's wins here
    if (dO.size(3) != dk) dO = at::constant_pad_nd(dO, {0, dk - dO.size(3)});        // kernel_fp16.cu:900-905
    dO = kernel_ready(dO);
    at::Tensor dQ = at::empty(Q.sizes(), Q.options()), dK = at::empty(K.sizes(), K.options()), dV = at::empty(V.sizes(), V.options());
    at::Tensor delta = at::empty({b, h, L.size(2)}, L.options());
    if (L.strides() != delta.strides()) L = L.contiguous();
    auto s3 = [&](const at::Tensor& t, int64_t* out) {
        out[0] = t.stride(0);
        out[1] = t.stride(h_ax);
        out[2] = t.stride(n_ax);
    };
    int64_t qs[3], ks[3], vs[3], os[3], gs[3], dqs[3], dks[3], dvs[3], ls[2] = {L.stride(0), L.stride(1)};
    s3(Q, qs); s3(K, ks); s3(V, vs); s3(O, os); s3(dO, gs); s3(dQ, dqs); s3(dK, dks); s3(dV, dvs);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
    const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(Q.device().index()).stream();
    at::Tensor ws;
    const size_t ws_bytes = causal ? 0 : cached_ws_bytes(1, dtype_code, Q.device().index(), b, h, act_n, act_nkv, dk);
    if (ws_bytes) ws = workspace(ws_bytes, Q, stream);
    else workspace_unused(Q.device().index(), stream);
    const int rc = fa2_bwd_ws(dtype_code, Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), L.data_ptr<float>(), dQ.data_ptr(),
                              dK.data_ptr(), dV.data_ptr(), delta.data_ptr<float>(), (int)b, (int)h, (int)act_n, (int)act_nkv, (int)dk, qs, ks, vs, os, gs,
                              dqs, dks, dvs, ls, (float)scale, causal ? 1 : 0, ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, (void*)stream);
    TORCH_CHECK(rc == 0, "fa2 call failed (", rc, "): ", fa2_error_string(rc));
    return {dQ.narrow(n_ax, 0, act_n).narrow(3, 0, act_d), dK.narrow(n_ax, 0, act_nkv).narrow(3, 0, act_d), dV.narrow(n_ax, 0, act_nkv).narrow(3, 0, act_d)};
}

// The autograd node of FlashAttentionFunction (rocwmma_fattn/FlashAttn.py:45-92 of the reference: forward saves q_pad, k_pad, v_pad, O, L; backward
// returns dQ, dK, dV) as a C++ node: the engine runs a ROCm node's backward on its device thread, and a Python node makes that thread take the
// GIL and walk the Python wrapper first — tens of microseconds a call, which SD-size training shapes (20 .. 60 us of kernels) are bound by.
// Same Br rule (FlashAttn.py:56-67), same saved tensors, same results as the Python class (tests/test_parity_gpu.py).
const auto op_forward = &forward;       // (the node's own members hide the names)
const auto op_backward = &backward;

struct AttentionNode : public torch::autograd::Function<AttentionNode> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor q, at::Tensor k, at::Tensor v, bool causal, double scale, bool permute_NH) {
        const int n_ax = permute_NH ? 1 : 2;
        const int64_t d = q.size(3), n = q.size(n_ax), n_kv = k.size(n_ax);
        // the forward of a call that will be differentiated: FA2_FLAG_EXACT_SCALE (the backward recomputes P from the scores L was formed from)
        std::vector<at::Tensor> r = op_forward(q, k, v, d > 384 ? 32 : 64, 128, (causal ? FA2_FLAG_CAUSAL : 0) | FA2_FLAG_EXACT_SCALE, scale, permute_NH);
        ctx->save_for_backward({r[1], r[2], r[3], r[4], r[5]});
        ctx->saved_data["n"] = n;
        ctx->saved_data["n_kv"] = n_kv;
        ctx->saved_data["d"] = d;
        ctx->saved_data["causal"] = causal;
        ctx->saved_data["scale"] = scale;
        ctx->saved_data["permute_NH"] = permute_NH;
        return r[0];
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto s = ctx->get_saved_variables();
        std::vector<at::Tensor> g = op_backward(s[0], s[1], s[2], s[3], grads[0], s[4], ctx->saved_data["n"].toInt(), ctx->saved_data["n_kv"].toInt(),
                                               ctx->saved_data["d"].toInt(), 128, 128, ctx->saved_data["causal"].toBool(),
                                               ctx->saved_data["scale"].toDouble(), ctx->saved_data["permute_NH"].toBool());
        return {g[0], g[1], g[2], at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor attention(at::Tensor q, at::Tensor k, at::Tensor v, bool causal, double scale, bool permute_NH) {
    return AttentionNode::apply(q, k, v, causal, scale, permute_NH);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled front end of the gfx950 FlashAttention-2 operator (forward of the reference's flash_attn_wmma module)";
    m.def("forward", &forward, "forward(q, k, v, Br, Bc, causal (bool, or the C-ABI's call flags), scale, permute_NH) -> [O_fwd, q_pad, k_pad, v_pad, O, L]");
    m.def("attention", &attention, "attention(q, k, v, causal, scale, permute_NH) -> O, differentiable (the C++ autograd node of FlashAttentionFunction)");
    m.def("workspace_pool_bytes", &workspace_pool_bytes, "bytes the per-stream scratch blocks of the split hold right now");
    m.def("backward", &backward, "backward(Q, K, V, O, dO, L, act_n, act_nkv, act_d, Br, Bc, causal, scale, permute_NH) -> [dQ, dK, dV]");
}
