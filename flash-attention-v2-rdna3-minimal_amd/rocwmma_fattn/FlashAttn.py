"""FlashAttention-2 forward operator for MI355X — same entry point as the reference's
rocwmma_fattn/FlashAttn.py (class FlashAttentionFunction, positional signature
`apply(q, k, v, mask, causal, scale, BNHD_fmt)`, FlashAttn.py:45-49), so bench_with_sdpa.py:99,
precision_test.py:63 and the external ComfyUI / sd-webui hooks keep working unchanged.

Layers (reference counterpart in brackets):
  FlashAttentionFunction.forward   [FlashAttn.py:47-76]      operator: shapes, default scale, save for bwd
  flash_attn_wmma.forward          [host.cpp:30-45 +
                                    kernel_fp16.cu:744-876]  dtype switch, D padding, O/L allocation,
                                                             6-tensor return, calls the C-ABI
  libfa2_gfx950.so : fa2_fwd_*     [fwd_kernel, kernel_fp16.cu:306-544]  hand-written gfx950 kernel

  FlashAttentionFunction.backward  [FlashAttn.py:78-92]      operator: dQ, dK, dV through flash_attn_wmma.backward
  libfa2_gfx950.so : fa2_bwd_*     [bwd_kernel, kernel_fp16.cu:547-740]  gfx950 backward kernels

The host is PyTorch-ROCm for memory and streams only; the compute is the C-ABI library
(include/fa2_gfx950.h).  There is no CPU path: tensors must live on a ROCm device.
"""
import os

import torch

from . import _fa2_lib

__all__ = ["FlashAttentionFunction", "flash_attn_wmma", "flash_attention"]


class _FlashAttnWmma:
    """Stand-in for the reference's JIT-built pybind module `flash_attn_wmma` (FlashAttn.py:23-41):
    same attribute names, argument order and return contract (host.cpp:60-64)."""

    @staticmethod
    def forward(q, k, v, Br, Bc, causal, scale, permute_NH):
        """forward() of the reference's module: the compiled front end (csrc/frontend.cpp, same logic in C++: ~6 us of host
        time per call instead of ~11) when it was built, else forward_py below."""
        # A caller of the reference's module-level API who pairs forward() with backward() himself (host.cpp:30-58) passes a plain bool: when an input
        # requires a gradient the forward is flagged FA2_FLAG_EXACT_SCALE for him — the backward recomputes P from scores scaled in f32, and an L formed
        # from a folded / rounded-P forward would not match them (2-4x the gradient tolerance at |logit| > 30: ADVICE r5).  Anyone else sets the flag.
        flags = _fa2_lib.call_flags(causal)
        if q.requires_grad or k.requires_grad or v.requires_grad:
            flags |= _fa2_lib.FA2_FLAG_EXACT_SCALE
        fe = _frontend()
        if fe is not None:
            return fe.forward(q, k, v, int(Br), int(Bc), flags, float(scale), bool(permute_NH))
        return _FlashAttnWmma.forward_py(q, k, v, Br, Bc, flags, scale, permute_NH)

    @staticmethod
    def forward_bias(q, k, v, bias, Br, Bc, causal, scale, permute_NH):
        """forward() with an attention bias / mask (extension: the reference reserves `mask` but ignores it, FlashAttn.py:49, :74,
        README.md:45).  `bias` broadcasts against [B, H, Nq, Nkv] like torch SDPA's attn_mask: bool = keep-mask, float = additive.
        Same 6-tensor return as forward()."""
        return _FlashAttnWmma.forward_py(q, k, v, Br, Bc, causal, scale, permute_NH, bias=bias)

    @staticmethod
    def forward_py(q, k, v, Br, Bc, causal, scale, permute_NH, bias=None):
        """Returns [O_fwd, q_pad, k_pad, v_pad, O, L] like forward_fp16/forward_bf16 (kernel_fp16.cu:744-876).
        O and L keep the reference's shapes — rows padded to a multiple of Br with a zero tail, O_fwd a view into
        O (kernel_fp16.cu:761, :793-796, :865-875) — but nothing is COPIED to get there: the gfx950 kernels mask
        ragged Nq / Nkv and any D that is a multiple of 8 in-kernel, so q_pad, k_pad, v_pad are the inputs
        themselves (made contiguous if their strides require it; the reference returns padded copies,
        kernel_fp16.cu:767-779).  Only a D that is not a multiple of 8 is zero-padded, to the next multiple of 8.
        Br sizes the N padding of O and L; Bc is accepted for signature compatibility.  `causal`: the reference's bool, or the C-ABI's call flags
        (_fa2_lib.FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE: the operator marks the forward of calls that will be differentiated)."""
        lib = _fa2_lib.load()
        flags = _fa2_lib.call_flags(causal)
        causal = bool(flags & _fa2_lib.FA2_FLAG_CAUSAL)
        if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
            raise RuntimeError("fa2: q, k, v must be 4-D ([B,H,N,D] or [B,N,H,D] with BNHD_fmt)")
        if not q.is_cuda or not k.is_cuda or not v.is_cuda:
            raise RuntimeError("fa2: q, k, v must be on a ROCm device (no CPU path in this operator)")
        if k.device != q.device or v.device != q.device:
            raise RuntimeError("fa2: q, k, v must be on the same device")
        # dtype switch (host.cpp:30-45): half stays half, everything else runs (and returns) as bf16
        if q.dtype == torch.float16:
            dtype_code = _fa2_lib.FA2_DTYPE_F16
            if k.dtype != q.dtype or v.dtype != q.dtype:
                raise RuntimeError("fa2: q, k, v must share one dtype")
        else:
            dtype_code = _fa2_lib.FA2_DTYPE_BF16
            if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
                q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)

        n_ax, h_ax = (1, 2) if permute_NH else (2, 1)
        qsh, ksh = q.shape, k.shape
        b, h, n, d = qsh[0], qsh[h_ax], qsh[n_ax], qsh[3]
        n_kv = ksh[n_ax]
        if ksh[0] != b or ksh[h_ax] != h or ksh[3] != d or v.shape != ksh:
            raise RuntimeError("fa2: inconsistent q/k/v shapes %s %s %s" % (tuple(q.shape), tuple(k.shape), tuple(v.shape)))
        if d > _MAX_HEAD_DIM:
            raise RuntimeError("fa2: head dim %d is larger than the largest gfx950 kernel" % d)
        d_pad = -d % 8
        d_kernel = d + d_pad                     # the head dim handed to the C-ABI (columns up to the kernel's are masked there)

        q_pad, k_pad, v_pad = q, k, v
        if d_pad:
            q_pad, k_pad, v_pad = (torch.nn.functional.pad(t, (0, d_pad)) for t in (q, k, v))
        q_pad, k_pad, v_pad = (_kernel_ready(t) for t in (q_pad, k_pad, v_pad))

        # outputs on q's device (kernel_fp16.cu:793-796): rows padded to a multiple of Br like the reference's, only
        # the padding tail is zero-filled (the kernel writes every real row)
        nq_pad = -n % int(Br)
        if nq_pad:
            oshape = (b, n + nq_pad, h, d_kernel) if permute_NH else (b, h, n + nq_pad, d_kernel)
            O = torch.empty(oshape, dtype=q_pad.dtype, device=q_pad.device)
            O.narrow(n_ax, n, nq_pad).zero_()
            L = torch.empty((b, h, n + nq_pad), dtype=torch.float32, device=q.device)
            L[:, :, n:].zero_()
        else:
            O = torch.empty_like(q_pad)
            if not _strides_ok(O):
                O = torch.empty(q_pad.shape, dtype=q_pad.dtype, device=q_pad.device)
            L = torch.empty((b, h, n), dtype=torch.float32, device=q.device)

        def s3(t):
            st = t.stride()
            key = (st[0], st[h_ax], st[n_ax])
            arr = _STRIDE_CACHE.get(key)
            if arr is None:
                if len(_STRIDE_CACHE) > 4096:
                    _STRIDE_CACHE.clear()
                arr = _STRIDE_CACHE[key] = _fa2_lib.strides3(*key)
            return arr

        dev = q.device.index
        args = (dtype_code, q_pad.data_ptr(), k_pad.data_ptr(), v_pad.data_ptr(), O.data_ptr(), L.data_ptr(),
                b, h, n, n_kv, d_kernel, s3(q_pad), s3(k_pad), s3(v_pad), s3(O),
                _fa2_lib.strides2(h * (n + nq_pad), n + nq_pad), float(scale), flags)
        fn = lib.fa2_fwd
        if bias is not None:
            bias_t, kind, bstr = _prepare_bias(bias, b, h, n, n_kv, q_pad.dtype, q.device)
            args += (bias_t.data_ptr(), kind, _fa2_lib.strides3(*bstr))
            fn = lib.fa2_fwd_bias
        if dev != _current_device():
            with torch.cuda.device(dev):
                rc = _launch_fwd(lib, fn, args, dev, q.device, bias is None and not causal)
        else:
            rc = _launch_fwd(lib, fn, args, dev, q.device, bias is None and not causal)
        if rc:
            _fa2_lib.check(rc)

        O_fwd = O                                # a view into the padded O (kernel_fp16.cu:865-875)
        if nq_pad:
            O_fwd = O_fwd.narrow(n_ax, 0, n)
        if d_pad:
            O_fwd = O_fwd[..., :d]
        return [O_fwd, q_pad, k_pad, v_pad, O, L]

    @staticmethod
    def backward(Q, K, V, O, dO, L, act_n, act_nkv, act_d, Br, Bc, causal, scale, permute_NH, bias=None):
        """Returns [dQ, dK, dV] sliced to the actual sizes, like backward_fp16/backward_bf16
        (host.cpp:47-58, kernel_fp16.cu:878-1028).  Q, K, V, O, L are the tensors the forward returned
        (D a multiple of 8); the gfx950 kernels take the actual Nq / Nkv / D and mask in-kernel, so nothing is
        padded here except a dO whose D differs from Q's."""
        if bias is None:
            fe = _frontend()
            if fe is not None and hasattr(fe, "backward"):      # the compiled front end (csrc/frontend.cpp): the same steps in C++
                return fe.backward(Q, K, V, O, dO, L, int(act_n), int(act_nkv), int(act_d), int(Br), int(Bc), bool(causal), float(scale), bool(permute_NH))
        return _FlashAttnWmma.backward_py(Q, K, V, O, dO, L, act_n, act_nkv, act_d, Br, Bc, causal, scale, permute_NH, bias)

    @staticmethod
    def backward_py(Q, K, V, O, dO, L, act_n, act_nkv, act_d, Br, Bc, causal, scale, permute_NH, bias=None):
        """backward() in Python (masked calls; every call when the compiled front end is absent)."""
        lib = _fa2_lib.load()
        if not (Q.is_cuda and dO.is_cuda):
            raise RuntimeError("fa2: tensors must be on a ROCm device (no CPU path in this operator)")
        n_ax, h_ax = (1, 2) if permute_NH else (2, 1)
        b, h, dk = Q.size(0), Q.size(h_ax), Q.size(3)
        act_n, act_nkv, act_d = int(act_n), int(act_nkv), int(act_d)
        dtype_code = _fa2_lib.FA2_DTYPE_F16 if Q.dtype == torch.float16 else _fa2_lib.FA2_DTYPE_BF16
        if dO.dtype != Q.dtype:
            dO = dO.to(Q.dtype)                       # host.cpp:47-58 dispatches on dO's dtype; Q's wins here
        if dO.size(3) != dk:                          # kernel_fp16.cu:900-905: dO is padded in D like Q
            dO = torch.nn.functional.pad(dO, (0, dk - dO.size(3)))
        dO = _kernel_ready(dO)
        dQ = torch.empty(Q.shape, dtype=Q.dtype, device=Q.device)   # every element [:act_n] is written by its owner
        dK = torch.empty(K.shape, dtype=K.dtype, device=K.device)
        dV = torch.empty(V.shape, dtype=V.dtype, device=V.device)
        delta = torch.empty((b, h, L.size(2)), dtype=torch.float32, device=Q.device)
        if L.stride() != delta.stride():
            L = L.contiguous()

        def s3(t):
            st = t.stride()
            return _fa2_lib.strides3(st[0], st[h_ax], st[n_ax])

        stream = _raw_stream(Q.device.index)
        args = (dtype_code, Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), L.data_ptr(),
                dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), delta.data_ptr(), b, h, act_n, act_nkv, dk,
                s3(Q), s3(K), s3(V), s3(O), s3(dO), s3(dQ), s3(dK), s3(dV),
                _fa2_lib.strides2(L.stride(0), L.stride(1)), float(scale), 1 if causal else 0)
        fn = lib.fa2_bwd
        if bias is not None:         # backward through forward_bias (extension, like the masked forward itself): same bias arguments
            bias_t, kind, bstr = _prepare_bias(bias, b, h, act_n, act_nkv, Q.dtype, Q.device)
            args += (bias_t.data_ptr(), kind, _fa2_lib.strides3(*bstr))
            fn = lib.fa2_bwd_bias
        if not causal:
            # scratch for the split of a partly filled last round of workgroups (fa2_bwd_ws / fa2_bwd_bias_ws, the backward's twins of fa2_fwd_ws)
            need = (lib.fa2_bwd_workspace_bytes if bias is None else lib.fa2_bwd_bias_workspace_bytes)(dtype_code, b, h, act_n, act_nkv, dk, 0)
            if need:
                ws = _workspace(need, Q.device, Q.device.index, stream)
                args += (ws.data_ptr(), need)
                fn = lib.fa2_bwd_ws if bias is None else lib.fa2_bwd_bias_ws
        args += (stream,)
        if Q.device.index != _current_device():
            with torch.cuda.device(Q.device):
                rc = fn(*args)
        else:
            rc = fn(*args)
        if rc:
            _fa2_lib.check(rc)
        if permute_NH:
            return [dQ[:, :act_n, :, :act_d], dK[:, :act_nkv, :, :act_d], dV[:, :act_nkv, :, :act_d]]
        return [dQ[:, :, :act_n, :act_d], dK[:, :, :act_nkv, :act_d], dV[:, :, :act_nkv, :act_d]]


_WS_POOL = {}            # (device index, raw stream) -> uint8 tensor: the split's scratch, allocated once per stream and reused (grown when a call needs more)
_WS_POOL_MAX = 8         # streams with a pooled workspace (least recently used dropped beyond that)
_WS_KEEP = 4 << 20       # a block up to this size stays whatever the next call needs


def _workspace_unused(dev, stream):
    """A call on `stream` that needs no scratch: a pooled block above _WS_KEEP is released (the allocator caches it; the next call that splits takes it
    back) — the operator's footprint follows the calls being made, not the largest one ever made."""
    ws = _WS_POOL.get((dev, stream))
    if ws is not None and ws.numel() > _WS_KEEP:
        del _WS_POOL[(dev, stream)]


def _workspace(need, device, dev, stream):
    """`need` bytes of scratch for a call on `stream`: ONE tensor per (device, stream), reused by every later call on that stream (kernels of one
    stream run in order, so the merge of call n has read the tiles before the parts of call n + 1 write them) — the operator's memory footprint is
    its outputs plus at most 64 MiB per stream in use, instead of a fresh block per call in flight (the reference records peak
    memory on every run, bench_with_sdpa.py:34).  While a graph is being captured the block comes from the allocator (the capture's private pool
    keeps it alive for the graph's replays); FA2_WS_POOL=0 restores per-call allocation."""
    if not _POOL_ON or torch.cuda.is_current_stream_capturing():
        return torch.empty(need, dtype=torch.uint8, device=device)
    key = (dev, stream)
    ws = _WS_POOL.pop(key, None)
    # grown when a call needs more — and (round 6) let go when a call needs less than half of a block above _WS_KEEP: the pool was a high-water mark,
    # every later call of the process carried the largest block any call had needed (the reference harness's D scan: +60 MB on every row)
    if ws is None or ws.numel() < need or (ws.numel() > _WS_KEEP and ws.numel() > 2 * need):
        ws = None                # (drop the old block first: the allocator can hand the same memory back)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        while len(_WS_POOL) >= _WS_POOL_MAX:
            _WS_POOL.pop(next(iter(_WS_POOL)))
    _WS_POOL[key] = ws       # (re-inserted: most recently used last)
    return ws


def workspace_pool_bytes():
    """Bytes the per-stream scratch blocks of this process hold right now (tools/scan_bench.py reports it beside the peak)."""
    n = sum(t.numel() for t in _WS_POOL.values())
    fe = _frontend()
    if fe is not None and hasattr(fe, "workspace_pool_bytes"):      # the compiled front end keeps its own blocks
        n += int(fe.workspace_pool_bytes())
    return n


_POOL_ON = os.environ.get("FA2_WS_POOL", "1") != "0"


def _launch_fwd(lib, fn, args, dev, device, may_split):
    """The C-ABI call on torch's current stream.  Non-causal, unbiased launches go through fa2_fwd_ws when the library can use a
    workspace for this shape (KV-split of the last, partly filled round of workgroups: include/fa2_gfx950.h): scratch memory from
    the per-stream pool above."""
    if may_split:
        # (the size is a function of dtype, shape, device and the library's options: asked once per shape — _fa2_lib.options() and
        #  _fa2_lib.set_option() drop the cache; a stale entry would only cost the split, the library re-plans every call itself)
        key = (args[0], args[6], args[7], args[8], args[9], args[10], dev)
        need = _WS_CACHE.get(key)
        if need is None:
            if len(_WS_CACHE) > 4096:
                _WS_CACHE.clear()
            need = _WS_CACHE[key] = lib.fa2_fwd_workspace_bytes(args[0], *args[6:11], 0)
        if need:
            stream = _raw_stream(dev)
            ws = _workspace(need, device, dev, stream)
            return lib.fa2_fwd_ws(*args, ws.data_ptr(), need, stream)
    stream = _raw_stream(dev)
    if _WS_POOL:
        _workspace_unused(dev, stream)
    return fn(*args, stream)


_FRONTEND = [False]      # False = not looked for yet, None = absent


def _frontend():
    """The optional compiled front end (rocwmma_fattn/_fa2_frontend.so, built by build.py); FA2_FRONTEND=py disables it."""
    if _FRONTEND[0] is False:
        mod = None
        if os.environ.get("FA2_FRONTEND", "") != "py":
            _fa2_lib.load()                       # the kernel library first: the front end links against it
            try:
                from . import _fa2_frontend as mod  # noqa: F401
            except ImportError:
                mod = None
        _FRONTEND[0] = mod
    return _FRONTEND[0]


_STRIDE_CACHE = {}       # (batch, head, row) element strides -> ctypes int64[3] (the arrays are read-only for the library)
_WS_CACHE = _fa2_lib.WS_CACHE      # (dtype, B, H, Nq, Nkv, D, device) -> fa2_fwd_workspace_bytes
_MAX_HEAD_DIM = 512      # largest kernel head dim, forward and backward (fa2_supported_head_dims)
_MAX_MASKED_BWD_HEAD_DIM = 256     # ... of the masked backward (fa2_bwd_bias)


def _raw_stream(device_index):
    """hipStream_t of torch's current stream on that device, as an integer (the cheap form of
    torch.cuda.current_stream(dev).cuda_stream: this is on the path of every call)."""
    return torch._C._cuda_getCurrentRawStream(device_index)


def _current_device():
    return torch._C._cuda_getDevice()


def _strides_ok(t):
    s0, s1, s2, s3 = t.stride()
    return s3 == 1 and not ((s0 | s1 | s2) & 7) and not (t.data_ptr() & 15)


def _prepare_bias(bias, b, h, n, n_kv, io_dtype, device):
    """Attention bias / mask -> (tensor kept alive by the caller, bias_kind, element strides (batch, head, row)) for fa2_fwd_bias.
    Broadcasting follows torch.nn.functional.scaled_dot_product_attention: a 2-, 3- or 4-D tensor is aligned on the right
    against [B, H, Nq, Nkv] and every dimension is 1 or full size; broadcast dimensions become stride 0 — nothing is expanded
    in memory unless the Nkv dimension itself is broadcast or strided."""
    if not torch.is_tensor(bias) or bias.dim() < 2 or bias.dim() > 4:
        raise RuntimeError("fa2: the attention mask must be a 2-, 3- or 4-D tensor broadcastable to [B, H, Nq, Nkv]")
    if bias.device != device:
        raise RuntimeError("fa2: the attention mask must be on the device of q")
    m = bias
    while m.dim() < 4:
        m = m.unsqueeze(0)
    for size, full in zip(m.shape, (b, h, n, n_kv)):
        if size != 1 and size != full:
            raise RuntimeError("fa2: attention mask of shape %s does not broadcast to %s" % (tuple(bias.shape), (b, h, n, n_kv)))
    if m.dtype == torch.bool:
        kind = _fa2_lib.FA2_BIAS_BOOL
    elif m.dtype == torch.float32:
        kind = _fa2_lib.FA2_BIAS_F32
    else:
        kind = _fa2_lib.FA2_BIAS_IO_DTYPE
        if m.dtype != io_dtype:
            m = m.to(io_dtype)
    if m.size(3) != n_kv:
        m = m.expand(m.size(0), m.size(1), m.size(2), n_kv).contiguous()
    elif m.stride(3) != 1 and n_kv > 1:
        m = m.contiguous()
    if kind == _fa2_lib.FA2_BIAS_BOOL:
        m = m.view(torch.uint8)
    strides = tuple(m.stride(i) if m.size(i) > 1 else 0 for i in range(3))
    return m, kind, strides


def _kernel_ready(t):
    """kernel_fp16.cu:780-787 makes a tensor contiguous iff its last stride is not 1; the gfx950
    kernel additionally wants 16-byte aligned rows (strides multiple of 8 elements)."""
    return t if _strides_ok(t) else t.contiguous()


flash_attn_wmma = _FlashAttnWmma()


class FlashAttentionFunction(torch.autograd.Function):

    @staticmethod
    @torch.no_grad()
    def forward(ctx, q, k, v, mask=None, causal=None, scale=None, BNHD_fmt=False, *args, **kwargs):
        # reference: FlashAttn.py:47-76.  `mask` is accepted and ignored there too (only stored).
        D = q.shape[3]
        N = q.shape[2]
        Nkv = k.shape[2]

        Br = 64
        Bc = 128

        if BNHD_fmt:
            N = q.shape[1]
            Nkv = k.shape[1]

        if scale is None:
            scale = D ** -0.5
        if D > 384:
            Br = 32
            Bc = 128

        # a call that will be differentiated scales the f32 product (FA2_FLAG_EXACT_SCALE: the reference kernel's contract, kernel_fp16.cu:164) whatever
        # option "fold" says: the backward then recomputes P from the very scores L was formed from
        # (q, k OR v: the reference looks at q alone, FlashAttn.py:73, and a call in which only K / V need gradients fails in its backward)
        needs_grad = ctx is not None and (q.requires_grad or k.requires_grad or v.requires_grad)
        flags = (_fa2_lib.FA2_FLAG_CAUSAL if causal else 0) | (_fa2_lib.FA2_FLAG_EXACT_SCALE if needs_grad else 0)
        ret = flash_attn_wmma.forward(q, k, v, Br, Bc, flags if needs_grad else bool(causal), scale, BNHD_fmt)

        o, q_bwd, k_bwd, v_bwd, o_bwd, L = ret

        if needs_grad:       # (ctx is None on the inference fast path below, e.g. under torch.no_grad())
            ctx.args = (causal, scale, mask, N, Nkv, D, BNHD_fmt)
            ctx.save_for_backward(q_bwd, k_bwd, v_bwd, o_bwd, L)
        return o

    @staticmethod
    @torch.no_grad()
    def backward(ctx, do):
        # reference: FlashAttn.py:78-92
        causal, scale, mask, N, Nkv, D, BNHD_fmt = ctx.args
        q, k, v, o, L = ctx.saved_tensors
        Br = 128
        Bc = 128
        dQ, dK, dV = flash_attn_wmma.backward(q, k, v, o, do, L, N, Nkv, D, Br, Bc, causal, scale, BNHD_fmt)
        return dQ, dK, dV, None, None, None, None


# Inference fast path: when nothing can require a gradient the autograd.Function machinery (a Python-side graph node
# per call, ~5 us) is skipped and forward() runs directly; `FlashAttentionFunction.apply(...)` keeps the reference's
# call shape and result either way (reference: FlashAttn.py:45-76).
_autograd_apply = FlashAttentionFunction.apply


def _apply(q, k, v, mask=None, causal=None, scale=None, BNHD_fmt=False, *args, **kwargs):
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        fe = _frontend()
        if fe is not None and q.requires_grad and not args and not kwargs and hasattr(fe, "attention"):
            # the same node in C++ (csrc/frontend.cpp::AttentionNode): its backward runs on the autograd engine's device thread without the GIL
            return fe.attention(q, k, v, bool(causal), float(q.shape[3] ** -0.5 if scale is None else scale), bool(BNHD_fmt))
        return _autograd_apply(q, k, v, mask, causal, scale, BNHD_fmt, *args, **kwargs)
    return FlashAttentionFunction.forward(None, q, k, v, mask, causal, scale, BNHD_fmt)


FlashAttentionFunction.apply = staticmethod(_apply)


class _MaskedAttentionFunction(torch.autograd.Function):
    """autograd node of flash_attention(mask=...): forward = fa2_fwd_bias, backward = fa2_bwd_bias (the mask is a constant: no gradient)."""

    @staticmethod
    @torch.no_grad()
    def forward(ctx, q, k, v, mask, causal, scale, BNHD_fmt):
        D = q.shape[3]
        # what fa2_bwd_bias would refuse is refused HERE, before the forward's work is done (a training step must not fail inside loss.backward())
        if D > _MAX_MASKED_BWD_HEAD_DIM:
            raise RuntimeError("fa2: a masked attention call that needs gradients supports head dims up to %d (got %d): fa2_bwd_bias has no kernel above"
                               % (_MAX_MASKED_BWD_HEAD_DIM, D))
        if torch.is_tensor(mask) and mask.dim() >= 2 and mask.shape[-2] > 1:
            n_ax_ = 1 if BNHD_fmt else 2
            if (q.shape[n_ax_] + 63) * k.shape[n_ax_] * (4 if mask.dtype == torch.float32 else 1 if mask.dtype == torch.bool else 2) >= 2 ** 31 - 1:
                raise RuntimeError("fa2: one (batch, head) slice of the attention mask must span < 2 GiB for the masked backward (fa2_bwd_bias)")
        Br = 32 if D > 384 else 64                      # FlashAttn.py:56-67
        o, q_bwd, k_bwd, v_bwd, o_bwd, L = flash_attn_wmma.forward_bias(q, k, v, mask, Br, 128, bool(causal), scale, BNHD_fmt)     # (the biased kernels never fold)
        n_ax = 1 if BNHD_fmt else 2
        ctx.args = (causal, scale, q.shape[n_ax], k.shape[n_ax], D, BNHD_fmt)
        ctx.save_for_backward(q_bwd, k_bwd, v_bwd, o_bwd, L, mask)
        return o

    @staticmethod
    @torch.no_grad()
    def backward(ctx, do):
        causal, scale, N, Nkv, D, BNHD_fmt = ctx.args
        q, k, v, o, L, mask = ctx.saved_tensors
        dQ, dK, dV = flash_attn_wmma.backward(q, k, v, o, do, L, N, Nkv, D, 128, 128, causal, scale, BNHD_fmt, bias=mask)
        return dQ, dK, dV, None, None, None, None


def flash_attention(q, k, v, mask=None, causal=False, scale=None, BNHD_fmt=False):
    """Forward attention that HONOURS `mask` — the extension the reference lists as to do (README.md:45; its
    FlashAttentionFunction accepts the argument and ignores it, FlashAttn.py:49, :74, and `FlashAttentionFunction.apply` here
    keeps doing exactly that so that existing call sites see no change).  `mask` follows
    torch.nn.functional.scaled_dot_product_attention(attn_mask=...): broadcastable to [B, H, Nq, Nkv]; bool = True where
    attention is allowed, float = added to the scaled scores.  mask=None is FlashAttentionFunction.apply.  Rows whose every
    position is masked return zeros.  Differentiable in q, k, v (C-ABI fa2_bwd_bias; head dims up to 256); the mask gets no gradient."""
    if mask is None:
        return FlashAttentionFunction.apply(q, k, v, None, causal, scale, BNHD_fmt)
    D = q.shape[3]
    if scale is None:
        scale = D ** -0.5
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _MaskedAttentionFunction.apply(q, k, v, mask, causal, scale, BNHD_fmt)
    Br = 32 if D > 384 else 64                      # FlashAttn.py:56-67
    return flash_attn_wmma.forward_bias(q, k, v, mask, Br, 128, bool(causal), scale, BNHD_fmt)[0]
