"""Attention hook in the call shape Stable-Diffusion front ends use.

The reference is consumed by two external repos that monkey-patch their host's attention function with
`FlashAttentionFunction.apply` (reference README.md:35-37: ComfyUI and sd-webui plugins; neither is in the
reference tree).  Both hosts hand attention three [B, N, heads*dim_head] tensors.  `attention_bnhd` is that
adapter for the gfx950 operator: it views the inputs as [B, N, H, D] — no transpose, no copy — and calls the
operator with BNHD_fmt=True, the zero-copy layout of rocwmma_fattn/kernel_fp16.cu:328-333
(bench_with_sdpa_BNHD.py:106 is the reference's own use of it).

    from rocwmma_fattn.sd_hook import attention_bnhd, install_comfyui, install_webui
    out = attention_bnhd(q, k, v, heads)          # q [B, Nq, H*D], k/v [B, Nkv, H*D] -> [B, Nq, H*D]
    install_comfyui()                             # optional: patch comfy.ldm.modules.attention.optimized_attention
    install_webui()                               # optional: patch (ldm|sgm).modules.attention.CrossAttention.forward (sd-webui)

Masks: the reference ignores its `mask` argument (FlashAttn.py:49/:74; README.md:45 lists it as to do), so its hooks have to
send masked calls back to the host's own attention.  Here a mask goes to the kernels (`flash_attention`, C-ABI fa2_fwd_bias):
bool = keep-mask, float = additive bias, in the shapes the SD hosts use — [Nq, Nkv], [B, Nq, Nkv] (batch first, as ComfyUI's
attention functions read a 3-D mask) or [B, H | 1, Nq, Nkv].
"""
import torch

from .FlashAttn import FlashAttentionFunction, flash_attention

__all__ = ["attention_bnhd", "install_comfyui", "install_webui"]

_MAX_HEAD_DIM = 512      # the forward kernels reach 512 (SD VAE attention: one head of 512)


def attention_bnhd(q, k, v, heads, mask=None, causal=False, scale=None, fallback=None):
    """q [B, Nq, heads*D], k, v [B, Nkv, heads*D] (any float dtype; non-half inputs run and return as bf16,
    host.cpp:42-45) -> [B, Nq, heads*D].  `mask`: None, or a bool keep-mask / additive float bias of shape [Nq, Nkv],
    [B, Nq, Nkv] or [B, H | 1, Nq, Nkv].  `fallback(q, k, v, heads, mask)` is used when the head dim exceeds the largest
    kernel, when a masked call needs a gradient at a head dim above 256 (the masked backward stops there) and when the mask does not
    broadcast to [B, H, Nq, Nkv]; without a fallback those cases raise.  (LoRA training through a masked cross-attention — head dims
    40..160 — runs the kernels: fa2_fwd_bias / fa2_bwd_bias.)"""
    b, nq, inner = q.shape
    d = inner // heads
    if d > _MAX_HEAD_DIM or inner != heads * d:
        if fallback is None:
            raise NotImplementedError("fa2 sd_hook: head dims > %d need the host's own attention" % _MAX_HEAD_DIM)
        return fallback(q, k, v, heads, mask)
    if mask is not None and fallback is not None:
        needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
        if (needs_grad and d > 256) or not _mask_broadcasts(mask, b, heads, nq, k.shape[1]):
            return fallback(q, k, v, heads, mask)
    if mask is not None and mask.dim() == 3:
        mask = mask.unsqueeze(1)                      # [B, Nq, Nkv] -> [B, 1, Nq, Nkv] (comfy.ldm.modules.attention reads it so)
    out_dtype = q.dtype
    q4 = q.reshape(b, nq, heads, d)
    k4 = k.reshape(b, k.shape[1], heads, d)
    v4 = v.reshape(b, v.shape[1], heads, d)
    if mask is not None:
        o = flash_attention(q4, k4, v4, mask, causal, scale, True)
    else:
        o = FlashAttentionFunction.apply(q4, k4, v4, None, causal, scale, True)      # BNHD_fmt=True
    o = o.reshape(b, nq, inner)
    return o if o.dtype == out_dtype or out_dtype not in (torch.float16, torch.bfloat16) else o.to(out_dtype)


def _mask_broadcasts(mask, b, heads, nq, nkv):
    """True if `mask` ([Nq, Nkv], [B, Nq, Nkv] or [B, H | 1, Nq, Nkv]) broadcasts to [B, heads, Nq, Nkv]."""
    shape = tuple(mask.shape)
    if len(shape) == 3:
        shape = (shape[0], 1) + shape[1:]
    if len(shape) not in (2, 4):
        return False
    want = (b, heads, nq, nkv)[4 - len(shape):]
    return all(s == w or s == 1 for s, w in zip(shape, want))


def install_comfyui():
    """Patch ComfyUI's optimized attention (comfy.ldm.modules.attention.optimized_attention) in place.
    Returns the original function.  ComfyUI is not a dependency: this raises ImportError when it is absent."""
    import comfy.ldm.modules.attention as attn_mod  # noqa: WPS433 (optional host package)
    original = attn_mod.optimized_attention

    def fa2_attention(q, k, v, heads, mask=None, attn_precision=None, skip_reshape=False, **kwargs):
        if skip_reshape or kwargs.get("skip_output_reshape"):
            return original(q, k, v, heads, mask=mask, attn_precision=attn_precision, skip_reshape=skip_reshape, **kwargs)
        return attention_bnhd(q, k, v, heads, mask=mask,
                              fallback=lambda q_, k_, v_, h_, m_: original(q_, k_, v_, h_, mask=m_, attn_precision=attn_precision))

    attn_mod.optimized_attention = fa2_attention
    return original


def webui_cross_attention_forward(self, x, context=None, mask=None, **kwargs):
    """Replacement for `CrossAttention.forward` of the ldm / sgm attention modules sd-webui runs (the call shape of its
    `sd_hijack_optimizations` functions): projections and output layer are the module's own, the attention in between is the
    gfx950 operator on the zero-copy [B, N, H, D] view.  A call this operator cannot serve (masked with gradients, oversize head
    dim) goes to torch's scaled_dot_product_attention, as sd-webui's own sdp optimisation does.
    `mask` is ldm's: a boolean PER-KEY mask [B, ...] (True = attend), flattened to [B, Nkv] and shared by heads and query rows
    (ldm/modules/attention.py: `rearrange(mask, 'b ... -> b (...)')`, `repeat(mask, 'b j -> (b h) () j')`) — it is handed on as
    the [B, 1, 1, Nkv] key-padding form the kernels serve with one load per KV tile.
    Not reproduced: sd-webui's hypernetwork application to `context` and its `upcast_attn` handling (both live in the
    functions of sd_hijack_optimizations this replaces); a UI with hypernetworks loaded should keep its own optimisation."""
    h = self.heads
    context = x if context is None else context
    q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)

    def sdpa(q_, k_, v_, heads, m_):
        b, _, inner = q_.shape
        t = lambda a: a.reshape(b, a.shape[1], heads, inner // heads).transpose(1, 2)  # noqa: E731
        if m_ is not None and m_.dim() == 3:
            m_ = m_.unsqueeze(1)
        o = torch.nn.functional.scaled_dot_product_attention(t(q_), t(k_), t(v_), attn_mask=m_)
        return o.transpose(1, 2).reshape(b, q_.shape[1], inner)

    if mask is not None:
        mask = mask.reshape(mask.shape[0], 1, 1, -1)          # [B, ...] per key -> [B, 1, 1, Nkv]
        if mask.dtype != torch.bool:
            mask = mask.to(torch.bool)
    out = attention_bnhd(q, k, v, h, mask=mask, fallback=sdpa)
    return self.to_out(out.to(x.dtype))


def install_webui():
    """Patch `CrossAttention.forward` of `ldm.modules.attention` (SD 1.x / 2.x) and, when present, `sgm.modules.attention`
    (SDXL) — the second front end the reference is consumed through (reference README.md:35-37).  Returns {module name: original
    forward}.  Neither package is a dependency: raises ImportError when none of them is importable."""
    import importlib
    patched = {}
    for name in ("ldm.modules.attention", "sgm.modules.attention"):
        try:
            mod = importlib.import_module(name)
        except ImportError:
            continue
        patched[name] = mod.CrossAttention.forward
        mod.CrossAttention.forward = webui_cross_attention_forward
    if not patched:
        raise ImportError("fa2 sd_hook: neither ldm.modules.attention nor sgm.modules.attention is importable")
    return patched
