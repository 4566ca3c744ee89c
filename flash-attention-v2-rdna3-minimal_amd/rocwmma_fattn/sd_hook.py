"""Attention hook in the call shape Stable-Diffusion front ends use.

The reference is consumed by two external repos that monkey-patch their host's attention function with
`FlashAttentionFunction.apply` (reference README.md:35-37: ComfyUI and sd-webui plugins; neither is in the
reference tree).  Both hosts hand attention three [B, N, heads*dim_head] tensors.  `attention_bnhd` is that
adapter for the gfx950 operator: it views the inputs as [B, N, H, D] — no transpose, no copy — and calls the
operator with BNHD_fmt=True, the zero-copy layout of rocwmma_fattn/kernel_fp16.cu:328-333
(bench_with_sdpa_BNHD.py:106 is the reference's own use of it).

    from rocwmma_fattn.sd_hook import attention_bnhd, install_comfyui
    out = attention_bnhd(q, k, v, heads)          # q [B, Nq, H*D], k/v [B, Nkv, H*D] -> [B, Nq, H*D]
    install_comfyui()                             # optional: patch comfy.ldm.modules.attention.optimized_attention

Masks: the reference ignores its `mask` argument (FlashAttn.py:49/:74; README.md:45 lists it as to do), so its hooks have to
send masked calls back to the host's own attention.  Here a mask goes to the kernels (`flash_attention`, C-ABI fa2_fwd_bias):
bool = keep-mask, float = additive bias, in the shapes the SD hosts use — [Nq, Nkv], [B, Nq, Nkv] (batch first, as ComfyUI's
attention functions read a 3-D mask) or [B, H | 1, Nq, Nkv].
"""
import torch

from .FlashAttn import FlashAttentionFunction, flash_attention

__all__ = ["attention_bnhd", "install_comfyui"]

_MAX_HEAD_DIM = 512      # the forward kernels reach 512 (SD VAE attention: one head of 512)


def attention_bnhd(q, k, v, heads, mask=None, causal=False, scale=None, fallback=None):
    """q [B, Nq, heads*D], k, v [B, Nkv, heads*D] (any float dtype; non-half inputs run and return as bf16,
    host.cpp:42-45) -> [B, Nq, heads*D].  `mask`: None, or a bool keep-mask / additive float bias of shape [Nq, Nkv],
    [B, Nq, Nkv] or [B, H | 1, Nq, Nkv].  `fallback(q, k, v, heads, mask)` is used when the head dim exceeds the largest
    kernel; without a fallback that case raises."""
    b, nq, inner = q.shape
    d = inner // heads
    if d > _MAX_HEAD_DIM or inner != heads * d:
        if fallback is None:
            raise NotImplementedError("fa2 sd_hook: head dims > %d need the host's own attention" % _MAX_HEAD_DIM)
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
