"""CPU tests of the Stable-Diffusion hook logic (rocwmma_fattn/sd_hook.py): the patches are installed into STUB host modules
(`comfy.ldm.modules.attention`, `ldm.modules.attention`) and the gfx950 operator is replaced by a dense torch attention with
the same call shape, so that what runs here is the hook's own code — reshape to the zero-copy [B, N, H, D] view, mask shapes,
the fallbacks (oversize head dim, masked call with gradients, non-broadcastable mask, skip_reshape) and un-patching.
The operator behind the hook is covered on the GPU (tests/test_parity_gpu.py, tests/test_bias_gpu.py)."""
import sys
import types

import pytest
import torch

from rocwmma_fattn import sd_hook


def _dense_bnhd(q, k, v, mask, causal, scale, bnhd):
    assert bnhd is True
    qt, kt, vt = (t.transpose(1, 2).float() for t in (q, k, v))
    s = qt @ kt.transpose(-1, -2) * (q.shape[-1] ** -0.5 if scale is None else scale)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else s + mask.float()
    return (torch.softmax(s, -1) @ vt).transpose(1, 2).to(q.dtype)


class _FakeFn:
    calls = []

    @staticmethod
    def apply(q, k, v, mask, causal, scale, bnhd):
        _FakeFn.calls.append("apply")
        return _dense_bnhd(q, k, v, None, causal, scale, bnhd)


def _fake_flash_attention(q, k, v, mask, causal, scale, bnhd):
    _FakeFn.calls.append("masked")
    return _dense_bnhd(q, k, v, mask, causal, scale, bnhd)


@pytest.fixture
def hook(monkeypatch):
    monkeypatch.setattr(sd_hook, "FlashAttentionFunction", _FakeFn)
    monkeypatch.setattr(sd_hook, "flash_attention", _fake_flash_attention)
    _FakeFn.calls = []
    return sd_hook


def _sdpa_flat(q, k, v, heads, mask=None):
    b, _, inner = q.shape
    t = lambda a: a.reshape(b, a.shape[1], heads, inner // heads).transpose(1, 2)  # noqa: E731
    if mask is not None and mask.dim() == 3:
        mask = mask.unsqueeze(1)
    o = torch.nn.functional.scaled_dot_product_attention(t(q), t(k), t(v), attn_mask=mask)
    return o.transpose(1, 2).reshape(b, q.shape[1], inner)


@pytest.fixture
def comfy_stub(monkeypatch):
    calls = []

    def optimized_attention(q, k, v, heads, mask=None, attn_precision=None, skip_reshape=False, **kwargs):
        calls.append(("original", skip_reshape, mask is not None))
        if skip_reshape:
            return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return _sdpa_flat(q, k, v, heads, mask)

    mods = {}
    for name in ("comfy", "comfy.ldm", "comfy.ldm.modules", "comfy.ldm.modules.attention"):
        mods[name] = types.ModuleType(name)
        monkeypatch.setitem(sys.modules, name, mods[name])
    mods["comfy"].ldm = mods["comfy.ldm"]
    mods["comfy.ldm"].modules = mods["comfy.ldm.modules"]
    mods["comfy.ldm.modules"].attention = mods["comfy.ldm.modules.attention"]
    mods["comfy.ldm.modules.attention"].optimized_attention = optimized_attention
    return mods["comfy.ldm.modules.attention"], calls


def test_install_comfyui_patches_and_serves_plain_and_masked_calls(hook, comfy_stub):
    attn_mod, calls = comfy_stub
    original = hook.install_comfyui()
    assert attn_mod.optimized_attention is not original
    g = torch.Generator().manual_seed(0)
    q, k, v = torch.randn(2, 50, 4 * 16, generator=g), torch.randn(2, 33, 4 * 16, generator=g), torch.randn(2, 33, 4 * 16, generator=g)
    want = _sdpa_flat(q, k, v, 4)
    got = attn_mod.optimized_attention(q, k, v, 4)
    assert _FakeFn.calls == ["apply"] and not calls
    assert torch.allclose(got, want, atol=1e-5)
    # masks in the three shapes ComfyUI passes: [Nq, Nkv], [B, Nq, Nkv] (batch first), [B, 1 | H, Nq, Nkv]
    for shape in ((50, 33), (2, 50, 33), (2, 1, 50, 33), (2, 4, 50, 33)):
        m = torch.rand(shape, generator=g) > 0.3
        m[..., 0] = True
        got = attn_mod.optimized_attention(q, k, v, 4, mask=m)
        assert torch.allclose(got, _sdpa_flat(q, k, v, 4, m), atol=1e-5), shape
    assert _FakeFn.calls.count("masked") == 4 and not calls
    attn_mod.optimized_attention = original       # un-patch = put the returned original back
    attn_mod.optimized_attention(q, k, v, 4)
    assert calls == [("original", False, False)]


def test_install_comfyui_fallbacks(hook, comfy_stub):
    attn_mod, calls = comfy_stub
    hook.install_comfyui()
    g = torch.Generator().manual_seed(1)
    # skip_reshape: the host hands [B, H, N, D] tensors and wants them back that way -> the host's own function
    q4 = torch.randn(1, 2, 10, 8, generator=g)
    out = attn_mod.optimized_attention(q4, q4, q4, 2, skip_reshape=True)
    assert out.shape == q4.shape and calls[-1] == ("original", True, False)
    # head dim beyond the largest kernel
    big = torch.randn(1, 6, 1024, generator=g)
    attn_mod.optimized_attention(big, big, big, 1)
    assert calls[-1] == ("original", False, False)
    # a masked call that needs a gradient (LoRA training through cross-attention) runs the kernels too (fa2_fwd_bias / fa2_bwd_bias) ...
    q = torch.randn(2, 12, 32, generator=g, requires_grad=True)
    k = torch.randn(2, 7, 32, generator=g)
    m = torch.ones(2, 12, 7, dtype=torch.bool)
    n_before = len(calls)
    out = attn_mod.optimized_attention(q, k, k, 2, mask=m)
    out.sum().backward()
    assert len(calls) == n_before and _FakeFn.calls[-1] == "masked" and q.grad is not None
    # ... except above head dim 256, where the masked backward stops: the host's own attention
    qb = torch.randn(1, 5, 320, generator=g, requires_grad=True)
    kb = torch.randn(1, 3, 320, generator=g)
    out = attn_mod.optimized_attention(qb, kb, kb, 1, mask=torch.ones(1, 5, 3, dtype=torch.bool))
    out.sum().backward()
    assert len(calls) == n_before + 1 and calls[-1] == ("original", False, True) and qb.grad is not None
    n_before += 1
    with torch.no_grad():
        attn_mod.optimized_attention(q, k, k, 2, mask=m)
    assert _FakeFn.calls[-1] == "masked" and len(calls) == n_before
    # a mask that does not broadcast to [B, H, Nq, Nkv] (batch 3 against batch 2) is the host's problem, not an exception here
    bad = torch.ones(3, 12, 7, dtype=torch.bool)
    with torch.no_grad(), pytest.raises(Exception):
        attn_mod.optimized_attention(q, k, k, 2, mask=bad)     # (the stub host's SDPA rejects it, like the real one)
    assert calls[-1] == ("original", False, True)


def test_attention_bnhd_without_fallback_raises(hook):
    q = torch.randn(1, 4, 1024)
    with pytest.raises(NotImplementedError):
        hook.attention_bnhd(q, q, q, 1)


def test_install_webui_patches_cross_attention(hook, monkeypatch):
    class CrossAttention(torch.nn.Module):
        def __init__(self, dim, ctx_dim, heads, dim_head):
            super().__init__()
            self.heads = heads
            self.to_q = torch.nn.Linear(dim, heads * dim_head, bias=False)
            self.to_k = torch.nn.Linear(ctx_dim, heads * dim_head, bias=False)
            self.to_v = torch.nn.Linear(ctx_dim, heads * dim_head, bias=False)
            self.to_out = torch.nn.Sequential(torch.nn.Linear(heads * dim_head, dim), torch.nn.Dropout(0.0))

        def forward(self, x, context=None, mask=None):
            # ldm's own mask semantics (ldm/modules/attention.py): a boolean PER-KEY mask [B, ...] -> [B, Nkv], shared by heads and query rows
            context = x if context is None else context
            if mask is not None:
                mask = mask.reshape(mask.shape[0], 1, 1, -1)
            return self.to_out(_sdpa_flat(self.to_q(x), self.to_k(context), self.to_v(context), self.heads, mask))

    for name in ("ldm", "ldm.modules", "ldm.modules.attention"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["ldm.modules.attention"].CrossAttention = CrossAttention
    monkeypatch.delitem(sys.modules, "sgm", raising=False)
    torch.manual_seed(3)
    layer, self_layer = CrossAttention(40, 24, 4, 8), CrossAttention(40, 40, 4, 8)
    x, ctx = torch.randn(2, 30, 40), torch.randn(2, 11, 24)
    key_mask = torch.ones(2, 11, dtype=torch.bool)
    key_mask[0, -3:] = False                                 # padded prompt tokens of batch 0
    want_self, want_cross, want_masked = self_layer(x), layer(x, ctx), layer(x, ctx, mask=key_mask)
    originals = hook.install_webui()
    assert list(originals) == ["ldm.modules.attention"]
    try:
        assert torch.allclose(self_layer(x), want_self, atol=1e-5) and torch.allclose(layer(x, context=ctx), want_cross, atol=1e-5)
        assert _FakeFn.calls == ["apply", "apply"]
        xg = x.clone().requires_grad_(True)              # masked + gradient: the kernels (head dim 8), with ldm's [B, Nkv] mask as [B, 1, 1, Nkv]
        got = layer(xg, context=ctx, mask=key_mask)
        assert torch.allclose(got, want_masked, atol=1e-5) and not torch.allclose(got, want_cross, atol=1e-3)
        got.sum().backward()
        assert xg.grad is not None and _FakeFn.calls == ["apply", "apply", "masked"]
        # B == Nq must not turn the per-key mask into a query-by-key one: [B, Nkv] = [2, 11] with Nq = 2
        x2 = torch.randn(2, 2, 40)
        CrossAttention.forward, patched = originals["ldm.modules.attention"], CrossAttention.forward
        want2 = layer(x2, ctx, mask=key_mask)
        CrossAttention.forward = patched
        assert torch.allclose(layer(x2, context=ctx, mask=key_mask), want2, atol=1e-5)
    finally:
        CrossAttention.forward = originals["ldm.modules.attention"]


def test_install_hooks_need_their_hosts(monkeypatch):
    for name in ("comfy", "ldm", "sgm"):
        monkeypatch.setitem(sys.modules, name, None)      # import -> ImportError
    with pytest.raises(ImportError):
        sd_hook.install_comfyui()
    with pytest.raises(ImportError):
        sd_hook.install_webui()
