"""GPU parity tests (`-m gpu`) of the attention bias / mask path: C-ABI `fa2_fwd_bias`, the operator-level `flash_attention`
and the SD hook with a mask — SURVEY §8 row f4, the `mask` argument the reference reserves and ignores
(rocwmma_fattn/FlashAttn.py:49, :74; README.md:45).  Checked against the C oracle run with the same bias (same precision contract),
dense float64 attention and the platform's own scaled_dot_product_attention.  Nothing here reads /root/reference."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import ATOL, FLOOR, LSE_TOL, RTOL
from oracle import fa2_oracle as fo
from rocwmma_fattn import _fa2_lib
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attention
from rocwmma_fattn.sd_hook import attention_bnhd

pytestmark = pytest.mark.gpu

TORCH_DT = {0: torch.float16, 1: torch.bfloat16}
KINDS = ("io", "f32", "bool")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run the CPU suite with -m 'not gpu'"
    return torch.device("cuda", 0)


def _bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def _inputs(B, H, Nq, Nkv, D, dt, seed):
    g = torch.Generator().manual_seed(seed)
    return tuple(torch.rand((B, H, n, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for n in (Nq, Nkv, Nkv))


def _make_bias(kind, shape, dt, seed, dead_rows=()):
    """A bias tensor of the given (broadcast) shape: additive kinds carry N(0,1) values with some -inf positions, the bool
    kind keeps ~70 % of the positions; `dead_rows` (flattened row indices) are masked completely."""
    g = torch.Generator().manual_seed(seed)
    keep = torch.rand(shape, generator=g) < 0.7
    keep[..., 0] |= True                                   # (rows listed in dead_rows are the only fully masked ones)
    flat = keep.view(-1, shape[-1])
    for r in dead_rows:
        flat[r % flat.shape[0]] = False
    if kind == "bool":
        return keep.to(_dev())
    vals = torch.randn(shape, generator=g)
    vals = torch.where(keep, vals, torch.full_like(vals, float("-inf")))
    return vals.to(torch.float32 if kind == "f32" else TORCH_DT[dt]).to(_dev())


def _bias_f32(bias):
    """What the kernel adds, as float32 numpy: the bias values as stored (16-bit kinds already rounded), bool -> 0 / -inf."""
    if bias.dtype == torch.bool:
        return np.where(bias.cpu().numpy(), 0.0, -np.inf).astype(np.float32)
    return bias.float().cpu().numpy()


def _cabi_forward_bias(q, k, v, bias, causal, scale=None):
    """Straight through the C-ABI (fa2_fwd_bias), caller-owned buffers, broadcast dimensions as stride 0."""
    lib = _fa2_lib.load(build_if_missing=False)
    B, H, N, D = q.shape
    Nkv = k.shape[2]
    o = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    m = bias
    while m.dim() < 4:
        m = m.unsqueeze(0)
    kind = {torch.bool: _fa2_lib.FA2_BIAS_BOOL, torch.float32: _fa2_lib.FA2_BIAS_F32}.get(m.dtype, _fa2_lib.FA2_BIAS_IO_DTYPE)
    assert m.stride(3) == 1 and m.size(3) == Nkv
    bs = _fa2_lib.strides3(*(m.stride(i) if m.size(i) > 1 else 0 for i in range(3)))
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    rc = lib.fa2_fwd_bias(0 if q.dtype == torch.float16 else 1, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(),
                          B, H, N, Nkv, D, s3(q), s3(k), s3(v), s3(o), _fa2_lib.strides2(lse.stride(0), lse.stride(1)),
                          float(D ** -0.5 if scale is None else scale), int(causal), m.data_ptr(), kind, bs,
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _fa2_lib.check(rc)
    torch.cuda.synchronize()
    return o, lse


def _check(o, lse, q, k, v, bias, dt, causal, scale=None):
    bf = _bias_f32(bias)
    o_ref_bits, lse_ref = fo.fwd_c(_bits(q), _bits(k), _bits(v), dt, causal, scale=scale, bias=bf)
    o_ref = fo.bits_to_f32(o_ref_bits, dt)
    got = o.float().cpu().numpy()
    assert np.isfinite(got).all()
    bad = np.abs(got - o_ref) > ATOL[dt] + RTOL[dt] * np.abs(o_ref)
    assert not bad.any(), "max diff %.3e at %s" % (np.abs(got - o_ref).max(), np.argwhere(bad)[:4])
    lse_got = lse.cpu().numpy()
    dead = np.isneginf(lse_ref)
    assert (np.isneginf(lse_got) == dead).all()
    assert np.abs(lse_got[~dead] - lse_ref[~dead]).max() <= LSE_TOL
    assert (got[dead] == 0).all()                                          # fully masked rows: zeros
    o_true, lse_true = fo.fwd_numpy(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), causal, scale, bias=bf)
    assert np.abs(got - o_true).max() <= 2 * FLOOR[dt]
    assert np.abs(lse_got[~dead] - lse_true[~dead]).max() <= LSE_TOL
    return dead


# shape (B, H, Nq, Nkv, D), bias shape: every broadcast pattern, ragged tiles, every kernel head dim, Nkv = 77 (odd row pitch)
CASES = [
    ((2, 3, 200, 77, 64), (2, 3, 200, 77)),
    ((2, 3, 200, 77, 64), (2, 1, 200, 77)),
    ((1, 4, 333, 257, 128), (1, 1, 333, 257)),
    ((2, 2, 130, 300, 40), (2, 1, 1, 300)),          # key-padding mask: one row per batch
    ((1, 5, 256, 256, 128), (256, 256)),             # 2-D, shared by every head
    ((2, 2, 64, 640, 160), (2, 2, 64, 640)),
    ((1, 2, 100, 130, 80), (2, 100, 130)),           # 3-D: aligned on the right like torch SDPA -> the head dimension
    ((1, 1, 96, 200, 512), (1, 1, 96, 200)),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dt", [0, 1])
def test_bias_kinds_and_broadcasts_against_oracle(case, kind, dt):
    (B, H, Nq, Nkv, D), bshape = case
    if D == 512 and (dt == 1 or kind != "io"):
        pytest.skip("one D = 512 case is enough (spilling kernel: slow to check)")
    q, k, v = _inputs(B, H, Nq, Nkv, D, dt, seed=11 + D)
    bias = _make_bias(kind, bshape, dt, seed=5 + Nkv, dead_rows=(3,) if kind != "f32" else ())
    o, lse = _cabi_forward_bias(q, k, v, bias, causal=False)
    dead = _check(o, lse, q, k, v, bias, dt, False)
    assert dead.any() == (kind != "f32")


@pytest.mark.parametrize("kind", KINDS)
def test_bias_with_causal_mask(kind):
    B, H, N, D = 2, 2, 300, 64
    q, k, v = _inputs(B, H, N, N, D, 0, seed=3)
    bias = _make_bias(kind, (B, 1, N, N), 0, seed=9)
    o, lse = _cabi_forward_bias(q, k, v, bias, causal=True)
    _check(o, lse, q, k, v, bias, 0, True)


def test_zero_bias_equals_the_unbiased_kernel_contract():
    """An all-zero bias (and an all-true mask) must reproduce unbiased attention: same oracle, same tolerances."""
    B, H, N, D = 1, 3, 384, 128
    q, k, v = _inputs(B, H, N, N, D, 0, seed=21)
    base = FlashAttentionFunction.apply(q, k, v, None, False)
    for bias in (torch.zeros((1, 1, N, N), dtype=torch.float16, device=_dev()), torch.ones((N, N), dtype=torch.bool, device=_dev())):
        o, _ = _cabi_forward_bias(q, k, v, bias, causal=False)
        assert (o.float() - base.float()).abs().max().item() <= 1e-3


def test_explicit_and_negative_scale_with_bias():
    B, H, N, D = 1, 2, 150, 64
    q, k, v = _inputs(B, H, N, N, D, 0, seed=4)
    bias = _make_bias("f32", (1, H, N, N), 0, seed=2)
    for scale in (0.3, -0.2):
        o, lse = _cabi_forward_bias(q, k, v, bias, causal=False, scale=scale)
        _check(o, lse, q, k, v, bias, 0, False, scale=scale)


def test_large_bias_moves_the_reference_max():
    """A bias that grows by far more than the deferred-rescale threshold from tile to tile forces the rescale branch."""
    B, H, N, D = 1, 2, 512, 64
    q, k, v = _inputs(B, H, N, N, D, 0, seed=8)
    ramp = (torch.arange(N, dtype=torch.float32) * 0.25).view(1, 1, 1, N).expand(1, 1, N, N).contiguous().to(_dev())
    o, lse = _cabi_forward_bias(q, k, v, ramp, causal=False)
    _check(o, lse, q, k, v, ramp, 0, False)


@pytest.mark.parametrize("kind", KINDS)
def test_operator_flash_attention_matches_sdpa(kind):
    """flash_attention(mask=...) against torch's scaled_dot_product_attention(attn_mask=...) on rows that attend somewhere;
    BHND and the zero-copy BNHD layout; masks that need no expansion stay views (stride-0 broadcast)."""
    B, H, Nq, Nkv, D = 2, 4, 260, 77, 64
    q, k, v = _inputs(B, H, Nq, Nkv, D, 0, seed=14)
    mask = _make_bias(kind, (B, 1, Nq, Nkv), 0, seed=6)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(),
                                                           attn_mask=mask if mask.dtype == torch.bool else mask.float())
    o = flash_attention(q, k, v, mask)
    assert o.shape == q.shape and o.dtype == q.dtype
    assert (o.float() - ref).abs().max().item() <= 2e-3
    qb, kb, vb = (t.transpose(1, 2).contiguous() for t in (q, k, v))        # [B, N, H, D]
    ob = flash_attention(qb, kb, vb, mask, False, None, True)
    assert torch.equal(ob.transpose(1, 2), o)
    assert torch.equal(flash_attention(q, k, v, None), FlashAttentionFunction.apply(q, k, v, None, False))
    # the reference's operator keeps ignoring the argument (FlashAttn.py:49, :74)
    assert torch.equal(FlashAttentionFunction.apply(q, k, v, mask, False), FlashAttentionFunction.apply(q, k, v, None, False))
    og = flash_attention(q.clone().requires_grad_(True), k, v, mask)          # inputs that need a gradient: the autograd node, same forward
    assert og.requires_grad and torch.equal(og.detach(), o)
    with pytest.raises(RuntimeError, match="broadcast"):
        flash_attention(q, k, v, mask[:, :, :5])


def test_non_contiguous_and_kv_broadcast_masks_are_made_kernel_ready():
    B, H, Nq, Nkv, D = 1, 2, 128, 96, 64
    q, k, v = _inputs(B, H, Nq, Nkv, D, 1, seed=15)
    wide = torch.randn((B, H, Nq, 2 * Nkv), device=_dev())
    strided = wide[..., ::2]                                                # Nkv stride 2
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=strided)
    assert (flash_attention(q, k, v, strided).float() - ref).abs().max().item() <= 1.6e-2
    per_row = torch.randn((B, H, Nq, 1), device=_dev())                     # broadcast over Nkv: softmax-invariant
    base = FlashAttentionFunction.apply(q, k, v, None, False)
    assert (flash_attention(q, k, v, per_row).float() - base.float()).abs().max().item() <= 1.6e-2
    half = torch.randn((B, 1, Nq, Nkv), device=_dev(), dtype=torch.float16) # fp16 bias with bf16 tensors: converted
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=half.float())
    assert (flash_attention(q, k, v, half).float() - ref).abs().max().item() <= 2.5e-2


def test_sd_hook_with_masks_matches_sdpa():
    """The shapes SD hosts hand over: [B, N, heads*D] tensors with a [B, Nq, Nkv] (batch-first 3-D), [Nq, Nkv] or 4-D mask."""
    B, heads, Nq, Nkv, D = 2, 8, 1024, 77, 40
    g = torch.Generator().manual_seed(31)
    q = torch.rand((B, Nq, heads * D), generator=g).half().to(_dev())
    k, v = (torch.rand((B, Nkv, heads * D), generator=g).half().to(_dev()) for _ in range(2))

    def sdpa(mask4):
        q4, k4, v4 = (t.view(B, -1, heads, D).transpose(1, 2).float() for t in (q, k, v))
        return torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, attn_mask=mask4).transpose(1, 2).reshape(B, Nq, heads * D)

    keep = torch.ones((B, Nq, Nkv), dtype=torch.bool, device=_dev())
    keep[0, :, 50:] = False                                                 # padded prompt tokens of sample 0
    keep[1, :, 70:] = False
    out = attention_bnhd(q, k, v, heads, mask=keep)
    assert out.shape == q.shape and (out.float() - sdpa(keep.unsqueeze(1))).abs().max().item() <= 2e-3
    add = torch.randn((Nq, Nkv), generator=g).half().to(_dev())
    out = attention_bnhd(q, k, v, heads, mask=add)
    assert (out.float() - sdpa(add.float())).abs().max().item() <= 2e-3
    add4 = torch.randn((B, heads, Nq, Nkv), generator=g).to(_dev())
    out = attention_bnhd(q, k, v, heads, mask=add4)
    assert (out.float() - sdpa(add4)).abs().max().item() <= 2e-3


@pytest.mark.parametrize("nkv,dtype", [(77, torch.float32), (76, torch.float32), (72, torch.float16), (76, torch.float16), (76, torch.bool), (80, torch.bool)])
def test_bias_memory_past_the_row_is_never_read(nkv, dtype):
    """Rows of Nkv elements inside a NaN- (bool: True-) poisoned buffer of pitch 128: none of the three load forms — guarded
    per-element loads (77), coalesced 16-byte tile loads (72, fp16), one load per group of four (76) — may use the neighbours' tail."""
    B, H, Nq, D = 1, 2, 200, 64
    q, k, v = _inputs(B, H, Nq, nkv, D, 0, seed=17)
    if dtype == torch.bool:
        buf = torch.ones((B, H, Nq, 128), dtype=torch.bool, device=_dev())
        buf[..., :nkv] = torch.rand((B, H, Nq, nkv), device=_dev()) < 0.7
        buf[..., 0] = True
    else:
        buf = torch.full((B, H, Nq, 128), float("nan"), device=_dev(), dtype=dtype)
        buf[..., :nkv] = torch.randn((B, H, Nq, nkv), device=_dev()).to(dtype)
    bias = buf[..., :nkv]                                                   # row pitch 128, last dim contiguous: passed as is
    o, lse = _cabi_forward_bias(q, k, v, bias, causal=False)
    _check(o, lse, q, k, v, bias, 0, False)


# ---------------------------------------------------------------- backward through the masked forward (fa2_bwd_bias)
BWD_BIAS_SHAPES = [
    # B, H, Nq, Nkv, D, bias shape: SD cross-attention key-padding mask, a dense per-head bias, ragged tiles, D = 160 (the 4-wave kernels)
    (2, 4, 256, 77, 64, (2, 1, 1, 77)),
    (1, 3, 130, 203, 128, (1, 3, 130, 203)),
    (2, 2, 96, 100, 40, (2, 1, 96, 100)),
    (1, 2, 70, 140, 160, (140,)),
]


@pytest.mark.parametrize("shape", BWD_BIAS_SHAPES)
@pytest.mark.parametrize("kind", ["bool", "io", "f32"])
@pytest.mark.parametrize("dt", [0, 1])
def test_masked_backward_against_oracle_and_autograd(shape, kind, dt):
    """flash_attention(mask=...) is differentiable in q, k, v: forward fa2_fwd_bias, backward fa2_bwd_bias (three HIP passes that add the
    bias to the recomputed scores).  Gradients against the C oracle's masked backward (same contract) and float64 autograd."""
    from rocwmma_fattn.FlashAttn import flash_attention
    B, H, Nq, Nkv, D, bshape = shape
    tdt = torch.float16 if dt == 0 else torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(sum(shape[:5]) + dt)
    q, k, v, do = (torch.randn((B, H, n, D), generator=g).to(tdt) for n in (Nq, Nkv, Nkv, Nq))
    if len(bshape) == 1:
        bshape = (1, 1, 1) + bshape
    if kind == "bool":
        mask = torch.rand(bshape, generator=g) > 0.35
        mask[..., 0] = True
        bias_f = torch.where(mask, 0.0, float("-inf")).float()
    else:
        mask = (torch.randn(bshape, generator=g) * 1.5).to(tdt if kind == "io" else torch.float32)
        bias_f = mask.float()
    dev = torch.device("cuda", 0)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    o = flash_attention(qd, kd, vd, mask.to(dev), False)
    o.backward(do.to(dev))
    torch.cuda.synchronize()
    # same-contract oracle: forward (for o, lse) and backward with the bias
    code = dt
    bits = lambda t: t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    bnp = np.broadcast_to(bias_f.numpy(), (B, H, Nq, Nkv))
    o_bits, lse = fo.fwd_c(bits(q), bits(k), bits(v), code, False, bias=bnp)
    want = fo.bwd_c(bits(q), bits(k), bits(v), o_bits, bits(do), lse, code, False, bias=bnp)
    truth = fo.bwd_numpy(q.float().numpy(), k.float().numpy(), v.float().numpy(), do.float().numpy(), False, bias=bnp)
    tol = 2e-3 if dt == 0 else 1.6e-2
    for name, got_t, w_bits, t64 in zip("qkv", (qd.grad, kd.grad, vd.grad), want, truth):
        got = got_t.float().cpu().numpy()
        w = fo.bits_to_f32(w_bits, code)
        assert np.isfinite(got).all(), name
        assert np.abs(got - w).max() <= tol * max(1.0, np.abs(w).max()), (name, np.abs(got - w).max())
        assert np.abs(got - t64).max() <= 2 * tol * max(1.0, np.abs(t64).max()), (name, np.abs(got - t64).max())


def test_masked_backward_causal_fully_masked_rows_and_hook():
    """bias + causal, a row with every key masked (zero output, zero gradients), and the SD hook's masked call with gradients
    (BNHD layout) against torch autograd through SDPA on fp32 inputs."""
    from rocwmma_fattn.FlashAttn import flash_attention
    from rocwmma_fattn.sd_hook import attention_bnhd
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(99)
    B, H, N, D = 1, 2, 192, 64
    q, k, v, do = (torch.randn((B, H, N, D), generator=g).half() for _ in range(4))
    keep = torch.rand((B, 1, N, N), generator=g) > 0.4
    keep[..., 0] = True
    keep[0, 0, 7, :] = False
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    o = flash_attention(qd, kd, vd, keep.to(dev), True)
    o.backward(do.to(dev))
    bias = np.broadcast_to(torch.where(keep, 0.0, float("-inf")).float().numpy(), (B, H, N, N))
    truth = fo.bwd_numpy(q.float().numpy(), k.float().numpy(), v.float().numpy(), do.float().numpy(), True, bias=bias)
    for got_t, t64 in zip((qd.grad, kd.grad, vd.grad), truth):
        got = got_t.float().cpu().numpy()
        assert np.isfinite(got).all() and np.abs(got - t64).max() <= 4e-3 * max(1.0, np.abs(t64).max())
    assert float(qd.grad[0, :, 7].abs().max()) == 0.0 and float(o.detach()[0, :, 7].abs().max()) == 0.0
    # the hook: [B, N, heads*D] tensors, boolean [B, Nq, Nkv] mask, gradients flow through the kernels
    x = torch.randn((2, 130, 4 * 40), generator=g).half()
    ctx = torch.randn((2, 77, 4 * 40), generator=g).half()
    m = torch.rand((2, 130, 77), generator=g) > 0.2
    m[..., 0] = True
    xd, cd = x.to(dev).requires_grad_(True), ctx.to(dev).requires_grad_(True)
    out = attention_bnhd(xd, cd, cd, 4, mask=m.to(dev))
    out.float().square().sum().backward()
    xf, cf = x.float().to(dev).requires_grad_(True), ctx.float().to(dev).requires_grad_(True)
    t = lambda a: a.reshape(2, a.shape[1], 4, 40).transpose(1, 2)  # noqa: E731
    ref = torch.nn.functional.scaled_dot_product_attention(t(xf), t(cf), t(cf), attn_mask=m.to(dev).unsqueeze(1)).transpose(1, 2).reshape(2, 130, 160)
    ref.square().sum().backward()
    assert float((out.float() - ref).abs().max()) <= 4e-3
    for got, want in ((xd.grad, xf.grad), (cd.grad, cf.grad)):
        assert float((got.float() - want).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bool", "io", "f32"])
@pytest.mark.parametrize("D,dt", [(64, torch.float16), (128, torch.bfloat16), (40, torch.float16)])
def test_dense_masks_on_grids_that_fill_the_chip_take_the_lds_dma_form(kind, D, dt):
    """A dense per-row mask whose geometry allows whole 16-byte granules, on a grid of more than 3/8 of the CUs' worth of 256-row workgroups, runs
    the 8-wave kernels with the bias tile staged by LDS-DMA into a swizzled wave-private image (fa2_fwd_kernel.hip.h, BIAS = 2; host.cpp
    bias_vec = 3).  Every broadcast pattern over batch / head, a padded row pitch, a ragged Nq, a masked tail tile (Nkv = 1000), causal on top,
    a fully masked row; against dense float64 attention with the same mask and against the 4-wave path (option rows = 128) on the same inputs."""
    from rocwmma_fattn import _fa2_lib
    from rocwmma_fattn.FlashAttn import flash_attention
    dev = torch.device("cuda", 0)
    B, H, Nq, Nkv = 2, 60, 1000, 1000                      # 2 * 60 * 4 = 480 workgroups of 256 rows
    g = torch.Generator(device=dev).manual_seed(D + len(kind))
    q, k, v = (torch.randn((B, H, n, D), generator=g, device=dev).to(dt) for n in (Nq, Nkv, Nkv))
    for (mb, mh, causal, pitch) in ((B, H, False, 1000), (1, 1, False, 1008), (B, 1, True, 1000), (1, H, False, 1024)):
        if kind == "bool":
            full = torch.rand((mb, mh, Nq, pitch), generator=g, device=dev) < 0.8
            full[:, :, 7] = False                           # a fully masked row
        else:
            full = (torch.randn((mb, mh, Nq, pitch), generator=g, device=dev) * 1.5).to(dt if kind == "io" else torch.float32)
        mask = full[..., :Nkv]                              # (a padded pitch: a view, its rows stay 16-byte aligned)
        o = flash_attention(q, k, v, mask, causal)
        with _fa2_lib.options(rows=128):
            o4 = flash_attention(q, k, v, mask, causal)
        s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * (D ** -0.5)
        s = s.masked_fill(~mask, float("-inf")) if kind == "bool" else s + mask.double()
        if causal:
            s = s.masked_fill(torch.ones(Nq, Nkv, dtype=torch.bool, device=dev).triu(1), float("-inf"))
        dead = torch.isinf(s).all(dim=-1, keepdim=True)
        truth = torch.matmul(torch.softmax(s.masked_fill(dead, 0.0), -1).masked_fill(dead, 0.0), v.double())
        tol = 2e-3 if dt == torch.float16 else 1.6e-2
        assert torch.isfinite(o.float()).all()
        assert float((o.double() - truth).abs().max()) <= tol * max(1.0, float(v.float().abs().max())), (kind, mb, mh, causal)
        assert float((o.float() - o4.float()).abs().max()) <= tol, (kind, mb, mh, causal)
        if kind == "bool":
            assert float(o[:, :, 7].float().abs().max()) == 0.0
