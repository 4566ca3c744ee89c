"""GPU parity tests of the hand-scheduled head-dim-256 forward (round 6: csrc/gen/fwd_m16_d256_gen.py, csrc/fa2_fwd_d256.hip.h; option "asm" bit 10):
through the C-ABI against the C oracle under the contract fa2_fwd_plan names, float64 attention, the compiler-scheduled kernel of the same call, and
NaN-poisoned memory around ragged shapes.  tests/test_asm_emu_d256.py emulates the same block instruction by instruction on the CPU."""
import numpy as np
import pytest
import torch

from oracle import fa2_oracle as fo
from rocwmma_fattn import _fa2_lib
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from test_parity_gpu import TORCH_DT, _assert_close_to_oracle, _cabi_forward, _dev, _plan

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, H, Nq, Nkv, causal
    (2, 8, 1024, 1024, False),
    (1, 24, 4096, 4096, False),          # the reference harness's D scan point (bench_with_sdpa.py:259-283)
    (2, 16, 2048, 2048, True),
    (1, 5, 1000, 1333, False),           # ragged Nq (last q block partly empty) and Nkv (masked last tile); heads not a multiple of 8
    (3, 3, 1500, 1500, True),            # ragged causal
    (1, 8, 512, 8192, False),            # long cross-attention-like sweep
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dt", [0, 1])
def test_head_dim_256_on_the_hand_scheduled_kernel(shape, dt):
    B, H, Nq, Nkv, causal = shape
    D = 256
    g = torch.Generator(device="cpu").manual_seed(256 + Nq + dt)
    q = torch.randn((B, H, Nq, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    k = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    v = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    plan = _plan(q, k, causal)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.rows == 128 and plan.contract == _fa2_lib.FA2_CONTRACT_LSUM_P16 and plan.heads_main == B * H, plan.as_dict()
    o, lse = _cabi_forward(q, k, v, causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    o2, lse2 = _cabi_forward(q, k, v, causal)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    # the compiler-scheduled kernel of the same call (option "asm" bit 10 clear): another contract (f32 row sums), same answer to rounding
    lib = _fa2_lib.load(build_if_missing=False)
    full = lib.fa2_get_option(b"asm")
    try:
        lib.fa2_set_option(b"asm", full & ~1024)
        assert _plan(q, k, causal).kernel != _fa2_lib.FA2_KERNEL_ASM
        o_hip, lse_hip = _cabi_forward(q, k, v, causal)
    finally:
        lib.fa2_set_option(b"asm", full)
    tol = 2.0 ** -9 if dt == 0 else 2.0 ** -6
    assert float((o.float() - o_hip.float()).abs().max()) <= 2 * tol * max(1.0, float(o_hip.float().abs().max()))
    assert float((lse - lse_hip).abs().max()) <= (1e-3 if dt == 0 else 6e-3)
    for (b, h) in ((0, 0), (B - 1, H - 1)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], dt, causal, plan=plan, head=b * H + h)


def test_head_dim_256_layouts_strides_and_memory_past_the_tails():
    """BNHD views (row pitch H * 512 bytes), tensors cut out of NaN-filled allocations (nothing past Nq / Nkv may enter or be written), through the
    operator; a row-padded K (pitch not a multiple of 512 bytes) and a differentiated call keep the compiler-scheduled kernels."""
    B, H, N, D = 2, 6, 1100, 256
    g = torch.Generator(device="cpu").manual_seed(9)
    big = {n: torch.full((B, N + 64, H, D), float("nan"), dtype=torch.float16, device=_dev()) for n in "qkv"}
    for n in "qkv":
        big[n][:, :N] = torch.randn((B, N, H, D), generator=g).half().to(_dev())
    q, k, v = (big[n][:, :N] for n in "qkv")                      # [B, N, H, D] views with NaN rows behind them
    plan = _fa2_lib.fwd_plan(q.transpose(1, 2), k.transpose(1, 2), False)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.rows == 128, plan.as_dict()
    o = FlashAttentionFunction.apply(q, k, v, None, False, None, True)
    torch.cuda.synchronize()
    assert o.shape == q.shape and torch.isfinite(o.float()).all()
    s = torch.einsum("bnhd,bmhd->bhnm", q.float(), k.float()) * D ** -0.5
    ref = torch.einsum("bhnm,bmhd->bnhd", torch.softmax(s, -1), v.float())
    assert float((o.float() - ref).abs().max()) <= 2e-3
    kp = torch.randn((B, H, N, D + 8), generator=g).half().to(_dev())[..., :D]
    assert _fa2_lib.fwd_plan(q.transpose(1, 2), kp, False).kernel != _fa2_lib.FA2_KERNEL_ASM
    assert _fa2_lib.fwd_plan(q.transpose(1, 2), k.transpose(1, 2), _fa2_lib.FA2_FLAG_EXACT_SCALE).kernel != _fa2_lib.FA2_KERNEL_ASM


def test_head_dim_256_large_logits():
    """N(0, 6^2) logits: the max-first bodies move their references (deferred at 2^14) — against the oracle and float64."""
    B, H, N, D = 1, 16, 2048, 256
    g = torch.Generator(device="cpu").manual_seed(11)
    q, k, v = ((6.0 ** 0.5 if i < 2 else 1.0) * torch.randn((B, H, N, D), generator=g) for i in range(3))
    for dt in (0, 1):
        qq, kk, vv = (t.to(TORCH_DT[dt]).to(_dev()) for t in (q, k, v))
        for causal in (False, True):
            plan = _plan(qq, kk, causal)
            assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM
            o, lse = _cabi_forward(qq, kk, vv, causal)
            sl = (slice(0, 1), slice(3, 4))
            _assert_close_to_oracle(o[sl], lse[sl], qq[sl], kk[sl], vv[sl], dt, causal, plan=plan, head=3)


@pytest.mark.parametrize("D", [136, 152, 160, 176, 192, 200, 224, 248])
@pytest.mark.parametrize("dt", [0, 1])
def test_head_dims_below_256_on_the_hand_scheduled_kernel(D, dt):
    """Head dims 136 .. 248 run ON the head-dim-256 body — with the k-steps and d groups that hold no real column left out of the instruction stream
    (ceil(D / 32) k-steps: 5 .. 8) — (generator opt=trim; the reference zero-pads D on the host, kernel_fp16.cu:763-779): rows of D
    columns at whatever pitch the tensors have, the padded columns of the Q / K / V images zero-filled by the loads themselves, D columns stored.  The
    tensors are column slices of NaN-filled allocations with a row pitch of D + 24 .. 40 elements: a granule fetched from the gap, or a column stored
    into it, would show."""
    B, H, N, causal = 2, 5, 1100, (D % 16 == 0 and D >= 176)
    g = torch.Generator(device="cpu").manual_seed(D + dt)
    pad = 24 + 8 * (D % 3)
    big = {n: torch.full((B, H, N + 3, D + pad), float("nan"), dtype=TORCH_DT[dt], device=_dev()) for n in "qkv"}
    for n in "qkv":
        big[n][:, :, :N, :D] = torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    q, k, v = (big[n][:, :, :N, :D] for n in "qkv")
    plan = _plan(q, k, causal)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.rows == 128 and plan.contract == _fa2_lib.FA2_CONTRACT_LSUM_P16, plan.as_dict()
    obig = torch.full((B, H, N + 3, D + pad), float("nan"), dtype=TORCH_DT[dt], device=_dev())
    o = obig[:, :, :N, :D]
    lse = torch.empty((B, H, N), dtype=torch.float32, device=_dev())
    lib = _fa2_lib.load(build_if_missing=False)
    import ctypes
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    _fa2_lib.check(lib.fa2_fwd(dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D, s3(q), s3(k), s3(v), s3(o),
                               _fa2_lib.strides2(lse.stride(0), lse.stride(1)), float(D ** -0.5), int(causal), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert torch.isnan(obig[:, :, N:].float()).all() and torch.isnan(obig[:, :, :, D:].float()).all()      # nothing stored past Nq or past column D
    qc, kc, vc = (t.contiguous() for t in (q, k, v))
    for (b, h) in ((0, 0), (B - 1, H - 1)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o[sl].contiguous(), lse[sl], qc[sl], kc[sl], vc[sl], dt, causal, plan=plan, head=b * H + h)
