"""KV-split of the last, partly filled round of forward workgroups (C-ABI fa2_fwd_ws / fa2_fwd_workspace_bytes, include/fa2_gfx950.h).

GPU tests (-m gpu): through the C-ABI against the oracle (oracle/, test infrastructure), against dense fp32 attention and against the
plain fa2_fwd call of the same inputs.  The reference has no counterpart: its launcher pads the grid to a multiple of its 96 CUs
(kernel_fp16.cu:808-813) and lets the padding blocks exit."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import ATOL, FLOOR, LSE_TOL, LSE_TOL_P16_BF16, RTOL
from oracle import fa2_oracle as fo
from rocwmma_fattn import _fa2_lib
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma

TORCH_DT = {0: torch.float16, 1: torch.bfloat16}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run the CPU suite with -m 'not gpu'"
    return torch.device("cuda", 0)


def _bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def _s3(t, bnhd=False):
    return _fa2_lib.strides3(t.stride(0), t.stride(2), t.stride(1)) if bnhd else _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))


def _fwd(q, k, v, ws_mode, bnhd=False, scale=None):
    """ws_mode: 'none' -> fa2_fwd; 'ws' -> fa2_fwd_ws with a workspace of exactly the advertised size (NaN-filled: every byte the
    merge reads must have been written by a part); 'small' -> fa2_fwd_ws with a workspace one byte short (must behave as fa2_fwd)."""
    lib = _fa2_lib.load(build_if_missing=False)
    if bnhd:
        B, N, H, D = q.shape
        Nkv = k.shape[1]
    else:
        B, H, N, D = q.shape
        Nkv = k.shape[2]
    dt = 0 if q.dtype == torch.float16 else 1
    o = torch.full_like(q, float("nan"))
    lse = torch.full((B, H, N), float("nan"), dtype=torch.float32, device=q.device)
    args = (dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, Nkv, D,
            _s3(q, bnhd), _s3(k, bnhd), _s3(v, bnhd), _s3(o, bnhd), _fa2_lib.strides2(lse.stride(0), lse.stride(1)),
            float(D ** -0.5 if scale is None else scale), 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    need = lib.fa2_fwd_workspace_bytes(dt, B, H, N, Nkv, D, 0)
    if ws_mode == "none":
        rc = lib.fa2_fwd(*args, stream)
    else:
        nbytes = need if ws_mode == "ws" else max(need - 1, 1)
        ws = torch.full(((nbytes + 3) // 4 + 4,), float("nan"), dtype=torch.float32, device=q.device)
        rc = lib.fa2_fwd_ws(*args, ws.data_ptr(), nbytes, stream)
    _fa2_lib.check(rc)
    torch.cuda.synchronize()
    return o, lse, need


def _plan_ws(q, k, bnhd, need):
    """fa2_fwd_plan of the fa2_fwd_ws call _fwd(q, k, v, "ws", bnhd) makes (workspace of `need` bytes)."""
    qv, kv = (q.transpose(1, 2), k.transpose(1, 2)) if bnhd else (q, k)          # [B, H, N, D] views carrying the call's strides
    return _fa2_lib.fwd_plan(qv, kv, False, workspace_bytes=need)


def _check_heads(o, lse, q, k, v, dt, heads, bnhd=False, plan=None):
    """sampled heads against the oracle under THE contract fa2_fwd_plan names for the call (whole items and KV-split parts run in one kernel:
    one contract per call; `plan` = _plan_ws(...) of the call that produced o)."""
    assert plan is not None and plan.kernel_tail == 0
    flags = (fo.PRESCALE_Q if plan.contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q else 0) | (fo.LSUM_P16 if plan.contract & _fa2_lib.FA2_CONTRACT_LSUM_P16 else 0)
    if dt == 0 and flags & fo.PRESCALE_Q:
        flags |= fo.PRESCALE_FUSED              # (the fp16 kernels round Q * c once, from the exact product: v_fma_mixlo_f16)
    for (b, h) in heads:
        if bnhd:
            sl = lambda t: t[b:b + 1, :, h:h + 1].transpose(1, 2).contiguous()  # noqa: E731
        else:
            sl = lambda t: t[b:b + 1, h:h + 1].contiguous()  # noqa: E731
        got = sl(o).float().cpu().numpy()
        assert np.isfinite(got).all()
        o_ref_bits, lse_ref = fo.fwd_c(_bits(sl(q)), _bits(sl(k)), _bits(sl(v)), dt, False, flags=flags)
        o_ref = fo.bits_to_f32(o_ref_bits, dt)
        diff = np.abs(got - o_ref)
        lse_err = np.abs(lse[b:b + 1, h:h + 1].cpu().numpy() - lse_ref).max()
        lse_tol = LSE_TOL_P16_BF16 if (dt == 1 and flags & fo.LSUM_P16) else LSE_TOL
        assert not (diff > ATOL[dt] + RTOL[dt] * np.abs(o_ref)).any() and lse_err <= lse_tol, \
            "head %s under contract %d (kernel %d): max |O diff| %.3g, max |LSE diff| %.3g" % ((b, h), plan.contract, plan.kernel, float(diff.max()), float(lse_err))


# shapes whose B*H*ceil(Nq/256) leaves a partly filled last round on 256 CUs:
#   (B, H, Nq, Nkv, D, dtype code, BNHD)
SPLIT_SHAPES = [
    (2, 10, 4096, 4096, 64, 0, False),      # SDXL 64x64 self-attention: 320 items -> 64 items x 4 parts; heads not a multiple of 8
    (1, 24, 3072, 3072, 64, 0, False),      # the reference harness's shape (bench_with_sdpa.py:52) at N = 3072: 288 items -> 32 x 6 parts
    (1, 24, 4096, 4096, 64, 1, False),      # 384 items -> 128 x 2, bf16
    (1, 24, 4096, 4096, 128, 0, False),     # D = 128: whole rounds on the hand-scheduled kernel, parts on the HIP kernel
    (1, 24, 4096, 4096, 128, 1, True),      # ... bf16, BNHD (zero-copy strides)
    (2, 10, 4000, 3990, 64, 0, False),      # ragged Nq (last q block partly empty) and ragged Nkv (the last part's last tile is masked)
    (2, 10, 4096, 4096, 40, 0, False),      # SD1.5's head dim on the D = 64 body (columns 40..63 zero-filled by the LDS-DMA)
    (3, 9, 3072, 2048, 96, 0, False),       # D = 96 on the D = 128 body, 324 items
    (3, 9, 3072, 2048, 80, 0, False),       # D = 80: the trimmed HIP kernels (no hand-scheduled body takes it: everything in one launch)
    (1, 40, 2048, 8192, 64, 0, True),       # cross-attention-like: Nkv != Nq, 320 items, BNHD
    # (round 6) grids that cover at most half of the CUs over a long sweep: EVERY item is split (no whole items at all), the parts run on the 8-wave kernel
    (1, 32, 1, 8192, 128, 0, False),        # decode-sized: one query row per head, 32 items x 8 parts; a part stores one row, not 256
    (4, 8, 1, 16384, 64, 1, False),
    (1, 32, 16, 8200, 128, 1, True),        # a few rows, ragged Nkv, BNHD
    (1, 8, 4096, 4096, 40, 0, False),       # SD 1.5 64 x 64 at batch 1: 128 items x 2 parts
    (1, 4, 2048, 2048, 128, 0, False),      # 32 items x 4 parts
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SPLIT_SHAPES)
def test_split_launches_against_oracle_dense_and_plain_call(shape):
    B, H, N, Nkv, D, dt, bnhd = shape
    g = torch.Generator(device="cpu").manual_seed(1000 + H + D)
    mk = lambda n: torch.randn(((B, n, H, D) if bnhd else (B, H, n, D)), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v = mk(N), mk(Nkv), mk(Nkv)
    o_ws, lse_ws, need = _fwd(q, k, v, "ws", bnhd)
    assert need > 0, "the shape is meant to be split on a 256-CU device"
    o_pl, lse_pl, _ = _fwd(q, k, v, "none", bnhd)
    assert torch.isfinite(o_ws.float()).all() and torch.isfinite(lse_ws).all()
    # against the plain call: the merge normalises the parts in f32 and rounds once, so results agree to one ulp of the I/O dtype
    ulp = 2.0 ** -10 if dt == 0 else 2.0 ** -7
    scale_o = max(1.0, float(o_pl.float().abs().max()))
    assert float((o_ws.float() - o_pl.float()).abs().max()) <= ulp * scale_o
    # (head dim 64, fp16: the whole rounds of the split launch run the folded-scale body, the plain call's 128-row kernels scale the f32 product)
    # (at head dim 128 both calls run the same contract, but their row sums add the ROUNDED P — FA2_CONTRACT_LSUM_P16 — of different KV partitions,
    #  each against its own reference: rounding-level differences of the LSE, 2^-12 per term in fp16, 2^-9 in bf16)
    assert float((lse_ws - lse_pl).abs().max()) <= (LSE_TOL if dt == 0 else LSE_TOL_P16_BF16)
    # a fair share of the elements must be bit-identical (the unsplit items are the same launch geometry)
    # (not at head dim 64 in fp16, where the two calls run different scaling contracts: see the LSE bound above)
    # (... nor where the plain call runs a head dim below the body's on the hand-scheduled kernel — folded scale — and the split call the HIP kernels:
    #  D = 96 fp16 since round 5)
    qv, kv = (q.transpose(1, 2), k.transpose(1, 2)) if bnhd else (q, k)
    other_contract = _fa2_lib.fwd_plan(qv, kv, False).contract != _plan_ws(q, k, bnhd, need).contract
    same = (o_ws.view(torch.int16) == o_pl.view(torch.int16)).float().mean().item()
    assert same > (0.2 if ((D == 64 and dt == 0) or other_contract) else 0.5), same
    # against the oracle on heads from the unsplit rounds and from the split tail (the last items of the order)
    heads = {(0, 0), (B - 1, H - 1), (B - 1, H - 2), (B // 2, H // 2), (B - 1, max(H - 8, 0))}
    plan = _plan_ws(q, k, bnhd, need)
    assert plan.nsplit > 1 and plan.split_items > 0
    every_item = plan.split_items == B * H * ((N + 255) // 256)     # (round 6) an underfilled grid: parts only, on the 8-wave kernel
    if every_item:      # ... or, where a part sweeps at least 24 tiles, inside the hand-scheduled persistent kernel like the parts of a last round
        long_parts = ((Nkv + 63) // 64) // plan.nsplit >= 24 and (D in (64, 128) or (D == 40 and dt == 0))
        assert plan.rows == 256 and ((plan.kernel == _fa2_lib.FA2_KERNEL_ASM) if long_parts else (plan.kernel == _fa2_lib.FA2_KERNEL_HIP_256 and plan.contract == 0)), plan.as_dict()
    elif D in (64, 128) or (D in (40, 96) and dt == 0):     # whole items and parts inside the hand-scheduled persistent kernel (round 5: also head dims just below a body's)
        assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM
    else:
        assert plan.kernel == _fa2_lib.FA2_KERNEL_HIP_256 and plan.contract == 0
    _check_heads(o_ws, lse_ws, q, k, v, dt, heads, bnhd, plan)
    # against dense fp32 attention on the whole tensor
    qf, kf, vf = (t.float().transpose(1, 2) if bnhd else t.float() for t in (q, k, v))
    s = torch.matmul(qf, kf.transpose(-1, -2)) * (D ** -0.5)
    truth = torch.matmul(torch.softmax(s, -1), vf)
    got = o_ws.float().transpose(1, 2) if bnhd else o_ws.float()
    assert float((got - truth).abs().max()) <= FLOOR[dt] * 2
    lse_true = torch.logsumexp(s, -1) * fo.LOG2E
    assert float((lse_ws - lse_true).abs().max()) <= 2e-3


@pytest.mark.gpu
def test_short_workspace_and_disabled_option_are_the_plain_call():
    B, H, N, D = 2, 10, 4096, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    q, k, v = (torch.randn((B, H, N, D), generator=g).half().to(_dev()) for _ in range(3))
    o_pl, lse_pl, need = _fwd(q, k, v, "none")
    assert need > 0
    o_sm, lse_sm, _ = _fwd(q, k, v, "small")
    assert torch.equal(o_sm.view(torch.int16), o_pl.view(torch.int16)) and torch.equal(lse_sm, lse_pl)
    lib = _fa2_lib.load(build_if_missing=False)
    with _fa2_lib.options(split=0):
        assert lib.fa2_fwd_workspace_bytes(0, B, H, N, N, D, 0) == 0
        o_off, lse_off, _ = _fwd(q, k, v, "ws")
    assert torch.equal(o_off.view(torch.int16), o_pl.view(torch.int16)) and torch.equal(lse_off, lse_pl)
    # shapes that never split: causal, exactly whole rounds, fewer items than CUs
    assert lib.fa2_fwd_workspace_bytes(0, B, H, N, N, D, 1) == 0
    assert lib.fa2_fwd_workspace_bytes(0, 2, 16, 4096, 4096, 128, 0) == 0
    assert lib.fa2_fwd_workspace_bytes(0, 2, 20, 1024, 1024, 64, 0) == 0


@pytest.mark.gpu
def test_operator_and_compiled_front_end_take_the_split_path():
    """FlashAttentionFunction.apply (compiled front end and Python front end) allocates the workspace itself; the result must be the
    C-ABI's split result bit for bit, forward only and with a backward on top (the saved LSE is the merged one)."""
    B, H, N, D = 2, 10, 4096, 64
    g = torch.Generator(device="cpu").manual_seed(6)
    q, k, v = (torch.randn((B, H, N, D), generator=g).half().to(_dev()) for _ in range(3))
    o_ws, lse_ws, need = _fwd(q, k, v, "ws")
    assert need > 0
    o_op = FlashAttentionFunction.apply(q, k, v, None, False)
    assert torch.equal(o_op.view(torch.int16), o_ws.view(torch.int16))
    ret = flash_attn_wmma.forward_py(q, k, v, 64, 128, False, D ** -0.5, False)
    assert torch.equal(ret[0].view(torch.int16), o_ws.view(torch.int16)) and torch.equal(ret[5], lse_ws)
    # backward through the split forward against float64 autograd on two heads
    qg, kg, vg = (t[:1, :2].clone().requires_grad_(True) for t in (q, k, v))
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    oa = FlashAttentionFunction.apply(qa, ka, va, None, False)
    do = torch.randn(oa.shape, generator=torch.Generator(device="cpu").manual_seed(7)).half().to(_dev())
    oa.backward(do)
    ref = torch.nn.functional.scaled_dot_product_attention(qg.double(), kg.double(), vg.double())
    ref.backward(do[:1, :2].double())
    for got, want in ((qa.grad, qg.grad), (ka.grad, kg.grad), (va.grad, vg.grad)):
        err = float((got[:1, :2].double() - want.double()).abs().max())
        assert err <= 2e-3 * max(1.0, float(want.abs().max())), err


def test_workspace_query_validates_like_the_call():
    """(CPU) the planning half of the call runs without a GPU: bad arguments give 0, and the size is what the header promises —
    never more than 64 MiB, a function of the shape only."""
    lib = _fa2_lib.load()
    assert lib.fa2_fwd_workspace_bytes(7, 2, 10, 4096, 4096, 64, 0) == 0          # bad dtype
    assert lib.fa2_fwd_workspace_bytes(0, 2, 10, 4096, 4096, 63, 0) == 0          # D not a multiple of 8
    assert lib.fa2_fwd_workspace_bytes(0, 0, 10, 4096, 4096, 64, 0) == 0
    for shape in [(2, 10, 4096, 4096, 64), (1, 24, 3072, 3072, 64), (1, 24, 4096, 4096, 128), (64, 16, 4096, 4096, 128), (1, 1, 70000, 4096, 64)]:
        n = lib.fa2_fwd_workspace_bytes(0, *shape, 0)
        assert n == lib.fa2_fwd_workspace_bytes(1, *shape, 0)
        assert 0 <= n <= 64 * 2 ** 20 and n % 16 == 0


# ---------------------------------------------------------------- backward: fa2_bwd_ws

def _fwd_bwd(q, k, v, do, ws_mode, bnhd=False):
    """forward (plain) + backward through the C-ABI; ws_mode 'none' -> fa2_bwd, 'ws' -> fa2_bwd_ws with a NaN-filled workspace of the advertised size."""
    lib = _fa2_lib.load(build_if_missing=False)
    if bnhd:
        B, N, H, D = q.shape
        Nkv = k.shape[1]
    else:
        B, H, N, D = q.shape
        Nkv = k.shape[2]
    dt = 0 if q.dtype == torch.float16 else 1
    scale = float(D ** -0.5)
    o = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    delta = torch.empty_like(lse)
    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
    s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _fa2_lib.check(lib.fa2_fwd(dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, Nkv, D,
                               _s3(q, bnhd), _s3(k, bnhd), _s3(v, bnhd), _s3(o, bnhd), s2, scale, 0, stream))
    args = (dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
            delta.data_ptr(), B, H, N, Nkv, D, _s3(q, bnhd), _s3(k, bnhd), _s3(v, bnhd), _s3(o, bnhd), _s3(do, bnhd), _s3(dq, bnhd), _s3(dk, bnhd),
            _s3(dv, bnhd), s2, scale, 0)
    need = lib.fa2_bwd_workspace_bytes(dt, B, H, N, Nkv, D, 0)
    if ws_mode == "none":
        rc = lib.fa2_bwd(*args, stream)
    else:
        ws = torch.full(((need + 3) // 4 + 4,), float("nan"), dtype=torch.float32, device=q.device)
        rc = lib.fa2_bwd_ws(*args, ws.data_ptr(), need, stream)
    _fa2_lib.check(rc)
    torch.cuda.synchronize()
    return o, lse, (dq, dk, dv), need


BWD_SPLIT_SHAPES = [
    (2, 10, 4096, 4096, 64, 0, False),      # SDXL 64x64: dQ pass and fused dK / dV pass both 320 workgroups
    (1, 24, 3072, 3072, 64, 1, False),      # bf16, 288 workgroups
    (3, 8, 4096, 4096, 40, 0, False),       # SD1.5's head dim on the D = 64 kernels, 384 workgroups
    (2, 20, 2048, 2048, 80, 0, False),      # D = 80 on the D = 128 HIP kernels: only the dQ pass splits (dK / dV run as wave pairs)
    (2, 10, 4000, 3990, 64, 0, False),      # ragged Nq and Nkv
    (1, 40, 2048, 4096, 64, 0, True),       # Nkv != Nq: the passes get different plans; BNHD
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", BWD_SPLIT_SHAPES)
def test_split_backward_against_oracle_autograd_and_plain_call(shape):
    from conftest import GRAD_TOL
    B, H, N, Nkv, D, dt, bnhd = shape
    g = torch.Generator(device="cpu").manual_seed(2000 + H + D)
    mk = lambda n: torch.randn(((B, n, H, D) if bnhd else (B, H, n, D)), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(N), mk(Nkv), mk(Nkv), mk(N)
    o, lse, gw, need = _fwd_bwd(q, k, v, do, "ws", bnhd)
    assert need > 0, "the shape is meant to be split on a 256-CU device"
    _, _, gp, _ = _fwd_bwd(q, k, v, do, "none", bnhd)
    for name, a, b_ in zip("qkv", gw, gp):
        assert torch.isfinite(a.float()).all(), "d%s has non-finite values (a part of the workspace was read before it was written?)" % name
        # the split changes the f32 summation order of the split rows only: one rounding of the I/O dtype at the gradient's scale
        tol = (2.0 ** -10 if dt == 0 else 2.0 ** -7) * max(1.0, float(b_.float().abs().max()))
        assert float((a.float() - b_.float()).abs().max()) <= tol, name
    # the oracle and float64 autograd on heads of the unsplit rounds and of the split tail
    for (b, h) in {(0, 0), (B - 1, H - 1), (B - 1, H - 2)}:
        sl = (lambda t: t[b:b + 1, :, h:h + 1].transpose(1, 2).contiguous()) if bnhd else (lambda t: t[b:b + 1, h:h + 1].contiguous())  # noqa: E731
        qs, ks, vs, dos, os_ = (sl(t) for t in (q, k, v, do, o))
        want = fo.bwd_c(_bits(qs), _bits(ks), _bits(vs), _bits(os_), _bits(dos), lse[b:b + 1, h:h + 1].cpu().numpy(), dt, False)
        for name, gt, w_bits in zip("qkv", gw, want):
            w = fo.bits_to_f32(w_bits, dt)
            got = sl(gt).float().cpu().numpy()
            assert np.abs(got - w).max() <= GRAD_TOL[dt] * max(1.0, np.abs(w).max()), (name, (b, h), np.abs(got - w).max())
        qd, kd, vd = (sl(t).double().requires_grad_(True) for t in (q, k, v))
        torch.nn.functional.scaled_dot_product_attention(qd, kd, vd).backward(dos.double())
        for name, gt, w in zip("qkv", gw, (qd.grad, kd.grad, vd.grad)):
            err = float((sl(gt).double() - w).abs().max())
            assert err <= GRAD_TOL[dt] * max(1.0, float(w.abs().max())), (name, (b, h), err)


@pytest.mark.gpu
def test_operator_backward_takes_the_split_path_and_option_turns_it_off():
    B, H, N, D = 2, 10, 4096, 64
    g = torch.Generator(device="cpu").manual_seed(8)
    q, k, v, do = (torch.randn((B, H, N, D), generator=g).half().to(_dev()) for _ in range(4))
    _, _, gw, need = _fwd_bwd(q, k, v, do, "ws")
    assert need > 0
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    with _fa2_lib.options(split=0):         # the same forward for both (the split forward's O differs by f32 rounding of the merge)
        oa = FlashAttentionFunction.apply(qa, ka, va, None, False)
    oa.backward(do)
    for a, b_ in zip(gw, (qa.grad, ka.grad, va.grad)):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16))
    lib = _fa2_lib.load(build_if_missing=False)
    with _fa2_lib.options(split=0):
        assert lib.fa2_bwd_workspace_bytes(0, B, H, N, N, D, 0) == 0
    assert lib.fa2_bwd_workspace_bytes(0, B, H, N, N, D, 1) == 0                  # causal
    assert lib.fa2_bwd_workspace_bytes(0, 2, 16, 4096, 4096, 128, 0) == 0         # the hand-scheduled passes


@pytest.mark.gpu
def test_split_launches_are_graph_capturable_and_stream_safe():
    """The operator keeps ONE scratch block per (device, stream) for the split and takes a block from torch's caching allocator while a stream is being
    captured, so a captured graph owns its scratch memory (graph pool) and two streams never share one: capture a forward of a split shape, replay
    it on new inputs, and run two streams concurrently."""
    B, H, N, D = 2, 10, 4096, 64
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(9)
    mk = lambda: torch.randn((B, H, N, D), generator=g).half().to(dev)  # noqa: E731
    q, k, v = mk(), mk(), mk()
    lib = _fa2_lib.load(build_if_missing=False)
    assert lib.fa2_fwd_workspace_bytes(0, B, H, N, N, D, 0) > 0
    want = FlashAttentionFunction.apply(q, k, v, None, False).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            FlashAttentionFunction.apply(q, k, v, None, False)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = FlashAttentionFunction.apply(q, k, v, None, False)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    q2 = mk()
    want2 = FlashAttentionFunction.apply(q2, k, v, None, False).clone()
    q.copy_(q2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want2)
    # two streams at once, different inputs: each call has its own workspace
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    qa, qb = mk(), mk()
    wa, wb = FlashAttentionFunction.apply(qa, k, v, None, False).clone(), FlashAttentionFunction.apply(qb, k, v, None, False).clone()
    torch.cuda.synchronize()
    outs = []
    for st, qq in ((s1, qa), (s2, qb)):
        with torch.cuda.stream(st):
            for _ in range(4):
                o = FlashAttentionFunction.apply(qq, k, v, None, False)
            outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], wa) and torch.equal(outs[1], wb)


@pytest.mark.gpu
def test_split_scratch_is_one_block_per_stream_not_one_per_call():
    """the operator used to take the split's workspace (up to 64 MiB) from the allocator on every call.  Now one block per
    (device, stream) is allocated on first use and reused: across 20 calls of a split shape the allocator's peak stays at inputs + one output
    set + the one block, the block's address does not change, and a second stream gets its own."""
    from rocwmma_fattn import FlashAttn
    B, H, N, D = 1, 24, 3584, 64                     # the reference harness's N-scan shape (bench_with_sdpa.py:52, :201-224): 336 items -> split
    dev = _dev()
    q, k, v = (torch.rand((B, H, N, D), device=dev).half() for _ in range(3))
    need = _fa2_lib.load(build_if_missing=False).fa2_fwd_workspace_bytes(0, B, H, N, N, D, 0)
    assert need > 0
    for python_front_end in (False, True):
        saved = FlashAttn._FRONTEND[0]
        if python_front_end:
            FlashAttn._FRONTEND[0] = None
        try:
            FlashAttentionFunction.apply(q, k, v, None, False)       # first use: the block exists from here on
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            for _ in range(20):
                o = FlashAttentionFunction.apply(q, k, v, None, False)
            torch.cuda.synchronize()
            out_bytes = o.numel() * 2 + B * H * N * 4
            assert torch.cuda.max_memory_allocated() - base <= 2 * out_bytes + (1 << 20), (torch.cuda.max_memory_allocated() - base, out_bytes, need)
            del o
        finally:
            FlashAttn._FRONTEND[0] = saved
    assert FlashAttn.workspace_pool_bytes() >= need               # (the Python front end's pool; the compiled one keeps its own)
    ptr = [t.data_ptr() for t in FlashAttn._WS_POOL.values()]
    FlashAttn._FRONTEND[0], saved = None, FlashAttn._FRONTEND[0]
    try:
        FlashAttentionFunction.apply(q, k, v, None, False)
        assert [t.data_ptr() for t in FlashAttn._WS_POOL.values()] == ptr
        s2 = torch.cuda.Stream()
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            FlashAttentionFunction.apply(q, k, v, None, False)
        torch.cuda.synchronize()
        assert len(FlashAttn._WS_POOL) == len(ptr) + 1
    finally:
        FlashAttn._FRONTEND[0] = saved


@pytest.mark.gpu
def test_split_scratch_follows_the_calls_being_made():
    """the per-stream block was a high-water mark — after one call that needed tens of MB every later call of the process carried
    them (the reference harness's D scan: +60 MB on every row, 4.3x torch SDPA's footprint at D = 16; bench_with_sdpa.py:34 records the peak per run).
    Now a call that needs less than half of a block above 4 MiB gets a right-sized one, and a call that needs none releases it; both front ends."""
    from rocwmma_fattn import FlashAttn
    lib = _fa2_lib.load(build_if_missing=False)
    dev = _dev()
    big = (1, 24, 5632, 64)                             # bench_with_sdpa.py's N scan: a large split
    small = (1, 24, 512, 64)
    need_big = lib.fa2_fwd_workspace_bytes(0, *big[:3], big[2], big[3], 0)
    need_small = lib.fa2_fwd_workspace_bytes(0, *small[:3], small[2], small[3], 0)
    assert need_big > (8 << 20) and need_small < need_big // 2, (need_big, need_small)
    qb, kb, vb = (torch.rand(big, device=dev).half() for _ in range(3))
    qs, ks, vs = (torch.rand(small, device=dev).half() for _ in range(3))
    # (the pools also hold the blocks earlier tests left on other streams: differences of the total, not the total)
    for python_front_end in (False, True):
        saved = FlashAttn._FRONTEND[0]
        if python_front_end:
            FlashAttn._FRONTEND[0] = None
        try:
            FlashAttentionFunction.apply(qb, kb, vb, None, False)
            torch.cuda.synchronize()
            p1 = FlashAttn.workspace_pool_bytes()
            o_ref = FlashAttentionFunction.apply(qs, ks, vs, None, False)                  # needs less than half (here: none): the block goes
            torch.cuda.synchronize()
            p2 = FlashAttn.workspace_pool_bytes()
            assert p1 - p2 >= need_big - need_small - (4 << 20) and p1 - p2 > 0, (p1, p2, need_big, need_small)
            FlashAttentionFunction.apply(qb, kb, vb, None, False)                           # ... a large call takes it back
            torch.cuda.synchronize()
            assert FlashAttn.workspace_pool_bytes() - p2 >= need_big - need_small - (4 << 20)
            FlashAttentionFunction.apply(qs, ks, vs, None, True)                            # a causal call: no split, no scratch
            torch.cuda.synchronize()
            assert FlashAttn.workspace_pool_bytes() <= p2 + (4 << 20)
            assert torch.equal(o_ref, FlashAttentionFunction.apply(qs, ks, vs, None, False))      # results do not depend on which block served a call
        finally:
            FlashAttn._FRONTEND[0] = saved


@pytest.mark.gpu
@pytest.mark.parametrize("D,dt", [(64, 0), (128, 1)])
def test_split_parts_with_one_workgroup_per_item(D, dt):
    """Option persist = 0 launches one workgroup per list entry of the hand-scheduled kernels — whole items and KV-split parts alike — instead of
    persistent workgroups: the arithmetic of an entry does not depend on how it was scheduled, so the merged result is bit-identical."""
    B, H, N = 1, 24, 4096 if D == 128 else 3072
    g = torch.Generator(device="cpu").manual_seed(12 + D)
    q, k, v = (torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(3))
    o1, l1, need = _fwd(q, k, v, "ws")
    assert need > 0
    with _fa2_lib.options(persist=0):
        o0, l0, _ = _fwd(q, k, v, "ws")
    assert torch.equal(o1.view(torch.int16), o0.view(torch.int16)) and torch.equal(l1, l0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 10, 4096, 77, 64, 0), (2, 8, 4096, 77, 40, 1), (1, 6, 2048, 300, 64, 0)])
def test_short_kv_backward_splits_every_kv_owner(shape):
    """Cross-attention backward (Nkv = 77): the KV-owned dK / dV pass has B*H workgroups in all, each sweeping every Q row — a tenth of the chip.
    With a workspace (fa2_bwd_ws) every one of them is split along its Q sweep (plan_tail_split, underfilled grids) and merged.  Against the plain
    call, the oracle and float64 autograd."""
    from conftest import GRAD_TOL
    B, H, N, Nkv, D, dt = shape
    g = torch.Generator(device="cpu").manual_seed(4000 + D + Nkv)
    q, do = (torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
    k, v = (torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
    o, lse, gw, need = _fwd_bwd(q, k, v, do, "ws")
    assert need > 0
    _, _, gp, _ = _fwd_bwd(q, k, v, do, "none")
    for name, a, b_ in zip("qkv", gw, gp):
        assert torch.isfinite(a.float()).all(), name
        tol = (2.0 ** -10 if dt == 0 else 2.0 ** -7) * max(1.0, float(b_.float().abs().max()))
        assert float((a.float() - b_.float()).abs().max()) <= tol, name
    for (b, h) in {(0, 0), (B - 1, H - 1)}:
        sl = (slice(b, b + 1), slice(h, h + 1))
        want = fo.bwd_c(_bits(q[sl]), _bits(k[sl]), _bits(v[sl]), _bits(o[sl]), _bits(do[sl]), lse[sl].cpu().numpy(), dt, False)
        for name, gt, w_bits in zip("qkv", gw, want):
            w = fo.bits_to_f32(w_bits, dt)
            assert np.abs(gt[sl].float().cpu().numpy() - w).max() <= GRAD_TOL[dt] * max(1.0, np.abs(w).max()), (name, (b, h))
        qd, kd, vd = (t[sl].double().requires_grad_(True) for t in (q, k, v))
        torch.nn.functional.scaled_dot_product_attention(qd, kd, vd).backward(do[sl].double())
        for name, gt, w in zip("qkv", gw, (qd.grad, kd.grad, vd.grad)):
            assert float((gt[sl].double() - w).abs().max()) <= GRAD_TOL[dt] * max(1.0, float(w.abs().max())), (name, (b, h))


MASKED_SPLIT_CASES = [
    # B, H, Nq, Nkv, D, dt, mask kind: "pad" = [B,1,1,Nkv] bool key-padding (row broadcast: one load per KV row in the dK / dV pass),
    # "add" = [B,1,Nq,Nkv] additive in the I/O dtype (per score; Nkv = 77: unaligned, 80: LDS-DMA tiles), "f32h" = [1,H,Nq,Nkv] float32
    (2, 10, 4096, 77, 64, 0, "pad"), (2, 8, 4096, 77, 40, 1, "pad"), (2, 10, 2048, 77, 64, 0, "add"), (2, 10, 2048, 80, 64, 0, "add"),
    (1, 6, 2048, 304, 64, 1, "f32h"), (2, 10, 4096, 4096, 64, 0, "pad"), (1, 24, 3072, 3072, 64, 0, "pad"), (1, 24, 4096, 4096, 128, 0, "pad"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", MASKED_SPLIT_CASES)
def test_masked_backward_with_workspace(case):
    """fa2_bwd_bias_ws: the masked backward under the split of fa2_bwd_ws — short-KV cross-attention with a key-padding mask (every KV-owner
    workgroup of the fused dK / dV pass split along its Q sweep), partly filled last rounds of the dQ pass, head dim 128 (dQ pass only).  Through the
    operator (flash_attention(mask=...)), split on against split off (same kernels, other f32 summation order) and against float64 autograd; fully
    masked keys get zero dK / dV."""
    from conftest import GRAD_TOL
    from rocwmma_fattn.FlashAttn import flash_attention
    B, H, N, Nkv, D, dt, kind = case
    g = torch.Generator(device="cpu").manual_seed(5000 + N + Nkv + D)
    q, do = (torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
    k, v = (torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
    if kind == "pad":
        keep = torch.rand((B, 1, 1, Nkv), generator=g) < 0.7
        keep[..., 0] = True
        mask = keep.to(_dev())
    elif kind == "add":
        mask = (torch.randn((B, 1, N, Nkv), generator=g) * 2).to(TORCH_DT[dt]).to(_dev())
    else:
        mask = (torch.randn((1, H, N, Nkv), generator=g) * 2).to(_dev())
    lib = _fa2_lib.load()
    need = lib.fa2_bwd_bias_workspace_bytes(dt, B, H, N, Nkv, D + (-D % 8), 0)
    assert need > 0, "the case is meant to split"
    grads = {}
    for mode in (1, 0):
        with _fa2_lib.options(split=mode):
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            flash_attention(qq, kk, vv, mask=mask).backward(do)
            torch.cuda.synchronize()
            grads[mode] = (qq.grad, kk.grad, vv.grad)
    for name, a, b_ in zip("qkv", grads[1], grads[0]):
        assert torch.isfinite(a.float()).all(), name
        tol = (2.0 ** -9 if dt == 0 else 2.0 ** -6) * max(1.0, float(b_.float().abs().max()))
        assert float((a.float() - b_.float()).abs().max()) <= tol, name
    if kind == "pad":
        dead = ~mask[:, 0, 0]                                    # [B, Nkv]
        for t in grads[1][1:]:
            assert float(t.float().abs().amax(dim=(1, 3))[dead].max() if dead.any() else 0.0) == 0.0
    for (b, h) in {(0, 0), (B - 1, H - 1)}:
        sl = (slice(b, b + 1), slice(h, h + 1))
        qd, kd, vd = (t[sl].double().requires_grad_(True) for t in (q, k, v))
        mb, mh = (b if mask.shape[0] > 1 else 0), (h if mask.shape[1] > 1 else 0)
        m = mask[mb:mb + 1, mh:mh + 1]
        m = m if m.dtype == torch.bool else m.double()
        torch.nn.functional.scaled_dot_product_attention(qd, kd, vd, attn_mask=m).backward(do[sl].double())
        for name, gt, w in zip("qkv", grads[1], (qd.grad, kd.grad, vd.grad)):
            assert float((gt[sl].double() - w).abs().max()) <= GRAD_TOL[dt] * max(1.0, float(w.abs().max())), (name, (b, h))
