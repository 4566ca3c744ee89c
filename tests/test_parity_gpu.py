"""GPU parity tests (`-m gpu`): the HIP path, called through the C-ABI (ctypes -> libfa2_gfx950.so),
against the golden fixtures, the C oracle on seeded inputs, and size-independent properties at the
BASELINE.json configurations.  Nothing here reads /root/reference."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import ATOL, FLOOR, LSE_TOL, LSE_TOL_P16_BF16, LSE_TRUTH_TOL, RTOL
from oracle import fa2_oracle as fo
from rocwmma_fattn import _fa2_lib
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma

pytestmark = pytest.mark.gpu

TORCH_DT = {0: torch.float16, 1: torch.bfloat16}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run the CPU suite with -m 'not gpu'"
    return torch.device("cuda", 0)


def _to_dev(bits, dt):
    return torch.from_numpy(bits.view(np.int16).copy()).view(TORCH_DT[dt]).to(_dev())


def _bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def _cabi_forward(q, k, v, causal, scale=None):
    """Straight through the C-ABI (fa2_fwd_f16 / fa2_fwd_bf16), caller-owned buffers."""
    lib = _fa2_lib.load(build_if_missing=False)
    B, H, N, D = q.shape
    Nkv = k.shape[2]
    o = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    fn = lib.fa2_fwd_f16 if q.dtype == torch.float16 else lib.fa2_fwd_bf16
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, Nkv, D,
            s3(q), s3(k), s3(v), s3(o), _fa2_lib.strides2(lse.stride(0), lse.stride(1)),
            float(D ** -0.5 if scale is None else scale), int(causal),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _fa2_lib.check(rc)
    torch.cuda.synchronize()
    return o, lse


def _plan(q, k, causal, scale=None, workspace_bytes=0):
    """fa2_fwd_plan of the call fa2_fwd*(q, k, v, causal, scale): the kernel(s) that serve it and their numerical contract."""
    return _fa2_lib.fwd_plan(q, k, causal, scale, workspace_bytes=workspace_bytes)


def _oracle_flags_of(contract):
    """FA2_CONTRACT_* bits (include/fa2_gfx950.h) -> the oracle's flags for the same contract."""
    return (fo.PRESCALE_Q if contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q else 0) | (fo.LSUM_P16 if contract & _fa2_lib.FA2_CONTRACT_LSUM_P16 else 0)


def _oracle_flags(q, k, causal, scale=None):
    """Oracle flags of the (single-launch) call fa2_fwd*(q, k, ...)."""
    pl = _plan(q, k, causal, scale)
    assert pl.heads_main == q.shape[0] * q.shape[1], "two launches: ask _plan() per head range"
    return _oracle_flags_of(pl.contract)


def _assert_close_to_oracle(o, lse, q, k, v, dt, causal, scale=None, plan=None, head=None):
    """Against the oracle run under THE contract of the kernel that served the call, as fa2_fwd_plan names it (one contract per launch;
    a call of two launches — head dims <= 64 with a nearly empty last round — is checked per head range).
      plan   the plan of the call that produced o (default: the plan of fa2_fwd*(q, k, v, causal, scale) itself — right whenever the
             tensors handed in ARE the call's tensors and no workspace was involved);
      head   o, lse, q, k, v are the [1, 1, ...] slice of flattened head `head` of that call.
    Tolerances: conftest (bf16 row sums of rounded P — LSUM_P16: LSE_TOL_P16_BF16; the kernel rounds P against its deferred reference maximum, the
    oracle against the running one: 2^-9 relative noise per term, a row with one to three visible keys shows all of it)."""
    B, H = q.shape[0], q.shape[1]
    if plan is None:
        assert head is None
        plan = _plan(q, k, causal, scale)
    got = o.float().cpu().numpy()
    assert np.isfinite(got).all()
    lse_np = lse.cpu().numpy()
    if head is not None:
        assert B == 1 and H == 1
        ranges = [(0, 1, plan.contract if head < plan.heads_main else plan.contract_tail)]
    else:
        assert plan.heads_main <= B * H
        ranges = [(0, plan.heads_main, plan.contract)]
        if plan.heads_main < B * H:
            ranges.append((plan.heads_main, B * H, plan.contract_tail))
    qb, kb, vb = (_bits(t).reshape((1, B * H) + tuple(t.shape[2:])) for t in (q, k, v))
    got = got.reshape((1, B * H) + got.shape[2:])
    lse_np = lse_np.reshape((1, B * H) + lse_np.shape[2:])
    for lo, hi, contract in ranges:
        flags = _oracle_flags_of(contract)
        if dt == 0 and flags & fo.PRESCALE_Q:
            flags |= fo.PRESCALE_FUSED          # (the fp16 kernels round Q * c once, from the exact product: v_fma_mixlo_f16)
        lse_tol = LSE_TOL_P16_BF16 if (dt == 1 and flags & fo.LSUM_P16) else LSE_TOL
        o_ref_bits, lse_ref = fo.fwd_c(np.ascontiguousarray(qb[:, lo:hi]), np.ascontiguousarray(kb[:, lo:hi]), np.ascontiguousarray(vb[:, lo:hi]),
                                       dt, causal, scale=scale, flags=flags)
        o_ref = fo.bits_to_f32(o_ref_bits, dt)
        diff = np.abs(got[:, lo:hi] - o_ref)
        bad = diff > ATOL[dt] + RTOL[dt] * np.abs(o_ref)
        lse_err = np.abs(lse_np[:, lo:hi] - lse_ref).max()
        assert not bad.any() and lse_err <= lse_tol, \
            "heads [%d, %d) under contract %d (kernel %d / %d): max |O diff| %.3g, max |LSE diff| %.3g (tol %.3g)" % (
                lo, hi, contract, plan.kernel, plan.kernel_tail, float(diff.max()), float(lse_err), lse_tol)


# ---------------------------------------------------------------- the default fp16 forward on large logits (round 6)

@pytest.mark.parametrize("causal", [0, 1])
def test_large_logits_do_not_cost_the_default_fp16_forward_a_second_sweep(causal):
    """Config 2's shape on N(0, amp^2) logits.  The default fp16 bodies (row sums on the matrix pipe, `asm` bit 9) never move the reference in their fast
    loop; until round 5 a score 16 octaves above it (amp >= 3: a few items per launch; amp >= 4: most) sent the item through a second sweep — 1.4 .. 1.7x
    (profiles/r19_growth_cliff.txt).  Now the tile is formed again in place and the sweep goes on on the max-first bodies (csrc/gen/fwd_m16_gen.py:
    lm_repair; profiles/r20_growth_cliff.txt: <= 1.09x the sum-check bodies on every row).  This times amp 4 and 8 against amp 1 and against the sum-check
    bodies on the same data, interleaved after a settle phase, and checks the large-logit results: deterministic, and right against the oracle."""
    lib = _fa2_lib.load(build_if_missing=False)
    B, H, N, D = 2, 16, 4096, 128
    full = lib.fa2_get_option(b"asm")
    data = {}
    for amp in (1.0, 4.0, 8.0):
        g = torch.Generator(device="cpu").manual_seed(5)
        q, k, v = ((amp ** 0.5 if i < 2 else 1.0) * torch.randn((B, H, N, D), generator=g) for i in range(3))
        data[amp] = tuple(t.to(torch.float16).to(_dev()) for t in (q, k, v))
    plan = _plan(*data[4.0][:2], causal)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.contract & _fa2_lib.FA2_CONTRACT_LSUM_P16, plan.as_dict()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    o = torch.empty_like(data[1.0][0])
    lse = torch.empty((B, H, N), dtype=torch.float32, device=_dev())
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731

    def call(amp):
        q, k, v = data[amp]
        _fa2_lib.check(lib.fa2_fwd_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D, s3(q), s3(k), s3(v), s3(o),
                                       _fa2_lib.strides2(lse.stride(0), lse.stride(1)), float(D ** -0.5), causal, stream))
    try:
        for _ in range(300):                      # settle the clock (after idle the chip boosts, overshoots and throttles for tens of ms)
            call(1.0)
        torch.cuda.synchronize()
        arms = [(1.0, full), (4.0, full), (8.0, full), (4.0, full & ~512), (8.0, full & ~512)]
        ts = {a: [] for a in arms}
        for _ in range(5):
            for arm in arms:
                lib.fa2_set_option(b"asm", arm[1])
                for _ in range(4):
                    call(arm[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(15):
                    call(arm[0])
                e1.record()
                torch.cuda.synchronize()
                ts[arm].append(e0.elapsed_time(e1) / 15)
    finally:
        lib.fa2_set_option(b"asm", full)
    t = {a: sorted(x)[len(x) // 2] for a, x in ts.items()}
    for amp, bound in ((4.0, 1.30), (8.0, 1.55)):
        # against benign data (measured 1.08 .. 1.15 at amp 4, 1.17 .. 1.42 at amp 8 — logits that large cost EVERY body kind its reference moves: the
        # sum-check bodies 1.10 .. 1.34 —; round 5: 1.56 .. 1.83 / 1.48 .. 1.70) ...
        assert t[(amp, full)] <= bound * t[(1.0, full)], t
        # ... and against the sum-check bodies on the same data (measured 1.04 .. 1.09; round 5: 1.29 .. 1.69)
        assert t[(amp, full)] <= 1.15 * t[(amp, full & ~512)], t
    q, k, v = data[4.0]
    o1, lse1 = _cabi_forward(q, k, v, causal)
    o2, lse2 = _cabi_forward(q, k, v, causal)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2) and torch.isfinite(o1.float()).all() and torch.isfinite(lse1).all()
    for (b, h) in ((0, 0), (1, 7), (1, 15)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o1[sl], lse1[sl], q[sl], k[sl], v[sl], 0, causal, plan=plan, head=b * H + h)


@pytest.mark.parametrize("D,dt,causal", [(112, 0, False), (56, 0, True), (120, 1, False), (64, 0, False), (64, 1, True)])
def test_large_logits_on_the_other_hand_scheduled_bodies(D, dt, causal):
    """The in-place repair (csrc/gen/fwd_m16_gen.py: lm_repair) on the kernel kinds config 2's guard above does not reach: head dims below a body's
    (the K fragments it reloads are masked per granule like the LDS-DMA's), head dim 64 folded and f32-scale, bf16 — N(0, 6^2) logits, against the
    oracle under the planned contract; deterministic."""
    B, H, N = 2, 33, 2048
    g = torch.Generator(device="cpu").manual_seed(40 + D + dt)
    q, k, v = ((6.0 ** 0.5 if i < 2 else 1.0) * torch.randn((B, H, N, D), generator=g) for i in range(3))
    q, k, v = (t.to(TORCH_DT[dt]).to(_dev()) for t in (q, k, v))
    with _fa2_lib.options(rows=256):
        plan = _plan(q, k, causal)
        assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.contract & _fa2_lib.FA2_CONTRACT_LSUM_P16 and plan.heads_main == B * H, plan.as_dict()
        o, lse = _cabi_forward(q, k, v, causal)
        o2, lse2 = _cabi_forward(q, k, v, causal)
    assert torch.equal(o, o2) and torch.equal(lse, lse2) and torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    for (b, h) in ((0, 0), (1, 17), (1, 32)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        if dt == 0:
            _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], dt, causal, plan=plan, head=b * H + h)
            continue
        # bf16 on logits this large: rows with one or two dominant keys carry the whole rounding of those P (2^-8 relative: 5.6e-3 of LSE, 2^-8 |v| of
        # O), and kernel and oracle round them against different references — the suite's bf16 bars (made for N(0,1) logits) plus that one ulp
        o_ref_bits, lse_ref = fo.fwd_c(_bits(q[sl]), _bits(k[sl]), _bits(v[sl]), dt, causal, flags=_oracle_flags_of(plan.contract))
        o_ref = fo.bits_to_f32(o_ref_bits, dt)
        assert np.abs(lse[sl].cpu().numpy() - lse_ref).max() <= 8e-3
        assert (np.abs(o[sl].float().cpu().numpy() - o_ref) <= ATOL[dt] + RTOL[dt] * np.abs(o_ref) + 2.0 ** -8 * float(v[sl].float().abs().max())).all()


# ---------------------------------------------------------------- golden fixtures

def test_golden_fixtures_through_cabi(golden):
    dt = golden["dtype"]
    q, k, v = (_to_dev(golden[t], dt) for t in "qkv")
    n = golden["N"]
    for causal, var in golden["variants"].items():
        o, lse = _cabi_forward(q, k, v, causal)
        got = o.float().cpu().numpy()
        o_ref = fo.bits_to_f32(var["o_ref"], dt)
        ref_err = np.abs(o_ref - var["o_true"]).max()
        err = np.abs(got - var["o_true"]).max()
        assert err <= max(2 * ref_err, FLOOR[dt]), (golden["name"], causal, err, ref_err)
        assert np.abs(lse.cpu().numpy() - var["lse2_true"]).max() <= LSE_TRUTH_TOL[dt]
        # the reference's L is natural-log, kept in the input dtype: compare coarsely after * log2(e)
        assert np.abs(lse.cpu().numpy() - var["l_ref"][:, :, :n] * fo.LOG2E).max() <= (2e-2 if dt == 0 else 1.6e-1)
        _assert_close_to_oracle(o, lse, q, k, v, dt, causal)


def test_golden_fixtures_through_operator(golden):
    """Same data through FlashAttentionFunction.apply, the reference's call shape (bench_with_sdpa.py:99)."""
    dt = golden["dtype"]
    q, k, v = (_to_dev(golden[t], dt) for t in "qkv")
    for causal, var in golden["variants"].items():
        o = FlashAttentionFunction.apply(q, k, v, None, causal)
        torch.cuda.synchronize()
        assert o.shape == q.shape and o.dtype == q.dtype and o.device == q.device
        err = np.abs(o.float().cpu().numpy() - var["o_true"]).max()
        ref_err = np.abs(fo.bits_to_f32(var["o_ref"], dt) - var["o_true"]).max()
        assert err <= max(2 * ref_err, FLOOR[dt])


# ---------------------------------------------------------------- seeded inputs vs the C oracle

SHAPES = [
    # B, H, Nq, Nkv, D
    (1, 1, 1, 1, 64), (1, 2, 1, 300, 128), (2, 3, 65, 1, 64), (1, 2, 31, 33, 64), (1, 3, 255, 257, 128),
    (2, 2, 256, 256, 128), (1, 2, 257, 511, 64), (1, 1, 700, 700, 128), (3, 5, 130, 77, 64), (1, 8, 512, 512, 128),
    (1, 2, 300, 300, 256), (2, 1, 65, 77, 256),     # D = 256: two 128-column halves per workgroup row
    (3, 7, 1537, 1234, 112),                        # the reference's one precision shape (precision_test.py:34-39; its D = 111 is
                                                    # zero-padded to the next multiple of 8 by the operator: test_reference_precision_shape)
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("causal", [False, True])
def test_seeded_shapes_against_oracle(shape, dt, causal):
    B, H, Nq, Nkv, D = shape
    g = torch.Generator(device="cpu").manual_seed(hash(shape) % 1000 + 17 * dt)
    q = torch.randn((B, H, Nq, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    k = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    v = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    o, lse = _cabi_forward(q, k, v, causal)
    _assert_close_to_oracle(o, lse, q, k, v, dt, causal)


def test_explicit_and_negative_scale():
    g = torch.Generator(device="cpu").manual_seed(3)
    q, k, v = (torch.randn((1, 2, 200, 64), generator=g).half().to(_dev()) for _ in range(3))
    lib = _fa2_lib.load(build_if_missing=False)
    # the coarse query: MAY fold at head dims 64 / 128 while 0 < scale * log2(e) <= 1 (the prescaled Q stays inside fp16's range)
    assert lib.fa2_fwd_prescales_q(64, 0.5) == 1 and lib.fa2_fwd_prescales_q(128, 0.5) == 1 and lib.fa2_fwd_prescales_q(64, 1.5) == 0
    assert lib.fa2_fwd_prescales_q(64, -1.0) == 0 and lib.fa2_fwd_prescales_q(256, 0.5) == 0
    for scale in (0.3, -0.2, 1.5, -2.0):
        for causal in (False, True):
            o, lse = _cabi_forward(q, k, v, causal, scale=scale)
            _assert_close_to_oracle(o, lse, q, k, v, 0, causal, scale=scale)


@pytest.mark.parametrize("D", [64, 128])
def test_large_scale_and_large_q_stay_finite_on_the_hand_scheduled_grid(D):
    """scale * log2(e) > 1 with |Q| near the top of fp16's range: a body that folded the scale into Q would round q * c to inf; the plan
    keeps the f32-scale body of the same schedule for such a scale (host.cpp: asm_folds) and the reference kernel's arithmetic (scale applied to
    the f32 product, kernel_fp16.cu:164) stays finite.  On a grid wide enough for the hand-scheduled kernel; K is tiny so the logits stay moderate."""
    B, H, N = 2, 16, 2048
    g = torch.Generator(device="cpu").manual_seed(91 + D)
    q = (torch.randn((B, H, N, D), generator=g) * 12000).clamp(-60000, 60000).half().to(_dev())
    k = (torch.randn((B, H, N, D), generator=g) * 2e-5).half().to(_dev())
    v = torch.randn((B, H, N, D), generator=g).half().to(_dev())
    for scale in (1.0, 2.5):
        plan = _plan(q, k, False, scale)
        assert plan.contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q == 0, plan.as_dict()
        if D == 128:
            assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM
        o, lse = _cabi_forward(q, k, v, False, scale=scale)
        assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
        for (b, h) in ((0, 0), (1, 15)):
            sl = (slice(b, b + 1), slice(h, h + 1))
            _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], 0, False, scale=scale, plan=plan, head=b * H + h)


def test_option_fold_2_folds_bf16_launches_too():
    """Option "fold" = 2: bf16 launches of the hand-scheduled bodies round Q * scale*log2(e) once to bf16 (opt-in: an 8-bit mantissa costs ~6e-3 of
    log2 LSE against float64) — config 3's shape, against the oracle under THAT contract at the usual tolerance, and back to the f32-scale body after."""
    B, H, N, D = 2, 16, 4096, 128
    g = torch.Generator(device=_dev()).manual_seed(55)
    q, k, v = (torch.rand((B, H, N, D), generator=g, device=_dev(), dtype=torch.float32).bfloat16() for _ in range(3))
    o0, lse0 = _cabi_forward(q, k, v, True)
    with _fa2_lib.options(fold=2):
        plan = _plan(q, k, True)
        assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.contract == _fa2_lib.FA2_CONTRACT_PRESCALE_Q | _fa2_lib.FA2_CONTRACT_LSUM_P16
        o, lse = _cabi_forward(q, k, v, True)
        for (b, h) in ((0, 0), (1, 15)):
            sl = (slice(b, b + 1), slice(h, h + 1))
            _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], 1, True, plan=plan, head=b * H + h)
    assert float((lse - lse0).abs().max()) <= 2e-2 and float((o.float() - o0.float()).abs().max()) <= 2 * FLOOR[1]
    o1, lse1 = _cabi_forward(q, k, v, True)
    assert torch.equal(o1, o0) and torch.equal(lse1, lse0)


def test_scale_zero_is_the_uniform_softmax():
    """scale == 0: every score is 0, O = mean of the visible V rows, LSE = log2(count) — what the reference's
    arithmetic gives (kernel_fp16.cu:449-479 with scale' = 0)."""
    g = torch.Generator(device="cpu").manual_seed(4)
    q = torch.randn((1, 2, 130, 128), generator=g).half().to(_dev())
    k = torch.randn((1, 2, 200, 128), generator=g).half().to(_dev())
    v = torch.randn((1, 2, 200, 128), generator=g).half().to(_dev())
    o, lse = _cabi_forward(q, k, v, False, scale=0.0)
    assert float((o.float() - v.float().mean(dim=2, keepdim=True)).abs().max()) <= 2e-3
    assert float((lse - np.log2(200.0)).abs().max()) <= 1e-4
    o, lse = _cabi_forward(q, k, v, True, scale=0.0)
    cnt = torch.arange(1, 131, device=_dev(), dtype=torch.float32)
    ref = v.float().cumsum(dim=2)[:, :, :130] / cnt[None, None, :, None]
    assert float((o.float() - ref).abs().max()) <= 2e-3
    assert float((lse - torch.log2(cnt)[None, None]).abs().max()) <= 1e-4


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("bnhd", [False, True])
def test_ragged_tail_ignores_memory_past_nkv(dt, bnhd):
    """K/V live inside a larger NaN-filled allocation: rows >= Nkv of the ragged last tile (and whatever follows the
    last head) must never reach the result — P is 0 there, but 0 * NaN = NaN, so the staged V rows have to be
    zero-filled by the bounds check, not merely masked in S."""
    B, H, Nq, Nkv, D = 2, 3, 200, 77 + 64, 128
    g = torch.Generator(device="cpu").manual_seed(21 + dt)
    tdt = TORCH_DT[dt]
    q = torch.randn((B, H, Nq, D), generator=g).to(tdt).to(_dev())
    kv = torch.randn((2, B, H, Nkv, D), generator=g).to(tdt)
    pad = 130                                                    # > one tile of poisoned rows after every head / batch
    if bnhd:   # [B, N, H, D] storage: row stride H*D, the rows past Nkv belong to the same batch's allocation
        big = torch.full((2, B, Nkv + pad, H, D), float("nan"), dtype=tdt)
        big[:, :, :Nkv] = kv.permute(0, 1, 3, 2, 4)
        big = big.to(_dev())
        k, v = (big[i, :, :Nkv].permute(0, 2, 1, 3) for i in range(2))   # BHND views with BNHD strides
    else:
        big = torch.full((2, B, H, Nkv + pad, D), float("nan"), dtype=tdt)
        big[:, :, :, :Nkv] = kv
        big = big.to(_dev())
        k, v = (big[i, :, :, :Nkv] for i in range(2))
    for causal in (False, True):
        o, lse = _cabi_forward(q, k, v, causal)
        assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
        _assert_close_to_oracle(o, lse, q, k.contiguous(), v.contiguous(), dt, causal)


def test_reference_precision_shape():
    """precision_test.py:34-39 of the reference: (B, H, N, D) = (3, 7, 1537, 111), Nkv = 1234, bf16, inputs * 1.2,
    through the operator (D = 111 is the one kind of head dim that is zero-padded on the host, to 112)."""
    g = torch.Generator(device="cpu").manual_seed(31)
    q = (torch.rand((3, 7, 1537, 111), generator=g) * 1.2).bfloat16().to(_dev())
    k = (torch.rand((3, 7, 1234, 111), generator=g) * 1.2).bfloat16().to(_dev())
    v = (torch.rand((3, 7, 1234, 111), generator=g) * 1.2).bfloat16().to(_dev())
    for causal in (False, True):
        o = FlashAttentionFunction.apply(q, k, v, None, causal, None, False)     # precision_test.py:63 call shape
        torch.cuda.synchronize()
        assert o.shape == q.shape and o.dtype == q.dtype
        o_ref_bits, _ = fo.fwd_c(_bits(q), _bits(k), _bits(v), 1, causal)
        o_ref = fo.bits_to_f32(o_ref_bits, 1)
        assert np.all(np.abs(o.float().cpu().numpy() - o_ref) <= ATOL[1] + RTOL[1] * np.abs(o_ref))
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (111 ** -0.5)
        if causal:
            s = s.masked_fill(torch.ones(1537, 1234, dtype=torch.bool, device=_dev()).triu(1), float("-inf"))
        truth = torch.matmul(torch.softmax(s, -1), v.float())
        assert float((o.float() - truth).abs().max()) <= FLOOR[1]


@pytest.mark.parametrize("D", [64, 128, 256])
def test_large_logits_and_forced_rescale(D):
    """Spike one K row so the running max jumps late in the sweep (the online-softmax rescale branch), and use logits
    of several hundred (cdna guide §5.4 rule 26: force the rare branch)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn((1, 2, 300, D), generator=g) * 3).half()
    k = (torch.randn((1, 2, 640, D), generator=g) * 3).half()
    v = torch.randn((1, 2, 640, D), generator=g).half()
    k[:, :, 517] = q[:, :, 11] * 4          # row 11's max jumps at kv tile 8
    k[:, :, 70] = q[:, :, 200] * 2
    q, k, v = q.to(_dev()), k.to(_dev()), v.to(_dev())
    o, lse = _cabi_forward(q, k, v, False)
    o_true, lse_true = fo.fwd_numpy(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), False)
    got = o.float().cpu().numpy()
    assert np.isfinite(got).all()
    # folded-scale build (opt-in): Q * scale*log2e is rounded to 16 bits, which on logits of several hundred is ~1e-2 in O and
    # ~0.1 in the log2 LSE against the float64 truth (measured 1.3e-2); the same-contract oracle below stays at the usual tolerance
    folded = _oracle_flags(q, k, False) & fo.PRESCALE_Q
    assert np.all(np.abs(got - o_true) <= (3e-2 if folded else 2e-3) + 4e-3 * np.abs(o_true))
    assert np.abs(lse.cpu().numpy() - lse_true).max() <= (0.3 if folded else 2e-2)
    _assert_close_to_oracle(o, lse, q, k, v, 0, False)
    # option "fold" = 0: every launch scales the f32 product like the reference kernel (kernel_fp16.cu:164) — the tight truth bounds hold
    with _fa2_lib.options(fold=0):
        assert _oracle_flags(q, k, False) & fo.PRESCALE_Q == 0
        o, lse = _cabi_forward(q, k, v, False)
        got = o.float().cpu().numpy()
        assert np.all(np.abs(got - o_true) <= 2e-3 + 4e-3 * np.abs(o_true))
        assert np.abs(lse.cpu().numpy() - lse_true).max() <= 2e-2
        _assert_close_to_oracle(o, lse, q, k, v, 0, False)


@pytest.mark.parametrize("dt,causal,D", [(0, False, 128), (1, False, 128), (0, True, 128), (0, False, 64), (1, True, 64)])
@pytest.mark.parametrize("kind", ["repair", "redo"])
def test_sum_check_bodies_repair_in_place_and_redo_in_safe_mode(dt, causal, D, kind):
    """The fast bodies of the hand-scheduled head-dim-128 kernels keep no running row maximum: a lane's row sum of a tile proves that no P of the
    tile overflows, and the reference moves, out of line, only when that check fails (csrc/gen/fwd_d128_gen.py: stream_exp_sum / rare_sum).
      repair  a key row raises some rows' scores 15 .. 127 log2 units above everything before it, late in the sweep: the reference is moved in
              place — rows of q block 0 and of q block 1 (whose next-tile scores get their shift a body later), one row twice;
      redo    ... by more than 2^7 log2 units: P overflows f32 itself, the wave raises the workgroup's flag and the shell runs the item again in
              safe mode (max-first bodies) — here on a grid with more items than workgroups, so that items redone are first items, second
              (prefetched) items and items with a successor.
    Both against float64 attention and the oracle under the planned contract (tests/test_asm_emu.py emulates the same two paths instruction by instruction)."""
    B, H, N = (1, 33, 2048) if causal else (1, 66, 1024)    # 264 items of 256 rows on 256 CUs (causal: the hand-scheduled kernel from 1792 keys)
    if D == 64 and not causal:
        H = 100                                             # (head dim 64 runs 128-row workgroups between 1 and 1.5 rounds: 400 items)
    g = torch.Generator(device="cpu").manual_seed(700 + dt + 2 * causal + D)
    q, k, v = (torch.randn((B, H, N, D), generator=g) for _ in range(3))
    if D == 64:
        q = q * 2 ** 0.5                                    # (the same logit statistics as at head dim 128: |q|^2 * scale = sqrt(D))
    if kind == "repair":
        k[:, :, 200] = q[:, :, 5] * 2.7
        k[:, :, 300] = q[:, :, 40] * 2.7
        k[:, :, 130] = q[:, :, 41] * 1.5
        k[:, :, 330] = q[:, :, 41] * 3.0
        k[:, :, 700] = q[:, :, 900] * 2.7                 # (a row of the last q block; causal: visible to it)
    else:
        # (inputs three times as large: besides the planted rows, ~1 row in 10^4 grows by 100+ octaves in one tile on its own — the first version of
        #  the repair wiped the rows that grew by 126 .. 127: its factor 2^-127 is no normal f32 and v_exp_f32 returns 0 for it)
        q, k = q * 3, k * 3
        k[:, :, 200] = q[:, :, 5] * 4
        k[:, :, 70] = q[:, :, 40] * 2
        k[:, :, 600] = q[:, :, 800] * 4
    q, k, v = (t.to(TORCH_DT[dt]) for t in (q, k, v))
    if kind == "redo":
        # growth of exactly 126.5 octaves over the row's tile-0 maximum: past what the in-place repair may take on (2^120), short of f32's range
        c = D ** -0.5 * fo.LOG2E
        for h in range(0, H, 5):
            for row, kv in ((300, 520), (77, 333)):
                qr = q[0, h, row].double()
                ref = float((k[0, h, :64].double() @ qr).max()) * c
                k[0, h, kv] = (q[0, h, row].double() * ((ref + 126.5) / (float((qr ** 2).sum()) * c))).to(TORCH_DT[dt])
    q, k, v = (t.to(_dev()) for t in (q, k, v))
    plan = _plan(q, k, causal)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.heads_main == B * H, plan.as_dict()
    o, lse = _cabi_forward(q, k, v, causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * D ** -0.5          # every head against dense fp32 attention (coarse), sampled heads below
    if causal:
        s = s.masked_fill(torch.ones(N, N, dtype=torch.bool, device=_dev()).triu(1), float("-inf"))
    dense = torch.matmul(torch.softmax(s, -1), v.float())
    assert float((o.float() - dense).abs().max()) <= (0.1 if kind == "redo" else (4e-2 if dt else 2e-2)), float((o.float() - dense).abs().max())
    del s, dense
    o2, lse2 = _cabi_forward(q, k, v, causal)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)            # the redo is deterministic too
    # through the operator: non-causal calls hand over a workspace, and the eight items of the partly filled last round may run as KV-split parts
    # inside the persistent kernel (a part that is redone rewrites its f32 tile in the workspace); merged in f32, rounded once: one ulp of the plain call
    o_op = FlashAttentionFunction.apply(q, k, v, None, causal)
    torch.cuda.synchronize()
    assert torch.isfinite(o_op.float()).all()
    assert float((o_op.float() - o.float()).abs().max()) <= (2.0 ** -9 if dt == 0 else 2.0 ** -6) * max(1.0, float(o.float().abs().max()))
    for (b, h) in ((0, 0), (0, 5), (0, H // 2), (0, H - 1)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], dt, causal, plan=plan, head=b * H + h)
        o_true, lse_true = fo.fwd_numpy(q[sl].float().cpu().numpy(), k[sl].float().cpu().numpy(), v[sl].float().cpu().numpy(), causal)
        got = o[sl].float().cpu().numpy()
        folded = plan.contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q
        # (against float64 the folded contract's one rounding of Q shows in proportion to the logits: 60 .. 600 log2 units here — the same-contract
        #  oracle above holds the usual tolerance; at head dim 64 this test's logits are sqrt(2) larger)
        o_tol = ((6e-2 if kind == "redo" else 1e-2) if folded else 4e-3) if dt == 0 else 3e-2
        assert np.all(np.abs(got - o_true) <= o_tol + 4e-3 * np.abs(o_true)), (h, float(np.abs(got - o_true).max()))
        assert np.abs(lse[sl].cpu().numpy() - lse_true).max() <= ((0.3 if folded else 2e-2) if kind == "redo" else 2e-2), h


# ---------------------------------------------------------------- operator contract (reference quirks)

def test_return_contract_shapes_without_padding_copies():
    """6-tensor return [O_fwd, q, k, v, O, L] (reference: kernel_fp16.cu:875): O and L are padded to a multiple of Br
    rows with a zero tail and O_fwd is a view into O, like the reference's (kernel_fp16.cu:761, :793-796, :865-875);
    but Q/K/V are never copied — the kernels mask ragged N and D = 40 in-kernel — so the returned q/k/v ARE the inputs.
    L is log2-domain f32."""
    g = torch.Generator(device="cpu").manual_seed(9)
    q = torch.rand((2, 3, 100, 40), generator=g).half().to(_dev())
    k = torch.rand((2, 3, 77, 40), generator=g).half().to(_dev())
    v = torch.rand((2, 3, 77, 40), generator=g).half().to(_dev())
    O_fwd, q_pad, k_pad, v_pad, O, L = flash_attn_wmma.forward(q, k, v, 64, 128, False, 40 ** -0.5, False)
    torch.cuda.synchronize()
    assert O_fwd.shape == q.shape and O.shape == (2, 3, 128, 40) and L.shape == (2, 3, 128)
    assert q_pad.data_ptr() == q.data_ptr() and k_pad.data_ptr() == k.data_ptr() and v_pad.data_ptr() == v.data_ptr()
    assert O_fwd.data_ptr() == O.data_ptr() and L.dtype == torch.float32 and L.device == q.device
    assert float(O[:, :, 100:].abs().max()) == 0.0 and float(L[:, :, 100:].abs().max()) == 0.0
    o_ref_bits, lse_ref = fo.fwd_c(_bits(q), _bits(k), _bits(v), 0, False, flags=_oracle_flags(q, k, False, 40 ** -0.5))
    assert np.all(np.abs(O_fwd.float().cpu().numpy() - fo.bits_to_f32(o_ref_bits, 0)) <= 2e-3)
    assert np.abs(L[:, :, :100].cpu().numpy() - lse_ref).max() <= LSE_TOL
    # a head dim that is not a multiple of 8 is the one case that is still zero-padded (to the next multiple of 8)
    q5, k5, v5 = (t[..., :37].contiguous() for t in (q, k, v))
    O_fwd, q_pad, k_pad, v_pad, O, L = flash_attn_wmma.forward(q5, k5, v5, 64, 128, True, 37 ** -0.5, False)
    torch.cuda.synchronize()
    assert O_fwd.shape == q5.shape and O.shape == (2, 3, 128, 40) and q_pad.shape == (2, 3, 100, 40) and k_pad.shape == (2, 3, 77, 40)
    assert O_fwd.data_ptr() == O.data_ptr()
    o_ref_bits, lse_ref = fo.fwd_c(_bits(q5), _bits(k5), _bits(v5), 0, True, flags=_oracle_flags(q_pad, k_pad, True, 37 ** -0.5))
    assert np.all(np.abs(O_fwd.float().cpu().numpy() - fo.bits_to_f32(o_ref_bits, 0)) <= 2e-3)
    assert np.abs(L[:, :, :100].cpu().numpy() - lse_ref).max() <= LSE_TOL


@pytest.mark.parametrize("D", [8, 40, 72, 80, 96, 120, 160, 200, 248])
@pytest.mark.parametrize("dt", [0, 1])
def test_head_dims_masked_in_kernel(D, dt):
    """Every multiple of 8 up to 256 runs on the next kernel head dim with columns >= D masked in-kernel
    (SD1.5: 40 / 80 / 160), straight through the C-ABI with unpadded tensors."""
    g = torch.Generator(device="cpu").manual_seed(100 + D)
    B, H, N, Nkv = 1, 3, 150, 203
    q = torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    k = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    v = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    for causal in (False, True):
        o, lse = _cabi_forward(q, k, v, causal)
        _assert_close_to_oracle(o, lse, q, k, v, dt, causal)


@pytest.mark.parametrize("D", [16, 32, 48, 56, 88, 96, 104, 136, 160, 168, 192, 216, 224, 232, 256])
def test_trimmed_kernels_on_grids_of_256_row_workgroups(D):
    """Head dims below the kernel's HD run TRIMMED instantiations (only the MFMA k-steps and O column blocks that hold real
    columns, fwd_hip.cpp: D <= 32 / 48 on the 64 kernel, <= 96 on the 128 kernel, <= 160 / 192 / 224 on the 256 kernel whose second
    column half then runs ceil((D - 128) / 32) blocks) — here on a grid wide enough for the 8-wave 256-row shape (the small-grid
    128-row shape is test_head_dims_masked_in_kernel), at every boundary of the dispatch and one dim beyond it, ragged N.  On such
    grids head dims 129..224 (causal: ..256) run ONE pass over all columns (HDV = 256) instead of two column halves."""
    dt = (D // 8) & 1
    g = torch.Generator(device="cpu").manual_seed(500 + D)
    B, H, N, Nkv = 2, 13, 1000, 1111           # 26 heads x 4 q blocks = 104 workgroups > 3/8 of the CUs
    q = torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    k = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    v = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    for causal in (False, True):
        o, lse = _cabi_forward(q, k, v, causal)
        _assert_close_to_oracle(o, lse, q, k, v, dt, causal)


@pytest.mark.parametrize("D", [264, 320, 328, 384, 448, 456, 512])
@pytest.mark.parametrize("dt", [0, 1])
def test_head_dims_above_256(D, dt):
    """The reference accepts any head dim (it pads D to a multiple of 32 and switches to Br = 32 above 384,
    FlashAttn.py:65-67); the SD VAE attention block is one head of D = 512.  Head dims in (256, 512] run on the
    D = 512 forward kernel: four 128-column slabs of O per 128-row Q block, columns >= D masked in-kernel."""
    g = torch.Generator(device="cpu").manual_seed(300 + D)
    B, H, N, Nkv = 1, 2, 200, 333
    q = torch.randn((B, H, N, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    k = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    v = torch.randn((B, H, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev())
    for causal in (False, True):
        o, lse = _cabi_forward(q, k, v, causal)
        _assert_close_to_oracle(o, lse, q, k, v, dt, causal)
    o_op = FlashAttentionFunction.apply(q, k, v, None, False)                  # operator path, Br = 32 padding above 384
    torch.cuda.synchronize()
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (D ** -0.5)
    truth = torch.matmul(torch.softmax(s, -1), v.float())
    assert float((o_op.float() - truth).abs().max()) <= FLOOR[dt] * 2


def test_vae_shaped_attention_d512():
    """SD VAE mid-block attention at a 512 x 512 image: 1 head, D = 512, N = 64 * 64 tokens, through the sd_hook adapter."""
    from rocwmma_fattn.sd_hook import attention_bnhd
    g = torch.Generator(device="cpu").manual_seed(15)
    q, k, v = (torch.randn((1, 4096, 512), generator=g).half().to(_dev()) for _ in range(3))
    o = attention_bnhd(q, k, v, 1)
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    assert float((o.float() - ref).abs().max()) <= ATOL[0] + RTOL[0] * float(ref.abs().max())


def test_head_dim_160_runs_on_the_256_kernel():
    """SD1.5's deepest attention level has D = 160 (the reference pads D to a multiple of 32, kernel_fp16.cu:763)."""
    g = torch.Generator(device="cpu").manual_seed(10)
    q, k, v = (torch.randn((2, 8, 256, 160), generator=g).half().to(_dev()) for _ in range(3))
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    torch.cuda.synchronize()
    assert o.shape == q.shape
    o_ref_bits, _ = fo.fwd_c(_bits(q), _bits(k), _bits(v), 0, False)
    o_ref = fo.bits_to_f32(o_ref_bits, 0)
    assert np.all(np.abs(o.float().cpu().numpy() - o_ref) <= ATOL[0] + RTOL[0] * np.abs(o_ref))


def test_bnhd_layout_is_zero_copy_and_matches_bhnd():
    """BNHD_fmt=True (reference: kernel_fp16.cu:328-333, bench_with_sdpa_BNHD.py:106)."""
    g = torch.Generator(device="cpu").manual_seed(11)
    q, k, v = (torch.rand((2, 4, 300, 64), generator=g).half().to(_dev()) for _ in range(3))
    o_bhnd = FlashAttentionFunction.apply(q, k, v, None, True)
    qn, kn, vn = (t.transpose(1, 2).contiguous() for t in (q, k, v))     # [B,N,H,D]
    o_bnhd = FlashAttentionFunction.apply(qn, kn, vn, None, True, None, True)
    torch.cuda.synchronize()
    assert o_bnhd.shape == qn.shape
    assert torch.equal(o_bnhd.transpose(1, 2), o_bhnd)


def test_sd_hook_bnhd_adapter_matches_sdpa():
    """The [B, N, H*D] call shape of ComfyUI / sd-webui (reference README.md:35-37) through the zero-copy BNHD path:
    SD1.5 self-attention (D=40 and D=160) and cross-attention with 77 text tokens."""
    from rocwmma_fattn.sd_hook import attention_bnhd
    g = torch.Generator(device="cpu").manual_seed(14)
    for (b, nq, nkv, heads, d) in ((2, 1024, 1024, 8, 40), (2, 256, 77, 8, 160), (1, 4096, 4096, 10, 64)):
        q = torch.randn((b, nq, heads * d), generator=g).half().to(_dev())
        k = torch.randn((b, nkv, heads * d), generator=g).half().to(_dev())
        v = torch.randn((b, nkv, heads * d), generator=g).half().to(_dev())
        o = attention_bnhd(q, k, v, heads)
        torch.cuda.synchronize()
        assert o.shape == q.shape and o.dtype == q.dtype
        split = lambda t: t.reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2).float()  # noqa: E731
        ref = torch.nn.functional.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(b, nq, heads * d)
        assert float((o.float() - ref).abs().max()) <= ATOL[0] + RTOL[0] * float(ref.abs().max())
    # head dims above the largest kernel go to the host's own attention (masks no longer do: tests/test_bias_gpu.py)
    wide = torch.zeros((1, 16, 1024), device=_dev(), dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        attention_bnhd(wide, wide, wide, 1)
    sentinel = object()
    assert attention_bnhd(wide, wide, wide, 1, fallback=lambda *a: sentinel) is sentinel
    with pytest.raises(RuntimeError, match="broadcast"):
        attention_bnhd(q, k, v, heads, mask=torch.ones(3, device=_dev()).bool())


def test_non_half_inputs_run_as_bf16_like_the_reference():
    """host.cpp:42-45: any other dtype is cast to bf16 and the bf16 result is returned."""
    g = torch.Generator(device="cpu").manual_seed(12)
    q, k, v = (torch.rand((1, 2, 64, 64), generator=g).to(_dev()) for _ in range(3))
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    assert o.dtype == torch.bfloat16
    o2 = FlashAttentionFunction.apply(q.bfloat16(), k.bfloat16(), v.bfloat16(), None, False)
    assert torch.equal(o, o2)


def test_requires_grad_saves_the_backward_tensors():
    q, k, v = (torch.rand((1, 2, 64, 64), device=_dev(), dtype=torch.float16, requires_grad=True) for _ in range(3))
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    assert o.requires_grad
    o.backward(torch.ones_like(o))          # backward parity itself: tests/test_backward_gpu.py
    assert q.grad is not None and k.grad.shape == k.shape and v.grad.dtype == v.dtype


def test_launch_is_on_the_callers_stream_and_device():
    s = torch.cuda.Stream()
    q, k, v = (torch.rand((1, 4, 512, 128), device=_dev(), dtype=torch.float16) for _ in range(3))
    ref = FlashAttentionFunction.apply(q, k, v, None, False)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        o = FlashAttentionFunction.apply(q, k, v, None, False)
    s.synchronize()
    assert torch.equal(o, ref)


def test_operator_is_graph_capturable():
    """Small (cross-attention-sized) calls are host-bound; the C-ABI launch only enqueues on the caller's stream, so a
    sequence of operator calls can be captured into a HIP graph and replayed without any host work."""
    g = torch.Generator(device="cpu").manual_seed(12)
    q = torch.randn((2, 4, 512, 64), generator=g).half().to(_dev())
    k = torch.randn((2, 4, 77, 64), generator=g).half().to(_dev())
    v = torch.randn((2, 4, 77, 64), generator=g).half().to(_dev())
    ref = FlashAttentionFunction.apply(q, k, v, None, False)          # warm-up outside the capture (library load, LDS opt-in)
    ref256 = FlashAttentionFunction.apply(q.repeat(1, 1, 1, 4), k.repeat(1, 1, 1, 4), v.repeat(1, 1, 1, 4), None, False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            o1 = FlashAttentionFunction.apply(q, k, v, None, False)
            o2 = FlashAttentionFunction.apply(o1, k, v, None, False)   # a dependent second call inside the same graph
    torch.cuda.current_stream().wait_stream(side)
    o1.zero_()
    o2.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o1, ref)
    assert torch.equal(o2, FlashAttentionFunction.apply(ref, k, v, None, False))
    q.copy_(q * 0.5)                                                   # new inputs in the captured buffers, replay again
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o1, FlashAttentionFunction.apply(q, k, v, None, False))
    del ref256


# ---------------------------------------------------------------- BASELINE.json configurations (full size)

CONFIGS = {
    "c2": (2, 16, 4096, 128, 0, False),
    "c3": (2, 16, 4096, 128, 1, True),
    "c4": (1, 32, 8192, 128, 0, True),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_config_properties(name):
    """At full size the oracle checks a SAMPLE of heads; the whole tensor is covered by properties:
    (a) determinism and batch/head independence: recomputing a head slice alone is bit-identical;
    (b) row-block independence: a Q row subset gives bit-identical rows;
    (c) rows are convex combinations: min(V) <= O <= max(V) per head and column, and constant V
        returns that constant; (d) causal row 0 equals V[0] exactly;
    (e) LSE consistency: lse equals the dense fp32 log2-sum-exp2 on sampled rows."""
    B, H, N, D, dt, causal = CONFIGS[name]
    g = torch.Generator(device=_dev()).manual_seed(1234)
    q, k, v = (torch.rand((B, H, N, D), generator=g, device=_dev(), dtype=torch.float32).to(TORCH_DT[dt]) for _ in range(3))
    o, lse = _cabi_forward(q, k, v, causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    # the BASELINE configurations run the hand-scheduled body (fp16: folded scale; bf16: f32 scale), one launch
    plan = _plan(q, k, causal)
    assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.heads_main == B * H
    assert plan.contract == (_fa2_lib.FA2_CONTRACT_PRESCALE_Q if dt == 0 else 0) | _fa2_lib.FA2_CONTRACT_LSUM_P16
    # sampled heads against the oracle (first, last, one in the middle)
    for (b, h) in {(0, 0), (B - 1, H - 1), (B // 2, H // 3)}:
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], dt, causal, plan=plan, head=b * H + h)
    # (a) head slice recomputed alone: bit-identical as long as the library picks the same kernel shape for the slice
    #     (it switches to 128-row workgroups when B*H*ceil(Nq/256) <= 96: 8 heads keep this slice above that)
    o_h, lse_h = _cabi_forward(q[:, 3:11].contiguous(), k[:, 3:11].contiguous(), v[:, 3:11].contiguous(), causal)
    assert torch.equal(o_h, o[:, 3:11]) and torch.equal(lse_h, lse[:, 3:11])
    #     ... and a slice small enough to run on the other workgroup shape agrees to rounding (fp16: the 128-row HIP kernel scales the f32
    #     product, the hand-scheduled body of the full launch folds the scale into Q: LSE_TOL instead of 1e-4)
    o_s, lse_s = _cabi_forward(q[:, 3:4].contiguous(), k[:, 3:4].contiguous(), v[:, 3:4].contiguous(), causal)
    assert float((o_s.float() - o[:, 3:4].float()).abs().max()) <= ATOL[dt]
    #     bf16: the full launch adds the rounded P into its row sums, on the matrix pipe (FA2_CONTRACT_LSUM_P16), the 128-row kernel the f32 P)
    assert float((lse_s - lse[:, 3:4]).abs().max()) <= (LSE_TOL if dt == 0 else LSE_TOL_P16_BF16)
    # (b) first rows recomputed alone (top-left causal alignment keeps the first rows unchanged)
    o_r, _ = _cabi_forward(q[:, :, :1024].contiguous(), k, v, causal)
    if causal:
        # (2048 rows: a causal launch of fewer than 1792 keys runs on the compiler-scheduled kernel — host.cpp, asm_kv_len_ok — which agrees
        # with the hand-scheduled one to rounding, not bit for bit)
        o_r2, _ = _cabi_forward(q[:, :, :2048].contiguous(), k[:, :, :2048].contiguous(), v[:, :, :2048].contiguous(), True)
        assert torch.equal(o_r2, o[:, :, :2048])
        o_r3, _ = _cabi_forward(q[:, :, :1024].contiguous(), k[:, :, :1024].contiguous(), v[:, :, :1024].contiguous(), True)
        assert float((o_r3.float() - o[:, :, :1024].float()).abs().max()) <= ATOL[dt]
    else:
        assert torch.equal(o_r, o[:, :, :1024])
    # (c) convexity
    vmin = v.float().amin(dim=2, keepdim=True)
    vmax = v.float().amax(dim=2, keepdim=True)
    ulp = 2.0 ** (-10 if dt == 0 else -7)
    assert bool(((o.float() >= vmin - ulp) & (o.float() <= vmax + ulp)).all())
    o_c, _ = _cabi_forward(q[:1, :2].contiguous(), k[:1, :2].contiguous(), torch.full_like(v[:1, :2], 0.75), causal)
    assert float((o_c.float() - 0.75).abs().max()) <= ulp
    # (d) causal first row
    if causal:
        assert torch.equal(o[:, :, 0], v[:, :, 0])
    # (e) LSE on sampled rows
    rows = torch.tensor([0, 1, 63, 64, 255, 256, N // 2 + 5, N - 1], device=_dev())
    s = torch.matmul(q[:, :, rows].float(), k.float().transpose(-1, -2)) * (D ** -0.5 * fo.LOG2E)
    if causal:
        col = torch.arange(N, device=_dev())
        s = s.masked_fill(col[None, None, None, :] > rows[None, None, :, None], float("-inf"))
    lse_rows = torch.logsumexp(s * 0.6931471805599453, dim=-1) * fo.LOG2E
    assert float((lse[:, :, rows] - lse_rows).abs().max()) <= LSE_TRUTH_TOL[dt]


def test_config5_shard_equals_slice_of_global_batch():
    """BASELINE config 5 splits B=64 over ranks; a rank's slab must equal the slice of the unsplit
    computation bit for bit (here: a 2-way split of B=4 at the full H16 N4096 D128 shape)."""
    from rocwmma_fattn.shard import shard_bounds
    g = torch.Generator(device=_dev()).manual_seed(77)
    q, k, v = (torch.rand((4, 16, 4096, 128), generator=g, device=_dev(), dtype=torch.float32).half() for _ in range(3))
    full = FlashAttentionFunction.apply(q, k, v, None, False)
    for rank in range(2):
        lo, hi = shard_bounds(4, 2, rank)
        part = FlashAttentionFunction.apply(q[lo:hi], k[lo:hi], v[lo:hi], None, False)
        assert torch.equal(part, full[lo:hi])


def test_config5_full_batch_and_per_rank_shard():
    """BASELINE config 5 at its real sizes on one GPU: the unsplit B=64 H16 N4096 D128 batch (4 x 1 GiB of tensors) and
    the B=8 slab one of 8 ranks owns.  The slab is bit-identical to its slice of the unsplit run, sampled heads match
    the oracle, and the whole output obeys the convexity property."""
    from rocwmma_fattn.shard import shard_bounds
    B, H, N, D = 64, 16, 4096, 128
    g = torch.Generator(device=_dev()).manual_seed(1239)
    q, k, v = (torch.rand((B, H, N, D), generator=g, device=_dev(), dtype=torch.float16) for _ in range(3))
    full = FlashAttentionFunction.apply(q, k, v, None, False)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    for rank in (0, 5, 7):
        lo, hi = shard_bounds(B, 8, rank)
        assert hi - lo == 8
        part = FlashAttentionFunction.apply(q[lo:hi], k[lo:hi], v[lo:hi], None, False)
        assert torch.equal(part, full[lo:hi])
    for (b, h) in ((0, 0), (63, 15), (37, 6)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        o_ref_bits, _ = fo.fwd_c(_bits(q[sl]), _bits(k[sl]), _bits(v[sl]), 0, False)
        o_ref = fo.bits_to_f32(o_ref_bits, 0)
        assert np.all(np.abs(full[sl].float().cpu().numpy() - o_ref) <= ATOL[0] + RTOL[0] * np.abs(o_ref))
    for b0 in range(0, B, 16):       # convexity, in slabs to bound the fp32 temporaries
        sl = slice(b0, b0 + 16)
        vmin = v[sl].amin(dim=2, keepdim=True).float() - 2.0 ** -10
        vmax = v[sl].amax(dim=2, keepdim=True).float() + 2.0 ** -10
        of = full[sl].float()
        assert bool(((of >= vmin) & (of <= vmax)).all())


def test_bench_two_ranks_on_one_gpu_dry_run():
    """The multi-rank control flow of bench.py (process group, barriers, max-over-ranks reduction, rank-0 JSON line)
    executed with the real operator: two ranks, both on cuda:0, gloo for the host-side collectives (NCCL refuses two
    ranks on one device).  The numbers are not a scaling claim; the line must parse and carry the contract's keys."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    for workload in ("c5",):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--workload", workload, "--backend", "gloo", "--same-device", "--steady-launches", "0", "--collectives"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, res.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["unit"] == "TFLOPS"
        assert rec["scaling"] == ("strong" if workload == "c5" else "weak")
        assert rec["config"]["global_batch"] == (64 if workload == "c5" else 4)
        assert "roofline" in rec and rec["check"]["max_abs_err_vs_dense_fp32"] <= rec["check"]["tol"]
        assert rec["collectives"]["backend"] == "gloo" and rec["collectives"]["scatter_qkv_ms"] > 0     # host-staged edge transfers ran (and were verified)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2 ...` with NO torch.distributed.run around it (the command shape the driver records): bench.py
    re-executes itself under the launcher on a free loopback port, the two ranks (both on cuda:0 over gloo here — NCCL refuses two
    ranks on one device) run and rank 0 prints the one JSON line with n_gpus = 2 and the ranks the group saw.  With no --workload an
    N > 1 run measures BASELINE.json's multi-GPU configuration — c5, B = 64 split over the ranks (strong scaling) — as `value` and
    carries the c2 weak-scaling figure under `weak_c2`; the line states the cold (pre-settle) number and every launch that preceded the timed region."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--same-device", "--steady-launches", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0
    assert rec["config"]["global_batch"] == 64 and rec["scaling"] == "strong" and rec["config"]["workload"].startswith("c5: B32 ")
    assert rec["weak_c2"]["global_batch"] == 4 and rec["weak_c2"]["scaling"] == "weak" and rec["weak_c2"]["value"] > 0
    assert rec["cold"]["value"] > 0 and rec["warmup_effective"] >= 1 + (1 + 3) + rec["settle"]["launches"] + 1
    assert rec["dist"]["world_size"] == 2 and sorted(r["rank"] for r in rec["dist"]["ranks"]) == [0, 1]


def test_bench_runs_the_rccl_path_with_one_rank():
    """`bench.py --force-dist --collectives` on the one GPU of this box: init_process_group("nccl", device_id=...), the device
    barriers, the MAX all-reduce of the timings and scatter_batch / gather_batch on DEVICE tensors all go through RCCL with a
    world of one rank — the calls the 2/4/8-GPU runs of the driver make, executed here so that they are not first run there."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c2",
           "--force-dist", "--collectives", "--steady-launches", "0", "--no-cpu-baseline", "--no-backward"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["dist"]["backend"] == "nccl" and "dry_run" not in rec
    assert rec["collectives"]["backend"] == "nccl" and rec["collectives"]["world"] == 1
    assert rec["collectives"]["scatter_qkv_ms"] > 0 and rec["collectives"]["gather_o_ms"] > 0


@pytest.mark.parametrize("shape", [(2, 10, 4096, 64), (2, 17, 4096, 64), (1, 24, 4096, 128), (3, 7, 2816, 128), (1, 20, 3500, 80)])
def test_tail_split_launches_cover_every_head(shape):
    """Grids whose last round of 256-row workgroups would be at most half full are issued as two launches (whole heads: the
    full rounds as they are, the remaining heads as 128-row workgroups — host.cpp tail_split_heads; B2 H17 N4096 D64 = 544
    workgroups is such a grid), and between one and one and a half rounds at head dims <= 64 the whole grid runs as 128-row
    workgroups (short_second_round: B2 H10 N4096 D64 = 320).  Every head must come out right, in particular the ones of the second launch."""
    B, H, N, D = shape
    g = torch.Generator(device="cpu").manual_seed(41 + H)
    q, k, v = (torch.randn((B, H, N, D), generator=g).half().to(_dev()) for _ in range(3))
    o, lse = _cabi_forward(q, k, v, False)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    plan = _plan(q, k, False)
    if shape == (2, 17, 4096, 64):          # 544 workgroups: two full rounds of the hand-scheduled body + the last 2 heads as 128-row workgroups
        assert plan.heads_main == 32 and plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.kernel_tail == _fa2_lib.FA2_KERNEL_HIP_128
        assert plan.contract == _fa2_lib.FA2_CONTRACT_PRESCALE_Q | _fa2_lib.FA2_CONTRACT_LSUM_P16 and plan.contract_tail == 0
    for (b, h) in {(0, 0), (B - 1, H - 1), (B - 1, H - 2), (B // 2, H // 2)}:
        sl = (slice(b, b + 1), slice(h, h + 1))
        _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], 0, False, plan=plan, head=b * H + h)
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (D ** -0.5)
    truth = torch.matmul(torch.softmax(s, -1), v.float())
    assert float((o.float() - truth).abs().max()) <= FLOOR[0] * 2


def _expand_and_call(fn):
    """Call a parametrized test function for the cartesian product of its @pytest.mark.parametrize lists (and every golden case
    for a `golden` argument) — used to re-run whole groups of tests under another library option in THIS process."""
    import inspect
    import itertools
    from conftest import GOLDEN_CASES, load_golden
    names, lists = [], []
    for m in getattr(fn, "pytestmark", []):
        if m.name == "parametrize":
            argn = [a.strip() for a in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
            names.append(argn)
            lists.append(list(m.args[1]))
    params = inspect.signature(fn).parameters
    n = 0
    for combo in itertools.product(*lists):
        kw = {}
        for argn, val in zip(names, combo):
            val = getattr(val, "values", val)      # pytest.param(...)
            if len(argn) == 1:
                kw[argn[0]] = val[0] if isinstance(val, tuple) and not isinstance(combo, tuple) else val
            else:
                kw.update(dict(zip(argn, val)))
        if "golden" in params:
            for name in GOLDEN_CASES:
                fn(golden=load_golden(name), **kw)
                n += 1
        else:
            fn(**kw)
            n += 1
    return n


def test_d128_asm_kernel_on_small_and_ragged_grids():
    """Grids of fewer than 97 workgroups run on the 128-row HIP kernel, so the seeded / golden / ragged / poisoned-tail cases
    above do not reach the hand-scheduled D = 128 kernel.  This re-runs them with the library option rows = 256 (fa2_set_option),
    which pins the 256-row shapes — i.e. every head-dim-128 case, ragged Nq / Nkv, cross-attention, causal, BNHD strides,
    NaN-poisoned tails, goes through the asm block on the GPU."""
    from rocwmma_fattn import _fa2_lib
    n = 0
    with _fa2_lib.options(rows=256):
        for fn in (test_golden_fixtures_through_cabi, test_golden_fixtures_through_operator, test_seeded_shapes_against_oracle,
                   test_explicit_and_negative_scale, test_scale_zero_is_the_uniform_softmax, test_ragged_tail_ignores_memory_past_nkv,
                   test_large_logits_and_forced_rescale, test_bnhd_d128_zero_copy, test_reference_precision_shape):
            n += _expand_and_call(fn)
    assert n >= 40, n
    assert _fa2_lib.load().fa2_get_option(b"rows") == 0


PERSISTENT_SHAPES = [
    # B, H, Nq, Nkv, BNHD: more items than CUs, so workgroups run several items and fetch across the item seam
    (3, 11, 2304, 2304, False),     # 297 items, head count not a multiple of 8 (the plain item -> head mapping), 36 KV tiles
    (5, 16, 1100, 1984, False),     # 400 items, ragged last q block (clamped rows), 31 KV tiles (ring parities flip at the seam)
    (40, 8, 512, 64, False),        # 640 items of ONE KV tile: the staging body is a head body
    (24, 8, 600, 100, True),        # 576 items of two tiles, ragged tail, BNHD strides
    (2, 16, 4096, 4096, False),     # config 2 itself: two items per workgroup
    (1, 9, 1280, 1280, False),      # five q blocks per head: a causal head's middle block is a unit of its own
]


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", PERSISTENT_SHAPES)
def test_d128_persistent_workgroups_match_one_workgroup_per_item(shape, causal):
    """D = 128 launches run persistent workgroups (grid = CUs; the last two bodies of an item fetch the next item's Q fragments and
    first K / V tiles; causal launches hand a workgroup pairs of q blocks (nqblk-1-i, i) of one head).  The arithmetic of an item does
    not depend on how it was scheduled, so the outputs must be BIT-IDENTICAL to a launch of one workgroup per item (library option
    persist = 0), and close to dense fp32."""
    from rocwmma_fattn import _fa2_lib
    B, H, Nq, Nkv, bnhd = shape
    if causal and Nkv < Nq:
        Nkv = Nq                      # (top-left aligned causal mask: rows past Nkv would see every key; keep the square case)
    g = torch.Generator(device="cpu").manual_seed(77)
    q = torch.randn((B, H, Nq, 128), generator=g).half().to(_dev())
    k = torch.randn((B, H, Nkv, 128), generator=g).half().to(_dev())
    v = torch.randn((B, H, Nkv, 128), generator=g).half().to(_dev())
    qq, kk, vv = (t.transpose(1, 2).contiguous() for t in (q, k, v)) if bnhd else (q, k, v)
    outs = {}
    for mode, persist in (("persist", 1), ("single", 0)):
        with _fa2_lib.options(rows=256, persist=persist):
            outs[mode] = FlashAttentionFunction.apply(qq, kk, vv, None, causal, None, bool(bnhd))
        torch.cuda.synchronize()
    assert torch.equal(outs["persist"], outs["single"])
    got = outs["persist"].float()
    if bnhd:
        got = got.transpose(1, 2)
    worst = 0.0
    for b in range(0, B, max(1, B // 3)):
        s = torch.matmul(q[b].float(), k[b].float().transpose(-1, -2)) * (128 ** -0.5)
        if causal:
            s = s.masked_fill(torch.ones(Nq, Nkv, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
        truth = torch.matmul(torch.softmax(s, -1), v[b].float())
        worst = max(worst, float((got[b] - truth).abs().max()))
    assert worst <= 2e-3, worst


def test_bnhd_d128_zero_copy():
    """[B, N, H, D] storage at head dim 128 (row stride H*D): the hand-scheduled kernel takes the same strides."""
    g = torch.Generator(device="cpu").manual_seed(16)
    q, k, v = (torch.rand((2, 16, 1100, 128), generator=g).half().to(_dev()) for _ in range(3))
    for causal in (False, True):
        o_bhnd = FlashAttentionFunction.apply(q, k, v, None, causal)
        qn, kn, vn = (t.transpose(1, 2).contiguous() for t in (q, k, v))     # [B,N,H,D]
        o_bnhd = FlashAttentionFunction.apply(qn, kn, vn, None, causal, None, True)
        torch.cuda.synchronize()
        assert torch.equal(o_bnhd.transpose(1, 2), o_bhnd)
        o_ref_bits, _ = fo.fwd_c(_bits(q[:1, :2]), _bits(k[:1, :2]), _bits(v[:1, :2]), 0, causal)
        o_ref = fo.bits_to_f32(o_ref_bits, 0)
        assert np.all(np.abs(o_bhnd[:1, :2].float().cpu().numpy() - o_ref) <= ATOL[0] + RTOL[0] * np.abs(o_ref))


def test_compiled_front_end_matches_python():
    """csrc/frontend.cpp is the Python forward of the reference's module transcribed to C++: same six tensors, bit for bit,
    for aligned, N-padded, D-padded (D = 37), BNHD, causal and non-half inputs."""
    from rocwmma_fattn.FlashAttn import _frontend
    fe = _frontend()
    if fe is None:
        pytest.skip("compiled front end not built")
    g = torch.Generator(device="cpu").manual_seed(23)
    cases = [((2, 3, 100, 40), (2, 3, 77, 40), torch.float16, False, False), ((1, 2, 256, 128), (1, 2, 256, 128), torch.bfloat16, True, False),
             ((2, 3, 100, 37), (2, 3, 77, 37), torch.float16, True, False), ((2, 130, 4, 64), (2, 90, 4, 64), torch.float16, False, True),
             ((1, 2, 64, 64), (1, 2, 64, 64), torch.float32, False, False), ((1, 1, 70, 320), (1, 1, 50, 320), torch.float16, False, False)]
    for qs, ks, dt, causal, bnhd in cases:
        q = torch.randn(qs, generator=g).to(dt).to(_dev())
        k = torch.randn(ks, generator=g).to(dt).to(_dev())
        v = torch.randn(ks, generator=g).to(dt).to(_dev())
        d = qs[3]
        br = 32 if d > 384 else 64
        a = fe.forward(q, k, v, br, 128, causal, d ** -0.5, bnhd)
        b = flash_attn_wmma.forward_py(q, k, v, br, 128, causal, d ** -0.5, bnhd)
        torch.cuda.synchronize()
        assert len(a) == len(b) == 6
        for x, y in zip(a, b):
            assert x.shape == y.shape and x.dtype == y.dtype and x.device == y.device and x.stride() == y.stride()
            assert torch.equal(x, y)
        assert a[0].data_ptr() == a[4].data_ptr()
    with pytest.raises(RuntimeError):
        fe.forward(q.cpu(), k.cpu(), v.cpu(), 64, 128, False, 0.1, False)


def test_head_dim_64_fp16_folded_scale_contract_on_large_logits():
    """fp16 launches of 256-row workgroups at head dim 64 run the hand-scheduled body that folds scale * log2(e) into Q (rounded once to fp16:
    the scaling contract of the reference's own oracle, pure_torch_ver.py:61; fa2_fwd_prescales_q(64, scale) == 1).  Against the oracle under
    THAT contract the usual tolerance holds whatever the logits; against float64 attention the rounding of Q shows in proportion to the logits:
    ~2e-4 of log2 LSE on N(0,1) inputs (logits of a few units), ~1e-2 in O at logits of several hundred — the documented price of the fold."""
    lib = _fa2_lib.load(build_if_missing=False)
    assert lib.fa2_fwd_prescales_q(64, 0.125) == 1
    B, H, N, D = 2, 16, 2048, 64                         # 256 workgroups of 256 rows: the hand-scheduled body
    pl = _fa2_lib.fwd_plan(torch.empty((B, H, N, D), dtype=torch.float16, device="meta"), torch.empty((B, H, N, D), dtype=torch.float16, device="meta"), False)
    assert pl.kernel == _fa2_lib.FA2_KERNEL_ASM and pl.contract == _fa2_lib.FA2_CONTRACT_PRESCALE_Q | _fa2_lib.FA2_CONTRACT_LSUM_P16
    for amp, o_tol, lse_tol in ((1.0, 1e-3, 1e-3), (3.0, 3e-2, 0.3)):
        g = torch.Generator(device="cpu").manual_seed(int(amp * 10))
        q = (torch.randn((B, H, N, D), generator=g) * amp).half().to(_dev())
        k = (torch.randn((B, H, N, D), generator=g) * amp).half().to(_dev())
        v = torch.randn((B, H, N, D), generator=g).half().to(_dev())
        o, lse = _cabi_forward(q, k, v, False)
        for (b, h) in ((0, 0), (1, 15)):
            sl = (slice(b, b + 1), slice(h, h + 1))
            o_ref_bits, lse_ref = fo.fwd_c(_bits(q[sl]), _bits(k[sl]), _bits(v[sl]), 0, False, flags=fo.PRESCALE_Q)
            o_ref = fo.bits_to_f32(o_ref_bits, 0)
            got = o[sl].float().cpu().numpy()
            assert np.all(np.abs(got - o_ref) <= ATOL[0] + RTOL[0] * np.abs(o_ref)), np.abs(got - o_ref).max()
            assert np.abs(lse[sl].cpu().numpy() - lse_ref).max() <= LSE_TOL
            o_true, lse_true = fo.fwd_numpy(q[sl].float().cpu().numpy(), k[sl].float().cpu().numpy(), v[sl].float().cpu().numpy(), False)
            assert np.all(np.abs(got - o_true) <= o_tol + 4e-3 * np.abs(o_true)), (amp, np.abs(got - o_true).max())
            assert np.abs(lse[sl].cpu().numpy() - lse_true).max() <= lse_tol, (amp, np.abs(lse[sl].cpu().numpy() - lse_true).max())


@pytest.mark.parametrize("D,dt", [(128, 0), (128, 1), (64, 0)])
def test_padded_row_pitch_on_grids_wide_enough_for_the_hand_scheduled_kernels(D, dt):
    """Column slices of wider matrices (row pitch D + 8 / D + 16 elements) on a grid of 256 workgroups of 256 rows.  The hand-scheduled bodies
    derive a wave's further LDS-DMA source offsets by flipping granule bits of the first, which only equals re-swizzling when the staged matrix's
    row pitch is a multiple of a tile row; host.cpp (asm_pitch_ok) sends other pitches to the HIP kernels.  The randomised sweep found the
    missing check once it drew grids this wide (profiles/fuzz_runs.md, row r06_fuzz_parity_seed5: LSE off by 2e-2 with a padded K).  Forward against the
    oracle and dense fp32, backward against float64 autograd, with every operand padded in turn."""
    from conftest import GRAD_TOL
    B, H, N = 2, 16, 2048
    g = torch.Generator(device="cpu").manual_seed(77 + D + dt)
    for pads in ((0, 8, 0, 0), (8, 0, 16, 0), (8, 8, 8, 8)):          # pitch padding of q, k, v, dO in elements
        wide = lambda p: torch.randn((B, H, N, D + p), generator=g).to(TORCH_DT[dt]).to(_dev())[..., :D]  # noqa: E731
        q, k, v, do = (wide(p) for p in pads)
        assert k.stride(2) == D + pads[1]
        qa, ka, va = (t.detach().requires_grad_(True) for t in (q, k, v))
        o = FlashAttentionFunction.apply(qa, ka, va, None, False)
        o.backward(do)
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (D ** -0.5)
        truth = torch.matmul(torch.softmax(s, -1), v.float())
        assert float((o.float() - truth).abs().max()) <= FLOOR[dt] * 2, pads
        lse = flash_attn_wmma.forward(q, k, v, 64, 128, False, D ** -0.5, False)[5]
        assert float((lse - torch.logsumexp(s, -1) * fo.LOG2E).abs().max()) <= LSE_TRUTH_TOL[dt], pads
        plan = _plan(q, k, False)
        if pads[1]:                          # a padded K pitch: never the hand-scheduled body (asm_pitch_ok)
            assert plan.kernel == _fa2_lib.FA2_KERNEL_HIP_256 and plan.contract == 0
        for (b, h) in ((0, 0), (1, 15)):
            sl = (slice(b, b + 1), slice(h, h + 1))
            _assert_close_to_oracle(o[sl].detach(), lse[sl], q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), dt, False, plan=plan, head=b * H + h)
            qd, kd, vd = (t[sl].double().requires_grad_(True) for t in (q, k, v))
            torch.nn.functional.scaled_dot_product_attention(qd, kd, vd).backward(do[sl].double())
            for name, got, want in (("dq", qa.grad, qd.grad), ("dk", ka.grad, kd.grad), ("dv", va.grad, vd.grad)):
                err = float((got[sl].double() - want).abs().max())
                assert err <= GRAD_TOL[dt] * max(1.0, float(want.abs().max())), (pads, name, err)


def test_compiled_backward_matches_python():
    """csrc/frontend.cpp::backward is _FlashAttnWmma.backward_py transcribed to C++ (unmasked calls): the same three tensors, bit for bit — aligned,
    ragged / D-padded (D = 37), BNHD, causal, a split last round (the workspace comes from torch's allocator on both sides)."""
    from rocwmma_fattn.FlashAttn import _frontend
    fe = _frontend()
    if fe is None or not hasattr(fe, "backward"):
        pytest.skip("compiled front end not built")
    g = torch.Generator(device="cpu").manual_seed(29)
    cases = [((2, 3, 100, 40), (2, 3, 77, 40), torch.float16, False, False), ((1, 2, 256, 128), (1, 2, 256, 128), torch.bfloat16, True, False),
             ((2, 3, 100, 37), (2, 3, 77, 37), torch.float16, True, False), ((2, 130, 4, 64), (2, 90, 4, 64), torch.float16, False, True),
             ((2, 10, 4096, 64), (2, 10, 4096, 64), torch.float16, False, False), ((2, 10, 2048, 64), (2, 10, 77, 64), torch.float16, False, False)]
    for qs, ks, dt, causal, bnhd in cases:
        q, k, v = (torch.randn(s, generator=g).to(dt).to(_dev()) for s in (qs, ks, ks))
        do = torch.randn(qs, generator=g).to(dt).to(_dev())
        d = qs[3]
        n, nkv = (qs[1], ks[1]) if bnhd else (qs[2], ks[2])
        o_fwd, qp, kp, vp, O, L = flash_attn_wmma.forward(q, k, v, 64, 128, causal, d ** -0.5, bnhd)
        a = fe.backward(qp, kp, vp, O, do, L, n, nkv, d, 128, 128, causal, d ** -0.5, bnhd)
        b = flash_attn_wmma.backward_py(qp, kp, vp, O, do, L, n, nkv, d, 128, 128, causal, d ** -0.5, bnhd)
        torch.cuda.synchronize()
        assert len(a) == len(b) == 3
        for x, y in zip(a, b):
            assert x.shape == y.shape and x.dtype == y.dtype and x.stride() == y.stride()
            assert torch.equal(x, y)


def test_compiled_autograd_node_matches_python_class():
    """FlashAttentionFunction.apply hands calls that need gradients to the C++ autograd node of the compiled front end (csrc/frontend.cpp::AttentionNode:
    the engine's device thread runs its backward without the GIL); the Python class (reference FlashAttn.py:45-92) stays as the fallback.  Same output,
    same gradients bit for bit — aligned, D-padded, BNHD, causal, explicit scale, fp32 inputs (run and returned as bf16, host.cpp:42-45), k / v without
    gradient — and calls where q needs no gradient keep the Python class (the reference saves its tensors only then, FlashAttn.py:70)."""
    from rocwmma_fattn import FlashAttn as FA
    fe = FA._frontend()
    if fe is None or not hasattr(fe, "attention"):
        pytest.skip("compiled front end not built")
    g = torch.Generator(device="cpu").manual_seed(31)
    cases = [((2, 3, 100, 40), (2, 3, 77, 40), torch.float16, False, False, None), ((1, 2, 256, 128), (1, 2, 256, 128), torch.bfloat16, True, False, None),
             ((2, 3, 100, 37), (2, 3, 77, 37), torch.float16, True, False, 0.3), ((2, 130, 4, 64), (2, 90, 4, 64), torch.float16, False, True, None),
             ((2, 10, 2048, 64), (2, 10, 77, 64), torch.float16, False, False, None), ((1, 2, 64, 64), (1, 2, 64, 64), torch.float32, False, False, None)]
    for qs, ks, dt, causal, bnhd, scale in cases:
        q, k, v = (torch.randn(s, generator=g).to(dt).to(_dev()).requires_grad_(True) for s in (qs, ks, ks))
        do = torch.randn(qs, generator=g).to(torch.bfloat16 if dt == torch.float32 else dt).to(_dev())
        res = []
        for fn in (FA._autograd_apply, FlashAttentionFunction.apply):
            q.grad = k.grad = v.grad = None
            o = fn(q, k, v, None, causal, scale, bnhd)
            o.backward(do)
            torch.cuda.synchronize()
            res.append((type(o.grad_fn).__name__, o.detach(), q.grad, k.grad, v.grad))
        assert res[0][0] == "FlashAttentionFunctionBackward" and res[1][0] != res[0][0], (res[0][0], res[1][0])
        for x, y in zip(res[0][1:], res[1][1:]):
            assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y)
    q, k, v = (torch.randn((1, 2, 128, 64), generator=g).half().to(_dev()) for _ in range(3))
    q.requires_grad_(True)
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    o.backward(torch.ones_like(o))
    assert q.grad is not None and k.grad is None and torch.isfinite(q.grad.float()).all()
    q2 = q.detach()
    k.requires_grad_(True)
    o = FlashAttentionFunction.apply(q2, k, v, None, False)
    assert type(o.grad_fn).__name__ == "FlashAttentionFunctionBackward"


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("causal", [False, True])
def test_both_mfma_tiles_of_the_head_dim_128_forward_hold_the_planned_contract(dt, causal):
    """Round 5: head dim 128 launches of whole items run bodies built on v_mfma_f32_16x16x32 by default (option "asm" bit 6; csrc/gen/fwd_m16_gen.py:
    a Q row is spread over four lanes there) and the 32x32x16 bodies otherwise (bit 6 clear; KV-split launches).  Same contract either way — both
    against the oracle under the contract fa2_fwd_plan names, under every `fold` setting, on a ragged shape with more items than workgroups (first
    items, prefetched items, a ragged last tile, rows past Nq) — and the two differ only in f32 summation order."""
    B, H, N, Nkv = (2, 40, 2000, 2000) if causal else (3, 30, 1000, 1111)          # 640 / 360 items of 256 rows; ragged Nq and Nkv
    g = torch.Generator(device="cpu").manual_seed(900 + dt + 2 * causal)
    q = torch.randn((B, H, N, 128), generator=g).to(TORCH_DT[dt]).to(_dev())
    k, v = (torch.randn((B, H, Nkv, 128), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
    for fold in (0, 1, 2):
        outs = {}
        for asm in (579, 67, 3):          # 16x16x32 bodies with the row sums on the matrix pipe (bits 6 + 9) / with the sum check (bit 6) / 32x32x16
            with _fa2_lib.options(asm=asm, fold=fold, rows=256):
                plan = _plan(q, k, causal)
                assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.heads_main == B * H
                folded = fold >= (2 if dt else 1)
                assert bool(plan.contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q) == folded
                # the folded 16x16x32 bodies keep their row sums on the matrix pipe: the sums of the ROUNDED P (csrc/gen/fwd_m16_gen.py, opt=lm)
                assert bool(plan.contract & _fa2_lib.FA2_CONTRACT_LSUM_P16) == (asm == 579)
                o, lse = _cabi_forward(q, k, v, causal)
                for head in (0, B * H // 2, B * H - 1):
                    b, h = divmod(head, H)
                    sl = (slice(b, b + 1), slice(h, h + 1))
                    _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl], dt, causal, plan=plan, head=head)
            outs[asm] = (o, lse)
        for asm in (579, 67):
            assert float((outs[asm][0].float() - outs[3][0].float()).abs().max()) <= (3.2e-2 if dt else 4e-3)
            assert float((outs[asm][1] - outs[3][1]).abs().max()) <= ((LSE_TOL_P16_BF16 if dt else 1e-3) if asm == 579 else 1e-4)


@pytest.mark.parametrize("pitch", [136, 160, 192])
def test_row_padded_v_on_the_hand_scheduled_head_dim_128_kernels(pitch):
    """A V whose rows are a column slice of a wider matrix (row pitch 136 / 160 / 192 elements: multiples of 8, 32 and 64) on a grid that reaches the
    hand-scheduled kernels.  The folded 16x16x32 bodies derive the source offsets of every other V piece by flipping a bit of the byte offset, which
    needs a pitch that is a multiple of 32 elements (fwd_asm.cpp keeps other pitches on the 32x32x16 body) — found by tools/fuzz_parity.py in round 5
    (profiles/fuzz_runs.md, row r17_fuzz_rows256_seed503: O wrong by 0.2 .. 4, LSE right)."""
    B, H, N = 2, 40, 1500
    g = torch.Generator(device="cpu").manual_seed(pitch)
    q, k = (torch.randn((B, H, N, 128), generator=g).half().to(_dev()) for _ in range(2))
    vw = torch.randn((B, H, N, pitch), generator=g).half().to(_dev())
    v = vw[..., :128]
    assert v.stride(2) == pitch
    for causal in (False, True):
        with _fa2_lib.options(rows=256):
            plan = _plan(q, k, causal)
            assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM
            o, lse = _cabi_forward(q, k, v, causal)
        for head in (0, B * H - 1):
            b, h = divmod(head, H)
            sl = (slice(b, b + 1), slice(h, h + 1))
            _assert_close_to_oracle(o[sl], lse[sl], q[sl], k[sl], v[sl].contiguous(), 0, causal, plan=plan, head=head)


@pytest.mark.gpu
@pytest.mark.parametrize("D,dt,causal", [(40, 0, False), (48, 0, True), (56, 0, False), (96, 0, False), (112, 0, True), (120, 0, False), (112, 1, False), (104, 1, True)])
def test_head_dims_below_the_hand_scheduled_bodys(D, dt, causal):
    """Round 5: head dims 40 .. 56 run on the head-dim-64 16x16x32 body and 96 .. 120 (f32 scale: 104 ..) on the 128 one — SD 1.5's D = 40 among them —
    with the padded columns of the Q / K / V images zero-filled by the LDS-DMA itself (csrc/gen/fwd_m16_gen.py: trim_offsets; host.cpp: plan_range).
    The plan must name the hand-scheduled kernel; results against the oracle under the planned contract; and — what the zero fill is for — with the
    tensors cut out of wider allocations whose gaps hold NaNs (row pitch D + 8 and 2 D: nothing of a gap or of a neighbouring row may enter a product),
    ragged Nq / Nkv included."""
    B, H, N, Nkv = 2, 9, 2000, 2100                        # 144 items of 256 rows on 256-row workgroups (rows = 256 pins the shape below)
    g = torch.Generator(device="cpu").manual_seed(40 + D)
    tdt = TORCH_DT[dt]
    with _fa2_lib.options(rows=256):
        for pitch in (D, D + 8, 2 * D):
            def cut(n):
                wide = torch.full((B, H, n, pitch), float("nan"), dtype=tdt)
                wide[..., :D] = torch.randn((B, H, n, D), generator=g).to(tdt)
                return wide.to(_dev())[..., :D]
            q, k, v = cut(N), cut(Nkv), cut(Nkv)
            plan = _plan(q, k, causal)
            assert plan.kernel == _fa2_lib.FA2_KERNEL_ASM and plan.heads_main == B * H, (D, pitch, plan.as_dict())
            o, lse = _cabi_forward(q, k, v, causal)
            assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all(), (D, pitch)
            for head in (0, B * H - 1):
                b, h = divmod(head, H)
                sl = (slice(b, b + 1), slice(h, h + 1))
                _assert_close_to_oracle(o[sl], lse[sl], q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), dt, causal, plan=plan, head=head)
        # bit 6 clear: the trimmed compiler-scheduled kernels, as before
        with _fa2_lib.options(asm=_fa2_lib.load().fa2_get_option(b"asm") & ~64):
            assert _plan(q, k, causal).kernel != _fa2_lib.FA2_KERNEL_ASM
        # the forward of a call that will be differentiated (FA2_FLAG_EXACT_SCALE): fp16 at 104 .. 120 runs the f32-scale 16x16x32 body with the sum check
        # (contract 0), everything else the compiler-scheduled kernels — either way the reference kernel's contract
        flags = (_fa2_lib.FA2_FLAG_CAUSAL if causal else 0) | _fa2_lib.FA2_FLAG_EXACT_SCALE
        plan = _fa2_lib.fwd_plan(q, k, flags)
        assert plan.contract == 0 and (plan.kernel == _fa2_lib.FA2_KERNEL_ASM) == (dt == 0 and D >= 104), (D, plan.as_dict())
        o, lse = _cabi_forward(q, k, v, flags)
        for head in (0, B * H - 1):
            b, h = divmod(head, H)
            sl = (slice(b, b + 1), slice(h, h + 1))
            _assert_close_to_oracle(o[sl], lse[sl], q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), dt, causal, plan=plan, head=head)
