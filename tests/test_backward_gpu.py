"""GPU parity tests of the backward path (`-m gpu`): fa2_bwd_* through the C-ABI and through
FlashAttentionFunction's autograd, against the golden fixtures (reference oracle backward), float64 autograd
and the C oracle.  Nothing here reads /root/reference."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import GRAD_TOL, grads_truth
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


def _cabi_fwd_bwd(q, k, v, do, causal, scale=None):
    """forward then backward straight through the C-ABI with caller-owned buffers."""
    lib = _fa2_lib.load(build_if_missing=False)
    B, H, N, D = q.shape
    Nkv = k.shape[2]
    scale = float(D ** -0.5 if scale is None else scale)
    o = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = 0 if q.dtype == torch.float16 else 1
    _fa2_lib.check(lib.fa2_fwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, Nkv, D,
                               s3(q), s3(k), s3(v), s3(o), s2, scale, int(causal), stream))
    fn = lib.fa2_bwd_f16 if q.dtype == torch.float16 else lib.fa2_bwd_bf16
    _fa2_lib.check(fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                      dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, N, Nkv, D,
                      s3(q), s3(k), s3(v), s3(o), s3(do), s3(dq), s3(dk), s3(dv), s2, scale, int(causal), stream))
    torch.cuda.synchronize()
    return o, lse, (dq, dk, dv)


def _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal, scale=None, independent=False):
    """independent=False: the oracle's backward is fed the kernel's own O and LSE (isolates the backward kernels from the forward's rounding);
    independent=True: it is fed the ORACLE's forward outputs — nothing of the kernel enters the expected gradients (the default form is
    partly self-referential; the forward is pinned separately, this closes the loop for the seeded shapes)."""
    if independent:
        o_bits, lse_np = fo.fwd_c(_bits(q), _bits(k), _bits(v), dt, causal, scale=scale)
        want = fo.bwd_c(_bits(q), _bits(k), _bits(v), o_bits, _bits(do), lse_np, dt, causal, scale=scale)
    else:
        want = fo.bwd_c(_bits(q), _bits(k), _bits(v), _bits(o), _bits(do), lse.cpu().numpy(), dt, causal, scale=scale)
    for name, g, w_bits in zip("qkv", grads, want):
        w = fo.bits_to_f32(w_bits, dt)
        got = g.float().cpu().numpy()
        assert np.isfinite(got).all(), "d%s has non-finite values" % name
        assert np.abs(got - w).max() <= GRAD_TOL[dt] * max(1.0, np.abs(w).max()), (name, np.abs(got - w).max())


def test_golden_backward_fixtures(golden):
    dt = golden["dtype"]
    q, k, v = (_to_dev(golden[t], dt) for t in "qkv")
    qf, kf, vf = (fo.bits_to_f32(golden[t], dt) for t in "qkv")
    for causal, var in golden["variants"].items():
        if "do" not in var:
            continue
        do = _to_dev(var["do"], dt)
        o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
        truth = grads_truth(qf, kf, vf, fo.bits_to_f32(var["do"], dt), causal)
        for name, g, g_true, ref_bits in zip("qkv", grads, truth, (var["dq_ref"], var["dk_ref"], var["dv_ref"])):
            got = g.float().cpu().numpy()
            ref_err = np.abs(fo.bits_to_f32(ref_bits, dt) - g_true).max()
            err = np.abs(got - g_true).max()
            assert err <= max(2 * ref_err, GRAD_TOL[dt] * max(1.0, np.abs(g_true).max())), (golden["name"], causal, name, err, ref_err)
        _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)


SHAPES = [(1, 1, 1, 1, 64), (1, 2, 1, 300, 128), (1, 2, 300, 129, 256), (2, 1, 128, 384, 160), (2, 3, 65, 1, 64), (1, 2, 31, 33, 64), (1, 3, 255, 257, 128),
          (2, 2, 256, 256, 128), (1, 2, 257, 511, 64), (1, 1, 700, 700, 128), (3, 5, 130, 77, 64), (1, 4, 512, 512, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("causal", [False, True])
def test_seeded_shapes_against_oracle(shape, dt, causal):
    B, H, Nq, Nkv, D = shape
    g = torch.Generator(device="cpu").manual_seed(hash(shape) % 1000 + 31 * dt)
    mk = lambda n: torch.randn((B, H, n, D), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(Nq), mk(Nkv), mk(Nkv), mk(Nq)
    o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
    _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)
    _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal, independent=True)


@pytest.mark.parametrize("D", [8, 40, 80, 104, 120, 136, 160, 200, 256])
@pytest.mark.parametrize("dt", [0, 1])
def test_backward_head_dims_masked_in_kernel(D, dt):
    """Multiples of 8 up to 256 run on the 64 / 128 / 256 kernels with columns >= D masked (SD1.5: 40, 80, 160), unpadded
    buffers; D > 128 takes the 4-wave single-stage kernels."""
    g = torch.Generator(device="cpu").manual_seed(200 + D)
    mk = lambda n: torch.randn((2, 2, n, D), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(130), mk(203), mk(203), mk(130)
    for causal in (False, True):
        o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
        _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)


@pytest.mark.parametrize("D", [16, 32, 48, 56, 96, 104, 160, 192, 224, 232])
def test_trimmed_backward_kernels_on_wide_grids(D):
    """Head dims well below the kernel's HD run TRIMMED instantiations of the passes (only the k-steps / accumulator column blocks that
    hold real columns, bwd_hip.cpp: D <= 32 / 48 on the 64 kernels, <= 96 on the 128 kernels — wave-pair dK / dV pass included —,
    <= 160 / 192 / 224 on the 256 kernels): every boundary of the dispatch and one dim beyond it, on a grid wide enough for the
    256-row dQ workgroups (test_backward_head_dims_masked_in_kernel covers the small-grid shapes), ragged N."""
    dt = (D // 8) & 1
    g = torch.Generator(device="cpu").manual_seed(600 + D)
    mk = lambda n: torch.randn((2, 17, n, D), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(1000), mk(900), mk(900), mk(1000)      # 34 heads x 4 q blocks = 136 workgroups of 256 rows
    for causal in (False, True):
        o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
        _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)


def test_autograd_through_the_operator_matches_torch():
    """FlashAttentionFunction.apply(...).backward(dO) — the reference's training call shape (bench_with_sdpa.py:89-96)."""
    g = torch.Generator(device="cpu").manual_seed(21)
    for dtype, causal, D, Nkv in ((torch.float16, False, 64, 300), (torch.bfloat16, True, 128, 200), (torch.float16, True, 40, 200)):
        q = torch.randn((2, 3, 200, D), generator=g).to(dtype).to(_dev()).requires_grad_(True)
        k = torch.randn((2, 3, Nkv, D), generator=g).to(dtype).to(_dev()).requires_grad_(True)
        v = torch.randn((2, 3, Nkv, D), generator=g).to(dtype).to(_dev()).requires_grad_(True)
        do = torch.randn((2, 3, 200, D), generator=g).to(dtype).to(_dev())
        o = FlashAttentionFunction.apply(q, k, v, None, causal)
        o.backward(do)
        torch.cuda.synchronize()
        assert q.grad.shape == q.shape and k.grad.shape == k.shape and v.grad.shape == v.shape
        truth = grads_truth(*(t.detach().float().cpu().numpy() for t in (q, k, v, do)), causal)
        dt = 0 if dtype == torch.float16 else 1
        for got, want in zip((q.grad, k.grad, v.grad), truth):
            assert np.abs(got.float().cpu().numpy() - want).max() <= GRAD_TOL[dt] * max(1.0, np.abs(want).max())


def test_calls_in_which_only_k_and_v_need_gradients():
    """The reference decides on q.requires_grad alone whether to save anything for the backward (FlashAttn.py:73-75): a call in which only K / V need
    gradients — a frozen query projection, a KV-cache being tuned — builds an autograd node whose backward finds no ctx.args.  Here any of q, k, v
    counts; both front ends, and the module-level forward() flags such a call FA2_FLAG_EXACT_SCALE on its own."""
    from rocwmma_fattn import FlashAttn
    g = torch.Generator(device="cpu").manual_seed(77)
    for python_front_end in (False, True):
        saved = FlashAttn._FRONTEND[0]
        if python_front_end:
            FlashAttn._FRONTEND[0] = None
        try:
            for dtype, causal, D in ((torch.float16, False, 64), (torch.bfloat16, True, 128)):
                q = torch.randn((2, 3, 200, D), generator=g).to(dtype).to(_dev())
                k = torch.randn((2, 3, 264, D), generator=g).to(dtype).to(_dev()).requires_grad_(True)
                v = torch.randn((2, 3, 264, D), generator=g).to(dtype).to(_dev()).requires_grad_(True)
                do = torch.randn((2, 3, 200, D), generator=g).to(dtype).to(_dev())
                o = FlashAttentionFunction.apply(q, k, v, None, causal)
                assert o.requires_grad
                o.backward(do)
                torch.cuda.synchronize()
                assert q.grad is None and k.grad.shape == k.shape and v.grad.shape == v.shape
                truth = grads_truth(*(t.detach().float().cpu().numpy() for t in (q, k, v, do)), causal)
                dt = 0 if dtype == torch.float16 else 1
                for got, want in zip((k.grad, v.grad), truth[1:]):
                    assert np.abs(got.float().cpu().numpy() - want).max() <= GRAD_TOL[dt] * max(1.0, np.abs(want).max())
        finally:
            FlashAttn._FRONTEND[0] = saved
    # the reference's module-level API (host.cpp:30-58), forward() with a plain bool: the plan of what it launches is the exact-scale one
    q = torch.randn((2, 16, 4096, 128), generator=g).half().to(_dev())
    k, v = (torch.randn((2, 16, 4096, 128), generator=g).half().to(_dev()).requires_grad_(True) for _ in range(2))
    o1 = flash_attn_wmma.forward(q, k, v, 64, 128, False, 128 ** -0.5, False)[0]
    o2 = flash_attn_wmma.forward(q, k.detach(), v.detach(), 64, 128, _fa2_lib.FA2_FLAG_EXACT_SCALE, 128 ** -0.5, False)[0]
    o3 = flash_attn_wmma.forward(q, k.detach(), v.detach(), 64, 128, False, 128 ** -0.5, False)[0]
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)          # (the default fp16 launch folds the scale: other bits)


def test_backward_is_deterministic_and_bnhd_matches_bhnd():
    g = torch.Generator(device="cpu").manual_seed(23)
    q, k, v, do = (torch.randn((2, 4, 300, 64), generator=g).half().to(_dev()) for _ in range(4))
    _, _, g1 = _cabi_fwd_bwd(q, k, v, do, True)
    _, _, g2 = _cabi_fwd_bwd(q, k, v, do, True)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)        # one owner per output element: no atomics, no run-to-run noise
    # BNHD: same numbers through the permuted layout (reference: kernel_fp16.cu:328-333)
    qn, kn, vn = (t.transpose(1, 2).contiguous() for t in (q, k, v))
    ret = flash_attn_wmma.forward(qn, kn, vn, 64, 128, True, 64 ** -0.5, True)
    dq, dk, dv = flash_attn_wmma.backward(ret[1], ret[2], ret[3], ret[4], do.transpose(1, 2).contiguous(), ret[5],
                                          300, 300, 64, 128, 128, True, 64 ** -0.5, True)
    torch.cuda.synchronize()
    for a, b in zip((dq, dk, dv), g1):
        assert torch.equal(a.transpose(1, 2), b)


def test_full_size_config2_backward_sampled_heads():
    """BASELINE config 2 shape: two heads checked against the oracle, the whole tensor for finiteness and for
    head-slice independence (a head recomputed alone is bit-identical)."""
    g = torch.Generator(device=_dev()).manual_seed(5)
    q, k, v, do = (torch.randn((2, 16, 4096, 128), generator=g, device=_dev(), dtype=torch.float32).half() for _ in range(4))
    o, lse, grads = _cabi_fwd_bwd(q, k, v, do, False)
    for t in grads:
        assert torch.isfinite(t.float()).all()
    sl = (slice(1, 2), slice(5, 6))
    _check_vs_oracle(q[sl], k[sl], v[sl], do[sl], o[sl], lse[sl], [t[sl] for t in grads], 0, False)
    # (8 heads: the forward keeps its 256-row kernel for the slice — it switches shape below 97 workgroups — so O and LSE,
    #  and with them the gradients, are bit-identical)
    _, _, part = _cabi_fwd_bwd(q[:, 3:11].contiguous(), k[:, 3:11].contiguous(), v[:, 3:11].contiguous(), do[:, 3:11].contiguous(), False)
    for a, b in zip(part, grads):
        assert torch.equal(a, b[:, 3:11])


def test_full_size_config3_backward_sampled_heads():
    """BASELINE config 3 shape (bf16, causal) through the backward: one head against the same-contract oracle, everything
    finite, and the causal structure itself — dK / dV of the last KV row depend on the last Q row only, so they equal the
    closed form (P = 1 on the diagonal of the last row: dV[last] = dO[last] * P, with P the last row's softmax weight)."""
    g = torch.Generator(device=_dev()).manual_seed(6)
    q, k, v, do = (torch.randn((2, 16, 4096, 128), generator=g, device=_dev(), dtype=torch.float32).bfloat16() for _ in range(4))
    o, lse, grads = _cabi_fwd_bwd(q, k, v, do, True)
    for t in grads:
        assert torch.isfinite(t.float()).all()
    sl = (slice(0, 1), slice(9, 10))
    _check_vs_oracle(q[sl], k[sl], v[sl], do[sl], o[sl], lse[sl], [t[sl] for t in grads], 1, True)
    # first Q row attends to the first KV row only: P = 1, dP - D = 0 up to f32 summation order -> dQ[0] vanishes
    assert float(grads[0][:, :, 0].float().abs().max()) <= 1e-3
    # last KV row is seen by the last Q row only: dV[last] = p * dO[last] with p = 2^(s*c - lse)
    s_last = (q[:, :, -1].float() * k[:, :, -1].float()).sum(-1) * (128 ** -0.5)
    p_last = torch.exp2(s_last * 1.4426950408889634 - lse[:, :, -1])
    want = p_last[..., None] * do[:, :, -1].float()
    assert float((grads[2][:, :, -1].float() - want).abs().max()) <= 1.6e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("D", [264, 320, 328, 384, 448, 456, 512])
@pytest.mark.parametrize("dt", [0, 1])
def test_backward_head_dims_above_256(D, dt):
    """Head dims above 256 (the forward reaches 512: the SD VAE attention block, the reference's D > 384 case, FlashAttn.py:65-67) run the
    slab kernels: 128-column slabs of dQ / dK / dV per workgroup, S and dP recomputed per slab over the whole head dim — so that
    o.backward() works on everything the forward accepts (reference: backward_fp16 pads D like the forward, kernel_fp16.cu:878-1028)."""
    g = torch.Generator(device="cpu").manual_seed(300 + D)
    mk = lambda n: (torch.randn((1, 2, n, D), generator=g) * 0.5).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(130), mk(203), mk(203), mk(130)
    for causal in (False, True):
        o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
        _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)


def test_autograd_at_head_dim_512_matches_torch():
    """o.backward() through FlashAttentionFunction at the VAE-sized head dim against float64 autograd of the dense formula."""
    g = torch.Generator(device="cpu").manual_seed(512)
    q, k, v, do = (torch.randn((1, 1, 160, 512), generator=g) * 0.3 for _ in range(4))
    qd, kd, vd = (t.half().to(_dev()).requires_grad_(True) for t in (q, k, v))
    o = FlashAttentionFunction.apply(qd, kd, vd, None, False)
    o.backward(do.half().to(_dev()))
    want = grads_truth(q.half().double().numpy(), k.half().double().numpy(), v.half().double().numpy(), do.half().double().numpy(), False)
    for name, got, w in zip("qkv", (qd.grad, kd.grad, vd.grad), want):
        err = float(np.abs(got.float().cpu().numpy() - w).max())
        assert err <= 2 * GRAD_TOL[0] * max(1.0, float(np.abs(w).max())), (name, err)


PAIR_SHAPES = [(1, 2, 129, 127, 128), (2, 3, 200, 333, 96), (1, 8, 640, 640, 128), (1, 1, 64, 700, 72), (2, 2, 513, 255, 120)]


@pytest.mark.parametrize("shape", PAIR_SHAPES)
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("causal", [False, True])
def test_wave_pair_dkdv_pass_against_oracle(shape, dt, causal):
    """Head dims 65..128 run dK and dV as ONE sweep of wave pairs (bwd_dkv_pair_kernel: 128 KV rows per workgroup, P crosses from
    the dV wave to the dK wave through LDS as the 16-bit fragments the dV product consumes).  Shapes chosen for it: KV rows that are
    not multiples of 128 or 32, one- and many-block sweeps, Nq shorter and longer than Nkv under the causal mask (pairs whose
    rows lie entirely above the first tile's q range), head dims with a masked column tail."""
    B, H, Nq, Nkv, D = shape
    g = torch.Generator(device="cpu").manual_seed(hash(shape) % 1000 + 7 * dt)
    mk = lambda n: torch.randn((B, H, n, D), generator=g).to(TORCH_DT[dt]).to(_dev())  # noqa: E731
    q, k, v, do = mk(Nq), mk(Nkv), mk(Nkv), mk(Nq)
    o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal)
    _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal)
    o2, lse2, grads2 = _cabi_fwd_bwd(q, k, v, do, causal)
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b), "the pair pass is not deterministic"


def test_pinned_workgroup_shapes_and_kernel_families():
    """Small grids run dQ as 128-row, 4-wave workgroups and head dim 128 runs the hand-scheduled bodies when the shape allows, so
    most parity cases above reach only one of the kernels that can serve them.  The library options (fa2_set_option) pin the other
    choices in this process: rows = 256 / 128 (the 8-wave / 4-wave dQ shapes) and asm = 1 (compiler-scheduled backward kernels,
    hand-scheduled forward) — the same cases must pass through each of them."""
    import sys
    from rocwmma_fattn import _fa2_lib
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from test_parity_gpu import _expand_and_call
    mod = sys.modules[__name__]
    groups = {"small": [n for n in dir(mod) if n.startswith("test_") and any(k in n for k in ("golden", "seeded", "wave_pair", "head_dims", "deterministic"))],
              "big": [n for n in dir(mod) if n.startswith("test_") and any(k in n for k in ("full_size", "autograd"))]}
    assert groups["small"] and groups["big"], groups
    n = 0
    for opts, names in (({"rows": 256}, groups["small"]), ({"rows": 128}, groups["big"]), ({"asm": 1}, groups["small"] + groups["big"]),
                        ({"asm": 1, "rows": 256}, groups["small"]), ({"asm": 67}, groups["small"] + groups["big"])):     # 67: the 32x32x16 dQ and dK / dV passes (bits 7, 8 clear)
        with _fa2_lib.options(**opts):
            for name in names:
                n += _expand_and_call(getattr(mod, name))
    assert n >= 20, n


@pytest.mark.parametrize("D,dt", [(128, 0), (128, 1), (64, 0), (80, 1)])
def test_backward_with_a_negative_scale_on_ragged_and_causal_shapes(D, dt):
    """Negative scales through every backward kernel family (the hand-scheduled D = 128 passes, the fused D <= 64 pass, the wave-pair pass),
    with a ragged Nkv and with the causal mask: masked scores must stay masked whatever the sign of scale * log2(e)
    (found by tools/fuzz_parity.py: the hand-scheduled passes turned them into +inf, profiles/fuzz_runs.md, row r06_fuzz_parity_seed6)."""
    for (N, Nkv, causal) in ((320, 300, False), (416, 416, True), (96, 77, False)):
        g = torch.Generator(device="cpu").manual_seed(N + D)
        q, do = (torch.randn((2, 3, N, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
        k, v = (torch.randn((2, 3, Nkv, D), generator=g).to(TORCH_DT[dt]).to(_dev()) for _ in range(2))
        scale = -(D ** -0.5)
        o, lse, grads = _cabi_fwd_bwd(q, k, v, do, causal, scale=scale)
        truth = grads_truth(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), do.float().cpu().numpy(), causal, scale=scale)
        for name, gt, g_true in zip("qkv", grads, truth):
            got = gt.float().cpu().numpy()
            assert np.isfinite(got).all(), (name, N, Nkv, causal)
            assert np.abs(got - g_true).max() <= GRAD_TOL[dt] * max(1.0, np.abs(g_true).max()), (name, N, Nkv, causal, np.abs(got - g_true).max())
        _check_vs_oracle(q, k, v, do, o, lse, grads, dt, causal, scale=scale)


@pytest.mark.parametrize("causal", [False, True])
def test_reference_precision_shape_backward(causal):
    """precision_test.py:34-39, :63, :72-98 of the reference: (B, H, N, D) = (3, 7, 1537, 111), Nkv = 1234, bf16, inputs * 1.2, through
    FlashAttentionFunction.apply(q, k, v, None, causal, None, False) and autograd with the reference's upstream gradient — ones, column 0 set
    to -2 — and its three comparisons: dQ, dK, dV.  (D = 111 is zero-padded to 112 by the operator, the one kind of head dim that is; the
    reference prints max differences and asserts nothing: the bounds here are the suite's.)  Checked against the C oracle's backward on sampled
    heads (whole tensor: 21 heads x 1.9 M scores would take the oracle minutes) and against float64 autograd on the whole tensor."""
    dt = 1
    g = torch.Generator(device="cpu").manual_seed(32)
    q = (torch.rand((3, 7, 1537, 111), generator=g) * 1.2).bfloat16().to(_dev())
    k = (torch.rand((3, 7, 1234, 111), generator=g) * 1.2).bfloat16().to(_dev())
    v = (torch.rand((3, 7, 1234, 111), generator=g) * 1.2).bfloat16().to(_dev())
    do = torch.ones_like(q)
    do[..., 0] = -2                                                    # precision_test.py:72-74
    qa, ka, va = (t.detach().requires_grad_(True) for t in (q, k, v))
    o = FlashAttentionFunction.apply(qa, ka, va, None, causal, None, False)
    o.backward(do)
    torch.cuda.synchronize()
    assert o.shape == q.shape and all(t.grad.shape == t.shape and t.grad.dtype == t.dtype for t in (qa, ka, va))
    # float64 autograd, whole tensor, on the device
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qd, kd.transpose(-1, -2)) * (111 ** -0.5)
    if causal:
        s = s.masked_fill(torch.ones(1537, 1234, dtype=torch.bool, device=_dev()).triu(1), float("-inf"))
    torch.matmul(torch.softmax(s, -1), vd).backward(do.double())
    for name, got, want in (("dq", qa.grad, qd.grad), ("dk", ka.grad, kd.grad), ("dv", va.grad, vd.grad)):
        assert torch.isfinite(got.float()).all(), name
        err = float((got.double() - want).abs().max())
        assert err <= GRAD_TOL[dt] * max(1.0, float(want.abs().max())), (name, err, float(want.abs().max()))
    # the C oracle's backward (the reference's recurrences, kernel_fp16.cu:547-740) on sampled heads, fed the kernel's own O and LSE of those heads
    ret = flash_attn_wmma.forward(q, k, v, 64, 128, causal, 111 ** -0.5, False)
    lse = ret[5][:, :, :1537]
    for (b, h) in ((0, 0), (2, 6), (1, 3)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        _check_vs_oracle(q[sl], k[sl], v[sl], do[sl], o[sl].detach(), lse[sl], [t.grad[sl] for t in (qa, ka, va)], dt, causal, scale=111 ** -0.5)


@pytest.mark.parametrize("causal", [False, True])
def test_differentiated_calls_scale_the_f32_product_on_large_logits(causal):
    """Option `fold` = 1 (the fp16 default) lets forward launches of the hand-scheduled bodies round Q * scale * log2(e) to fp16 once, and round 4
    let the dK / dV pass do the same to K (`kfold`): the P a backward pass recomputes then differs from the P behind the saved L by ~|logit| * 2^-11
    relative — measured here at 3x the amplitude of the suite's other cases (logits of +-30 and more) on a grid that reaches the hand-scheduled
    D = 128 passes: dQ 2.4e-3 .. 4.1e-3 of the largest gradient under fold = 1 against 0.7e-3 .. 1.1e-3 with every pass scaling the f32 product
    (GRAD_TOL: 2e-3).  Since round 5 the operator flags the forward of a call that will be differentiated FA2_FLAG_EXACT_SCALE and
    `kfold` is off by default: forward and both backward passes form P from the same scores.  Checked against float64 autograd — nothing of the
    kernel enters the expectation — through the operator under the DEFAULT options, and the plan of the flagged call must say so."""
    dt = 0
    B, H, N, D = 2, 8, 1024, 128
    g = torch.Generator(device="cpu").manual_seed(77 + int(causal))
    q, k = (3.0 * torch.randn((B, H, N, D), generator=g)).half().to(_dev()), (3.0 * torch.randn((B, H, N, D), generator=g)).half().to(_dev())
    v, do = torch.randn((B, H, N, D), generator=g).half().to(_dev()), torch.randn((B, H, N, D), generator=g).half().to(_dev())
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qd, kd.transpose(-1, -2)) * (D ** -0.5)
    assert float(s.detach().abs().max()) > 30.0                          # the logits this test is about
    if causal:
        s = s.masked_fill(torch.ones(N, N, dtype=torch.bool, device=_dev()).triu(1), float("-inf"))
    torch.matmul(torch.softmax(s, -1), vd).backward(do.double())
    with _fa2_lib.options(rows=256):                                     # (the hand-scheduled kernels on this grid; fold / kfold at their defaults)
        assert _fa2_lib.load().fa2_get_option(b"fold") == 1 and _fa2_lib.load().fa2_get_option(b"kfold") == 0
        plain = _fa2_lib.fwd_plan(q, k, causal)
        flagged = _fa2_lib.fwd_plan(q, k, (_fa2_lib.FA2_FLAG_CAUSAL if causal else 0) | _fa2_lib.FA2_FLAG_EXACT_SCALE)
        assert plain.kernel == flagged.kernel == _fa2_lib.FA2_KERNEL_ASM
        assert plain.contract == _fa2_lib.FA2_CONTRACT_PRESCALE_Q | _fa2_lib.FA2_CONTRACT_LSUM_P16 and flagged.contract == 0
        qa, ka, va = (t.detach().requires_grad_(True) for t in (q, k, v))
        o = FlashAttentionFunction.apply(qa, ka, va, None, causal)
        o.backward(do)
        o_inf = FlashAttentionFunction.apply(q, k, v, None, causal)      # the same call without gradients: the folded body
        torch.cuda.synchronize()
    for name, got, want in zip("qkv", (qa.grad, ka.grad, va.grad), (qd.grad, kd.grad, vd.grad)):
        assert torch.isfinite(got.float()).all(), name
        err = float((got.double() - want).abs().max()) / max(1.0, float(want.abs().max()))
        assert err <= GRAD_TOL[dt], (name, err)
    # the two forward contracts differ by the rounding of Q * c, and only by that (profiles/r16_fold_evidence.txt has the table against float64)
    assert not torch.equal(o.detach(), o_inf) and float((o.detach().float() - o_inf.float()).abs().max()) <= 2e-2


def test_kfold_option_keeps_the_suite_tolerance_on_unit_scale_inputs():
    """Option `kfold` = 1 (round 4's default, opt-in since round 5): the hand-scheduled dK / dV pass takes its P from K * scale * log2(e) rounded once to
    fp16 (generator option `kfold`, emulated in tests/test_asm_emu_bwd.py).  On N(0,1) inputs — logits of a few units — it holds the suite's tolerance
    against the C oracle fed the kernel's own O and L, and differs from the f32-scale pass."""
    dt = 0
    g = torch.Generator(device="cpu").manual_seed(91)
    mk = lambda: torch.randn((1, 8, 1024, 128), generator=g).half().to(_dev())  # noqa: E731
    q, k, v, do = mk(), mk(), mk(), mk()
    res = {}
    for kfold in (0, 1):
        with _fa2_lib.options(kfold=kfold, rows=256):
            o, lse, grads = _cabi_fwd_bwd(q, k, v, do, False)
        _check_vs_oracle(q[:, :2], k[:, :2], v[:, :2], do[:, :2], o[:, :2], lse[:, :2], [t[:, :2] for t in grads], dt, False)
        res[kfold] = grads
    assert not torch.equal(res[0][1], res[1][1])              # dK came from another body
    assert torch.equal(res[0][0], res[1][0])                  # the dQ pass is the same
