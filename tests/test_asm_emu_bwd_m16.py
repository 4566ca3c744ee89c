"""CPU check of the backward passes built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py, csrc/gen/bwd_dkv_m16_gen.py; option "asm" bits 7 / 8): the
generated instruction lists run on the functional emulator (tools/asm_emu.py) for one workgroup, with the operand values the shells (fa2_bwd_d128.hip.h,
M16 = true) compute, and must reproduce float64 gradients (and delta) of dense attention with no modelled hazard.  Same case lists as the 32x32x16 passes
(tests/test_asm_emu_bwd.py): every head / tail body, the fast loop in both parities, causal diagonals, ragged tails, clamped rows, bf16."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "tools"))
import asm_emu_bwd as harness  # noqa: E402
from test_asm_emu_bwd import DKV_CASES, DQ_CASES  # noqa: E402


@pytest.fixture(autouse=True)
def _m16():
    saved = harness.DQ_M16, harness.DKV_M16
    harness.DQ_M16 = harness.DKV_M16 = True
    yield
    harness.DQ_M16, harness.DKV_M16 = saved


@pytest.mark.parametrize("case", DQ_CASES)
def test_dq16_block_matches_dense_gradients(case):
    nq, nkv, qblk, causal, bf16 = case
    err, derr, m, ref = harness.check_dq(nq, nkv, qblk, causal, bf16=bf16, seed=nq + nkv + qblk, verbose=False)
    assert not m.errors, m.errors[:5]
    scale = max(1.0, float(abs(ref["dq"]).max()))
    assert err <= (8e-3 if bf16 else 1e-3) * scale, err
    assert derr <= 1e-5 * max(1.0, float(abs(ref["delta"]).max())), derr


@pytest.mark.parametrize("causal", [False, True])
def test_dq16_block_with_a_negative_scale(causal):
    import numpy as np
    rng = np.random.default_rng(13 + causal)
    q, k, v, do = (rng.standard_normal((n, 128)) for n in (320, 300, 300, 320))
    dq, delta, m, ref = harness.run_dq(q, k, v, do, 0, causal, scale=-0.11)
    assert not m.errors, m.errors[:5]
    assert np.isfinite(dq).all() and np.abs(dq - ref["dq"]).max() <= 1e-3 * max(1.0, float(np.abs(ref["dq"]).max()))


@pytest.mark.parametrize("case", DKV_CASES)
def test_dkv16_block_matches_dense_gradients(case):
    nq, nkv, kblk, causal, bf16 = case
    ek, ev, m, ref = harness.check_dkv(nq, nkv, kblk, causal, bf16=bf16, seed=nq + nkv + kblk, verbose=False)
    assert not m.errors, m.errors[:5]
    tol = 8e-3 if bf16 else 1e-3
    assert ek <= tol * max(1.0, float(abs(ref["dk"]).max())), ek
    assert ev <= tol * max(1.0, float(abs(ref["dv"]).max())), ev


@pytest.mark.parametrize("causal", [False, True])
def test_dkv16_block_with_a_negative_scale(causal):
    import numpy as np
    rng = np.random.default_rng(17 + causal)
    q, k, v, do = (rng.standard_normal((n, 128)) for n in (320, 256, 256, 320))
    dk, dv, m, ref = harness.run_dkv(q, k, v, do, 1, causal, scale=-0.11)
    assert not m.errors, m.errors[:5]
    for got, want in ((dk, ref["dk"]), (dv, ref["dv"])):
        assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-3 * max(1.0, float(np.abs(want).max()))


def test_dkv16_generator_rejects_schedules_that_stage_behind_the_book_keeping():
    import bwd_dkv_m16_gen as gen16
    gen16.GenDKV16(False, dma=(16.0, 38.0)).build()          # a later window that stays clear of the book-keeping gaps (40 .. 63)
    for w in ((30.0, 52.0), (40.0, 62.0)):
        with pytest.raises(ValueError, match="illegal schedule"):
            gen16.GenDKV16(False, dma=w).build()


@pytest.mark.parametrize("kind", ["dq", "dkv"])
def test_generated_m16_backward_text_assembles_for_gfx950(kind, tmp_path):
    import re
    import shutil
    import subprocess
    import bwd_d128_gen as gen
    import bwd_dkv_m16_gen
    import bwd_dq_m16_gen
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    cls, n_v = (bwd_dq_m16_gen.GenDQ16, gen.DQ.N_VARGS) if kind == "dq" else (bwd_dkv_m16_gen.GenDKV16, gen.KV.N_VARGS)
    for bf16 in (False, True):
        prog = cls(bf16).build()
        widths = {}
        for ins in prog.ins:
            for o in ins.ops:
                if isinstance(o, gen.Arg) and o.kind == "s":
                    widths[o.n] = o.width
        assert min(widths) >= n_v
        subst, nxt = {}, 0
        for n in sorted(widths):
            w = widths[n]
            nxt = (nxt + w - 1) // w * w
            subst[n] = "s%d" % nxt if w == 1 else "s[%d:%d]" % (nxt, nxt + w - 1)
            nxt += w
        assert nxt <= 60
        text = "\n".join(prog.text_lines())
        text = re.sub(r"%(\d+)", lambda m: subst.get(int(m.group(1)), "v%s" % m.group(1)), text.replace("%=", "0"))
        src = tmp_path / ("%s16_%d.s" % (kind, bf16))
        src.write_text(text + "\n")
        res = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[:2000]
