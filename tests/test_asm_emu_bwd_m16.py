"""CPU check of the dQ pass built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py; option "asm" bit 7): the generated instruction list runs on the
functional emulator (tools/asm_emu.py) for one workgroup, with the operand values the shell (bwd_dq_d128_kernel<..., M16 = true>, fa2_bwd_d128.hip.h)
computes, and must reproduce float64 dQ and delta of dense attention with no modelled hazard.  Same case list as the 32x32x16 pass
(tests/test_asm_emu_bwd.py): every head / tail body, the fast loop in both parities, causal diagonals, ragged tails, clamped rows, bf16."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "tools"))
import asm_emu_bwd as harness  # noqa: E402
from test_asm_emu_bwd import DQ_CASES  # noqa: E402


@pytest.fixture(autouse=True)
def _dq16():
    saved = harness.DQ_M16
    harness.DQ_M16 = True
    yield
    harness.DQ_M16 = saved


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


def test_generated_dq16_text_assembles_for_gfx950(tmp_path):
    import re
    import shutil
    import subprocess
    import bwd_d128_gen as gen
    import bwd_dq_m16_gen as gen16
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    n_v = gen.DQ.N_VARGS
    for bf16 in (False, True):
        prog = gen16.GenDQ16(bf16).build()
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
        src = tmp_path / ("dq16_%d.s" % bf16)
        src.write_text(text + "\n")
        res = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[:2000]
