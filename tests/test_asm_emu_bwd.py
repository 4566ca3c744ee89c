"""CPU check of the hand-scheduled D = 128 backward blocks (csrc/gen/bwd_d128_gen.py): the generated instruction lists — the same
objects that are rendered into the inline-asm bodies of fa2_bwd_d128.hip.h — run on the functional emulator (tools/asm_emu.py) for
one workgroup and must reproduce float64 gradients of dense attention, with no hazard the emulator models (loads read before their
s_waitcnt, MFMA results read too early, VALU->permlane/MFMA wait states, LDS races between waves inside a barrier epoch).  Cases
cover every body variant: head / tail bodies for 1, 2, 3 and more tiles, the fast loop in both parities, causal diagonals and ragged
tails (masked bodies), waves that finish early and only stage, clamped rows, bf16."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "tools"))
import asm_emu_bwd as harness  # noqa: E402

DQ_CASES = [
    # Nq, Nkv, q block, causal, bf16
    (256, 256, 0, False, False),
    (256, 32, 0, False, False),         # one tile: H1, H2b, TC
    (256, 64, 0, False, False),         # two tiles: H2m, TB2, TC
    (256, 96, 0, False, True),          # three tiles: H2, TB3, TB2, TC
    (256, 1, 0, False, False),
    (256, 100, 0, False, False),        # ragged tail
    (256, 640, 0, False, False),        # fast loop, both parities
    (200, 333, 0, False, False),        # clamped Q rows + ragged tail
    (512, 512, 1, True, False),         # causal: waves finish at different tiles (stage-and-sync bodies)
    (300, 300, 1, True, True),
    (256, 77, 0, True, False),          # causal + short KV
    (512, 512, 0, True, False),         # causal, first q block: waves with 2, 4, 6, 8 tiles
]


@pytest.mark.parametrize("case", DQ_CASES)
def test_dq_block_matches_dense_gradients(case):
    nq, nkv, qblk, causal, bf16 = case
    err, derr, m, ref = harness.check_dq(nq, nkv, qblk, causal, bf16=bf16, seed=nq + nkv + qblk, verbose=False)
    assert not m.errors, m.errors[:5]
    scale = max(1.0, float(abs(ref["dq"]).max()))
    assert err <= (8e-3 if bf16 else 1e-3) * scale, err
    assert derr <= 1e-5 * max(1.0, float(abs(ref["delta"]).max())), derr


DKV_CASES = [
    # Nq, Nkv, kv block (128 rows), causal, bf16
    (256, 128, 0, False, False),
    (32, 128, 0, False, False),         # one tile
    (64, 128, 0, False, False),         # two tiles
    (96, 128, 0, False, True),          # three tiles, bf16 (the dS side unpacks P with shifts instead of v_fma_mix)
    (32, 77, 0, False, True),           # clamped KV rows
    (640, 128, 0, False, False),        # fast loop, both parities, Q ring wrap-around
    (320, 200, 1, False, False),        # second kv block, ragged Nkv
    (512, 512, 1, True, False),         # causal: sweep starts at the block's diagonal
    (288, 300, 2, True, True),
    (256, 256, 0, True, False),
    (512, 512, 3, True, False),         # last block: short sweep, masked bodies only
    (640, 256, 1, True, True),
    (128, 256, 1, True, False),         # causal block whose rows all lie past Nq: one fully masked tile, zeros
]


@pytest.mark.parametrize("case", DKV_CASES)
def test_dkv_block_matches_dense_gradients(case):
    nq, nkv, kblk, causal, bf16 = case
    ek, ev, m, ref = harness.check_dkv(nq, nkv, kblk, causal, bf16=bf16, seed=nq + nkv + kblk, verbose=False)
    assert not m.errors, m.errors[:5]
    tol = 8e-3 if bf16 else 1e-3
    assert ek <= tol * max(1.0, float(abs(ref["dk"]).max())), ek
    assert ev <= tol * max(1.0, float(abs(ref["dv"]).max())), ev


@pytest.mark.parametrize("kind", ["dq", "dkv"])
def test_generated_backward_text_assembles_for_gfx950(kind, tmp_path):
    """Every line of the rendered bodies goes through the gfx950 assembler (operand classes, constant-bus limits, literals)."""
    import re
    import shutil
    import subprocess
    import bwd_d128_gen as gen
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    cls, n_v = (gen.GenDQ, gen.DQ.N_VARGS) if kind == "dq" else (gen.GenDKV, gen.KV.N_VARGS)
    sregs = iter(range(0, 60))

    def operand(n):
        if n < n_v:
            return "v%d" % n
        return None

    for bf16 in (False, True):
        prog = cls(bf16).build()
        widths = {}
        for ins in prog.ins:
            for o in ins.ops:
                if isinstance(o, gen.Arg) and o.kind == "s":
                    widths[o.n] = o.width
        subst, nxt = {}, 0
        for n in sorted(widths):
            w = widths[n]
            nxt = (nxt + w - 1) // w * w
            subst[n] = "s%d" % nxt if w == 1 else "s[%d:%d]" % (nxt, nxt + w - 1)
            nxt += w
        assert nxt <= 60
        text = "\n".join(prog.text_lines())
        text = re.sub(r"%(\d+)", lambda m: subst.get(int(m.group(1)), "v%s" % m.group(1)), text.replace("%=", "0"))
        src = tmp_path / ("%s_%d.s" % (kind, bf16))
        src.write_text(text + "\n")
        res = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[:2000]


@pytest.mark.parametrize("causal", [False, True])
def test_backward_blocks_with_a_negative_scale(causal):
    """scale < 0 (the reference takes any float, FlashAttn.py:61): scale * log2(e) is negative, so the masked scores of a ragged or causal tile must
    enter the fma as +inf to come out as -inf (P = 0).  The randomised GPU sweep found them entering as -inf (P = inf, dQ non-finite); both
    hand-scheduled passes, ragged Nkv and the causal diagonal."""
    import numpy as np
    import asm_emu_bwd as hb
    rng = np.random.default_rng(11 + causal)
    q, k, v, do = (rng.standard_normal((n, 128)) for n in (320, 300, 300, 320))
    dq, delta, m, ref = hb.run_dq(q, k, v, do, 0, causal, scale=-0.11)
    assert not m.errors, m.errors[:5]
    assert np.isfinite(dq).all() and np.abs(dq - ref["dq"]).max() <= 1e-3 * max(1.0, float(np.abs(ref["dq"]).max()))
    dk, dv, m, ref = hb.run_dkv(q, k[:256], v[:256], do, 1, causal, scale=-0.11)
    assert not m.errors, m.errors[:5]
    for got, want in ((dk, ref["dk"]), (dv, ref["dv"])):
        assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-3 * max(1.0, float(np.abs(want).max()))


def test_generator_rejects_schedules_that_stage_behind_the_book_keeping():
    """A schedule window is an input of the generators (window sweeps: tools/kbench.py, tools/bwd_bench.py).  The filler streams are written against
    the running state a body is entered with — tile offsets of the LDS-DMA pieces, M0's ring slot, the moving read addresses — and the book-keeping
    that advances it rides in late gaps: an LDS-DMA window that reaches those gaps stages the NEXT tile's data into a slot this tile's readers are
    still on.  Round 3 found that by wrong gradients on the GPU (profiles/r09_experiments.txt item 1); now the generator refuses the schedule
    (BodyEmitter.check_running_state), and — with the check switched off — the emulator run shows what it would have computed."""
    import bwd_d128_gen as gen
    for bf16 in (False, True):                       # the shipped windows are legal
        gen.GenDQ(bf16).build()
        gen.GenDKV(bf16).build()
    gen.GenDKV(False, dma=(8.0, 20.0)).build()       # ... and so is a later window that stays clear of the book-keeping gaps (20..31)
    for cls, kw in ((gen.GenDKV, dict(dma=(14.0, 26.0))), (gen.GenDKV, dict(dma=(20.0, 31.0))), (gen.GenDQ, dict(dma=(40.0, 48.0)))):
        with pytest.raises(ValueError, match="illegal schedule"):
            cls(False, **kw).build()
    # the emulator on the rejected dK / dV schedule, check off: no modelled hazard fires (the instruction stream is self-consistent), the gradients are wrong
    saved = dict(harness._PROGS)
    try:
        harness._PROGS[("dkv", False)] = gen.GenDKV(False, dma=(14.0, 26.0), opt=("nocheck",)).build()
        ek, ev, m, ref = harness.check_dkv(640, 128, 0, False, seed=1, verbose=False)
        assert max(ek, ev) > 0.1 * max(1.0, float(abs(ref["dk"]).max()))
    finally:
        harness._PROGS.clear()
        harness._PROGS.update(saved)
