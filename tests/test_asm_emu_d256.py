"""The hand-scheduled head-dim-256 forward block (csrc/gen/fwd_m16_d256_gen.py, round 6) on the instruction-level emulator (tools/asm_emu.py): every
body variant against float64 attention with the hazard model on — head bodies only, the dispatch loop, ragged Nq / Nkv, the causal diagonal with waves
that finish at different tiles, large logits (reference moves, deferred at 2^14) — stores confined to the workgroup's rows, and the text through the
gfx950 assembler.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
import asm_emu_d256 as harness  # noqa: E402

CASES = [
    # Nq, Nkv, q block, causal, bf16, amp
    (128, 64, 0, False, False, 1.0),          # one tile: H1, H2b, TC
    (128, 128, 0, False, True, 1.0),
    (128, 192, 0, False, False, 1.0),         # first TA body (guarded staging)
    (128, 640, 0, False, False, 1.0),         # TF bodies: staging piece by piece between the other streams
    (100, 333, 0, False, False, 1.0),         # ragged Nq (rows >= Nq are neither loaded nor stored) and Nkv (masked last tile)
    (128, 77, 0, True, False, 1.0),
    (256, 256, 1, True, False, 1.0),          # causal: waves 0 / 1 finish a tile ahead of 2 / 3 (stage-only bodies)
    (512, 512, 3, True, True, 1.0),
    (200, 200, 1, True, False, 1.0),
    (128, 704, 0, False, False, 6.0),         # N(0, 6^2) logits: the references move
    (256, 704, 1, True, True, 8.0),
]


@pytest.mark.parametrize("case", CASES)
def test_d256_block_matches_dense_attention(case):
    Nq, Nkv, qblk, causal, bf16, amp = case
    err, lerr, m = harness.check(Nq, Nkv, qblk, causal, bf16=bf16, seed=Nq + Nkv, amp=amp, verbose=False)
    assert not m.errors, m.errors[:5]
    big = amp > 1.0
    assert err <= (8e-3 if bf16 else 1e-3) * (2 if big else 1) and lerr <= ((6e-3 if big else 4e-3) if bf16 else 1e-3), (err, lerr)


@pytest.mark.parametrize("d", [136, 160, 192, 248])
def test_head_dims_below_256_on_the_same_body(d):
    """opt=trim: rows of d columns at a pitch of 2 d bytes (the XOR derivation of the piece offsets does not hold: the general form), a granule the row
    does not have is marked out of range and reads as zero; Q fragments and O stores masked alike.  The emulated matrices are contiguous, so a granule
    fetched from the neighbouring row, or a stray column stored, would show in O / in the store check of the harness."""
    saved = harness.DTRIM
    harness.DTRIM = d
    harness._PROGS.clear()
    try:
        for (Nq, Nkv, qblk, causal, bf16) in ((128, 333, 0, False, False), (200, 200, 1, True, True)):
            err, lerr, m = harness.check(Nq, Nkv, qblk, causal, bf16=bf16, seed=d + Nq, verbose=False)
            assert not m.errors, m.errors[:5]
            assert err <= (8e-3 if bf16 else 1e-3) and lerr <= (4e-3 if bf16 else 1e-3), (err, lerr)
    finally:
        harness.DTRIM = saved
        harness._PROGS.clear()


def test_trimmed_bodies_leave_out_the_k_steps_without_a_real_column():
    """ceil(D / 32) k-steps of Q.K^T and twice as many d groups of O: 84 / 100 / 116 / 132 MFMAs per tile for head dims <= 160 / 192 / 224 / 256 (the
    launcher picks the body, fwd_asm.cpp); a larger body than needed gives the same result (its extra k-steps multiply zero-filled columns)."""
    import fwd_m16_d256_gen as gen
    for nks, want in ((5, 84), (6, 100), (7, 116), (8, 132)):
        prog = gen.Gen256(False, nks=nks, opt=("trim",)).build()
        names = [i.ops[0].name if i.op == "label" else None for i in prog.ins]
        ops = [i.op for i in prog.ins[names.index("ta_e"):names.index("tb_e")]]
        assert sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == want
    saved = harness.DTRIM, harness.NKS
    try:
        outs = []
        for nks in (5, 7):
            harness.DTRIM, harness.NKS = 152, nks
            harness._PROGS.clear()
            rng = np.random.default_rng(3)
            q, k, v = rng.standard_normal((128, 152)), rng.standard_normal((200, 152)), rng.standard_normal((200, 152))
            o, lse, m = harness.run_block(q, k, v, 0, False)
            assert not m.errors
            outs.append((o, lse))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    finally:
        harness.DTRIM, harness.NKS = saved
        harness._PROGS.clear()


def test_d256_body_is_max_first_and_fits_the_instruction_formats():
    """132 MFMAs per full body (64 P.V + 64 Q.K^T + 4 row-sum links), every DS offset inside 16 bits and every MUBUF offset inside 12 (the assembler
    truncates the latter silently — the first GPU run of this kernel raced because of it; tools/asm_emu.py refuses both now), no fast loop."""
    import fwd_m16_d256_gen as gen
    g = gen.Gen256(False)
    prog = g.build()
    names = [i.ops[0].name if i.op == "label" else None for i in prog.ins]
    lo, hi = names.index("ta_e"), names.index("tb_e")
    ops = [i.op for i in prog.ins[lo:hi]]
    assert sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 132
    assert "fast0" not in names and not any(n and n.startswith("lm_repair") for n in names)
    for i in prog.ins:
        if i.op.startswith("ds_"):
            assert 0 <= i.mods.get("offset", 0) <= 0xffff, i.text()
        if i.op.startswith("buffer_"):
            assert 0 <= i.mods.get("offset", 0) <= 0xfff, i.text()


def test_d256_text_assembles_for_gfx950(tmp_path):
    import fwd_d128_gen as base
    import fwd_m16_d256_gen as gen
    llvm_mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(llvm_mc):
        pytest.skip("llvm-mc not installed")
    for bf16, opt, nks in ((False, (), 8), (True, (), 8), (False, ("trim",), 8), (True, ("trim",), 5), (False, ("trim",), 6), (False, ("trim",), 7)):
        prog = gen.Gen256(bf16, nks=nks, opt=opt).build()
        text = "\n".join(prog.text_lines())
        # inline-asm operands -> plain registers of the right width (the assembler checks syntax, operand classes and encodings)
        wide = {4: "s[8:11]", 5: "s[12:15]", 6: "s[16:19]", 25: "s[20:23]"}
        sregs = {3: "s24", 13: "s25", 14: "s26", 15: "s27", 16: "s28", 17: "s29", 18: "s30", 19: "s31", 20: "s32", 22: "s33", 23: "s34", 24: "s35", 26: "s36"}
        for n in sorted(set(range(gen.N_ARGS)), reverse=True):
            rep = wide.get(n) or sregs.get(n) or "v%d" % (n if n < 16 else n - 12)
            text = text.replace("%%%d" % n, rep)
        text = text.replace("%=", "0")
        path = tmp_path / ("d256_%d_%d_%d.s" % (bf16, len(opt), nks))
        path.write_text(text + "\n")
        res = subprocess.run([llvm_mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(path)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
