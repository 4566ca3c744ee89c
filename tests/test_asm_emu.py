"""CPU check of the hand-scheduled D = 128 forward block (csrc/gen/fwd_d128_gen.py): the generated instruction list —
the same objects that are rendered into the inline-asm body of fa2_fwd_d128.hip.h — runs on the functional emulator
(tools/asm_emu.py) for one workgroup and must reproduce dense float64 attention, with no hazard the emulator models
(loads read before their s_waitcnt, MFMA results read too early, VALU->permlane/MFMA wait states, LDS races between
waves inside a barrier epoch).  Cases cover every body variant of the generator: head/tail bodies for 1, 2, 3 and more
KV tiles, the fast loop in both parities, causal diagonal and ragged-tail masks, waves that finish early and only stage,
clamped Q rows, bf16, and the forced rescale branch."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "tools"))
import asm_emu_harness as harness  # noqa: E402

CASES = [
    # Nq, Nkv, q block, causal, bf16, spike
    (256, 256, 0, False, False, False),
    (256, 64, 0, False, False, False),        # one tile: H1m, H2b, TC
    (256, 100, 0, False, False, False),       # two tiles, ragged
    (256, 1, 0, False, False, False),
    (256, 192, 0, False, True, False),        # three tiles
    (256, 640, 0, False, False, False),       # fast loop, both parities
    (200, 333, 0, False, False, False),       # clamped Q rows + ragged tail
    (512, 512, 1, True, False, False),        # causal: waves finish at different tiles
    (300, 300, 1, True, True, False),
    (256, 77, 0, True, False, False),         # causal + short KV
    (256, 704, 0, False, False, True),        # spiked K rows (scores > 2^7 log2 units above the reference: a sum-check body meets P = inf -> the item is redone in safe mode)
    (256, 704, 0, False, False, 2),           # moderate spikes (15 .. 127 log2 units): the in-place repair of the sum-check bodies, q block 0 and 1, twice in one row
    (256, 704, 0, False, True, 2),
    (512, 768, 1, True, False, 2),            # ... with waves that finish at different tiles
    (256, 704, 0, False, False, 3),           # growth of 119 .. 127.4 octaves: at the edge of f32 the repair factor 2^-d is no normal number any more -> redo
    (256, 704, 0, False, True, 3),
]


@pytest.fixture(params=[(), ("ct",)], ids=["f32-scale", "folded-scale"])
def variant(request):
    """The body the library ships — scale applied to the f32 scores — and the generator's folded-scale variant (Q * scale*log2e
    rounded once, the running reference in the C operand of the first QK^T k-step; not shipped since 0.7, kept correct)."""
    saved = harness.OPT
    harness.OPT = request.param
    harness._PROGS.clear()
    yield request.param
    harness.OPT = saved
    harness._PROGS.clear()


@pytest.mark.parametrize("case", CASES)
def test_generated_block_matches_dense_attention(case, variant):
    nq, nkv, qblk, causal, bf16, spike = case
    err, lse_err, m = harness.check(nq, nkv, qblk, causal, bf16=bf16, spike=spike, seed=nq + nkv, verbose=False)
    assert not m.errors, m.errors[:5]
    # (spiked rows: O is one V row of magnitude 2 .. 4, whose 16-bit rounding alone is 1e-3 / 8e-3)
    assert err <= ((1.6e-2 if spike else 4e-3) if bf16 else (3e-3 if spike else 1e-3)), err
    assert lse_err <= 1e-4, lse_err
    # the head-dim-128 fast bodies are sum-check bodies: P = inf (the hard spikes) must have sent the item through the safe-mode redo, once;
    # everything else — the moderate spikes included — is repaired in place
    assert m.redos == (1 if spike in (True, 3) else 0), m.redos


def test_sum_check_bodies_drop_the_row_max_stream():
    """What the sum-check variant is for: the fast bodies of the head-dim-128 kernels issue no v_max3 and no v_permlane32_swap (the row-max stream of
    the max-first bodies: 2 x 22 instructions per tile), three instructions per q block decide instead; opt=maxfirst restores the old bodies."""
    import fwd_d128_gen as gen
    for opt in ((), ("ct",)):
        new, old = gen.Gen(False, opt=opt), gen.Gen(False, opt=opt + ("maxfirst",))
        pn, po = new.build(), old.build()

        def fast_loop(prog):
            names = [i.ops[0].name if i.op == "label" else None for i in prog.ins]
            a, b = names.index("fast0"), names.index("dispatch")
            return prog.ins[a:b]
        fn, fo_ = fast_loop(pn), fast_loop(po)
        count = lambda body, op: sum(1 for i in body if i.op == op)  # noqa: E731
        assert count(fn, "v_max3_f32") == 0 and count(fn, "v_permlane32_swap_b32") == 0
        assert count(fo_, "v_max3_f32") == 2 * 2 * 15 and count(fn, "v_exp_f32") == count(fo_, "v_exp_f32") == 2 * 64
        assert count(fn, "v_mfma_f32_32x32x16_f16") == count(fo_, "v_mfma_f32_32x32x16_f16") == 128
        n_new = sum(1 for i in fn if i.op not in ("label", "raw"))
        n_old = sum(1 for i in fo_ if i.op not in ("label", "raw"))
        assert n_old - n_new >= 2 * 36, (n_old, n_new)                   # two bodies, >= 36 issue slots each
        # a check lands after its q block's PV MFMAs (the rare block rescales those accumulators); when it lands past MFMA 32 + 2 qb the next tile's
        # scores were formed with the old C tuple and get their shift at the start of the next body (S_FIX / rare_fix): both cases are emulated above
        assert all(g >= 16 * qb + 15 for (name, qb), g in new.check_gaps.items())


SEAMS = [
    # Nq, [Nkv per item], [q block per item], bf16
    (512, [256, 256], [0, 1], False),
    (512, [64, 64, 64], [0, 1, 0], False),        # one tile per item: the staging body is a head body
    (512, [100, 128], [1, 0], False),             # two tiles, ragged first item
    (768, [640, 640, 600], [0, 2, 1], False),     # fast loop on both sides of two seams
    (256, [192, 192], [0, 0], True),              # odd number of tiles (the parities of the rings flip at the seam)
]


@pytest.mark.parametrize("seam", SEAMS)
def test_persistent_workgroup_seams(seam, variant):
    """One workgroup runs the statement for several items in a row (fa2_fwd_d128.hip.h's persistent loop): the last two bodies
    of an item stage the next item's Q fragments and first K / V tiles, the next statement skips its loads.  Every item must
    match dense attention and the emulator must see no hazard (loads in flight across the seam, LDS ring reuse)."""
    import numpy as np
    nq, nkvs, qblks, bf16 = seam
    rng = np.random.default_rng(nq + sum(nkvs))
    items = [(rng.standard_normal((nq, 128)), rng.standard_normal((nkv, 128)), rng.standard_normal((nkv, 128)), qb)
             for nkv, qb in zip(nkvs, qblks)]
    outs, m = harness.run_items(items, False, bf16=bf16)
    assert not m.errors, m.errors[:5]
    for (q, k, v, qb), (o, lse) in zip(items, outs):
        r0 = qb * 256
        o_ref, lse_ref = harness.dense(q[r0:r0 + o.shape[0]], k, v, False, bf16=bf16, row0=r0, pre=bool(variant))
        assert np.abs(o - o_ref).max() <= (4e-3 if bf16 else 1e-3)
        assert np.abs(lse - lse_ref).max() <= 1e-4


@pytest.mark.parametrize("hd", [128, 64])
def test_tail_prescale_option_is_kept_correct(hd):
    """opt=qpre (measured, not shipped — profiles/r16_kbench_eq2_tail_prescale_ab.txt): folded-scale kernels prescale the NEXT item's Q fragments between the
    MFMAs of the item's last body (TCQ / STQ variants) and the next statement's entry skips its prescale.  Non-causal and causal seams (waves that
    finish early run the STQ variant), items of one tile (the seam is a head body)."""
    import numpy as np
    saved = harness.HD, harness.OPT
    harness.HD, harness.OPT = hd, ("ct", "qpre")
    harness._PROGS.clear()
    try:
        rng = np.random.default_rng(hd + 7)
        items = [(rng.standard_normal((512, hd)), rng.standard_normal((nkv, hd)), rng.standard_normal((nkv, hd)), qb) for nkv, qb in zip([640, 64, 200], [0, 1, 0])]
        outs, m = harness.run_items(items, False)
        assert not m.errors, m.errors[:5]
        for (q, k, v, qb), (o, lse) in zip(items, outs):
            o_ref, lse_ref = harness.dense(q[qb * 256:qb * 256 + o.shape[0]], k, v, False, row0=qb * 256, pre=True)
            assert np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= 1e-4
        q, k, v = (rng.standard_normal((768, hd)) for _ in range(3))
        outs, m = harness.run_items([(q, k, v, 2), (q, k, v, 0), (q, k, v, 1)], True)
        assert not m.errors, m.errors[:5]
        o_ref, _ = harness.dense(q, k, v, True, pre=True)
        for qb, (o, _) in zip([2, 0, 1], outs):
            assert np.abs(o - o_ref[qb * 256:qb * 256 + 256]).max() <= 1e-3
    finally:
        harness.HD, harness.OPT = saved
        harness._PROGS.clear()


D64_CASES = [
    # Nq, Nkv, q block, causal, bf16, spike: the head-dim-64 body of the same generator (hd=64: 16 + 8 PV-phase MFMAs — the row sums
    # ride the matrix pipe — and 16 QK-phase MFMAs per tile, 128-byte tile rows)
    (256, 256, 0, False, False, False),
    (256, 64, 0, False, False, False),
    (256, 100, 0, False, True, False),
    (256, 640, 0, False, False, False),
    (200, 333, 0, False, False, False),
    (512, 512, 1, True, False, False),
    (256, 77, 0, True, True, False),
    (256, 704, 0, False, False, True),        # hard spikes: the sum-check bodies (head dim 64 too since 0.9: f32 row sums, no lmfma) meet P = inf -> safe-mode redo
    (256, 704, 0, False, False, 2),           # moderate spikes: repaired in place
    (256, 704, 0, False, True, 3),            # growth of 121 .. 126.6 octaves: redo
]


@pytest.fixture(params=[(), ("ct",)], ids=["f32-scale", "folded-scale"])
def hd64(request):
    """Both head-dim-64 bodies: the library ships the folded-scale one for fp16 (Q * scale*log2e rounded once to fp16, pure_torch_ver.py:61;
    the reference maximum enters the first QK^T k-step as its C operand) and the f32-scale one for bf16; either is a correct body in both dtypes."""
    saved = harness.HD, harness.OPT
    harness.HD, harness.OPT = 64, request.param
    harness._PROGS.clear()
    yield request.param
    harness.HD, harness.OPT = saved
    harness._PROGS.clear()


@pytest.mark.parametrize("case", D64_CASES)
def test_generated_d64_block_matches_dense_attention(case, hd64):
    nq, nkv, qblk, causal, bf16, spike = case
    err, lse_err, m = harness.check(nq, nkv, qblk, causal, bf16=bf16, spike=spike, seed=nq + nkv, verbose=False)
    assert not m.errors, m.errors[:5]
    assert err <= ((1.6e-2 if spike else 6e-3) if bf16 else (3e-3 if spike else 1e-3)), err
    assert lse_err <= 1e-4, lse_err                                # (f32 row sums since 0.9; the lmfma bodies of 0.8 summed the ROUNDED P: 4e-3 / 1e-3)
    assert m.redos == (1 if spike in (True, 3) else 0), m.redos


def test_d64_persistent_items_and_causal_pairs(hd64):
    import numpy as np
    rng = np.random.default_rng(5)
    items = [(rng.standard_normal((768, 64)), rng.standard_normal((nk, 64)), rng.standard_normal((nk, 64)), qb) for nk, qb in zip([640, 640, 600], [0, 2, 1])]
    outs, m = harness.run_items(items, False)
    assert not m.errors, m.errors[:5]
    for (q, k, v, qb), (o, lse) in zip(items, outs):
        o_ref, _ = harness.dense(q[qb * 256:qb * 256 + o.shape[0]], k, v, False, row0=qb * 256, pre=bool(hd64))
        assert np.abs(o - o_ref).max() <= 1e-3
    q, k, v = (rng.standard_normal((1024, 64)) for _ in range(3))
    outs, m = harness.run_items([(q, k, v, 3), (q, k, v, 0)], True)
    assert not m.errors, m.errors[:5]
    o_ref, _ = harness.dense(q, k, v, True, pre=bool(hd64))
    for qb, (o, _) in zip([3, 0], outs):
        assert np.abs(o - o_ref[qb * 256:qb * 256 + 256]).max() <= 1e-3


CAUSAL_SEAMS = [
    # Nq = Nkv, [q block per item]: the pairs a causal launch hands to one workgroup — a long block, then its short partner
    (1024, [3, 0], False),            # 16 tiles, then 4: the ring parities and the per-wave tile counts differ across the seam
    (768, [2, 0, 1], True),           # odd block count: (2, 0) is a unit, the middle block comes alone in the next unit
    (512, [1, 0, 1, 0], False),       # two units in a row
]


@pytest.mark.parametrize("seam", CAUSAL_SEAMS)
def test_persistent_workgroup_causal_pairs(seam):
    """Causal launches hand a workgroup PAIRS of q blocks of one head (fa2_fwd_d128.hip.h): items of different length run back to
    back through the item seam (the next item's Q / K(0) / K(1) / V(0) are fetched while waves of the current item that finished
    early only stage and sync).  Every item must match dense causal attention, with no hazard."""
    import numpy as np
    n, qblks, bf16 = seam
    rng = np.random.default_rng(n + len(qblks))
    q, k, v = rng.standard_normal((n, 128)), rng.standard_normal((n, 128)), rng.standard_normal((n, 128))
    items = [(q, k, v, qb) for qb in qblks]
    outs, m = harness.run_items(items, True, bf16=bf16)
    assert not m.errors, m.errors[:5]
    for (_, _, _, qb), (o, lse) in zip(items, outs):
        r0 = qb * 256
        o_ref, lse_ref = harness.dense(q, k, v, True, bf16=bf16)
        assert np.abs(o - o_ref[r0:r0 + o.shape[0]]).max() <= (1e-2 if bf16 else 1e-3)      # (bf16: randn V, rows with one or two visible keys)
        assert np.abs(lse - lse_ref[r0:r0 + o.shape[0]]).max() <= 1e-4


def test_emulator_flags_a_missing_wait():
    """The checker itself: drop the lgkmcnt wait in front of the QK^T phase and the emulator must object."""
    import fwd_d128_gen as gen
    prog = gen.Gen(False, opt=harness.OPT).build()
    idx = [i for i, ins in enumerate(prog.ins) if ins.op == "s_waitcnt" and ins.mods == {"lgkmcnt": 0}]
    assert idx
    key = (False, harness.HD, harness.M16)            # (the harness's program cache: dtype, head dim, MFMA tile)
    saved = harness._PROGS.get(key)
    try:
        for i in idx[:-1]:
            prog.ins[i].mods = {"lgkmcnt": 15}
        harness._PROGS[key] = prog
        _, _, m = harness.check(256, 256, 0, False, verbose=False)
        assert any("in flight" in e for e in m.errors)
    except Exception as e:      # the emulator may also abort on the error flood
        assert "in flight" in str(e)
    finally:
        if saved is not None:
            harness._PROGS[key] = saved
        else:
            harness._PROGS.pop(key, None)


@pytest.mark.parametrize("opt", [(), ("ct",)])
def test_generated_text_assembles_for_gfx950(opt, tmp_path):
    """Every line of the rendered body goes through the gfx950 assembler (operand classes, constant-bus limits, offsets):
    the emulator interprets instruction objects, so this is the check that the TEXT is legal."""
    import re
    import shutil
    import subprocess
    import fwd_d128_gen as gen
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    subst = {0: "v0", 1: "v1", 2: "v2", 3: "s2", 4: "s[36:39]", 5: "s[4:7]", 6: "s[8:11]", 7: "v6", 8: "v7", 9: "v8", 10: "v9", 11: "v10",
             12: "v11", 13: "s12", 14: "s13", 15: "s14", 16: "s15", 17: "s16", 18: "s17", 19: "s18", 20: "s19", 21: "v12", 22: "s20",
             23: "s21", 24: "s3", 25: "s[40:43]", 26: "s[24:27]", 27: "s[28:31]", 28: "s[32:33]"}
    for bf16, hd in ((False, 128), (True, 128), (False, 64), (True, 64)):
        text = "\n".join(gen.Gen(bf16, hd=hd, opt=opt).build().text_lines())
        text = re.sub(r"%(\d+)", lambda m: subst[int(m.group(1))], text.replace("%=", "0"))
        src = tmp_path / ("body_%d_%d.s" % (bf16, hd))
        src.write_text(text + "\n")
        res = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[:2000]


@pytest.mark.parametrize("hd,opt", [(128, ("ct",)), (128, ()), (64, ("ct",)), (64, ())])
def test_kv_split_part_epilogue(hd, opt):
    """KV-split parts in the persistent workgroups (fa2_fwd_ws): an item flagged as a part (flag bit 3) sweeps a KV range and its epilogue stores
    the NORMALISED f32 tile straight to the workspace (1-KiB stores in the layout of the HIP kernels' parts) instead of the 16-bit tile in LDS.
    A whole item, then two parts of another item through the item seam: every output must match dense attention over its own KV range, the part
    tiles must be fully written (the workspace starts NaN-filled), and the emulator must see no hazard (stores in flight at the seam)."""
    import numpy as np
    saved = harness.HD, harness.OPT
    harness.HD, harness.OPT = hd, opt
    harness._PROGS.clear()
    try:
        rng = np.random.default_rng(hd)
        q, k, v = rng.standard_normal((512, hd)), rng.standard_normal((448, hd)), rng.standard_normal((448, hd))
        items = [(q, k, v, 1), (q, k[:256], v[:256], 0, True), (q, k[256:], v[256:], 0, True)]      # parts: tiles [0, 4) and [4, 7) of q block 0
        outs, m = harness.run_items(items, False)
        assert not m.errors, m.errors[:5]
        for (item, (o, lse)) in zip(items, outs):
            qq, kk, vv, qb = item[:4]
            o_ref, lse_ref = harness.dense(qq[qb * 256:qb * 256 + 256], kk, vv, False, pre=bool(opt))
            assert np.isfinite(o).all()
            assert np.abs(o - o_ref).max() <= 1e-3
            assert np.abs(lse - lse_ref).max() <= (1e-3 if hd == 64 else 1e-4)
        # merged like fwd_combine_kernel, the two parts are the whole item
        (o1, l1), (o2, l2) = outs[1], outs[2]
        lse = np.logaddexp2(l1, l2)
        o = o1 * np.exp2(l1 - lse)[:, None] + o2 * np.exp2(l2 - lse)[:, None]
        o_ref, lse_ref = harness.dense(q[:256], k, v, False, pre=bool(opt))
        assert np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= 1e-3
    finally:
        harness.HD, harness.OPT = saved
        harness._PROGS.clear()
