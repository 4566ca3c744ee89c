"""The forward body built on v_mfma_f32_16x16x32 (csrc/gen/fwd_m16_gen.py, round 5) on the instruction-level emulator (tools/asm_emu.py): the same
checks the 32x32x16 bodies get in tests/test_asm_emu.py — every body variant against float64 attention with the hazard model on, persistent item seams
(non-causal and causal pairs), the sum-check repair / redo paths — plus the text through the gfx950 assembler.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
import asm_emu_harness as harness  # noqa: E402


# ("ct", "lm"): the folded bodies as shipped — row sums on the matrix pipe, fast bodies without adds or a check (FA2_CONTRACT_LSUM_P16: the LSE carries
# the rounding of P; what the check guarded against sends the item through the safe-mode redo); ("ct",): the folded bodies with the sum check (gen opt=nolm)
# ("lm",): the f32-scale bodies with the row sums on the matrix pipe (constants in a[224:255], K fragments in the 32-register pool): what bf16 launches and
# every other f32-scale call that is not flagged FA2_FLAG_EXACT_SCALE run
@pytest.fixture(autouse=True, params=[(), ("ct",), ("ct", "lm"), ("lm",)], ids=["f32-scale", "folded-scale", "folded-lm", "f32-scale-lm"])
def m16(request):
    saved = harness.HD, harness.OPT, harness.M16
    harness.HD, harness.OPT, harness.M16 = 128, request.param, True
    harness._PROGS.clear()
    yield request.param
    harness.HD, harness.OPT, harness.M16 = saved
    harness._PROGS.clear()


CASES = [
    # Nq, Nkv, q block, causal, bf16, spike
    (256, 64, 0, False, False, False),         # one tile: head bodies + TC only
    (256, 128, 0, False, False, False),
    (256, 192, 0, False, True, False),
    (256, 256, 0, False, False, False),        # first fast body
    (256, 640, 0, False, True, False),
    (200, 333, 0, False, False, False),        # ragged Nq and Nkv
    (256, 77, 0, True, False, False),          # causal + ragged: the limits of both kinds
    (512, 512, 1, True, False, False),         # causal diagonal: waves finish at different tiles
    (1024, 1024, 3, True, True, False),
    (256, 704, 0, False, False, True),         # hard spikes: P = inf in a sum-check body -> flag -> safe-mode redo
    (256, 704, 0, False, False, 2),            # moderate spikes: repaired in place, rows of both q groups of a lane
    (512, 768, 1, True, False, 2),
    (256, 704, 0, False, True, 3),             # growth of 121 .. 126.6 octaves: redo
]


@pytest.mark.parametrize("case", CASES)
def test_m16_block_matches_dense_attention(case, m16):
    Nq, Nkv, qblk, causal, bf16, spike = case
    err, lerr, m = harness.check(Nq, Nkv, qblk, causal, bf16=bf16, seed=Nq + Nkv, spike=spike, verbose=False)
    assert not m.errors, m.errors[:5]
    tol = (8e-3 if bf16 else 1e-3) * (3 if spike else 1)
    lm = "lm" in m16
    # (lm, bf16: a row with one dominant key carries the rounding of that P — up to log2(1 + 2^-8) = 5.6e-3 — unless the reference sits exactly on its score:
    #  tests/conftest.py LSE_TOL_P16_BF16; the spike cases make such rows)
    assert err <= tol and lerr <= (((6e-3 if spike else 4e-3) if bf16 else 1e-3) if lm else 1e-4), (err, lerr)
    if spike == 2 and not lm:
        assert m.redos == 0            # repaired in place by the sum check's rare block
    if spike in (True, 3):
        # the sum-check bodies cannot repair P = inf / growth beyond 120 octaves: redo.  The lm bodies (round 6) see the row sums of a tile AHEAD of its
        # P.V, form the tile again in place with the max-first streams and finish the sweep on the max-first bodies: nothing is redone (Gen16.lm_repair)
        assert m.redos == (0 if lm else 1)
    if lm and spike:
        assert getattr(m, "repairs", 0) >= 1 or spike == 2


def test_m16_fast_bodies_carry_no_cross_lane_instruction():
    """What makes the other MFMA tile affordable: a Q row is spread over four lanes, and the fast bodies never look across them — per-lane partial
    sums and a per-lane sum check; every permlane swap of the block is in a head / tail body, an out-of-line block or the epilogue."""
    import fwd_m16_gen
    g = fwd_m16_gen.Gen16(False, opt=harness.OPT)
    prog = g.build()
    names = [i.ops[0].name if i.op == "label" else None for i in prog.ins]
    lo, hi = names.index("fast0"), names.index("dispatch")
    ops = [i.op for i in prog.ins[lo:hi]]
    lm = "lm" in harness.OPT
    assert sum(o.startswith("v_mfma_f32_16x16x32") for o in ops) == 2 * (136 if lm else 128)
    assert not any(o.startswith("v_permlane") for o in ops)
    if lm:
        # exp + pack and nothing else: no add, no row maximum; per tile ONE look at the wave's four row sums of the tile just packed — v_max3 + v_max +
        # v_cmp + branch, behind the body's last MFMA and AHEAD of the tile's P.V (round 6: the in-place repair, Gen16.lm_repair, is out of line)
        assert not any(o in ("v_add_f32", "v_cmp_nge_f32") for o in ops)
        assert [o for o in ops if o.startswith("v_max") or o.startswith("v_cmp") or o == "s_cbranch_vccnz"] == 2 * ["v_max3_f32", "v_max_f32", "v_cmp_ngt_f32", "s_cbranch_vccnz"]
    else:
        assert not any(o.startswith("v_max") for o in ops)
        assert sum(o == "v_exp_f32" for o in ops) == 2 * 64 and sum(o.startswith("v_cvt_pk") for o in ops) == 2 * 32


SEAMS = [
    (512, [256, 256], [0, 1], False, False),
    (512, [64, 64, 64], [0, 1, 0], False, False),        # one tile per item: the seam is a head body
    (768, [640, 640, 600], [0, 2, 1], False, False),     # the next item's Q fully staged by the fast bodies
    (256, [192, 192], [0, 0], False, True),
    (1024, [1024, 1024], [3, 0], True, False),           # causal pair: a long block, then its short partner
    (768, [768, 768, 768], [2, 0, 1], True, True),
]


@pytest.mark.parametrize("seam", SEAMS)
def test_m16_persistent_workgroup_seams(seam):
    nq, nkvs, qblks, causal, bf16 = seam
    rng = np.random.default_rng(nq + sum(nkvs))
    if causal:
        q, k, v = (rng.standard_normal((nq, 128)) for _ in range(3))
        items = [(q, k, v, qb) for qb in qblks]
    else:
        items = [(rng.standard_normal((nq, 128)), rng.standard_normal((nkv, 128)), rng.standard_normal((nkv, 128)), qb) for nkv, qb in zip(nkvs, qblks)]
    outs, m = harness.run_items(items, causal, bf16=bf16)
    assert not m.errors, m.errors[:5]
    for (q, k, v, qb), (o, lse) in zip(items, outs):
        r0 = qb * 256
        o_ref, lse_ref = harness.dense(q[r0:r0 + o.shape[0]], k, v, causal, bf16=bf16, row0=r0, pre="ct" in harness.OPT)
        assert np.abs(o - o_ref).max() <= (8e-3 if bf16 else 1.1e-3)
        assert np.abs(lse - lse_ref).max() <= ((4e-3 if bf16 else 1e-3) if "lm" in harness.OPT else 1e-4)


def _assemble(hd, tmp_path):
    import re
    import shutil
    import subprocess
    import fwd_m16_gen
    mc = shutil.which("llvm-mc") or "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    subst = {0: "v0", 1: "v1", 2: "v2", 3: "s2", 4: "s[36:39]", 5: "s[4:7]", 6: "s[8:11]", 7: "v6", 8: "v7", 9: "v8", 10: "v9", 11: "v10",
             12: "v11", 13: "s12", 14: "s13", 15: "s14", 16: "s15", 17: "s16", 18: "s17", 19: "s18", 20: "s19", 21: "v12", 22: "s20",
             23: "s21", 24: "s3", 25: "s[40:43]", 26: "s[24:27]", 27: "s[28:31]", 28: "s[32:33]"}
    for bf16 in (False, True):
        text = "\n".join(fwd_m16_gen.Gen16(bf16, hd=hd, opt=harness.OPT).build().text_lines())
        text = re.sub(r"%(\d+)", lambda m_: subst[int(m_.group(1))], text.replace("%=", "0"))
        src = tmp_path / ("m16_%d_%d.s" % (hd, bf16))
        src.write_text(text + "\n")
        res = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[:2000]


def test_m16_text_assembles_for_gfx950(tmp_path):
    _assemble(128, tmp_path)


def test_m16_kv_split_part_epilogue():
    """KV-split parts (fa2_fwd_ws; flag bit 3): a whole item, then two parts of another item through the item seam — each part leaves its NORMALISED f32
    tile in its workspace tile, in the layout of every part in this library (float (((dt*4 + g) * 256 + row) * 8 + 4*hi + e)), and its partial LSE;
    merged like fwd_combine_kernel they are the whole item.  The workspace starts NaN-filled: every element must be written."""
    rng = np.random.default_rng(128)
    q, k, v = rng.standard_normal((512, 128)), rng.standard_normal((448, 128)), rng.standard_normal((448, 128))
    items = [(q, k, v, 1), (q, k[:256], v[:256], 0, True), (q, k[256:], v[256:], 0, True)]      # parts: tiles [0, 4) and [4, 7) of q block 0
    outs, m = harness.run_items(items, False)
    assert not m.errors, m.errors[:5]
    pre = "ct" in harness.OPT
    for (item, (o, lse)) in zip(items, outs):
        qq, kk, vv, qb = item[:4]
        o_ref, lse_ref = harness.dense(qq[qb * 256:qb * 256 + 256], kk, vv, False, pre=pre)
        assert np.isfinite(o).all()
        assert np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if "lm" in harness.OPT else 1e-4)
    (o1, l1), (o2, l2) = outs[1], outs[2]
    lse = np.logaddexp2(l1, l2)
    o = o1 * np.exp2(l1 - lse)[:, None] + o2 * np.exp2(l2 - lse)[:, None]
    o_ref, lse_ref = harness.dense(q[:256], k, v, False, pre=pre)
    assert np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= 1e-3


def test_m16_generator_rejects_a_v_read_window_that_reaches_into_the_pv_phase():
    """The V^T read window is an input (tools/kbench.py sweeps it): a read placed ahead of the last P.V MFMA of the body that takes its registers would hand
    that MFMA the next tile's fragment.  The generator refuses such a schedule (Gen16.body) instead of leaving it to a wrong result on the GPU."""
    import fwd_m16_gen
    fwd_m16_gen.Gen16(False, opt=("ct", "lm"), lm_vread=(56.0, 110.0)).build()         # k-step 0's registers are free from gap 54 on
    with pytest.raises(ValueError, match="illegal schedule"):
        fwd_m16_gen.Gen16(False, opt=("ct", "lm"), lm_vread=(30.0, 60.0)).build()


# ---- head dim 64 (same generator, hd = 64): the folded body with the row sums on the matrix pipe is what the library ships for the launches that fold
# the scale (fwd_asm.cpp); the other two variants are generated and kept correct (they lose to the 32x32x16 body on the GPU)
@pytest.fixture(params=[(), ("ct",), ("ct", "lm")], ids=["d64-f32-scale", "d64-folded", "d64-folded-lm"])
def m16_d64(request, m16):
    if m16:
        pytest.skip("one pass over the head-dim-64 variants is enough")      # (the autouse fixture above runs every test once per head-dim-128 variant)
    harness.HD, harness.OPT = 64, request.param
    harness._PROGS.clear()
    yield request.param


CASES_D64 = [
    (256, 64, 0, False, False, False),
    (256, 640, 0, False, True, False),
    (200, 333, 0, False, False, False),
    (256, 77, 0, True, False, False),
    (512, 512, 1, True, False, False),
    (256, 704, 0, False, False, True),         # hard spikes: safe-mode redo
]


@pytest.mark.parametrize("case", CASES_D64)
def test_m16_head_dim_64_block_matches_dense_attention(case, m16_d64):
    Nq, Nkv, qblk, causal, bf16, spike = case
    err, lerr, m = harness.check(Nq, Nkv, qblk, causal, bf16=bf16, seed=Nq + Nkv, spike=spike, verbose=False)
    assert not m.errors, m.errors[:5]
    lm = "lm" in m16_d64
    assert err <= (8e-3 if bf16 else 1e-3) * (3 if spike else 1) and lerr <= ((4e-3 if bf16 else 1e-3) if lm else 1e-4), (err, lerr)
    if spike:
        assert m.redos == (0 if lm else 1) and (not lm or m.repairs >= 1)       # (lm, round 6: repaired in place — Gen16.lm_repair)


def test_m16_head_dim_64_persistent_seams_and_text(m16_d64, tmp_path):
    rng = np.random.default_rng(64)
    items = [(rng.standard_normal((768, 64)), rng.standard_normal((nkv, 64)), rng.standard_normal((nkv, 64)), qb) for nkv, qb in ((640, 0), (64, 2), (600, 1))]
    outs, m = harness.run_items(items, False)
    assert not m.errors, m.errors[:5]
    for (q, k, v, qb), (o, lse) in zip(items, outs):
        o_ref, lse_ref = harness.dense(q[qb * 256:qb * 256 + o.shape[0]], k, v, False, row0=qb * 256, pre="ct" in harness.OPT)
        assert np.abs(o - o_ref).max() <= 1.1e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if "lm" in m16_d64 else 1e-4)
    q, k, v = (rng.standard_normal((768, 64)) for _ in range(3))
    outs, m = harness.run_items([(q, k, v, 2), (q, k, v, 0)], True)                  # a causal pair unit
    assert not m.errors, m.errors[:5]
    for qb, (o, lse) in zip((2, 0), outs):
        o_ref, lse_ref = harness.dense(q[qb * 256:qb * 256 + 256], k, v, True, row0=qb * 256, pre="ct" in harness.OPT)
        assert np.abs(o - o_ref).max() <= 1.1e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if "lm" in m16_d64 else 1e-4)
    _assemble(64, tmp_path)


def test_m16_head_dim_64_kv_split_part_epilogue(m16_d64):
    """KV-split parts at head dim 64 (SDXL's 64 x 64 self-attention is 320 workgroups: fa2_fwd_ws splits its last round): a whole item, then two parts of
    another through the item seam; merged like fwd_combine_kernel the parts are the whole item."""
    rng = np.random.default_rng(65)
    q, k, v = rng.standard_normal((512, 64)), rng.standard_normal((448, 64)), rng.standard_normal((448, 64))
    items = [(q, k, v, 1), (q, k[:256], v[:256], 0, True), (q, k[256:], v[256:], 0, True)]
    outs, m = harness.run_items(items, False)
    assert not m.errors, m.errors[:5]
    pre = "ct" in harness.OPT
    ltol = 1e-3 if "lm" in m16_d64 else 1e-4
    for (item, (o, lse)) in zip(items, outs):
        qq, kk, vv, qb = item[:4]
        o_ref, lse_ref = harness.dense(qq[qb * 256:qb * 256 + 256], kk, vv, False, pre=pre)
        assert np.isfinite(o).all() and np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= ltol
    (o1, l1), (o2, l2) = outs[1], outs[2]
    lse = np.logaddexp2(l1, l2)
    o = o1 * np.exp2(l1 - lse)[:, None] + o2 * np.exp2(l2 - lse)[:, None]
    o_ref, lse_ref = harness.dense(q[:256], k, v, False, pre=pre)
    assert np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= 1e-3


def test_m16_items_after_a_redo_start_in_safe_mode():
    """The shell's sticky bit (fa2_fwd_d128.hip.h): once an item of a persistent workgroup went through the safe-mode redo, the items the workgroup
    takes after it start in safe mode — prefetched through the item seam like any other (flag bits 0 and 4 together) — and are exact."""
    rng = np.random.default_rng(99)
    q = rng.standard_normal((512, 128)) * 3
    k = rng.standard_normal((704, 128)) * 3
    v = rng.standard_normal((704, 128))
    k[200] = q[5] * 4                       # a hard spike in the first item: P = inf in a fast body -> flag -> redo
    items = [(q, k, v, 0), (q, k, v, 1), (rng.standard_normal((256, 128)), rng.standard_normal((320, 128)), rng.standard_normal((320, 128)), 0)]
    outs, m = harness.run_items(items, False)
    assert not m.errors, m.errors[:5]
    lm = "lm" in harness.OPT
    if lm:
        # round 6: the tile is repaired in place (no redo); the wave asks for the sticky bit all the same, and the items after it never enter a fast body
        assert m.redos == 0 and m.repairs >= 1 and [f & 48 for f in m.item_flags] == [0, 48, 48]
    else:
        assert m.redos == 1                 # the first item only; the two after it never entered a fast body
    for (qq, kk, vv, qb), (o, lse) in zip(items, outs):
        o_ref, lse_ref = harness.dense(qq[qb * 256:qb * 256 + 256], kk, vv, False, pre="ct" in harness.OPT)
        assert np.abs(o - o_ref).max() <= 3e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if lm else 1e-4)


REPAIRS = [
    # opt, head dim of the body, D (None: the body's), bf16, causal, Nq, Nkv, q block, spikes [(row, kv, factor)], the repair blocks that must run
    (("ct", "lm"), 128, None, False, False, 256, 704, 0, [(5, 200, 2.7)], {"lm_repair_0"}),             # tile 3: found by F0 (t = 2)
    (("ct", "lm"), 128, None, False, False, 256, 704, 0, [(70, 150, 2.7)], {"lm_repair_1"}),            # tile 2: found by F1 (t = 1), wave 1
    (("ct", "lm"), 128, None, False, False, 256, 704, 0, [(5, 200, 2.7), (40, 130, 2.0), (200, 330, 3.0)], {"lm_repair_0", "lm_repair_1"}),
    (("ct", "lm"), 128, None, True, False, 256, 704, 0, [(5, 200, 5.0)], {"lm_repair_0"}),              # bf16 folded: a share beyond 2^40
    (("lm",), 128, None, True, False, 256, 704, 0, [(5, 200, 5.0), (130, 150, 5.0)], {"lm_repair_0", "lm_repair_1"}),   # the f32-scale bodies (bf16's default)
    (("lm",), 128, None, False, False, 256, 704, 0, [(5, 200, 2.7)], {"lm_repair_0"}),
    (("ct", "lm"), 128, None, False, True, 1024, 1024, 3, [(800, 200, 2.7), (1000, 400, 2.7)], {"lm_repair_0", "lm_repair_1"}),   # causal: waves of different sweep lengths
    (("ct", "lm"), 128, 96, False, False, 256, 704, 0, [(5, 200, 3.0)], {"lm_repair_0"}),               # a head dim below the body's: the padded columns read as zeros
    (("lm",), 128, 120, True, False, 256, 704, 0, [(5, 150, 6.0)], {"lm_repair_1"}),
    (("ct", "lm"), 64, None, False, False, 256, 704, 0, [(5, 200, 3.5), (70, 150, 3.5)], {"lm_repair_0", "lm_repair_1"}),
    (("ct", "lm"), 64, 40, False, False, 256, 704, 0, [(5, 200, 5.0)], {"lm_repair_0"}),
    (("lm",), 64, None, True, False, 256, 704, 0, [(5, 200, 8.0)], {"lm_repair_0"}),
]


@pytest.mark.parametrize("case", REPAIRS)
def test_m16_lm_bodies_repair_an_overflowing_tile_in_place(case, m16):
    """Round 6.  The fast bodies of the lm kernels never move the reference; a score 16 octaves above it (fp16: P = inf) used to send the whole ITEM through
    a second, safe-mode sweep.  Now the row-sum links of a tile ride ahead of its P.V, the fast body looks at them, and a wave that finds inf / NaN / a
    share beyond 2^40 forms that one tile again — K fragments straight from memory, the max-first streams — and finishes its sweep on the max-first
    bodies (Gen16.lm_repair): exact results, nothing redone, the sticky request raised.  Both parities of the fast loop, every kernel kind that ships."""
    if m16:
        pytest.skip("one pass is enough")
    import asm_emu
    opt, hd, d, bf16, causal, Nq, Nkv, qblk, spikes, blocks = case
    saved = harness.HD, harness.OPT, harness.DTRIM
    harness.HD, harness.OPT, harness.DTRIM = hd, opt, d
    harness._PROGS.clear()
    hits = set()
    step = asm_emu.Machine.step

    def counting_step(self, w):
        ins = self.ins[w.pc]
        if ins.op == "label" and ins.ops[0].name in ("lm_repair_0", "lm_repair_1"):
            hits.add(ins.ops[0].name)
        return step(self, w)
    asm_emu.Machine.step = counting_step
    try:
        D = d or hd
        rng = np.random.default_rng(Nq + Nkv + D)
        q, k, v = rng.standard_normal((Nq, D)), rng.standard_normal((Nkv, D)), rng.standard_normal((Nkv, D))
        for (row, kv, f) in spikes:
            k[kv] = q[row] * f * (128.0 / D) ** 0.5
        # a second item behind it: starts in safe mode (sticky), prefetched through the seam like any other
        items = [(q, k, v, qblk), (q, k[:320], v[:320], 0)]
        outs, m = harness.run_items(items, causal, bf16=bf16)
        assert not m.errors, m.errors[:5]
        assert m.redos == 0 and m.repairs >= 1 and hits == blocks, (m.redos, m.repairs, hits)
        assert [f & 48 for f in m.item_flags] == [0, 48]
        for (qq, kk, vv, qb), (o, lse) in zip(items, outs):
            r0 = qb * 256
            o_ref, lse_ref = harness.dense(qq[r0:r0 + o.shape[0]], kk, vv, causal, bf16=bf16, row0=r0, pre="ct" in opt)
            assert np.isfinite(o).all() and np.isfinite(lse).all()
            assert np.abs(o - o_ref).max() <= (2.4e-2 if bf16 else 3e-3) and np.abs(lse - lse_ref).max() <= (6e-3 if bf16 else 1e-3)
    finally:
        asm_emu.Machine.step = step
        harness.HD, harness.OPT, harness.DTRIM = saved
        harness._PROGS.clear()


@pytest.mark.parametrize("hd,d,opt", [(64, 40, ("ct", "lm")), (64, 56, ("ct", "lm")), (64, 8, ("ct", "lm")), (128, 96, ("ct", "lm")), (128, 120, ("lm",)), (128, 104, ())])
def test_m16_head_dims_below_the_bodys(hd, d, opt, m16):
    """Head dims below the body's (SD 1.5's D = 40 on the 64 body, 96 .. 120 on the 128 body; host.cpp: plan_range): rows of 2 D bytes at any pitch, the
    LDS-DMA offsets in their general form, and a granule the row does not have gets an offset beyond the descriptor — the image's padded columns are
    zero-filled by the load itself (Gen16.trim_offsets).  The emulated matrices sit between NaN guard bands and are contiguous, so a granule that were
    fetched from the neighbouring row, or a stray column of V, would show: O's padded columns must be exact zeros, the real ones match float64."""
    if m16:
        pytest.skip("one pass is enough")
    saved = harness.HD, harness.OPT, harness.DTRIM
    harness.HD, harness.OPT, harness.DTRIM = hd, opt, d
    harness._PROGS.clear()
    try:
        rng = np.random.default_rng(hd + d)
        q, k, v = rng.standard_normal((456, d)), rng.standard_normal((333, d)), rng.standard_normal((333, d))
        items = [(q, k, v, 1), (q, k, v, 0)]                       # ragged Nq and Nkv, two items through the seam
        outs, m = harness.run_items(items, False)
        assert not m.errors, m.errors[:5]
        for (qq, kk, vv, qb), (o, lse) in zip(items, outs):
            o_ref, lse_ref = harness.dense(qq[qb * 256:qb * 256 + o.shape[0]], kk, vv, False, row0=qb * 256, pre="ct" in opt)
            assert o.shape[1] == d and np.abs(o - o_ref).max() <= 1e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if "lm" in opt else 1e-4)
        q, k, v = (rng.standard_normal((512, d)) for _ in range(3))
        outs, m = harness.run_items([(q, k, v, 1), (q, k, v, 0)], True)            # a causal pair unit
        assert not m.errors, m.errors[:5]
        for qb, (o, lse) in zip((1, 0), outs):
            o_ref, lse_ref = harness.dense(q[qb * 256:qb * 256 + 256], k, v, True, row0=qb * 256, pre="ct" in opt)
            assert np.abs(o - o_ref).max() <= 1.1e-3 and np.abs(lse - lse_ref).max() <= (1e-3 if "lm" in opt else 1e-4)
    finally:
        harness.HD, harness.OPT, harness.DTRIM = saved
        harness._PROGS.clear()
