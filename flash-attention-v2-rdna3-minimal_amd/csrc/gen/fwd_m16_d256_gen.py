#!/usr/bin/env python3
"""Forward main block for head dims 136 .. 256 on v_mfma_f32_16x16x32 (round 6) — the bodies, streams and scheduler of fwd_m16_gen.py (reference
counterpart: kernel_fp16.cu:381-508) at another geometry.

Why.  The compiler-scheduled kernels serve head dims above 128 at 0.28 - 0.35 of the MFMA peak (D = 256 non-causal recomputes Q.K^T per 128-column slab
of O).  tools/ubench/mfma_shape_probe.py priced the hand-scheduled structure at this head dim (profiles/r20_mfma_probe_d256_geometry.json): one wave per
SIMD can keep O for 32 Q rows (128 accumulator registers + 64 of Q fragments), a wave-tile of 32 rows x 64 keys is the FLOPs of the head-dim-128 body's
64 x 64 with half the exp / pack work and twice the fragment reads, and the synthetic body runs within 3.5 % of the head-dim-128 one per FLOP.

Shape: workgroup = 4 waves = 128 Q rows, one wave per SIMD, wave = 32 rows = two 16-row groups qg (ONE "q block" of the base generators); KV tiles of 64.
    S^T tile (kg, qg) = sum_ks K[kg rows, 32 d] . Q^T[ks, qg]          8 x 2 x 4 = 64 MFMAs per tile
    O^T tile (dg, qg) += sum_kvs V^T[16 d, 32 kv] . P^T[kvs, qg]       16 x 2 x 2 = 64 MFMAs, + 4 row-sum links (opt=lm)       -> a 132-gap body
  registers: O a[0:127], Q fragments a[128:191], K fragments in a three-k-step pool a[192:239], the links' constants a[240:247];
             S / P banks v[16:79], ALL of a tile's V^T fragments v[80:207] (32 x 4), addresses / softmax state / scratch v[208:255].
  LDS: 2 x K + 2 x V tiles of 32 KiB = 128 KiB and nothing else — no epilogue image (O leaves the registers by bounds-checked buffer stores: 8 bytes
  per lane and (d group, q group)), no Q image (the fragments come straight from memory, once per item).
Bodies are max-first only (the reference kernel's own recurrence, kernel_fp16.cu:434-490; deferred rescale at 2^14): at half the exp / pack work per
MFMA the row-max stream costs half what it does at head dim 128, and the kernel carries no fast loop, no repair and no redo.  f32 scale (the reference
kernel's contract), row sums of the rounded P on the matrix pipe (FA2_CONTRACT_LSUM_P16).  One item per workgroup (no seams): an item is 2 x the
work per tile of a head-dim-128 one.
opt=trim: head dims 136 .. 248 on the same body (the reference zero-pads D on the host, kernel_fp16.cu:763-779).  The rows really have D columns at
whatever pitch the caller's tensors have, so a piece's LDS-DMA source offset takes its general form — piece 0's + ((gl_i - gl_0) << 4) + i * (two rows),
gl_i = the logical granule the lane's slot holds under the image's swizzle — and a granule the row does not have (gl_i >= D / 8) gets an offset beyond
every descriptor: the load returns zeros, the image's padded columns are zero and nothing of a neighbouring row enters a product (fwd_m16_gen.py:
trim_offsets has the argument).  Q fragments and O stores are masked the same way.  The body runs at D / 256 of its rate.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
import fwd_d128_gen as base  # noqa: E402
import fwd_m16_gen as m16  # noqa: E402
import sched  # noqa: E402
from fwd_d128_gen import (A_C, A_KD0, A_KR0, A_KRS, A_KTILE, A_LDSW, A_LIM0, A_LIM1, A_LSE0, A_NTW, A_NTWG, A_VD0, A_VR0, A_VRS, A_VTILE,  # noqa: E402,F401
                          NEG_INF, S_D, S_FLAG, S_KOFF, S_NOVM, S_T, S_TMP, S_TMP2, S_VOFF, S_WAVE, SB)
from fwd_m16_gen import EPX, FS, KR, LS, LSV, MC, TMP, VR  # noqa: E402
from isa import A, V, Arg, Ins, Label, M0, mk  # noqa: E402

# ---- operands (fa2_fwd_d256.hip.h): the numbers the inherited streams use keep their meaning (5 .. 17, 20), the others are this kernel's
A_Q0 = Arg(2)                                      # per-lane byte offset of this lane's 16 bytes (k-step 0) of row n = lane % 16 of a 16-row group: n * pitch + 16 (lane / 16)
A_QW = Arg(3, "s")                                 # byte offset of the wave's first Q row from the head base
A_QRS = Arg(4, "s", 4)                             # buffer descriptor of this head's Q matrix (rows >= Nq read as zeros)
A_KROW2, A_VROW2 = Arg(18, "s"), Arg(19, "s")      # 2 * row bytes - 1024: source stride between the DMA pieces of a wave (a piece is two rows of 512 B)
A_OO0 = Arg(21)                                    # per-lane byte offset into O: n * pitch + 8 (lane / 16)
A_OW = Arg(22, "s")                                # byte offset of the wave's first O row from the head base
A_QT16, A_OT16 = Arg(23, "s"), Arg(24, "s")        # 16 * Q / O row bytes: the second q group
A_ORS = Arg(25, "s", 4)                            # buffer descriptor of this head's O matrix (rows >= Nq are not stored)
A_NG = Arg(26, "s")                                # opt=trim (head dims 136 .. 248): the 16-byte granules a row really has (D / 8)
N_ARGS = 27

KD0, VD0, DT0, DT1 = V(220), V(221), V(222), V(223)   # LDS-DMA source offsets of piece 0 (K, V) and two scratch registers for the other pieces
KG0, VG0, MARK, G4R = V(224), V(225), V(226), V(227)  # opt=trim: the logical granule this lane's slot holds in piece 0 of the K / V image; 0x80000000; lane / 16
VRO = EPX                                          # read addresses of the odd d groups (the conflict-free V image: fwd_m16_gen.Gen16.stream_vread)
ONES = [A(240 + 4 * qg, 4) for qg in range(2)]     # the constant A tuples of the row-sum links


class Geo256:
    HD, ROWB = 256, 512
    SLOT_B = 64 * ROWB                             # one K (or V) tile image: 32 KiB
    NP, RPP = SLOT_B // 4096, 1024 // ROWB         # 8 LDS-DMA pieces of 1 KiB per wave and tile, two tile rows per piece
    K_SLOT, V_BASE = 0, 2 * SLOT_B
    FAIL_OFF = 4 * SLOT_B                          # (flag words: kept for the inherited checks, never read by the shell)
    LDS_BYTES = FAIL_OFF + 16


def OACC(dg, qg):
    return A(4 * (16 * qg + dg), 4)


def QF(qg, ks):
    return A(128 + 4 * (8 * qg + ks), 4)


def KF(kg, ks):                                    # three k-step slots: a read is issued a whole k-step (8 MFMAs) + one slot ahead of its first use
    return A(192 + 16 * (ks % 3) + 4 * kg, 4)


def VF(dg, kvs):
    return V(80 + 4 * (16 * kvs + dg), 4)


class Gen256(m16.Gen16):
    def __init__(self, bf16=False, nks=8, **cfg):
        """nks: 32-column k-steps of Q.K^T the body really runs (opt=trim: 5 .. 8 for head dims <= 160 / 192 / 224 / 256; the d groups of O follow: 2 nks).
        The images, rings and the LDS-DMA are the head-dim-256 ones whatever nks is; k-steps and d groups that hold no real column — their MFMAs, their
        fragment reads — are simply not there: 84 / 100 / 116 / 132 MFMAs per tile."""
        opt = tuple(o for o in cfg.get("opt", ()) if o != "lm") + ("lm",)
        user = dict(cfg)
        user["opt"] = opt
        super().__init__(bf16, hd=128, **user)
        assert not self.ct, "the head-dim-256 bodies scale the f32 product"
        self.trim = "trim" in self.opt
        assert 5 <= nks <= 8 and (nks == 8 or self.trim)
        self.g = Geo256()
        self.NKS16, self.NDG = nks, 2 * nks
        self.kf16, self.vf16, self.oacc16, self.qf16 = KF, VF, OACC, QF
        self.ones16 = ONES
        self.pool = True
        self.vflip, self.vro = True, VRO
        self.npv, self.nqk = 4 * self.NDG, 8 * self.NKS16 + 4
        self.ng = self.npv + self.nqk
        npv, ng = float(self.npv), float(self.ng)
        windows = {"m": (4.0, 24.0), "mmask": (4.0, 40.0), "e": (24.0, ng - 20.0), "vread": (npv + 2.0, ng - 2.0), "kread_ct": (8.0, npv - 4.0), "dma": (2.0, npv), "dmaf": (2.0, ng - 24.0)}
        for k, w in windows.items():
            if k not in cfg:
                self.cfg[k] = w
            elif nks != 8:                   # a window given for the 132-gap body (tools/kbench.py sweeps): the shorter bodies take it to scale
                self.cfg[k] = (cfg[k][0] * ng / 132.0, cfg[k][1] * ng / 132.0)

    # ------------------------------------------------------------------ MFMA lists (one q block)
    def qk_mfmas(self, par):
        out = []
        for ks in range(self.NKS16):
            for h in range(2):
                for kg in range(4):
                    dst = SB(0, par).sub(16 * h + 4 * kg, 4)
                    out.append(mk(self.mfma, dst, KF(kg, ks), QF(h, ks), 0 if ks == 0 else dst, tag="mfma"))
        return out

    def sum_links(self, par_s, mode="acc"):
        b = SB(0, par_s)
        return [mk(self.mfma, LSV, ONES[h], b.sub(16 * h + 8 * kvs, 4), LSV, tag="mfma") for kvs in range(2) for h in range(2)]

    def qk_phase(self, par, s1, s2, fast=False):
        qk = self.qk_mfmas(par) if s2 else [None] * (8 * self.NKS16)
        links = self.sum_links(par ^ 1) if s1 else [None] * 4
        at = [self.ng - 12 + 3 * j for j in range(4)]          # behind the exp / pack stream (cfg e), two MFMAs behind the last link
        out, qi, li = [], 0, 0
        for gap in range(self.npv, self.ng):
            if li < 4 and gap == at[li]:
                out.append(links[li])
                li += 1
            else:
                out.append(qk[qi])
                qi += 1
        return out

    # ------------------------------------------------------------------ filler streams that differ
    def stream_kread(self, par):
        g = self.g
        return [mk("ds_read_b128", KF(kg, ks), KR[ks & 3], tag="lds", offset=g.K_SLOT + par * g.SLOT_B + kg * 16 * g.ROWB + 256 * (ks >> 2))
                for ks in range(self.NKS16) for kg in range(4)]

    def place_pool_kreads(self, load, slots, par):
        """k-steps 0 .. 2 are read during the P.V phase, k-step ks >= 3 goes into the slot of ks - 3 as soon as that k-step's eight MFMAs are issued"""
        kr = self.stream_kread(par)
        self.place(load, slots, kr[:12], self.cfg["kread_ct"][0], self.cfg["kread_ct"][1], 3)
        for ks in range(3, self.NKS16):
            g0 = self.npv + 8 * (ks - 3) + 7
            for kg in range(4):
                it = kr[4 * ks + kg]
                load[g0] += base._weight(it)
                slots[g0].append((g0 + 0.5 + 0.1 * kg, 3, it))

    def stream_vread(self, par):
        g = self.g
        out = []
        for kvs in range(2):
            for dg in range(self.NDG):
                j = dg >> 1
                adr = (VRO if dg & 1 else VR)[j & 3]
                off = par * g.SLOT_B + 32 * kvs * g.ROWB + 256 * (j >> 2)       # (V_BASE is in the address registers: a DS offset has 16 bits)
                out.append(mk("ds_read_b64_tr_b16", VF(dg, kvs).sub(0, 2), adr, tag="lds", offset=off))
                out.append(mk("ds_read_b64_tr_b16", VF(dg, kvs).sub(2, 2), adr, tag="lds", offset=off + 16 * g.ROWB))
        return out

    def dma_group(self, which, slot_par, guarded, ahead):
        """The 8 LDS-DMA pieces of this wave's quarter (16 rows) of one K or V tile (tile index = t + ahead).  Piece i = rows 2 i, 2 i + 1 of the quarter:
        its source offset is piece 0's with the image's swizzle bits flipped (K: granule ^ (row & 15) -> byte-offset bits 5 .. 7 = 2 i; V: chunk ^ (row & 3)
        -> bit 7 = i & 1, and the flipped 32-byte half of rows with (row >> 2) & 1 -> bit 5 = (i >> 1) & 1) plus i * (2 rows) — formed on the fly, two
        scratch registers in turn (the instruction offset 1024 i advances the LDS and the source address alike: *_ROW2 is biased by it)."""
        g = self.g
        rs, d0, soff, row2 = (A_KRS, KD0, S_KOFF, A_KROW2) if which == "k" else (A_VRS, VD0, S_VOFF, A_VROW2)
        lbase = (g.K_SLOT if which == "k" else g.V_BASE) + slot_par * g.SLOT_B
        groups = [[mk("s_add_u32", M0, A_LDSW, lbase, tag="salu"), mk("s_mov_b32", S_TMP, 0 if self.trim else soff, tag="salu")]]
        for i in range(g.NP):
            out = []
            flip = (32 * i) if which == "k" else (128 * (i & 1)) | (32 * ((i >> 1) & 1))
            if i == 0 and not self.trim:
                reg = d0
                out.append(mk("s_nop", 0, tag="salu"))
            else:
                reg = (DT0, DT1)[i & 1]
                if i:
                    out.append(mk("s_add_u32", S_TMP, S_TMP, row2, tag="salu"))
                if i == 4:      # a MUBUF instruction offset has 12 bits: the second half of the quarter through M0 and the scalar offset
                    out.append(mk("s_add_u32", M0, A_LDSW, lbase + 4096, tag="salu"))
                    out.append(mk("s_add_u32", S_TMP, S_TMP, 4096, tag="salu"))
                if self.trim:
                    # general form: gl_i = gl_0 ^ (flip >> 4);  offset = piece 0's + ((gl_i - gl_0) << 4), or beyond every descriptor if the row has no such granule
                    g0 = KG0 if which == "k" else VG0
                    out.append(mk("v_xor_b32", reg, flip >> 4, g0, tag="valu"))
                    out.append(mk("s_nop", 0, tag="salu"))
                    out.append(mk("v_cmp_gt_u32", m16.VCC, A_NG, reg, tag="valu"))
                    out.append(mk("v_sub_u32", reg, reg, g0, tag="valu"))
                    out.append(mk("s_nop", 0, tag="salu"))
                    out.append(mk("v_lshl_add_u32", reg, reg, 4, d0, tag="valu"))
                    out.append(mk("s_nop", 0, tag="salu"))
                    # (the pieces' stride, 2 rows - 1024, is NEGATIVE for rows below 512 bytes: it goes into the per-lane offset, where offset +
                    #  instruction offset wrap to the right sum — a scalar offset that has wrapped puts every lane out of range on the hardware)
                    out.append(mk("v_add_u32", reg, S_TMP, reg, tag="valu"))
                    out.append(mk("s_nop", 0, tag="salu"))
                    out.append(mk("v_cndmask_b32", reg, MARK, reg, m16.VCC, tag="valu"))
                else:
                    out.append(mk("v_xor_b32", reg, flip, d0, tag="valu"))
                out.append(mk("s_nop", 0, tag="salu"))
            out.append(mk("buffer_load_dwordx4", reg, rs, soff if self.trim else S_TMP, tag="dma", offen=True, offset=1024 * (i & 3), lds=True))
            groups.append(out)
        if not guarded:
            # main-loop bodies (tile t + ahead exists): one group per piece, in this order — M0, the running scalar offset, the two scratch registers and
            # VCC (trim) belong to the ONE stream the K and V groups of a body form; eight 1 KiB pieces back to back hold up the wave's issue (round 6:
            # the staging instructions as two blocks cost 14.5 % of the launch, profiles/r22_d256_body_ablation.txt)
            return groups
        skip = self.p.fresh("dma_skip")
        flat = [mk("s_add_u32", S_TMP2, S_T, ahead, tag="salu"), mk("s_cmp_lt_i32", S_TMP2, A_NTWG, tag="salu"),
                mk("s_cbranch_scc0", Label(skip), tag="branch")]
        for grp in groups:
            flat.extend(grp)
        flat.append(Ins("label", (Label(skip),)))
        return [flat]             # one atomic group: the guard's SCC and branch must not be interleaved with other streams

    def rare_rescale(self, lab):
        r = [Ins("label", (Label(lab),))]
        r.append(mk("s_nop", 15))
        r.append(mk("s_nop", 15))
        for h in range(2):
            r.append(mk("v_mul_f32", LS[h][0], LS[h][0], FS[h][0]))
            for dg in range(self.NDG):
                acc = OACC(dg, h)
                for j in range(4):
                    r.append(mk("v_accvgpr_read_b32", TMP[j], acc[j]))
                r.append(mk("s_nop", 1))
                for j in range(4):
                    r.append(mk("v_mul_f32", TMP[j], TMP[j], FS[h][0]))
                r.append(mk("s_nop", 1))
                for j in range(4):
                    r.append(mk("v_accvgpr_write_b32", acc[j], TMP[j]))
        r.append(mk("s_mov_b32", S_FLAG, 0))
        r.append(mk("s_nop", 7))
        r.append(mk("s_branch", Label(lab + "_ret")))
        return r

    # ------------------------------------------------------------------ one body: B(t) = P.V(t) | softmax(t + 1) | Q.K^T(t + 2) + the links of tile t + 1
    def body(self, par, pv=True, s1=True, s2=True, masked=False, guarded=True, name="body", first=False, dma=True, **kw):
        p, cfg, ng = self.p, self.cfg, self.ng
        start = len(p.ins)
        abl = set(cfg["abl"]) if name.startswith(("TA", "TF")) else set()       # timing-only ablations of the main-loop bodies (tools/kbench.py d256gen=abl=...)
        mf = (self.pv_mfmas(par, 0) if pv else [None] * self.npv) + self.qk_phase(par, s1, s2)
        assert len(mf) == ng
        if "mfma" in abl:
            mf = [None] * ng
        if not pv:
            p.emit("s_nop", 15)
            p.emit("s_nop", 15)
        load = [0.0] * ng
        slots = [[] for _ in range(ng)]
        if s1 and "max" not in abl:
            mw = cfg["mmask"] if masked else cfg["m"]
            self.place(load, slots, self.stream_max(0, par ^ 1, masked, first), mw[0], mw[1], 0)
        if dma and "dma" not in abl:
            dw = cfg["dma"] if guarded else cfg["dmaf"]
            self.place(load, slots, self.dma_group("k", par ^ 1, guarded, 3) + self.dma_group("v", par, guarded, 2), dw[0], dw[1], 2)
        if s2 and "kread" not in abl:
            self.place_pool_kreads(load, slots, par)
        if s1 and "vread" not in abl:
            self.place(load, slots, self.stream_vread(par ^ 1), cfg["vread"][0], cfg["vread"][1], 4)
        if s1 and "exp" not in abl:
            self.place(load, slots, self.stream_exp(0, par ^ 1), (cfg["mmask"] if masked else cfg["m"])[1], cfg["e"][1], 5)
        self.last_load = load
        for g in range(ng):
            slots[g].sort(key=lambda x: (x[0], x[1]))
        body_start = len(p.ins)
        for g in range(ng):
            if g == self.npv:
                lab = p.fresh("rare_r")
                p.emit("s_cmp_lg_u32", S_FLAG, 0)
                p.emit("s_cbranch_scc1", Label(lab))
                p.label(lab + "_ret")
                self.rare.append(self.rare_rescale(lab))
            if mf[g] is not None:
                p.ins.append(mf[g])
            for (_, _, item) in slots[g]:
                p.ins.extend(item if isinstance(item, list) else [item])
        p.ins[body_start:] = self.lds_waits(p.ins[body_start:])
        p.emit("s_add_u32", S_T, S_T, 1)
        p.emit("s_add_u32", S_KOFF, S_KOFF, A_KTILE)
        p.emit("s_add_u32", S_VOFF, S_VOFF, A_VTILE)
        if "vmwait" in abl:
            p.emit("s_waitcnt", lgkmcnt=0)
        else:
            p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        if "bar" not in abl:
            p.emit("s_barrier")
        # legality (Gen16.body): no V^T read ahead of the last P.V MFMA that takes its registers, no link ahead of a pack of its registers
        ins = [x for x in p.ins[start:] if x.op != "label"]
        last = {}
        for i, x in enumerate(ins):
            if x.op.startswith("v_mfma"):
                for (kind, lo, hi) in base.Gen._regs(x)[0]:
                    for r in range(lo, hi):
                        last[(kind, r)] = i
        for i, x in enumerate(ins):
            if x.op == "ds_read_b64_tr_b16":
                for (kind, lo, hi) in base.Gen._regs(x)[1]:
                    if any(last.get((kind, r), -1) > i for r in range(lo, hi)):
                        raise ValueError("illegal schedule: a V^T fragment read (%s) ahead of a P.V MFMA of this body that reads its registers" % x)
            if x.op.startswith("v_mfma") and x.ops[1].kind == "a" and x.ops[1].idx >= 240:
                lo, hi = x.ops[2].idx, x.ops[2].idx + x.ops[2].n
                for j in range(i + 1, len(ins)):
                    if ins[j].op == self.cvt and ins[j].ops[0].kind == "v" and lo <= ins[j].ops[0].idx < hi:
                        raise ValueError("illegal schedule: a row-sum link (%s) ahead of a pack of its P registers (%s)" % (x, ins[j]))

    # ------------------------------------------------------------------ whole block
    def build(self):
        p, g = self.p, self.g
        bf16_ = self.bf16
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        # Q fragments straight from memory, issued first: lane (n, g4) takes, of row 16 qg + n, the 16 bytes at column 64 ks + 16 g4
        p.emit("s_mov_b32", S_TMP, A_QW)
        if self.trim:
            p.emit("v_mbcnt_lo_u32_b32", TMP[7], -1, 0)
            p.emit("v_mov_b32", MARK, 0x80000000)
            p.emit("v_mbcnt_hi_u32_b32", TMP[7], -1, TMP[7])
            p.emit("s_nop", 0)
            p.emit("v_lshrrev_b32", G4R, 4, TMP[7])                         # g4 = lane / 16
            # this lane's slot of piece 0 holds the logical granule ... of the K image (granule ^ (row & 15); row & 15 = lane / 32 in piece 0) ...
            p.emit("v_and_b32", TMP[0], 31, TMP[7])
            p.emit("v_lshrrev_b32", TMP[1], 5, TMP[7])
            p.emit("s_nop", 0)
            p.emit("v_xor_b32", KG0, TMP[0], TMP[1])
            # ... and of the V image: chunk ^ (row & 3) -> granule bits 2 .. 3 ^ (lane / 32); no flipped half in piece 0 (rows 16 w, 16 w + 1: (row >> 2) & 1 = 0)
            p.emit("v_lshlrev_b32", TMP[1], 2, TMP[1])
            p.emit("s_nop", 0)
            p.emit("v_xor_b32", VG0, TMP[0], TMP[1])
            for ks in range(self.NKS16):                                     # Q: granule 4 ks + g4 of the row
                p.emit("v_add_u32", TMP[2], 4 * ks, G4R)
                p.emit("s_nop", 0)
                p.emit("v_cmp_gt_u32", m16.VCC, A_NG, TMP[2])
                p.emit("s_nop", 0)
                p.emit("v_cndmask_b32", V(16 + ks), MARK, A_Q0, m16.VCC)    # (the S banks are idle until H1: eight offset registers)
            p.emit("s_nop", 0)
        for qg in range(2):
            if qg:
                p.emit("s_add_u32", S_TMP, S_TMP, A_QT16)
            for ks in range(self.NKS16):
                p.emit("buffer_load_dwordx4", QF(qg, ks), V(16 + ks) if self.trim else A_Q0, A_QRS, S_TMP, offen=True, offset=64 * ks)
        for ks in range(4):
            p.emit("v_xor_b32", KR[ks], ks << 6, A_KR0)
        for j in range(4):
            p.emit("v_xor_b32", VR[j], j << 6, A_VR0)
            p.emit("v_xor_b32", VRO[j], (j << 6) | 32, A_VR0)
        p.emit("s_nop", 0)
        for j in range(4):
            p.emit("v_add_u32", VR[j], g.V_BASE, VR[j])
            p.emit("v_add_u32", VRO[j], g.V_BASE, VRO[j])
        # the constant A tuples of the row-sum links: 0.25 on the lanes of rows m = lane % 16 with m % 4 == qg (fwd_m16_gen.py)
        p.emit("v_lshrrev_b32", TMP[0], g.ROWB.bit_length() - 1, A_KR0)
        p.emit("v_mov_b32", TMP[1], 0x3e803e80 if bf16_ else 0x34003400)
        p.emit("v_and_b32", TMP[0], 3, TMP[0])
        for qg in range(2):
            p.emit("v_cmp_eq_u32", m16.VCC, qg, TMP[0])
            p.emit("v_cndmask_b32", TMP[2], 0, TMP[1], m16.VCC)
            p.emit("s_nop", 0)
            for i in range(4):
                p.emit("v_accvgpr_write_b32", ONES[qg][i], TMP[2])
        p.emit("s_lshr_b32", S_WAVE, A_LDSW, (g.SLOT_B // 4).bit_length() - 1)
        p.emit("v_mov_b32", KD0, A_KD0)
        p.emit("v_mov_b32", VD0, A_VD0)
        p.emit("s_mov_b32", S_T, -3)          # (the three staging groups below are bodies' groups with t + ahead = 0, 0, 1)
        p.emit("s_mov_b32", S_FLAG, 0)
        p.emit("s_mov_b32", S_NOVM, 0)
        p.emit("s_mov_b32", S_KOFF, 0)
        p.emit("s_mov_b32", S_VOFF, 0)
        for ins in self.dma_group("k", 0, True, 3)[0]:       # K(0) -> K slot 0
            p.ins.append(ins)
        for ins in self.dma_group("v", 0, True, 3)[0]:       # V(0) -> V slot 0   (a body stages V two tiles ahead, into slot `par`)
            p.ins.append(ins)
        p.emit("s_mov_b32", S_T, -2)
        p.emit("s_mov_b32", S_KOFF, A_KTILE)
        for ins in self.dma_group("k", 1, True, 3)[0]:       # K(1) -> K slot 1 (if the workgroup has a second tile)
            p.ins.append(ins)
        # running state of body t = -2: it stages K(t + 3) = K(1)?  No: K(1) is on its way already — H1 carries no staging (dma=False), H2 stages K(2), V(1)
        p.emit("s_mov_b32", S_KOFF, A_KTILE)
        p.emit("s_mov_b32", S_VOFF, 0)
        for h in range(2):
            p.emit("v_mov_b32", MC[h][0], NEG_INF)
            p.emit("v_mov_b32", LS[h][0], 0)
            p.emit("v_mov_b32", FS[h][0], 1.0)
        p.emit("v_mov_b32", LSV[2], 0)
        p.emit("v_mov_b32", LSV[3], 0)
        for i in range(128):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")

        # ---- head bodies, then the dispatch loop over max-first bodies (fwd_m16_gen.Gen16.build without its fast loop)
        self.body(0, pv=False, s1=False, s2=True, name="H1", dma=False)
        # H1 advanced S_KOFF / S_VOFF like every body: H2 (t = -1) stages K(2) and V(1)
        p.emit("s_cmp_eq_u32", A_NTW, 1)
        p.emit("s_cbranch_scc1", Label("h2b"))
        self.body(1, pv=False, s1=True, s2=True, name="H2", first=True)
        p.emit("s_branch", Label("dispatch"))
        p.label("h2b")
        self.body(1, pv=False, s1=True, s2=False, masked=True, name="H2b", first=True)
        p.label("dispatch")
        p.emit("s_cmp_ge_i32", S_T, A_NTWG)
        p.emit("s_cbranch_scc1", Label("epilogue"))
        p.emit("s_sub_u32", S_D, A_NTW, S_T)
        p.emit("s_and_b32", S_TMP, S_T, 1)
        p.emit("s_cmp_eq_u32", S_TMP, 1)
        p.emit("s_cbranch_scc1", Label("disp_odd"))
        for par, suffix in ((0, "e"), (1, "o")):
            if par == 1:
                p.label("disp_odd")
            p.emit("s_cmp_ge_i32", S_D, 4)           # t + 3 < ntw <= ntwg: the tiles this body stages exist, its staging needs no guard
            p.emit("s_cbranch_scc1", Label("tf_" + suffix))
            p.emit("s_cmp_ge_i32", S_D, 3)
            p.emit("s_cbranch_scc1", Label("ta_" + suffix))
            p.emit("s_cmp_eq_u32", S_D, 2)
            p.emit("s_cbranch_scc1", Label("tb_" + suffix))
            p.emit("s_cmp_eq_u32", S_D, 1)
            p.emit("s_cbranch_scc1", Label("tc_" + suffix))
            self.body(par, pv=False, s1=False, s2=False, name="ST%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tf_" + suffix)
            self.body(par, name="TF%d" % par, guarded=False)
            p.emit("s_branch", Label("dispatch"))
            p.label("ta_" + suffix)
            self.body(par, name="TA%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tb_" + suffix)
            self.body(par, s2=False, masked=True, name="TB%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tc_" + suffix)
            self.body(par, s1=False, s2=False, name="TC%d" % par)
            p.emit("s_branch", Label("dispatch"))

        # ---- epilogue: row sums over the row's four lanes, O / l -> 16 bit -> memory (rows >= Nq fall outside the descriptor), LSE out
        p.label("epilogue")
        p.emit("s_nop", 15)
        lse = [TMP[4], TMP[5]]
        inv = [FS[0][0], FS[1][0]]
        for h in range(2):
            lt, t = TMP[0], TMP[1]
            p.emit("v_mov_b32", lt, LS[h][0])
            for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
                p.emit("v_mov_b32", t, lt)
                p.emit("s_nop", 1)
                p.emit(op, lt, t)
                p.emit("v_add_f32", lt, lt, t)
                p.emit("s_nop", 0)
            p.emit("v_rcp_f32", inv[h], lt)
            p.emit("v_log_f32", t, lt)
            p.emit("s_nop", 0)
            p.emit("v_add_f32", lse[h], MC[h][0], t)
        p.emit("s_nop", 0)
        p.emit("s_mov_b32", S_TMP, A_OW)
        if self.trim:
            p.emit("s_lshl_b32", S_D, A_NG, 1)      # D / 4
        for qg in range(2):
            if qg:
                p.emit("s_add_u32", S_TMP, S_TMP, A_OT16)
            for dg in range(self.NDG):
                acc = OACC(dg, qg)
                t0 = 2 * (dg & 1)                       # two scratch pairs in turn: a store's data registers are not rewritten right behind it
                for j in range(4):
                    p.emit("v_accvgpr_read_b32", TMP[j] if t0 == 0 else EPX[j], acc[j])
                src = TMP if t0 == 0 else EPX
                p.emit("s_nop", 0)
                for j in range(4):
                    p.emit("v_mul_f32", src[j], src[j], inv[qg])
                p.emit("s_nop", 0)
                p.emit(self.cvt, src[0], src[0], src[1])
                p.emit(self.cvt, src[1], src[2], src[3])
                p.emit("s_nop", 0)
                # 8 bytes: d = 16 dg + 4 g4 .. + 3 of row 16 qg + n
                oo = A_OO0
                if self.trim:                       # ... if the row has them: 4 dg + g4 < D / 4
                    oo = src[2]
                    p.emit("v_lshl_add_u32", src[3], dg, 2, G4R)
                    p.emit("s_nop", 0)
                    p.emit("v_cmp_gt_u32", m16.VCC, S_D, src[3])
                    p.emit("s_nop", 0)
                    p.emit("v_cndmask_b32", oo, MARK, A_OO0, m16.VCC)
                    p.emit("s_nop", 0)
                p.emit("buffer_store_dwordx2", V(src[0].idx, 2), oo, A_ORS, S_TMP, offen=True, offset=32 * dg)
        # the LSE leaves in ONE register: lane l = 16 g4 + n (l < 32) hands over row l of the wave
        p.emit("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
        p.emit("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])
        p.emit("s_nop", 0)
        p.emit("v_lshrrev_b32", TMP[0], 4, TMP[0])
        p.emit("s_nop", 0)
        p.emit("v_cmp_eq_u32", m16.VCC, 1, TMP[0])
        p.emit("s_nop", 0)
        p.emit("v_cndmask_b32", TMP[1], lse[0], lse[1], m16.VCC)
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        p.emit("v_mov_b32", A_LSE0, TMP[1])
        p.emit("s_branch", Label("end"))
        for r in self.rare:
            p.extend(r)
        p.label("end")
        return p


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    ap.add_argument("--opt", default="")
    ap.add_argument("--probe", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cfg = base.parse_opts(a.opt)
    if base.is_probe(cfg) and not a.probe:
        sys.exit("fwd_m16_d256_gen.py: %r contains timing-probe options; they need --probe" % a.opt)
    for bf16, trim, nks in ((b, t, n) for b in (False, True) for (t, n) in ((False, 8), (True, 8), (True, 7), (True, 6), (True, 5))):
        c = dict(cfg)
        c["opt"] = tuple(o for o in cfg.get("opt", ()) if o != "trim") + (("trim",) if trim else ())
        prog = Gen256(bf16, nks=nks, **c).build()
        path = os.path.join(a.out, "fa2_fwd_m16_d256_%s%s.inc" % ("bf16" if bf16 else "f16", ("_trim%d" % nks) if trim else ""))
        base.write_atomic(path, "// GENERATED by csrc/gen/fwd_m16_d256_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + base.render_inline(prog))
        print(path, len(prog.ins), "instructions")


if __name__ == "__main__":
    main()
