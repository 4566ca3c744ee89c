#!/usr/bin/env python3
"""dQ pass of the D = 128 backward built on v_mfma_f32_16x16x32 (round 5) — the pipeline, images, staging and operand list of GenDQ
(bwd_d128_gen.py; reference counterpart: bwd_kernel, kernel_fp16.cu:547-740), another MFMA tile.

Why: the chip is power-limited under these kernels and the 16x16x32 form does the same FLOPs for fewer joules; a synthetic body
with this pass's filler mix runs 10.9 % faster with it (tools/ubench/mfma_shape_probe.py: bwd_dq_like_*, profiles/r17_mfma_shape_probe_bwd.json).

Shape: workgroup = 4 waves = 256 Q rows, wave = 64 rows = four 16-row groups qg; KV tiles of 32 rows = two 16-row groups kg.  MFMA layouts as in
fwd_m16_gen.py: A[m][k]: lane l holds m = l % 16, k = 8 (l / 16) .. +7;  B[k][n]: n = l % 16;  D[m][n]: n = l % 16, m = 4 (l / 16) + i.
    S^T[kv,q]  tile (kg, qg) = sum_ks K[kg rows, 32 d] . Q^T[ks, qg]          2 x 4 x 4 = 32 MFMAs per tile
    dP^T[kv,q] tile (kg, qg) = sum_ks V[kg rows, 32 d] . dO^T[ks, qg]         32
    dQ^T[d,q]  tile (dg, qg) += K^T[16 d, 32 kv] . dS^T[32 kv, qg]             8 x 4 = 32           -> 96 MFMAs per body (GenDQ: 48 of twice the size)
A lane (n, g = l / 16) holds, of Q row 16 qg + n, the scores kv = 16 kg + 4 g + i: L and delta are per ROW, four rows per lane.  A "q block" qb of
GenDQ is a pair of q groups (qg = 2 qb + h): bank register e = 8 h + 4 kg + i; dS is packed in place into registers 8 h .. 8 h + 3 — the B operand
of the dQ product, whose k-slot (g, j) stands for kv = 16 (j >> 2) + 4 g + (j & 3): the K^T fragment is two transposed reads, rows 4 g .. +3 and 16
further down.  Fragment loads of Q / dO / O at the entry: lane (n, g) takes, of row 16 qg + n, the 16 bytes at column 64 ks + 16 g; the row offsets
are formed here from the operands ROW0 (= qw0 + n), Nq - 1 (rows past Nq read the last row, as in GenDQ) and the three row pitches.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
import bwd_d128_gen as base  # noqa: E402
import sched  # noqa: E402
from bwd_d128_gen import DQ  # noqa: E402
from isa import A, V, Arg, Ins, Label, Neg, VCC, mk  # noqa: E402

NEG_INF = float("-inf")

# ---- operands: GenDQ's list, position for position (fa2_bwd_d128.hip.h hands both kernels the same 34), the per-lane row offsets re-read:
A_D0 = DQ.A_D0                         # out: delta of row l of the wave (lane l); %1 (A_D1) is not written
A_ROW0, A_G16 = Arg(2), Arg(3)         # this lane's row of q group 0 (qw0 + lane % 16); 16 * (lane / 16)
A_NQ1 = Arg(4)                         # Nq - 1 (the same value in every lane)
A_QROWB, A_GROWB, A_OROWB = Arg(5), Arg(6), Arg(7)     # row pitches of Q / dO / O in bytes (the same value in every lane)
A_LIM0, A_LIM1 = DQ.A_LIM0, DQ.A_LIM1  # masks of the wave's last two tiles: row 16 qg + n keeps kv_local = 16 kg + 4 g + i (of the LAST tile; the one before: + 32)
                                       # iff 16 kg + i <= min(LIM0 + 16 qg, LIM1)

# ---- register map (what differs from DQ's; KD / VD / TD, the SGPRs and the LDS layout are DQ's: GenDQ.dma_group is inherited)
KR = [V(216 + i) for i in range(4)]                # row-fragment read addresses, k-step ks (32 head-dim columns)
NL = [V(220 + i) for i in range(4)]                # -LSE of this lane's row of q group qg
VR = DQ.VR                                         # transposed read addresses, 64-byte chunk dg >> 1 (v224..227), even d groups ...
VRO = DQ.VRB                                       # ... and odd ones (v246..249): the other 32-byte half, whose side depends on the lane's row (stream_trread)
DD = [V(234 + i) for i in range(4)]                # delta of this lane's row of q group qg
TMP = DQ.TMP                                       # v238..245
LIMQ = [[V(250), V(251)], [V(252), V(253)]]        # masked bodies: limits of the two rows of q block qb


def SB(par, qb):
    return DQ.SB(par, qb)


def DB(par, qb):
    return DQ.DB(par, qb)


def RP(x, kg, ks):                                 # row-fragment pool (x = 0: K, 1: V): two k-step slots of (2 matrices x 2 kv groups) x 4 registers
    return V(152 + 16 * (ks % 2) + 8 * x + 4 * kg, 4)


def TP(dg):                                        # K^T fragments of one tile: 16 d rows x 32 kv
    return V(184 + 4 * dg, 4)


def ACC(dg, qg):
    return A(4 * (8 * qg + dg), 4)


def QF(qg, ks):
    return A(128 + 4 * (4 * qg + ks), 4)


def GF(qg, ks):
    return A(192 + 4 * (4 * qg + ks), 4)


class GenDQ16(base.GenDQ):
    DEFAULTS = {"valu": (2.0, 94.0), "rowread": (0.0, 30.0), "trread": (34.0, 94.0), "dma": (2.0, 28.0), "opt": (), "abl": ()}

    def __init__(self, bf16=False, **cfg):
        super().__init__(bf16, **cfg)
        assert "uni" not in self.opt
        self.mfma = "v_mfma_f32_16x16x32_bf16" if bf16 else "v_mfma_f32_16x16x32_f16"

    # ------------------------------------------------------------------ MFMA lists
    def dq_mfmas(self, par):
        out = []
        for dg in range(8):
            for qb in range(2):
                for h in range(2):
                    acc = ACC(dg, 2 * qb + h)
                    out.append(mk(self.mfma, acc, TP(dg), DB(par, qb).sub(8 * h, 4), acc, tag="mfma"))
        return out

    def sdp_mfmas(self, par):
        out = []
        for ks in range(4):
            for x, bank, frag in ((0, SB, QF), (1, DB, GF)):
                for qb in range(2):
                    for h in range(2):
                        for kg in range(2):
                            dst = bank(par, qb).sub(8 * h + 4 * kg, 4)
                            out.append(mk(self.mfma, dst, RP(x, kg, ks), frag(2 * qb + h, ks), 0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    def stream_valu(self, qb, par, masked, off):
        """P = 2^(S c - L), dS = P (dP - delta), pairs packed in place in the dP bank; per row h of the q block.  masked: scores with kv beyond the
        row's limit become -inf first (off = 0: the wave's last tile, 32: the one before it)."""
        s, d = SB(par, qb), DB(par, qb)
        out = []
        if masked:
            lim = LIMQ[qb]
            t2 = TMP[4 * qb]
            for h in range(2):
                out.append(mk("v_add_u32", lim[h], 16 * (2 * qb + h), A_LIM0, tag="valu"))
            for h in range(2):
                out.append(mk("v_min_i32", lim[h], lim[h], A_LIM1, tag="valu"))
            # (-inf for c >= 0, +inf for a negative scale: GenDQ.stream_valu)
            out.append([mk("v_mov_b32", t2, DQ.A_C, tag="valu"), mk("v_and_b32", t2, 0x80000000, t2, tag="valu"),
                        mk("v_xor_b32", t2, 0xff800000, t2, tag="valu")])
            for h in range(2):
                for kg in range(2):
                    for i in range(4):
                        e = 8 * h + 4 * kg + i
                        out.append([mk("v_cmp_le_i32", VCC, 16 * kg + i - off, lim[h], tag="valu"), mk("v_cndmask_b32", s[e], t2, s[e], VCC, tag="valu")])
        prs = [8 * h + 2 * k for k in range(4) for h in range(2)]          # the 8 register pairs, rows alternating
        for k in range(8 + 4):
            F, E, Sb, Mu, C = [], [], [], [], []
            if k < 8:
                e0 = prs[k]
                qg = 2 * qb + e0 // 8
                for e in (e0, e0 + 1):
                    F.append(mk("v_fma_f32", s[e], s[e], DQ.A_C, NL[qg], tag="valu"))
                    Sb.append(mk("v_sub_f32", d[e], d[e], DD[qg], tag="valu"))
            if 0 <= k - 1 < 8:
                e0 = prs[k - 1]
                for e in (e0, e0 + 1):
                    E.append(mk("v_exp_f32", s[e], s[e], tag="trans"))
            if 0 <= k - 2 < 8:
                e0 = prs[k - 2]
                for e in (e0, e0 + 1):
                    Mu.append(mk("v_mul_f32", d[e], s[e], d[e], tag="valu"))
            if 0 <= k - 3 < 8:
                e = prs[k - 3]                        # e = 8 h + 4 kg + i (i even) -> packed word 8 h + 2 kg + i / 2
                C.append(mk(self.cvt, d[8 * (e // 8) + (e % 8) // 2], d[e], d[e + 1], tag="valu"))
            out += F + E + Sb + Mu + C
        return out

    def row_read(self, x, kg, ks, par):
        off = DQ.ROW_RING + par * DQ.ROW_SLOT + x * DQ.V_IN_SLOT + kg * 16 * 256
        return mk("ds_read_b128", RP(x, kg, ks), KR[ks], tag="lds", offset=off)

    def stream_trread(self, par):
        out = []
        # A read's lanes 0..31 are the 16-lane groups g = 0, 1: rows 4 g + (n >> 2), i.e. rows r and r + 4 of the tile in one cycle.  The "tr" image
        # (64-byte chunk ^ (row & 3)) kept those in the same banks: a 2-way conflict on every read, 26 % of the pass's LDS cycles
        # (profiles/r19_bwd_c2_pmc.txt).  Round 6: the image flips the 32-byte half of a chunk for rows with (row >> 2) & 1 (fa2_bwd_d128.hip.h: td0, vr0),
        # and the odd d groups — the other half, on a side that depends on the lane's row — are read through a second address set.
        for dg in range(8):
            off = DQ.TR_RING + par * DQ.TR_SLOT
            adr = (VRO if dg & 1 else VR)[dg >> 1]
            out.append(mk("ds_read_b64_tr_b16", TP(dg).sub(0, 2), adr, tag="lds", offset=off))
            out.append(mk("ds_read_b64_tr_b16", TP(dg).sub(2, 2), adr, tag="lds", offset=off + 16 * 256))
        return out

    # ------------------------------------------------------------------ one body
    def body(self, par, dq=True, s1=True, s2=True, masked=False, off=0, guarded=True, dma=True, name="body"):
        """B(t) with t & 1 == par (GenDQ.body): dq: dQ(t), 32 MFMAs; s1: the VALU work and the transposed reads of tile t+1; s2: the row reads and
        S / dP of tile t+2, 64 MFMAs."""
        p, cfg = self.p, self.cfg
        abl = set(cfg["abl"]) if name.startswith("F") else set()
        ng = 96
        mf = (self.dq_mfmas(par) if dq else [None] * 32) + (self.sdp_mfmas(par) if s2 else [None] * 64)
        if "mfma" in abl:
            mf = [None] * ng
        load = [0.0] * ng
        slots = [[] for _ in range(ng)]
        pre = []
        if not dq:
            pre += [mk("s_nop", 15), mk("s_nop", 15)]
        if dma and "dma" not in abl:
            grp = self.dma_group("k", par ^ 1, guarded, 3) + self.dma_group("v", par ^ 1, guarded, 3) + self.dma_group("t", par, guarded, 2)
            sched.place(load, slots, grp, cfg["dma"][0], cfg["dma"][1], 2)
        if s2 and "rowread" not in abl:
            # pool of two k-step slots: k-steps 0, 1 are read during the dQ phase, k-step ks >= 2 into the slot of ks - 2 once that k-step's 16 MFMAs are issued
            first = [self.row_read(x, kg, ks, par) for ks in range(2) for x in range(2) for kg in range(2)]
            sched.place(load, slots, first, cfg["rowread"][0], cfg["rowread"][1], 3)
            for ks in range(2, 4):
                g = 32 + 16 * (ks - 2) + 15
                for j, (x, kg) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                    it = self.row_read(x, kg, ks, par)
                    load[g] += sched.weight(it)
                    slots[g].append((g + 0.5 + 0.1 * j, 3, it))
        if s1 and "trread" not in abl:
            sched.place(load, slots, self.stream_trread(par ^ 1), cfg["trread"][0], cfg["trread"][1], 4)
        if s1 and "valu" not in abl:
            w = cfg["valu"]
            sched.place(load, slots, self.stream_valu(0, par ^ 1, masked, off), w[0], w[1] - 2.0, 5)
            sched.place(load, slots, self.stream_valu(1, par ^ 1, masked, off), w[0], w[1], 6)
        self.last_load = load
        boundary = {}
        if s2:
            boundary[32] = [mk("s_nop", 1)]
        bk = [mk("s_add_u32", DQ.S_T, DQ.S_T, 1, tag="salu"), mk("s_add_u32", DQ.S_KOFF, DQ.S_KOFF, DQ.A_KTILE, tag="salu"),
              mk("s_add_u32", DQ.S_VOFF, DQ.S_VOFF, DQ.A_VTILE, tag="salu"), mk("s_add_u32", DQ.S_TOFF, DQ.S_TOFF, DQ.A_KTILE, tag="salu")]
        sched.place(load, slots, bk, 88.0, 95.0, 9)
        post = [mk("s_waitcnt", vmcnt=0, lgkmcnt=0)]
        if "barrier" not in abl:
            post.append(mk("s_barrier"))
        self.emit_body(p, mf, slots, pre=pre, boundary=boundary, post=post, bookkeeping=bk, name="dQ16 body " + name)

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(4):
            p.emit("v_xor_b32", KR[ks], ks << 6, DQ.A_KR0)
        for j in range(4):
            p.emit("v_xor_b32", VR[j], j << 6, DQ.A_VR0)
            p.emit("v_xor_b32", VRO[j], (j << 6) | 32, DQ.A_VR0)
        # ---- fragment loads.  Row offsets of this lane's four rows: min(ROW0 + 16 qg, Nq - 1) * pitch + 16 g, per matrix (T[0..3]: Q, T[4..7]: dO;
        #      O after the dO loads are issued, in T[4..7] again; the L offsets reuse the clamped rows)
        tg, to = V(24, 64), V(88, 64)              # dO / O pass through the (still unused) S / dP banks for delta
        rows = [V(250 + i) for i in range(4)]      # (LIMQ registers: free until the first masked body)
        for qg in range(4):
            p.emit("v_add_u32", rows[qg], 16 * qg, A_ROW0)
        p.emit("s_nop", 0)
        for qg in range(4):
            p.emit("v_min_u32", rows[qg], rows[qg], A_NQ1)
        p.emit("s_nop", 0)
        for qg in range(4):
            p.emit("v_mul_lo_u32", TMP[qg], rows[qg], A_QROWB)
            p.emit("v_mul_lo_u32", TMP[4 + qg], rows[qg], A_GROWB)
        p.emit("s_nop", 0)
        for qg in range(4):
            p.emit("v_add_u32", TMP[qg], TMP[qg], A_G16)
            p.emit("v_add_u32", TMP[4 + qg], TMP[4 + qg], A_G16)
        p.emit("s_nop", 0)
        for qg in range(4):
            for ks in range(4):
                p.emit("global_load_dwordx4", QF(qg, ks), TMP[qg], DQ.A_QB, offset=64 * ks)
        for qg in range(4):
            for ks in range(4):
                p.emit("global_load_dwordx4", tg.sub(16 * qg + 4 * ks, 4), TMP[4 + qg], DQ.A_GB, offset=64 * ks)
        for qg in range(4):
            p.emit("v_mul_lo_u32", TMP[qg], rows[qg], A_OROWB)          # (the Q loads have read their addresses: VMEM issues in order)
        p.emit("s_nop", 0)
        for qg in range(4):
            p.emit("v_add_u32", TMP[qg], TMP[qg], A_G16)
            p.emit("v_lshlrev_b32", rows[qg], 2, rows[qg])                # byte offset of the row's LSE
        p.emit("s_nop", 0)
        for qg in range(4):
            for ks in range(4):
                p.emit("global_load_dwordx4", to.sub(16 * qg + 4 * ks, 4), TMP[qg], DQ.A_OB, offset=64 * ks)
        for qg in range(4):
            p.emit("global_load_dword", NL[qg], rows[qg], DQ.A_LB)
        # ---- DMA source offsets (GenDQ.build)
        p.emit("v_mov_b32", DQ.KD[0], DQ.A_KD0)
        p.emit("v_mov_b32", DQ.VD[0], DQ.A_VD0)
        p.emit("v_mov_b32", DQ.TD[0], DQ.A_TD0)
        p.emit("v_xor_b32", DQ.KD[1], 64, DQ.A_KD0)
        p.emit("v_xor_b32", DQ.VD[1], 64, DQ.A_VD0)
        p.emit("v_xor_b32", DQ.TD[1], 32, DQ.A_TD0)                     # piece 1: rows 4 further down — the flipped half of the "tr" image
        p.emit("s_nop", 0)
        p.emit("v_add_u32", DQ.TD[1], DQ.A_KROW4, DQ.TD[1])
        p.emit("s_nop", 0)
        p.emit("v_add_u32", DQ.KD[1], DQ.A_KROW4, DQ.KD[1])
        p.emit("v_add_u32", DQ.VD[1], DQ.A_VROW4, DQ.VD[1])
        p.emit("s_mov_b32", DQ.S_T, -2)
        p.emit("s_mov_b32", DQ.S_KOFF, 0)
        p.emit("s_mov_b32", DQ.S_VOFF, 0)
        p.emit("s_mov_b32", DQ.S_TOFF, 0)
        for which, rs, vd, bs in (("k", DQ.A_KRS, DQ.KD, DQ.ROW_RING), ("v", DQ.A_VRS, DQ.VD, DQ.ROW_RING + DQ.V_IN_SLOT)):
            p.emit("s_add_u32", base.M0, DQ.A_LDSW, bs)
            p.emit("s_nop", 0)
            for i in range(2):
                p.emit("buffer_load_dwordx4", vd[i], rs, DQ.S_KOFF, offen=True, offset=1024 * i, lds=True)
        p.emit("s_mov_b32", DQ.S_KOFF, DQ.A_KTILE)
        p.emit("s_mov_b32", DQ.S_VOFF, DQ.A_VTILE)
        for i in range(128):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        # ---- delta = rowsum(dO * O): this lane's slices of its four rows, then across the row's four lanes; the 4 DMA pieces issued last may keep flying
        p.emit("s_waitcnt", vmcnt=4)
        for qg in range(4):
            acc = DD[qg]
            p.emit("v_mov_b32", acc, 0)
            p.emit("s_nop", 0)
            for i in range(16):
                p.emit(self.dot2, acc, tg[16 * qg + i], to[16 * qg + i], acc)
            p.emit("s_nop", 3)      # a DOT result read by another kind of VALU instruction: 3 wait states, not interlocked (GenDQ.build)
            for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
                p.emit("v_mov_b32", TMP[0], acc)
                p.emit("s_nop", 1)
                p.emit(op, acc, TMP[0])
                p.emit("s_nop", 0)
                p.emit("v_add_f32", acc, acc, TMP[0])
                p.emit("s_nop", 0)
            p.emit("v_sub_f32", NL[qg], 0, NL[qg])                       # -LSE
        for i in range(64):                                           # dO fragments -> their AGPRs
            p.emit("v_accvgpr_write_b32", A(192 + i), tg[i])
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")

        # ---- head bodies, fast loop, tail dispatch: GenDQ's structure
        self.body(0, dq=False, s1=False, s2=True, name="H1")
        p.emit("s_cmp_ge_i32", DQ.A_NTW, 3)
        p.emit("s_cbranch_scc1", Label("h2"))
        p.emit("s_cmp_eq_u32", DQ.A_NTW, 2)
        p.emit("s_cbranch_scc1", Label("h2m"))
        self.body(1, dq=False, s1=True, s2=False, masked=True, off=0, name="H2b")
        p.emit("s_branch", Label("main"))
        p.label("h2m")
        self.body(1, dq=False, s1=True, s2=True, masked=True, off=32, name="H2m")
        p.emit("s_branch", Label("main"))
        p.label("h2")
        self.body(1, dq=False, s1=True, s2=True, name="H2")
        p.label("main")
        p.emit("s_sub_u32", DQ.S_NFAST, DQ.A_NTW, 3)
        p.emit("s_cmp_gt_i32", DQ.S_NFAST, 0)
        p.emit("s_cbranch_scc0", Label("dispatch"))
        p.label("fast0")
        self.body(0, guarded=False, name="F0")
        p.emit("s_sub_u32", DQ.S_NFAST, DQ.S_NFAST, 1)
        p.emit("s_cmp_gt_i32", DQ.S_NFAST, 0)
        p.emit("s_cbranch_scc0", Label("dispatch"))
        self.body(1, guarded=False, name="F1")
        p.emit("s_sub_u32", DQ.S_NFAST, DQ.S_NFAST, 1)
        p.emit("s_cmp_gt_i32", DQ.S_NFAST, 0)
        p.emit("s_cbranch_scc1", Label("fast0"))
        p.label("dispatch")
        p.emit("s_cmp_ge_i32", DQ.S_T, DQ.A_NTWG)
        p.emit("s_cbranch_scc1", Label("epilogue"))
        p.emit("s_sub_u32", DQ.S_D, DQ.A_NTW, DQ.S_T)
        p.emit("s_and_b32", DQ.S_TMP, DQ.S_T, 1)
        p.emit("s_cmp_eq_u32", DQ.S_TMP, 1)
        p.emit("s_cbranch_scc1", Label("disp_odd"))
        for par, sfx in ((0, "e"), (1, "o")):
            if par == 1:
                p.label("disp_odd")
            p.emit("s_cmp_eq_u32", DQ.S_D, 3)
            p.emit("s_cbranch_scc1", Label("tb3_" + sfx))
            p.emit("s_cmp_eq_u32", DQ.S_D, 2)
            p.emit("s_cbranch_scc1", Label("tb2_" + sfx))
            p.emit("s_cmp_eq_u32", DQ.S_D, 1)
            p.emit("s_cbranch_scc1", Label("tc_" + sfx))
            self.body(par, dq=False, s1=False, s2=False, name="ST%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tb3_" + sfx)
            self.body(par, masked=True, off=32, name="TB3%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tb2_" + sfx)
            self.body(par, s2=False, masked=True, off=0, name="TB2%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tc_" + sfx)
            self.body(par, s1=False, s2=False, name="TC%d" % par)
            p.emit("s_branch", Label("dispatch"))

        # ---- epilogue: dQ = acc * scale -> 16 bit -> the wave's LDS image (rows of 272 B); delta out in ONE register (lane l = row l of the wave)
        p.label("epilogue")
        p.emit("s_nop", 15)
        T = TMP
        for qg in range(4):
            for dg in range(8):
                acc = ACC(dg, qg)
                for j in range(4):
                    p.emit("v_accvgpr_read_b32", T[j], acc[j])
                p.emit("s_nop", 0)
                for j in range(4):
                    p.emit("v_mul_f32", T[j], DQ.A_SCALE, T[j])
                p.emit("s_nop", 0)
                p.emit(self.cvt, T[0], T[0], T[1])
                p.emit(self.cvt, T[1], T[2], T[3])
                p.emit("s_nop", 0)
                p.emit("ds_write_b64", DQ.A_EPI, V(T[0].idx, 2), offset=16 * qg * DQ.EPI_ROWB + 32 * dg)
        p.emit("v_mbcnt_lo_u32_b32", T[0], -1, 0)
        p.emit("v_mbcnt_hi_u32_b32", T[0], -1, T[0])
        p.emit("s_nop", 0)
        p.emit("v_lshrrev_b32", T[0], 4, T[0])
        p.emit("s_nop", 0)
        p.emit("v_mov_b32", T[1], DD[0])
        for qg in range(1, 4):
            p.emit("v_cmp_eq_u32", VCC, qg, T[0])
            p.emit("v_cndmask_b32", T[1], T[1], DD[qg], VCC)
        p.emit("s_waitcnt", lgkmcnt=0)
        p.emit("v_mul_f32", A_D0, DQ.A_DSIGN, T[1])
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
    cfg = base.parse_opts(a.opt) if hasattr(base, "parse_opts") else {}
    for bf16 in (False, True):
        prog = GenDQ16(bf16, **cfg).build()
        path = os.path.join(a.out, "fa2_bwd_dq_m16_%s.inc" % ("bf16" if bf16 else "f16"))
        base.write_atomic(path, "// GENERATED by csrc/gen/bwd_dq_m16_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + base.render_inline(prog, "fa2dq16"))
        print(path, len(prog.ins), "instructions")


if __name__ == "__main__":
    main()
