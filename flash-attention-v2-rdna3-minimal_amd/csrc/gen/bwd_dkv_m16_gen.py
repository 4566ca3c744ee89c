#!/usr/bin/env python3
"""dK / dV pass of the D = 128 backward built on v_mfma_f32_16x16x32 (round 5) — the pipeline, rings, roles, hand-over and book-keeping of
bwd_d128_gen.GenDKV (reference counterpart: bwd_kernel, kernel_fp16.cu:547-740), another MFMA tile.

Why: the forward's finding — on a power-limited chip the 16x16x32 form does the same FLOPs for fewer joules — paid on the dQ pass
(bwd_dq_m16_gen.py: -7 .. 8 % of the pass); this is the other pass.

Shape (workgroup = 4 waves = two wave pairs = 128 KV rows; a wave owns 64 KV rows = four 16-row groups kvg; Q is swept in tiles of 32 rows = two groups qg):
    P side    S[q,kv] tile (qg, kvg) = sum_ks Q[qg rows, 32 d of k-step ks] . K^T[ks, kvg]      2 x 4 x 4 = 32 MFMAs per tile (A: row reads, B: own K rows)
              dV^T[d,kv] tile (dg, kvg) += dO^T[16 d of dg, 32 q] . P[32 q, kvg]                 8 x 4     = 32 MFMAs (A: transposed reads, B: P packed in place)
    dS side   dP[q,kv] likewise from dO rows and own V rows (-delta is the C operand of k-step 0), dK^T += Q^T . dS
  MFMA layouts: A[m][k]: lane l holds m = l % 16, k = 8 (l / 16) .. +7;  B[k][n]: n = l % 16, same k;  D[m][n]: n = l % 16, m = 4 (l / 16) + i.
  A lane (n, g) holds, of KV row 16 kvg + n, the scores of q = 16 qg + 4 g + i.  Bank register e = 8 (kvg & 1) + 4 qg + i inside half kvb = kvg >> 1 of a
  32-register bank — so that the packed P / dS of KV group kvg lands in registers 8 kvg .. +3 in MFMA k-slot order (slot (g, j) stands for q = 16 (j >> 2) +
  4 g + (j & 3)) by the very in-place formula of the 32 x 32 layout (8 (e // 8) + (e % 8) // 2): the exp / pack / hand-over streams carry over, and the dS
  side reads its partner's words lane for lane.  L / -delta of a tile: 8 values per lane (q = 16 qg + 4 g + i), two ds_read_b128.  The transposed fragment of
  (dg, the tile's one k-step) is two transposed reads, rows 4 g .. +3 and 16 further down.
Not in this generator: "kfold", "qsplit" (those launches keep the 32 x 32 bodies).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
import bwd_d128_gen as base  # noqa: E402
import sched  # noqa: E402
from bwd_d128_gen import KV  # noqa: E402
from isa import A, V, Arg, Ins, Label, M0, Neg, VCC, mk  # noqa: E402

# ---- operands: GenDKV's list (fa2_bwd_d128.hip.h) with four positions re-read
A_FO = [Arg(0), Arg(1), Arg(7), Arg(10)]           # byte offset of this lane's 16 bytes (k-step 0) of its own K / V row of group kvg: clamp(row) * pitch + 16 g
A_LIM0 = Arg(6)                                    # P side, causal: kv row of group 0 minus the first tile's q0 minus 4 g; -2^30 otherwise
# (A_KR0: n * 256 + ((g ^ f(n)) << 4);  A_VR0: q * 256 + ((((n & 3) >> 1) ^ f(q)) << 4) + 8 (n & 1), q = 4 g + (n >> 2), f = f_swz16;  A_LDA: 16 g + 512 role;
#  A_EPI: wave * 64 * 272 + n * 272 + 8 g)

# ---- registers (KV's map where the meaning is the same: banks v16..79, row fragments v112..143, transposed fragments v144..175, PR, TMP, XA, QD, GD)
KR16 = [V(176 + i) for i in range(4)]              # row read addresses, k-step ks (32 head-dim columns)
VR16 = KV.VR + KV.VRB                              # transposed read addresses, d group dg
LIMT16 = [V(96 + i) for i in range(4)]             # P side, masked bodies: the limit of KV group kvg relative to the tile
L4R = V(100)                                       # lane * 4: source offset of the 4-byte LDS-DMA that stages L / -delta


def LR16(par):                                     # L (P side) / -delta (dS side) of a tile in the accumulator's register layout: q = 16 qg + 4 g + i -> 4 qg + i
    return V(80 + 8 * par, 8)


def RP16(qg, ks):
    return V(112 + 4 * (4 * qg + ks), 4)


def TP16(dg):
    return V(144 + 4 * dg, 4)


def ACC16(kvg, dg):
    return A(4 * (8 * kvg + dg), 4)


def FF16(kvg, ks):
    return A(128 + 4 * (4 * kvg + ks), 4)


def TILE(par, qg, kvg):
    return V(16 + 32 * par + 8 * kvg + 4 * qg, 4)


def PACKED(par, kvg):
    return V(16 + 32 * par + 8 * kvg, 4)


class GenDKV16(base.GenDKV):
    # windows of a 64-gap body: GenDKV's, in units of the shorter MFMA
    DEFAULTS = {"valu_p": (2.0, 62.0), "valu_s": (2.0, 62.0), "rowread": (0.0, 30.0), "trread": (32.0, 62.0), "dma": (2.0, 24.0), "lread": (40.0, 62.0),
                "lread_p": (2.0, 24.0), "opt": (), "abl": ()}

    def __init__(self, bf16=False, **cfg):
        super().__init__(bf16, **cfg)
        assert not self.kfold and not self.qsplit, "the 16x16x32 dK / dV generator has no kfold / qsplit bodies"
        self.mfma = "v_mfma_f32_16x16x32_bf16" if bf16 else "v_mfma_f32_16x16x32_f16"

    # ------------------------------------------------------------------ MFMA lists
    def acc_mfmas(self, par):
        """phase A: ACC[kvg][dg] += T(dg) . X(kvg), X = the packed fragments in bank `par`"""
        out = []
        for dg in range(8):
            for kvg in range(4):
                out.append(mk(self.mfma, ACC16(kvg, dg), TP16(dg), PACKED(par, kvg), ACC16(kvg, dg), tag="mfma"))
        return out

    def row_mfmas(self, par, cinit):
        """phase B: tile (qg, kvg) of bank `par` = R(qg, ks) . F(kvg, ks) over the 4 k-steps (dS side: starting from -delta)"""
        out = []
        for ks in range(4):
            for qg in range(2):
                for kvg in range(4):
                    dst = TILE(par, qg, kvg)
                    c0 = LR16(par).sub(4 * qg, 4) if cinit else 0
                    out.append(mk(self.mfma, dst, RP16(qg, ks), FF16(kvg, ks), c0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    def stream_p(self, kvb, par, masked):
        """P side: P = 2^(S c - L[q]) for half kvb of the bank (KV groups 2 kvb, 2 kvb + 1), pairs packed in place, then each group's fragment to the slot."""
        s, L = KV.BK(par, kvb), LR16(par)
        out = []
        if masked:      # causal: q (tile-local: 16 qg + i, + 4 g folded into the limit) must be >= this lane's kv row
            t2 = KV.TMP[4 * kvb]
            out.append([mk("v_mov_b32", t2, KV.A_C, tag="valu"), mk("v_and_b32", t2, 0x80000000, t2, tag="valu"),
                        mk("v_xor_b32", t2, 0xff800000, t2, tag="valu")])
            for e in range(16):
                kvg, qg, i = 2 * kvb + (e >> 3), (e >> 2) & 1, e & 3
                out.append([mk("v_cmp_ge_i32", VCC, 16 * qg + i, LIMT16[kvg], tag="valu"),
                            mk("v_cndmask_b32", s[e], t2, s[e], VCC, tag="valu")])
        for k in range(8 + 2):
            F, E, C = [], [], []
            if k < 8:
                for e in (2 * k, 2 * k + 1):
                    F.append(mk("v_fma_f32", s[e], s[e], KV.A_C, Neg(L[e % 8]), tag="valu"))
            if 0 <= k - 1 < 8:
                for e in (2 * (k - 1), 2 * (k - 1) + 1):
                    E.append(mk("v_exp_f32", s[e], s[e], tag="trans"))
            if 0 <= k - 2 < 8:
                e = 2 * (k - 2)
                C.append(mk(self.cvt, s[8 * (e // 8) + (e % 8) // 2], s[e], s[e + 1], tag="valu"))
            out += F + E + C
            if k - 2 in (3, 7):      # a KV group's four words are packed: hand them to the partner wave
                h = (k - 2) // 4
                out.append(mk("ds_write_b128", KV.XA, s.sub(8 * h, 4), tag="lds", offset=par * 4096 + 1024 * (2 * kvb + h)))
        return out

    # stream_pread, stream_ds: GenDKV's (the same bank layout, the same packed words)

    def stream_rowread(self, base_):
        return [mk("ds_read_b128", RP16(qg, ks), KR16[ks], tag="lds", offset=base_ + 4096 * qg) for ks in range(4) for qg in range(2)]

    def stream_trread(self, base_):
        out = []
        for dg in range(8):
            out.append(mk("ds_read_b64_tr_b16", TP16(dg).sub(0, 2), VR16[dg], tag="lds", offset=base_))
            out.append(mk("ds_read_b64_tr_b16", TP16(dg).sub(2, 2), VR16[dg], tag="lds", offset=base_ + 16 * 256))
        return out

    def stream_dma(self, par, P=True):
        out = [[mk("s_mov_b32", M0, KV.S_M0Q, tag="salu"), mk("s_nop", 0, tag="salu")]]
        for i in range(2):
            out.append(mk("buffer_load_dwordx4", KV.QD[i], KV.A_QRS, KV.S_QOFF, tag="dma", offen=True, offset=1024 * i, lds=True))
        out.append([mk("s_add_u32", M0, KV.A_LDSW, KV.G_RING + par * KV.SLOT, tag="salu"), mk("s_nop", 0, tag="salu")])
        for i in range(2):
            out.append(mk("buffer_load_dwordx4", KV.GD[i], KV.A_GRS, KV.S_GOFF, tag="dma", offen=True, offset=1024 * i, lds=True))
        skip = self.p.fresh("ld_skip")
        out.append([mk("s_cmp_eq_u32", KV.A_LDM0, 0, tag="salu"), mk("s_cbranch_scc1", Label(skip), tag="branch"),
                    mk("s_add_u32", M0, KV.A_LDM0, (par ^ 1) * 256, tag="salu"), mk("s_nop", 0, tag="salu"),
                    mk("buffer_load_dword", L4R, KV.A_LRS, KV.S_LOFF, tag="dma", offen=True, lds=True),
                    Ins("label", (Label(skip),))])
        return out

    def stream_lread(self, par):
        return [mk("ds_read_b128", LR16(par).sub(4 * qg, 4), KV.A_LDA, tag="lds", offset=KV.LD_BASE + par * 256 + 64 * qg) for qg in range(2)]

    # ------------------------------------------------------------------ one body (GenDKV.body at twice the gaps)
    def body(self, role, par, acc=True, valu=True, row=True, masked=False, tr=True, rr=True, name="body"):
        p, cfg = self.p, self.cfg
        abl = set(cfg["abl"]) if name.startswith("F") else set()
        ng = 64
        P = role == 0
        if P:
            mf = (self.acc_mfmas(par) if acc else [None] * 32) + (self.row_mfmas(par, False) if row else [None] * 32)
        else:
            mf = (self.acc_mfmas(par ^ 1) if acc else [None] * 32) + (self.row_mfmas(par ^ 1, True) if row else [None] * 32)
        if "mfma" in abl:
            mf = [None] * ng
        load = [0.0] * ng
        slots = [[] for _ in range(ng)]
        pre = []
        if not acc:
            pre += [mk("s_nop", 15), mk("s_nop", 15)]
        if P and masked and valu:
            # this lane's limits relative to tile t+1: LIM0 + 16 kvg - 32 (t + 1)
            pre.append(mk("s_add_u32", KV.S_TMP, KV.S_T, 1))
            pre.append(mk("s_lshl_b32", KV.S_TMP, KV.S_TMP, 5))
            for kvg in range(4):
                pre.append(mk("v_subrev_u32", LIMT16[kvg], KV.S_TMP, A_LIM0))
            pre.append(mk("s_nop", 0))
            for kvg in range(1, 4):
                pre.append(mk("v_add_u32", LIMT16[kvg], 16 * kvg, LIMT16[kvg]))
        if "dma" not in abl:
            sched.place(load, slots, self.stream_dma(par, P), cfg["dma"][0], cfg["dma"][1], 2)
            sched.place(load, slots, self.stream_lread(par), cfg["lread"][0], cfg["lread"][1], 8)
        if P:
            if rr and "rowread" not in abl:
                sched.place(load, slots, self.stream_rowread(KV.Q_RING), cfg["rowread"][0], cfg["rowread"][1], 3)
            if tr and "trread" not in abl:
                sched.place(load, slots, self.stream_trread(KV.G_RING + (par ^ 1) * KV.SLOT), cfg["trread"][0], cfg["trread"][1], 4)
            if valu and "valu" not in abl:
                w = cfg["valu_p"]
                sched.place(load, slots, self.stream_p(0, par ^ 1, masked), w[0], w[1] - 2.0, 5)
                sched.place(load, slots, self.stream_p(1, par ^ 1, masked), w[0], w[1], 6)
        else:
            if valu and "valu" not in abl:
                sched.place(load, slots, self.stream_pread(par), 0.0, 2.0, 1)
                w = cfg["valu_s"]
                sched.place(load, slots, self.stream_ds(0, par), w[0], w[1] - 2.0, 5)
                sched.place(load, slots, self.stream_ds(1, par), w[0], w[1], 6)
            if rr and "rowread" not in abl:
                sched.place(load, slots, self.stream_rowread(KV.G_RING + (par ^ 1) * KV.SLOT), cfg["rowread"][0], cfg["rowread"][1], 3)
            if tr and "trread" not in abl:
                sched.place(load, slots, self.stream_trread(KV.Q_RING), cfg["trread"][0], cfg["trread"][1] - 6.0, 4)
        bk1 = [mk("s_add_u32", KV.S_T, KV.S_T, 1, tag="salu"), mk("s_add_u32", KV.S_QOFF, KV.S_QOFF, KV.A_QTILE, tag="salu"),
               mk("s_add_u32", KV.S_GOFF, KV.S_GOFF, KV.A_GTILE, tag="salu"), mk("s_add_u32", KV.S_LOFF, KV.S_LOFF, 128, tag="salu"),
               mk("s_add_u32", KV.S_QSLOT, KV.S_QSLOT, 1, tag="salu"), mk("s_and_b32", KV.S_QSLOT, KV.S_QSLOT, 3, tag="salu"),
               [mk("s_cmp_eq_u32", KV.S_QSLOT, 0 if P else 2, tag="salu"), mk("s_cselect_b32", KV.S_BUMP, 4 * KV.SLOT, 0, tag="salu")],
               mk("s_sub_u32", KV.S_BUMP, KV.SLOT, KV.S_BUMP, tag="salu"),
               mk("s_add_u32", KV.S_TMP2, KV.S_QSLOT, 1, tag="salu"), mk("s_and_b32", KV.S_TMP2, KV.S_TMP2, 3, tag="salu"),
               mk("s_lshl_b32", KV.S_TMP2, KV.S_TMP2, 13, tag="salu"), mk("s_add_u32", KV.S_M0Q, KV.S_TMP2, KV.A_LDSWQ, tag="salu")]
        moving = KR16 if P else VR16
        bk2 = [mk("v_add_u32", r, KV.S_BUMP, r, tag="valu") for r in moving]
        if "bk" in abl:
            bk1, bk2 = [], []
        sched.place(load, slots, bk1, 40.0, 55.0, 9)
        sched.place(load, slots, bk2, 56.0, 63.0, 9)
        self.last_load = load
        post = [mk("s_waitcnt", vmcnt=0, lgkmcnt=0)]
        if "barrier" not in abl:
            post.append(mk("s_barrier"))
        self.emit_body(p, mf, slots, pre=pre, post=post, bookkeeping=bk1 + bk2, name="dK/dV 16x16 body %s (%s side)" % (name, "P" if P else "dS"))

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(4):
            p.emit("v_xor_b32", KR16[ks], ks << 6, KV.A_KR0)
        for dg in range(8):
            p.emit("v_xor_b32", VR16[dg], dg << 5, KV.A_VR0)
        p.emit("v_mbcnt_lo_u32_b32", L4R, -1, 0)
        p.emit("s_nop", 0)
        p.emit("v_mbcnt_hi_u32_b32", L4R, -1, L4R)
        # own rows (K or V) -> B fragments in AGPRs
        for kvg in range(4):
            for ks in range(4):
                p.emit("global_load_dwordx4", FF16(kvg, ks), A_FO[kvg], KV.A_FB, offset=64 * ks)
        p.emit("v_lshlrev_b32", L4R, 2, L4R)
        # DMA source offsets of piece 1: rows 4 further down flip bit 3 of this generator's granule swizzle (bwd_d128_gen.f_swz16: (r & 7) << 1)
        p.emit("v_mov_b32", KV.QD[0], KV.A_QD0)
        p.emit("v_mov_b32", KV.GD[0], KV.A_GD0)
        p.emit("v_xor_b32", KV.QD[1], 128, KV.A_QD0)
        p.emit("v_xor_b32", KV.GD[1], 128, KV.A_GD0)
        p.emit("v_mov_b32", KV.XA, KV.A_PXA)
        p.emit("v_add_u32", KV.QD[1], KV.A_QROW4, KV.QD[1])
        p.emit("v_add_u32", KV.GD[1], KV.A_GROW4, KV.GD[1])
        p.emit("s_mov_b32", KV.S_T, -2)
        p.emit("s_mov_b32", KV.S_QOFF, KV.A_QOFF0)
        # Q(0) -> Q ring slot 0
        p.emit("s_add_u32", M0, KV.A_LDSWQ, KV.Q_RING)
        p.emit("s_nop", 0)
        for i in range(2):
            p.emit("buffer_load_dwordx4", KV.QD[i], KV.A_QRS, KV.S_QOFF, offen=True, offset=1024 * i, lds=True)
        p.emit("s_mov_b32", KV.S_LOFF, KV.A_LOFF0)
        p.emit("s_cmp_eq_u32", KV.A_LDM0, 0)
        p.emit("s_cbranch_scc1", Label("no_ld0"))
        p.emit("s_mov_b32", M0, KV.A_LDM0)
        p.emit("s_nop", 0)
        p.emit("buffer_load_dword", L4R, KV.A_LRS, KV.S_LOFF, offen=True, lds=True)
        p.label("no_ld0")
        # running state of body -2 (GenDKV.build)
        p.emit("s_add_u32", KV.S_QOFF, KV.S_QOFF, KV.A_QTILE)
        p.emit("s_mov_b32", KV.S_GOFF, KV.A_GOFF0)
        p.emit("s_add_u32", KV.S_LOFF, KV.S_LOFF, 128)
        p.emit("s_mov_b32", KV.S_QSLOT, 0)
        p.emit("s_add_u32", KV.S_M0Q, KV.A_LDSWQ, KV.Q_RING + KV.SLOT)
        for i in range(128):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        p.emit("s_cmp_eq_u32", KV.A_ROLE, 1)
        p.emit("s_cbranch_scc1", Label("role_s"))
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")
        self.role_code(0)
        p.emit("s_branch", Label("epilogue"))
        p.label("role_s")
        for r in VR16:
            p.emit("v_add_u32", r, 2 * KV.SLOT, r)
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")
        self.role_code(1)

        # ---- epilogue (both roles): acc * factor -> 16 bit -> wave-private LDS image (rows of 272 B) over the dead rings; this lane holds
        # d = 16 dg + 4 g .. + 3 of KV row 16 kvg + n: 8 bytes per (kvg, dg)
        p.label("epilogue")
        p.emit("s_nop", 15)
        T = KV.TMP
        for kvg in range(4):
            for dg in range(0, 8, 2):
                for x in range(2):
                    acc = ACC16(kvg, dg + x)
                    for j in range(4):
                        p.emit("v_accvgpr_read_b32", T[4 * x + j], acc[j])
                p.emit("s_nop", 0)
                for j in range(8):
                    p.emit("v_mul_f32", T[j], KV.A_OSCALE, T[j])
                p.emit("s_nop", 0)
                p.emit(self.cvt, T[0], T[0], T[1])
                p.emit(self.cvt, T[1], T[2], T[3])
                p.emit(self.cvt, T[2], T[4], T[5])
                p.emit(self.cvt, T[3], T[6], T[7])
                p.emit("s_nop", 1)
                p.emit("ds_write_b64", KV.A_EPI, V(T[0].idx, 2), offset=16 * kvg * KV.EPI_ROWB + 32 * dg)
                p.emit("ds_write_b64", KV.A_EPI, V(T[2].idx, 2), offset=16 * kvg * KV.EPI_ROWB + 32 * (dg + 1))
                p.emit("s_nop", 1)
        p.emit("s_waitcnt", lgkmcnt=0)
        return p


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    ap.add_argument("--opt", default="", help="schedule windows / options (bwd_d128_gen.parse_opts)")
    ap.add_argument("--probe", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cfg = base.parse_opts(a.opt)
    if "abl" in cfg and not a.probe:
        sys.exit("bwd_dkv_m16_gen.py: %r contains timing-probe options; they need --probe" % a.opt)
    for bf16 in (False, True):
        prog = GenDKV16(bf16, **cfg).build()
        fn = "fa2_bwd_dkv_m16_%s.inc" % ("bf16" if bf16 else "f16")
        base.write_atomic(os.path.join(a.out, fn),
                          "// GENERATED by csrc/gen/bwd_dkv_m16_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + base.render_inline(prog, "fa2dkv16"))
        print(fn, len(prog.ins), "instructions")


if __name__ == "__main__":
    main()
