#!/usr/bin/env python3
"""Forward main block for D = 128 built on v_mfma_f32_16x16x32 (round 5) — the same software pipeline, staging, seams and sum-check bodies as
fwd_d128_gen.py (reference counterpart: kernel_fp16.cu:381-508), another MFMA tile.

Why.  The chip is power-limited under this kernel: a launch takes what its ENERGY takes.  A loop of nothing but MFMAs sustains
1 981 TF with the 16x16x32 form against 1 728 with 32x32x16 on the same operands (it moves half the accumulator bytes per FLOP), and a synthetic
tile body with every filler of the real one runs 8.4 % faster (tools/ubench/mfma_shape_probe.py, profiles/r16_mfma_shape_probe.json).

Shape (one workgroup = 4 waves = 256 Q rows, one wave per SIMD, wave = 64 Q rows = four 16-row groups qg, KV tiles of 64 = four 16-row groups kg):
    S^T[kv,q] tile (kg, qg) = sum_ks  K[kg rows, 32 d of k-step ks] . Q^T[ks, qg]      4 x 4 x 4 = 64 MFMAs per tile, 4 registers per tile
    O^T[d,q]  tile (dg, qg) += sum_kvs V^T[16 d of dg, 32 kv of kvs] . P^T[kvs, qg]      8 x 4 x 2 = 64 MFMAs per tile
  MFMA layouts: A[m][k]: lane l holds m = l % 16, k = 8 (l / 16) .. +7;  B[k][n]: n = l % 16, same k;  D[m][n]: n = l % 16, m = 4 (l / 16) + i.
  So a lane (n, g = l / 16) holds, of Q row 16 qg + n, the scores kv = 16 kg + 4 g + i — a ROW IS SPREAD OVER FOUR LANES (g = 0..3).  Nothing in the
  fast bodies cares: they keep per-lane partial row sums (summed across the four lanes once, in the epilogue) and the sum check is per lane; only
  the max-first head / tail bodies and the rare repair reduce across lanes (v_permlane16_swap + v_permlane32_swap).
  A "q block" qb of the base generator is a PAIR of q groups (qg = 2 qb + h): the S / P bank of (qb, parity) keeps register e = 16 h + 4 kg + i, and
  the packed P of k-step kvs (kv groups 2 kvs, 2 kvs + 1) lands in registers 16 h + 8 kvs .. +3 — the very formula of the 32 x 32 layout
  (8 (e // 8) + (e % 8) // 2), so exp / pack streams carry over; a lane's bank now belongs to TWO rows (h = 0, 1): reference, sums and limits are
  per row.  The MFMA k-slot (g, j) of P.V stands for kv = 32 kvs + 16 (j >> 2) + 4 g + (j & 3): the V^T fragment is two transposed reads, rows
  32 kvs + 4 g .. +3 and 16 further down — what ds_read_b64_tr_b16 delivers per 16-lane group anyway.

opt=ct: the folded scale (the base generator's scheme: Q * scale*log2e rounded once to the I/O dtype, the running reference enters the first Q.K^T
k-step as its C operand — one 4-register tuple per q group here, all four registers of a 16 x 16 tile belong to one row).  It matters MORE with this
tile: 128 MFMA issues per body instead of 64 make the body issue-bound (the fillers alone take 76 % of a launch, profiles/r16_kbench_m16_sweep.txt),
and the fold takes 64 of them out.  The C tuples take v[176:191]: the V^T fragments of k-step 1 move to a[224:255], the K fragments to a
32-register pool (k-steps 2, 3 are read into the slots of 0, 1 once those MFMAs are issued; counted lgkmcnt waits, sched.lds_waits).

Not in this generator (the 32 x 32 kernels keep those launches): head dim 64.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
import fwd_d128_gen as base  # noqa: E402
from fwd_d128_gen import (A_C, A_EPI, A_FLAGS, A_WSB, A_KD0, A_KR0, A_KRS, A_KROW4, A_KTILE, A_LDSW, A_LIM0, A_LIM1, A_LSE0, A_NKRS, A_NQRS,  # noqa: E402,F401
                          A_NQW, A_NTW, A_NTWG, A_NVRS, A_QD0, A_QRS, A_QT16, A_QW, A_VD0, A_VR0, A_VROW4, A_VRS, A_VTILE, KD, NEG_INF,
                          PSUM_MAX, QD, SB, S_D, S_FLAG, S_FIX, S_KOFF, S_NFAST, S_NOVM, S_PF, S_QH, S_QM0, S_QSB, S_QSOFF, S_SUM, S_T, S_TMP,
                          S_TMP2, S_VOFF, S_WAVE, VBASE, VD)
from isa import A, V, Ins, Label, M0, Neg, VCC, mk  # noqa: E402

# Deferred-rescale threshold of the max-first bodies, log2 units: the reference of a row moves when a score outgrows it by more than this, so P <= 2^14 —
# inside fp16 (65504) with two octaves to spare; rounding is relative and O, l are f32, so where the reference sits changes nothing else.  The base
# generator's 8 (and the HIP kernels') dates from round 1; since round 6 the max-first bodies also carry the rest of a sweep whose fast bodies met a P
# beyond the 16-bit range (Gen16.lm_repair), on exactly the data whose row maxima keep climbing: on N(0, 6^2) logits a wave moves a reference (rare_m +
# the O rescale, ~half a body) in 15 % of its tiles at 8 and in 4.6 % at 14 (a simulation of the recurrence; tools/growth_cliff.py has the GPU times)
THR = 14.0

# ---- register map (everything the base generator does not fix)
KR = [V(208 + i) for i in range(4)]                # K / Q fragment read addresses, k-step ks (32 head-dim columns each)
VR = [V(212 + i) for i in range(4)]                # V^T fragment read addresses, 64-byte chunk dg >> 1
LA = [V(228), V(230)]                              # running partial row sums of THIS lane: row h = 0 of q block qb ...
LB = [V(229), V(231)]                              # ... and row h = 1
MCA = [V(232), V(234)]                             # reference (m * c, log2 units) of row h = 0 of q block qb ...
MCB = [V(233), V(235)]                             # ... h = 1
FSA = [V(236), V(238)]                             # pending O rescale factors (max-first bodies), rows h = 0 / 1
FSB = [V(237), V(239)]
TMP = [V(240 + i) for i in range(8)]               # scratch (row-max chains, tile sums, rescale block, epilogue)
EPX = [V(248 + i) for i in range(4)]               # epilogue: 1 / l of the wave's four rows of this lane; rare_sum scratch
LIMQ = [[V(216), V(217)], [V(218), V(219)]]        # masked bodies: kv limits of this lane's two rows of q block qb (the two blocks' streams interleave)
# (v252..255: QD of the base generator — the staging of the next item's Q runs through every body)
MC = [MCA, MCB]
FS = [FSA, FSB]
LS = [LA, LB]


def OACC(dg, qg, ndg=8):                           # (ndg: 16-column d groups of O = head dim / 16; nks: 32-column k-steps of Q.K^T = head dim / 32)
    return A(4 * (ndg * qg + dg), 4)


def QF(qg, ks, nks=4):
    return A(128 + 4 * (nks * qg + ks), 4)


def KF(kg, ks, nks=4):
    return A(192 + 4 * (nks * kg + ks), 4)


def VF(dg, kvs, ndg=8):
    return V(144 + 4 * (ndg * kvs + dg), 4)


# "ct" (folded scale): C tuples in v[176:191], so the V^T fragments of k-step 1 live in a[224:255] and the K fragments in a 32-register pool
def KF_POOL(kg, ks):
    return A(192 + 16 * (ks % 2) + 4 * kg, 4)


def VF_CT(dg, kvs):
    return V(144 + 4 * dg, 4) if kvs == 0 else A(224 + 4 * dg, 4)


VRO = [V(196 + i) for i in range(4)]               # "ct": read addresses of the ODD d groups (see stream_vread)
# "lm" (with "ct"): the row sums ride the matrix pipe.  One more MFMA per (q group, P.V k-step): A = a constant tuple that holds 0.25 in the rows m with
# m % 4 == qg and 0 elsewhere, B = the packed P the P.V product consumes, C = D = LSV — so register qg of LSV gains a quarter of the row sum of row
# 16 qg + n on each of the row's four lanes: the very registers, and the very meaning (a lane's share; the epilogue adds the four lanes up), of the
# partial sums the v_add chains keep.  The constants take the registers the repair blocks of the sum check no longer need.
ONES16 = [V(192, 4), V(200, 4), V(204, 4), V(248, 4)]
LSV = V(228, 4)                                    # = LS[h][qb] at index 2 qb + h
TS = V(240, 4)                                     # = TMP[0:4]: the other end of the fast bodies' row-sum chain (Gen16.sum_links)
LM_MAX = float(2.0 ** 40)                          # a lane's share of a row sum at or above this (or NaN): redo the item in safe mode (see Gen16.__init__)
CT16 = [V(176 + 4 * qg, 4) for qg in range(4)]     # -(reference) of this lane's row of q group qg, four copies (the C operand of a 16 x 16 tile)
DSH16 = [[V(192), V(193)], [V(194), V(195)]]       # [h][qb]: pending shift of the NEXT tile's scores of row h of q block qb (see rare_fix)


class Gen16(base.Gen):
    # schedule windows of a 128-gap body: the base generator's, in units of the shorter MFMA
    # (measured, tools/kbench.py, folded body, c2 / b8 / c4 us on one box: dma 8:40 + vread 66:100 203.6 / 837.3 / 422.4; dma 4:36 + vread 66:80 200.6 / 823.2 /
    #  413.6 — the LDS-DMA pieces want to be early, the V^T reads compact since their image is conflict-free; K-read and exp / pack windows are flat:
    #  profiles/r16_kbench_m16_sweep.txt, r16_kbench_m16_windows2.txt)
    DEFAULTS16 = {"m": (4.0, 20.0), "e": (20.0, 128.0), "vread": (66.0, 80.0), "kread": (0.0, 48.0), "dma": (4.0, 36.0), "mmask": (4.0, 48.0),
                  "se0": (0.0, 96.0), "se1": (16.0, 120.0), "sc0": (96.0, 116.0), "sc1": (120.0, 128.0)}

    # head dim 64 (a 64-gap body, 72 with opt=lm): the same windows at half the length
    DEFAULTS16_64 = {"m": (2.0, 12.0), "e": (12.0, 64.0), "vread": (34.0, 44.0), "kread": (0.0, 24.0), "dma": (2.0, 20.0), "mmask": (2.0, 24.0),
                     "se0": (0.0, 48.0), "se1": (8.0, 60.0), "sc0": (48.0, 58.0), "sc1": (60.0, 64.0)}

    def __init__(self, bf16=False, hd=128, **cfg):
        opt = tuple(cfg.get("opt", ()))
        assert "lmfma" not in opt, "the 16x16x32 generator has no lmfma bodies"
        user = {(k[4:] if k.startswith("d64_") else k): v for k, v in cfg.items() if hd == 64 or not k.startswith("d64_")}
        super().__init__(bf16, hd=hd, **cfg)
        self.NKS16, self.NDG = hd // 32, hd // 16         # k-steps of Q.K^T (32 head-dim columns each), 16-column d groups of O
        nks, ndg = self.NKS16, self.NDG
        self.cfg.update({k: v for k, v in (self.DEFAULTS16 if hd == 128 else self.DEFAULTS16_64).items() if k not in user})
        if "kread_ct" not in user:
            # "ct": gap window of the K reads of k-steps 0, 1 (2, 3 follow their pool slots; head dim 64 has no others)
            self.cfg["kread_ct"] = (16.0, 64.0) if hd == 128 else (8.0, 32.0)
        self.kf16 = KF_POOL if self.ct else (lambda kg, ks: KF(kg, ks, nks))
        self.vf16 = VF_CT if self.ct else (lambda dg, kvs: VF(dg, kvs, ndg))
        self.oacc16 = lambda dg, qg: OACC(dg, qg, ndg)
        self.qf16 = lambda qg, ks: QF(qg, ks, nks)
        self.mfma = "v_mfma_f32_16x16x32_bf16" if bf16 else "v_mfma_f32_16x16x32_f16"
        # opt=lm: row sums on the matrix pipe (ONES16 above), in EVERY body, and fast bodies that are exp + pack and nothing else: no adds, no check, no
        # repair blocks.  What the sum check guarded against — a P beyond the 16-bit type's range, an accumulator beyond f32's — is caught later and
        # coarser: an fp16 P above 65504 packs to inf and the row sum it enters stays inf; a lane's share is compared with 2^40 wherever it is about to be
        # scaled down (rare_rescale) and in the epilogue, and a share that fails sends the ITEM through the safe-mode redo (max-first bodies only) the sum
        # check uses for growth beyond 120 octaves.  Below the threshold every P is below 2^42 and every accumulator finite (2^42 * |V| * Nkv).  The
        # price is the in-place repair: a row whose scores outgrow the reference of its first tiles by 16 octaves (fp16) costs its item a second sweep.
        # Contract: FA2_CONTRACT_LSUM_P16 — l adds the ROUNDED P (what P.V consumes; the weights O applies then sum to exactly one).
        # Measured: tools/ubench/mfma_shape_probe.py (profiles/r18_mfma_sum_probe.json): 64 v_add_f32 out, 8 MFMAs in: -3.9 % on the synthetic body;
        # the kernel (tools/kbench.py, one box, profiles/r18_kbench_lm_*.txt): c2 196.6 -> 188.6 us, B8 788.8 -> 751.5, c4 387.9 -> 372.4 (-4.0 .. 4.7 %).
        # Windows (same files): the exp + pack streams anywhere (no check to stay ahead of, no packs to hold back), the V^T reads spread over 74 .. 110;
        # every variant within 0.5 % of the best — the optimum is flat.
        # Without opt=ct (the f32-scale bodies: fast bodies are fma + exp + pack) the constants live in a[224:255]: at head dim 128 the K fragments
        # shrink to the 32-register pool of the ct bodies (k-steps 2, 3 are read into the slots of 0, 1), at head dim 64 those registers are free.
        self.lm = "lm" in self.opt
        self.repairs = set()
        # the conflict-free V image (32-byte halves flipped for rows with (row >> 2) & 1, odd d groups read through a second address set: stream_vread):
        # the folded bodies since round 5; round 6: the f32-scale lm bodies too — bf16's default, config 3: SQ_LDS_BANK_CONFLICT 4.6e6 of 1.37e7 LDS
        # cycles per launch with the plain image (profiles/r20_c3_summary.txt) — whose second address set takes v[248:251] (free without the sum check)
        self.vflip = self.ct or self.lm
        self.vro = VRO if self.ct else EPX
        self.ones16 = ONES16 if self.ct else [A(224 + 4 * qg, 4) for qg in range(4)]
        if self.lm and not self.ct:
            self.kf16 = KF_POOL
            self.pool = hd == 128
        # (round 6) the eight row-sum links of the tile a body's softmax streams produce (tile t + 1) ride in the LAST 24 gaps of that body's Q.K^T phase
        # — behind the pack stream, AHEAD of P.V(t + 1) — so a fast body can look at the sums before a P beyond the 16-bit range has touched O: see
        # qk_phase, body_end and lm_repair
        self.npv, self.nqk = 8 * ndg, 16 * nks + (8 if self.lm else 0)
        self.ng = self.npv + self.nqk
        if self.lm:
            # link gaps lk0 .. lk3 (q block 0: P.V k-step 0, 1; q block 1: k-step 0, 1): a link needs the packs of ITS registers only, and a pack stream
            # finishes k-step 0 of both rows at its half — so only the last two links wait for the end of q block 1's stream
            # Measured against the round-5 library (links inside the NEXT body's P.V phase, no per-tile check), one box, tools/kbench.py
            # (profiles/r20_kbench_windows*.txt): links in the last 24 gaps + the round-5 windows squeezed in front of them -2.2 % (c2), -4 % (head dim 64);
            # these link gaps -1.1 %; with the V^T reads moved into the gaps the squeeze left empty (vread 104:136 / 52:72) -0.5 % (c2), -0.3 % (c4),
            # -0.6 .. 1.7 % (head dim 64): what the repair costs data that never needs it.
            lmw = ({"e": (20.0, 110.0), "vread": (104.0, 136.0), "se0": (0.0, 96.0), "se1": (24.0, 128.0), "kread_ct": (16.0, 60.0),
                    "lk0": (100.0, 103.0), "lk1": (106.0, 109.0), "lk2": (112.0, 115.0), "lk3": (130.0, 132.0)} if hd == 128 else
                   {"e": (12.0, 46.0), "vread": (52.0, 72.0), "se0": (0.0, 50.0), "se1": (8.0, 66.0), "kread_ct": (8.0, 30.0),
                    "lk0": (52.0, 54.0), "lk1": (56.0, 58.0), "lk2": (60.0, 62.0), "lk3": (67.0, 69.0)})
            for k, w in lmw.items():
                if k not in user:
                    self.cfg[k] = w
            for k, w in cfg.items():          # "lm_<key>" (head dim 128) / "d64_lm_<key>" (64): a schedule tunable of the lm bodies only (window sweeps, tools/kbench.py)
                if hd == 128 and k.startswith("lm_"):
                    self.cfg[k[3:]] = w
                elif hd == 64 and k.startswith("d64_lm_"):
                    self.cfg[k[7:]] = w

    def body(self, par, *args, **kw):
        start = len(self.p.ins)
        super().body(par, *args, **kw)
        # legality of the V^T read window (an input: tools/kbench.py sweeps it): the fragments read in this body are P.V(t+1)'s — a read ahead of the last
        # P.V(t) MFMA that takes its registers would hand that MFMA the next tile's data
        ins = [x for x in self.p.ins[start:] if x.op != "label"]
        last = {}
        for i, x in enumerate(ins):
            if x.op.startswith("v_mfma"):
                for (kind, lo, hi) in base.Gen._regs(x)[0]:
                    for r in range(lo, hi):
                        last[(kind, r)] = i
        if self.lm:
            # the row-sum links read the packed P of the tile this body's softmax streams produce: every pack that writes a link's B registers must be
            # ahead of it (the exp / pack windows are inputs: tools/kbench.py sweeps them)
            ones = {(o.kind, o.idx) for o in self.ones16}
            for i, x in enumerate(ins):
                if x.op.startswith("v_mfma") and (x.ops[1].kind, x.ops[1].idx) in ones:
                    lo, hi = x.ops[2].idx, x.ops[2].idx + x.ops[2].n
                    for j in range(i + 1, len(ins)):
                        if ins[j].op == self.cvt and lo <= ins[j].ops[0].idx < hi and ins[j].ops[0].kind == "v":
                            raise ValueError("illegal schedule: a row-sum link (%s) ahead of a pack of its P registers (%s)" % (x, ins[j]))
        for i, x in enumerate(ins):
            if x.op == "ds_read_b64_tr_b16":
                for (kind, lo, hi) in base.Gen._regs(x)[1]:
                    if any(last.get((kind, r), -1) > i for r in range(lo, hi)):
                        raise ValueError("illegal schedule: a V^T fragment read (%s) ahead of a P.V MFMA of this body that reads its registers" % x)

    # ------------------------------------------------------------------ MFMA lists
    def pv_mfmas(self, par, qb):
        out = []
        b = SB(qb, par)
        for kvs in range(2):
            for dg in range(self.NDG):
                for h in range(2):
                    acc = self.oacc16(dg, 2 * qb + h)
                    out.append(mk(self.mfma, acc, self.vf16(dg, kvs), b.sub(16 * h + 8 * kvs, 4), acc, tag="mfma"))
        return out

    def sum_links(self, par_s, mode):
        """opt=lm: the eight row-sum MFMAs of the tile whose packed P sits in the banks of parity par_s — (q block, P.V k-step, h), the order the pack
        streams finish the registers in.  mode 'acc': D = C = LSV (max-first bodies: their P cannot overflow); 'to_ts': the chain starts from LSV and
        ends in TS; 'to_lsv': from TS into LSV — the fast bodies alternate, so the tuple a chain started from still holds the sums of the tiles before
        when the check behind the chain (body_end) fails."""
        src, dst = {"acc": (LSV, LSV), "to_ts": (LSV, TS), "to_lsv": (TS, LSV)}[mode]
        out = []
        for qb in range(2):
            b = SB(qb, par_s)
            for kvs in range(2):
                for h in range(2):
                    out.append(mk(self.mfma, dst, self.ones16[2 * qb + h], b.sub(16 * h + 8 * kvs, 4), dst if out else src, tag="mfma"))
        return out

    def qk_phase(self, par, s1, s2, fast):
        nq = 16 * self.NKS16
        qk = self.qk_mfmas(par) if s2 else [None] * nq
        if not self.lm:
            return qk
        links = self.sum_links(par ^ 1, ("to_ts" if par == 0 else "to_lsv") if fast else "acc") if s1 else [None] * 8
        # the links sit in the gaps cfg lk0 .. lk3 name (two each, chain order), the Q.K^T MFMAs in the phase's other gaps in their own order.  Behind
        # Q.K^T MFMA 31 at the earliest: place_pool_kreads counts on the first two k-steps sitting in gaps npv .. npv + 31
        # (the max-first bodies run both q blocks' exp / pack streams over one window, cfg e, that ends ahead of the last 24 gaps: their links sit there)
        at = [int(x) for k in ("lk0", "lk1", "lk2", "lk3") for x in self.cfg[k]] if fast else [self.ng - 24 + 3 * j for j in range(8)]
        assert at == sorted(set(at)) and self.npv + (32 if self.pool else 0) <= at[0] and at[-1] < self.ng, at
        out, qi, li = [], 0, 0
        for gap in range(self.npv, self.ng):
            if li < 8 and gap == at[li]:
                out.append(links[li])
                li += 1
            else:
                out.append(qk[qi])
                qi += 1
        assert qi == nq and li == 8
        return out

    def body_end(self, par, name, fast, s1):
        """opt=lm, fast bodies: did the tile just packed (t + 1) overflow?  An fp16 P beyond 65504 is inf, its row sum inf, the other registers of the link's
        tuple NaN (0 * inf); a share at or beyond LM_MAX fails too.  Nothing of that tile has touched O yet: lm_repair recomputes it in place."""
        if not (self.lm and fast and s1):
            return
        p = self.p
        x = TS if par == 0 else LSV
        lab = "lm_repair_%d" % par
        p.emit("v_max3_f32", TMP[4], x[0], x[1], x[2])
        p.emit("s_nop", 0)
        p.emit("v_max_f32", TMP[4], TMP[4], x[3])
        p.emit("s_nop", 0)
        p.emit("v_cmp_ngt_f32", VCC, LM_MAX, TMP[4])
        p.emit("s_cbranch_vccnz", Label(lab))
        p.label(lab + "_ret")
        if lab not in self.repairs:
            self.repairs.add(lab)
            self.rare.append(self.lm_repair(lab, par))

    def qk_mfmas(self, par):
        out = []
        for ks in range(self.NKS16):
            for qb in range(2):
                for h in range(2):
                    for kg in range(4):
                        dst = SB(qb, par).sub(16 * h + 4 * kg, 4)
                        c0 = CT16[2 * qb + h] if self.ct else 0
                        out.append(mk(self.mfma, dst, self.kf16(kg, ks), self.qf16(2 * qb + h, ks), c0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    @staticmethod
    def _order():
        """the 16 register pairs of a bank, rows h = 0 / 1 alternating (two dependent chains never back to back)"""
        return [(16 * h + 2 * k) for k in range(8) for h in range(2)]

    def stream_exp(self, qb, par):
        """max-first bodies: P = 2^(S*c - m*c) in place, per-row sum chains, pairs packed in place; skewed by pair."""
        b = SB(qb, par)
        prs = self._order()
        out = []
        for k in range(16 + 3):
            F, E, Ad, C = [], [], [], []
            if k < 16 and not self.fold:
                e = prs[k]
                for x in (e, e + 1):
                    F.append(mk("v_fma_f32", b[x], b[x], A_C, Neg(MC[e // 16][qb]), tag="valu"))
            if 0 <= k - 1 < 16:
                e = prs[k - 1]
                E += [mk("v_exp_f32", b[e], b[e], tag="trans"), mk("v_exp_f32", b[e + 1], b[e + 1], tag="trans")]
            if 0 <= k - 2 < 16 and not self.lm:
                e = prs[k - 2]
                Ad.append(mk("v_add_f32", LS[e // 16][qb], LS[e // 16][qb], b[e], tag="valu"))
            if 0 <= k - 3 < 16:
                e = prs[k - 3]
                if not self.lm:
                    Ad.append(mk("v_add_f32", LS[e // 16][qb], LS[e // 16][qb], b[e + 1], tag="valu"))
                C.append(mk(self.cvt, b[8 * (e // 8) + (e % 8) // 2], b[e], b[e + 1], tag="valu"))
            out += F + E + Ad + C
        return out

    def stream_exp_sum(self, qb, par):
        """fast bodies (no row-max stream; base.Gen.stream_exp_sum has the argument): P against the current references of the lane's two rows, the
        tile's sums per row (ta: h = 0, tb: h = 1) started afresh, added to the running sums, and ONE check: ta + tb bounds each of this lane's 32 P."""
        b = SB(qb, par)
        if self.lm:
            # exp and pack, nothing else (the row sums: pv_mfmas of the NEXT body; what replaces the check: __init__)
            prs = self._order()
            out = []
            for k in range(16 + 3):
                if k < 16 and not self.fold:
                    e = prs[k]
                    for x in (e, e + 1):
                        out.append(mk("v_fma_f32", b[x], b[x], A_C, Neg(MC[e // 16][qb]), tag="valu"))
                if 0 <= k - 1 < 16:
                    e = prs[k - 1]
                    out += [mk("v_exp_f32", b[e], b[e], tag="trans"), mk("v_exp_f32", b[e + 1], b[e + 1], tag="trans")]
                if 0 <= k - 3 < 16:
                    e = prs[k - 3]
                    out.append(mk(self.cvt, b[8 * (e // 8) + (e % 8) // 2], b[e], b[e + 1], tag="valu"))
            return out
        ta, tb, ts = TMP[4 * qb], TMP[4 * qb + 1], TMP[4 * qb + 2]
        tsum = [ta, tb]
        prs = self._order()
        out = []
        for k in range(16 + 3):
            F, E, Ad = [], [], []
            if k < 16 and not self.fold:
                e = prs[k]
                for x in (e, e + 1):
                    F.append(mk("v_fma_f32", b[x], b[x], A_C, Neg(MC[e // 16][qb]), tag="valu"))
            if 0 <= k - 1 < 16:
                e = prs[k - 1]
                E += [mk("v_exp_f32", b[e], b[e], tag="trans"), mk("v_exp_f32", b[e + 1], b[e + 1], tag="trans")]
            if "pkadd" in self.opt:
                # opt=pkadd: a pair (two consecutive registers of one row) enters its row's sums with ONE packed add — two partial sums per row (the four
                # scratch registers of the q block).  An anti-lever beside the energy-bound 32 x 32 body (-6.4 %); this body is issue-bound.
                if 0 <= k - 2 < 16:
                    e = prs[k - 2]
                    tp = V(TMP[4 * qb + 2 * (e // 16)].idx, 2)
                    if e % 16 == 2:               # the row's second pair: both are through the exp stage now
                        Ad.append(mk("v_pk_add_f32", tp, b.sub(e - 2, 2), b.sub(e, 2), tag="valu"))
                    elif e % 16 != 0:
                        Ad.append(mk("v_pk_add_f32", tp, tp, b.sub(e, 2), tag="valu"))
                out += F + E + Ad
                continue
            if 0 <= k - 2 < 16:
                e = prs[k - 2]
                t = tsum[e // 16]
                if e % 16 == 0:                   # the row's first pair starts its chain
                    Ad.append(mk("v_add_f32", t, b[e], b[e + 1], tag="valu"))
                else:
                    Ad.append(mk("v_add_f32", t, t, b[e], tag="valu"))
            if 0 <= k - 3 < 16:
                e = prs[k - 3]
                if e % 16 != 0:
                    Ad.append(mk("v_add_f32", tsum[e // 16], tsum[e // 16], b[e + 1], tag="valu"))
            out += F + E + Ad
        if "pkadd" in self.opt:
            # row h: partial sums in TMP[4 qb + 2 h], +1 -> ta / tb (the registers the check and the rare block know)
            t0, t1, t2_, t3 = (TMP[4 * qb + i] for i in range(4))
            out.append(mk("v_add_f32", t0, t0, t1, tag="valu"))      # ta = row 0
            out.append(mk("v_add_f32", t1, t2_, t3, tag="valu"))     # tb = row 1
        out.append(mk("v_add_f32", LA[qb], LA[qb], ta, tag="valu"))
        out.append(mk("v_add_f32", LB[qb], LB[qb], tb, tag="valu"))
        out.append(mk("v_add_f32", ts, ta, tb, tag="valu"))
        lab = self.p.fresh("rare_s")
        out.append([mk("s_nop", 0, tag="salu"), mk("v_cmp_nge_f32", VCC, PSUM_MAX, ts, tag="valu"),
                    mk("s_cbranch_vccnz", Label(lab), tag="branch"), Ins("label", (Label(lab + "_ret"),))])
        self.pending_rare_sum.append((lab, qb, par))
        return out

    # stream_pack: the base generator's (the packed P of k-step kvs of row h lands in registers 16 h + 8 kvs .. + 3 by the same formula)
    def stream_pack(self, qb, par):
        return [] if self.lm else super().stream_pack(qb, par)          # (lm: packed by stream_exp_sum, a pair behind its exps)

    def trim_offsets(self, image, dst, a0, s_piece):
        """Head dims BELOW the body's (flag bits 8 .. 12 = nG, the 16-byte granules a row really has; D = 40 / 48 / 56 on the head-dim-64 body, 72 .. 120 on
        the 128 one): the LDS-DMA source offsets of an image's pieces in their general form.  A lane fills LDS slot (row r0 + RPP i, slot dslot) of piece i
        with the row's LOGICAL granule gl_i (the inverse of the image's read swizzle).  The fast path derives piece i from piece 0 by flipping offset bits,
        which needs a row pitch that is a multiple of the image row; here the pitch is whatever the caller's rows are (80 bytes at D = 40), so the offset
        is re-derived: a0 + ((gl_i - gl_0) << 4) + i * (RPP rows), and a granule the row does not have (gl_i >= nG) gets an offset beyond every
        descriptor (bit 31): the load returns zeros — the image's padded columns are zero, so are the Q fragments read from it, and nothing of a
        neighbouring row enters a product.  image: 'k' (also Q: the same swizzle), 'v'.  s_piece: SGPR holding RPP * row bytes - 1024 (the source stride
        between two pieces, minus the 1024 the instruction offset adds).  Uses TMP[0:8], S_D, S_TMP (after the caller is done with it)."""
        p, g = self.p, self.g
        G = g.ROWB // 16
        rpb = max(1, 256 // g.ROWB)
        kmask, vmask = min(G, 16) - 1, min(g.ROWB // 64, 4) - 1
        lane, dslot, r0, mark, gl0, t, t2 = TMP[7], TMP[6], TMP[5], TMP[4], TMP[3], TMP[0], TMP[1]
        p.emit("v_mbcnt_lo_u32_b32", lane, -1, 0)
        p.emit("v_mov_b32", mark, 0x80000000)
        p.emit("v_mbcnt_hi_u32_b32", lane, -1, lane)
        p.emit("s_lshr_b32", S_D, A_FLAGS, 8)
        p.emit("s_and_b32", S_D, S_D, 31)
        p.emit("v_and_b32", dslot, G - 1, lane)
        p.emit("v_lshrrev_b32", r0, G.bit_length() - 1, lane)

        def granule(dst_, i):
            """dst_ = the logical granule this lane's slot of piece i holds"""
            p.emit("v_add_u32", t2, g.RPP * i, r0)                         # row of the piece's lane (wave offsets are multiples of 16: they drop out)
            p.emit("s_nop", 0)
            if image == "k":
                if rpb > 1:
                    p.emit("v_lshrrev_b32", t2, rpb.bit_length() - 1, t2)
                    p.emit("s_nop", 0)
                p.emit("v_and_b32", t2, kmask, t2)
                p.emit("s_nop", 0)
                p.emit("v_xor_b32", dst_, t2, dslot)
            else:
                p.emit("v_lshrrev_b32", dst_, rpb.bit_length() - 1, t2) if rpb > 1 else p.emit("v_mov_b32", dst_, t2)
                p.emit("s_nop", 0)
                p.emit("v_and_b32", dst_, vmask, dst_)
                p.emit("s_nop", 0)
                p.emit("v_lshlrev_b32", dst_, 2, dst_)                       # chunk swizzle: granule bits 2 ..
                if self.vflip:                                              # ... and the flipped 32-byte halves of the conflict-free V image: granule bit 1
                    p.emit("v_lshrrev_b32", t2, 2, t2)
                    p.emit("s_nop", 0)
                    p.emit("v_and_b32", t2, 1, t2)
                    p.emit("s_nop", 0)
                    p.emit("v_lshl_or_b32", dst_, t2, 1, dst_)
                p.emit("s_nop", 0)
                p.emit("v_xor_b32", dst_, dst_, dslot)
            p.emit("s_nop", 0)

        granule(gl0, 0)
        p.emit("s_mov_b32", S_TMP, 0)
        for i in range(len(dst)):
            if i:
                granule(t, i)
                p.emit("s_add_u32", S_TMP, S_TMP, s_piece)
                p.emit("v_sub_u32", t2, t, gl0)
                p.emit("s_nop", 0)
                p.emit("v_lshlrev_b32", t2, 4, t2)
                p.emit("s_nop", 0)
                p.emit("v_add_u32", dst[i], t2, a0)
                p.emit("s_nop", 0)
                p.emit("v_add_u32", dst[i], S_TMP, dst[i])
                gi = t
            else:
                p.emit("v_mov_b32", dst[0], a0)
                gi = gl0
            p.emit("v_cmp_gt_u32", VCC, S_D, gi)                            # the row has this granule
            p.emit("s_nop", 0)
            p.emit("v_cndmask_b32", dst[i], mark, dst[i], VCC)

    def trim_test(self, lab):
        """scc1 -> the launch's head dim is the body's (flag bits 8 .. 12 hold 0 or G): the fast derivation of the piece offsets stands"""
        p = self.p
        p.emit("s_lshr_b32", S_D, A_FLAGS, 8)
        p.emit("s_and_b32", S_D, S_D, 31)
        p.emit("s_cmp_eq_u32", S_D, 0)
        p.emit("s_cbranch_scc1", Label(lab))
        p.emit("s_cmp_ge_u32", S_D, self.g.ROWB // 16)
        p.emit("s_cbranch_scc1", Label(lab))

    def lm_fail_check(self, r, x, t, t2):
        """appends to r: a lane's share x of a row sum at or above LM_MAX, or NaN -> the wave raises its flag word (the shell redoes the item in safe mode)"""
        ok = self.p.fresh("lm_ok")
        r.append(mk("v_cmp_ngt_f32", VCC, LM_MAX, x))
        r.append(mk("s_cbranch_vccz", Label(ok)))
        r.append(mk("v_mov_b32", t, S_WAVE))
        r.append(mk("v_mov_b32", t2, 1))
        r.append(mk("v_lshlrev_b32", t, 2, t))
        r.append(mk("s_nop", 0))
        r.append(mk("v_add_u32", t, self.g.FAIL_OFF, t))
        r.append(mk("s_nop", 0))
        r.append(mk("ds_write_b32", t, t2))
        r.append(mk("s_waitcnt", lgkmcnt=0))
        r.append(Ins("label", (Label(ok),)))

    def lm_repair(self, lab, par):
        """Out of line, end of fast body t (parity par; round 6): the row-sum chain of tile t + 1 came back inf / NaN / beyond LM_MAX — some P of that tile left
        the 16-bit range (fp16: a score 16 octaves above its row's reference, which the fast bodies never move).  Nothing of the tile has touched O or the
        sums yet (the links ride AHEAD of P.V(t + 1): qk_phase; the tuple the chain started from is intact: sum_links), so the tile is formed again IN
        PLACE — the reference kernel's own recurrence (kernel_fp16.cu:434-490) for this one tile — and no work is thrown away:
          K(t + 1) has left its LDS slot (K(t + 3) is landing there), so its fragments come straight from memory (L2: the tile was staged two bodies ago)
          in the MFMA A layout, two k-steps at a time through the K fragment registers (idle between the bodies); Q.K^T(t + 1) again into the tile's
          banks; the max-first streams of the head / tail bodies (stream_max -> the reference moves, stream_exp) on them, emitted in line; the scores of
          tile t + 2 — all four k-steps done against the OLD reference — get the shift; O and the sums are rescaled at once (rare_rescale, in line); the
          eight links again.  From here on the wave leaves the fast loop (S_NFAST = 0: the max-first bodies carry the rest of its sweep, ~12 % slower)
          and asks the shell (flag word value 2: sticky, no redo) to start the workgroup's later items in safe mode: data that outgrew the fast
          bodies once tends to do it again.
        Cost: ~2 body times, once per wave and item at most."""
        g = self.g
        ps = par ^ 1
        r = [Ins("label", (Label(lab),))]
        e = r.append
        e(mk("s_nop", 15))
        e(mk("s_nop", 15))
        if par == 1:                        # the chain ran TS -> LSV: TS holds the sums of the tiles before
            for i in range(4):
                e(mk("v_mov_b32", LSV[i], TS[i]))
        # the sticky request first: a fail check further down (rare_rescale; the bodies behind the loop) overwrites it with the redo request, which implies it
        e(mk("v_mov_b32", TMP[4], S_WAVE))
        e(mk("v_mov_b32", TMP[5], 2))
        e(mk("v_lshlrev_b32", TMP[4], 2, TMP[4]))
        e(mk("s_nop", 0))
        e(mk("v_add_u32", TMP[4], g.FAIL_OFF, TMP[4]))
        e(mk("s_nop", 0))
        e(mk("ds_write_b32", TMP[4], TMP[5]))
        e(mk("s_waitcnt", lgkmcnt=0))
        # ---- per-lane byte offsets of the K fragments: row n = lane % 16 of a 16-row kv group, granule 4 ks + g4 (g4 = lane / 16) of the row
        lane, n, g4, t, mark = TMP[7], TMP[6], TMP[5], TMP[4], FS[0][0]
        s_ng, s_tile, s_grp, s_off = S_SUM                                     # (trace builds only use them otherwise; the 16x16 generator has none)
        e(mk("v_mbcnt_lo_u32_b32", lane, -1, 0))
        e(mk("s_lshr_b32", s_grp, A_KTILE, 6))                                 # K row pitch in bytes
        e(mk("v_mbcnt_hi_u32_b32", lane, -1, lane))
        e(mk("s_lshr_b32", s_ng, A_FLAGS, 8))
        e(mk("s_and_b32", s_ng, s_ng, 31))
        e(mk("s_cmp_eq_u32", s_ng, 0))
        e(mk("s_cselect_b32", s_ng, 64, s_ng))                                 # granules a row really has (trimmed head dims: trim_offsets; 0 = all)
        e(mk("v_and_b32", n, 15, lane))
        e(mk("v_lshrrev_b32", g4, 4, lane))
        e(mk("v_mov_b32", mark, 0x80000000))
        e(mk("v_mul_lo_u32", n, n, s_grp))
        e(mk("v_lshlrev_b32", t, 4, g4))
        e(mk("s_nop", 0))
        e(mk("v_add_u32", n, n, t))
        e(mk("s_sub_u32", s_tile, S_KOFF, A_KTILE))                            # S_KOFF: tile t + 3 (the LDS-DMA runs three tiles ahead)
        e(mk("s_sub_u32", s_tile, s_tile, A_KTILE))
        e(mk("s_lshl_b32", s_grp, s_grp, 4))                                   # 16 rows
        for ks in range(self.NKS16):
            e(mk("v_add_u32", t, 4 * ks, g4))
            e(mk("v_add_u32", TMP[ks], 64 * ks, n))
            e(mk("v_cmp_gt_u32", VCC, s_ng, t))                                # the row has this granule; otherwise an offset beyond every descriptor: zeros
            e(mk("s_nop", 0))
            e(mk("v_cndmask_b32", TMP[ks], mark, TMP[ks], VCC))
        qk = self.qk_mfmas(ps)                                                 # ks major, 16 per k-step
        for k0 in range(0, self.NKS16, 2):
            e(mk("s_mov_b32", s_off, s_tile))
            for kg in range(4):
                for ks in (k0, k0 + 1):
                    e(mk("buffer_load_dwordx4", self.kf16(kg, ks), TMP[ks], A_KRS, s_off, offen=True))
                if kg < 3:
                    e(mk("s_add_u32", s_off, s_off, s_grp))
            e(mk("s_waitcnt", vmcnt=0))
            r.extend(qk[16 * k0:16 * k0 + 32])
            e(mk("s_nop", 15))
        e(mk("s_nop", 15))
        # ---- the tile's softmax, max-first: the reference moves (rare_m: MC, the C tuples, the bank's scores, FS, S_FLAG) ...
        if self.fold:
            for qb in range(2):
                for h in range(2):
                    e(mk("v_mov_b32", LIMQ[qb][h], MC[h][qb]))                 # (the limits of the masked bodies are set where they are used)

        def inline(items):
            for it in items:
                for x in (it if isinstance(it, list) else [it]):
                    e(x)
                    if x.op != "label" and not x.op.startswith("s_"):
                        e(mk("s_nop", 1))
        for qb in range(2):
            inline(self.stream_max(qb, ps, False))
        if self.fold:
            # ... the scores of tile t + 2 were formed against the old references
            for qb in range(2):
                b2 = SB(qb, par)
                for h in range(2):
                    e(mk("v_sub_f32", LIMQ[qb][h], MC[h][qb], LIMQ[qb][h]))
                    e(mk("s_nop", 1))
                    for i in range(16):
                        e(mk("v_sub_f32", b2[16 * h + i], b2[16 * h + i], LIMQ[qb][h]))
        # ... everything accumulated at the old references (O, the sums through tile t) is scaled now
        rr = self.p.fresh("rare_r")
        r.extend(self.rare_rescale(rr))
        e(Ins("label", (Label(rr + "_ret"),)))
        for qb in range(2):
            inline(self.stream_exp(qb, ps))
        e(mk("s_nop", 7))
        for x in self.sum_links(ps, "acc"):
            e(x)
            e(mk("s_nop", 7))
        e(mk("s_nop", 15))
        if par == 0:                        # (F0's exit copies TS to LSV)
            for i in range(4):
                e(mk("v_mov_b32", TS[i], LSV[i]))
        e(mk("s_mov_b32", S_NFAST, 0))
        e(mk("s_branch", Label(lab + "_ret")))
        return r

    def _row_reduce_max(self, r, x, t):
        """x = max of x over the four lanes of a row (l % 16 equal): exchange with the lane 16 away, then 32 away (t: scratch)"""
        for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
            r.append(mk("v_mov_b32", t, x))
            r.append(mk("s_nop", 1))
            r.append(mk(op, x, t))
            r.append(mk("v_max_f32", x, x, t))
            r.append(mk("s_nop", 0))

    def rare_sum(self, lab, qb, par, fix):
        """Out of line, sum-check bodies (base.Gen.rare_sum has the scheme): the two rows of this lane get their P maximum — over the row's four lanes —,
        the references move by d = max(0, ceil(log2 max P)) octaves per row and everything at the old references (this tile's P, the running sums, the O
        accumulators of the q block: all of PV(t) was issued gaps ago) is multiplied by 2^-d.  Growth of 120 octaves and more: flag, redo in safe mode."""
        b = SB(qb, par)
        mx = [TMP[4 * qb], TMP[4 * qb + 1]]            # (the tile's sum chains, already added to the running sums)
        t, t2 = TMP[4 * qb + 2], TMP[4 * qb + 3]
        f = [EPX[0], EPX[1]]                           # the rows' factors 2^-d
        scr = [EPX[2], EPX[3]]
        r = [Ins("label", (Label(lab),))]
        r.append(mk("v_mov_b32", t2, t))               # the tile sum the check read: it carries an inf / NaN the maxima may drop
        for h in range(2):
            r.append(mk("v_max3_f32", mx[h], b[16 * h], b[16 * h + 1], b[16 * h + 2]))
        for i in range(6):
            for h in range(2):
                r.append(mk("v_max3_f32", mx[h], mx[h], b[16 * h + 3 + 2 * i], b[16 * h + 4 + 2 * i]))
        for h in range(2):
            r.append(mk("v_max_f32", mx[h], mx[h], b[16 * h + 15]))
        r.append(mk("s_nop", 0))
        for h in range(2):
            self._row_reduce_max(r, mx[h], t)
        r.append(mk("v_max_f32", t, mx[0], mx[1]))
        r.append(mk("s_nop", 0))
        r.append(mk("v_add_f32", t2, t2, t))
        fail = lab + "_fail"
        r.append(mk("s_nop", 0))
        r.append(mk("v_cmp_ngt_f32", VCC, float(2.0 ** 120), t2))
        r.append(mk("s_cbranch_vccnz", Label(fail)))
        for h in range(2):
            r.append(mk("v_log_f32", t, mx[h]))
            r.append(mk("s_nop", 0))
            r.append(mk("v_max_f32", t, 0, t))                          # rows that stayed below their reference keep it (d = 0)
            r.append(mk("s_nop", 0))
            r.append(mk("v_ceil_f32", t, t))                            # whole octaves: every factor below is an exact power of two
            r.append(mk("s_nop", 0))
            r.append(mk("v_exp_f32", f[h], Neg(t)))
            r.append(mk("v_add_f32", MC[h][qb], MC[h][qb], t))
            r.append(mk("s_nop", 0))
            for e in range(16):
                r.append(mk("v_mul_f32", b[16 * h + e], b[16 * h + e], f[h]))
            r.append(mk("v_mul_f32", LS[h][qb], LS[h][qb], f[h]))
            if self.ct:
                # the C tuple of the coming Q.K^T products; `fix`: the first k-step of the NEXT tile was issued with the old tuple already — those scores
                # get the shift at the start of the next body (S_FIX, rare_fix)
                r.append(mk("v_sub_f32", scr[0], 0, MC[h][qb]))
                r.append(mk("s_nop", 0))
                for i in range(4):
                    r.append(mk("v_mov_b32", CT16[2 * qb + h][i], scr[0]))
                if fix:
                    r.append(mk("v_mov_b32", DSH16[h][qb], t))
        if self.ct and fix:
            r.append(mk("s_or_b32", S_FIX, S_FIX, 1 << qb))
        r.append(mk("s_nop", 15))
        r.append(mk("s_nop", 15))
        for h in range(2):
            for dg in range(self.NDG):
                acc = self.oacc16(dg, 2 * qb + h)
                for j in range(0, 4, 2):
                    for x in range(2):
                        r.append(mk("v_accvgpr_read_b32", scr[x], acc[j + x]))
                    r.append(mk("s_nop", 1))
                    for x in range(2):
                        r.append(mk("v_mul_f32", scr[x], scr[x], f[h]))
                    r.append(mk("s_nop", 1))
                    for x in range(2):
                        r.append(mk("v_accvgpr_write_b32", acc[j + x], scr[x]))
        r.append(mk("s_nop", 7))
        r.append(mk("s_branch", Label(lab + "_ret")))
        r.append(Ins("label", (Label(fail),)))
        r.append(mk("v_mov_b32", t, S_WAVE))
        r.append(mk("v_mov_b32", t2, 1))
        r.append(mk("v_lshlrev_b32", t, 2, t))
        r.append(mk("s_nop", 0))
        r.append(mk("v_add_u32", t, self.g.FAIL_OFF, t))
        r.append(mk("s_nop", 0))
        r.append(mk("ds_write_b32", t, t2))
        r.append(mk("s_waitcnt", lgkmcnt=0))
        r.append(mk("s_branch", Label(lab + "_ret")))
        return r

    def stream_max(self, qb, par, masked, first=False):
        """mask (tail bodies) -> maxima of this lane's 16 scores of each of its two rows -> reduce over the row's four lanes -> rescale decision."""
        b = SB(qb, par)
        out = []
        mx = [TMP[4 * qb], TMP[4 * qb + 1]]
        t, t2 = TMP[4 * qb + 2], TMP[4 * qb + 3]
        if masked:
            # element (kg, i) of row h is kv_local = 16 kg + i (4 g is folded into the limits): kept iff <= the row's limit
            # lim(row h of q block qb) = min(A_LIM0 + 16 (2 qb + h), A_LIM1)   (A_LIM0: causal diagonal of q group 0, A_LIM1: ragged Nkv; fa2_fwd_m16.hip.h)
            lim = LIMQ[qb]
            for h in range(2):
                out.append(mk("v_add_u32", lim[h], 16 * (2 * qb + h), A_LIM0, tag="valu"))
            for h in range(2):
                out.append(mk("v_min_i32", lim[h], lim[h], A_LIM1, tag="valu"))
            out.append(mk("v_mov_b32", t2, NEG_INF, tag="valu"))
            for h in range(2):
                for kg in range(4):
                    for i in range(4):
                        out.append([mk("v_cmp_le_i32", VCC, 16 * kg + i, lim[h], tag="valu"),
                                    mk("v_cndmask_b32", b[16 * h + 4 * kg + i], t2, b[16 * h + 4 * kg + i], VCC, tag="valu")])
        for h in range(2):
            out.append(mk("v_max3_f32", mx[h], b[16 * h], b[16 * h + 1], b[16 * h + 2], tag="valu"))
        for i in range(6):
            for h in range(2):
                out.append(mk("v_max3_f32", mx[h], mx[h], b[16 * h + 3 + 2 * i], b[16 * h + 4 + 2 * i], tag="valu"))
        for h in range(2):
            out.append(mk("v_max_f32", mx[h], mx[h], b[16 * h + 15], tag="valu"))
        for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
            out.append(mk("v_mov_b32", t, mx[0], tag="valu"))
            out.append(mk("v_mov_b32", t2, mx[1], tag="valu"))
            out.append([mk("s_nop", 1, tag="salu"), mk(op, mx[0], t, tag="valu"), mk(op, mx[1], t2, tag="valu")])
            out.append(mk("v_max_f32", mx[0], mx[0], t, tag="valu"))
            out.append(mk("v_max_f32", mx[1], mx[1], t2, tag="valu"))
        lab = self.p.fresh("rare_m")
        if self.fold:
            # S already is (score - reference) in log2 units: a row's maximum IS its growth.  Tile 0 adopts its own maxima whatever their sign (the
            # reference starts at 0, not at -inf: it travels through the MFMA as a 16-bit-exact f32 term)
            if first:
                out.append([mk("s_branch", Label(lab), tag="branch"), Ins("label", (Label(lab + "_ret"),))])
            else:
                out.append(mk("v_max_f32", t, mx[0], mx[1], tag="valu"))
                out.append([mk("s_nop", 0, tag="salu"), mk("v_cmp_lt_f32", VCC, THR, t, tag="valu"),
                            mk("s_cbranch_vccnz", Label(lab), tag="branch"), Ins("label", (Label(lab + "_ret"),))])
            r = [Ins("label", (Label(lab),))]
            for h in range(2):
                if not first:
                    r.append(mk("v_max_f32", mx[h], 0, mx[h]))              # a row that did not grow keeps its reference
                    r.append(mk("s_nop", 0))
                r.append(mk("v_add_f32", t, MC[h][qb], mx[h]))               # new reference
                r.append(mk("s_nop", 0))
                r.append(mk("v_sub_f32", t2, t, MC[h][qb]))                  # the shift as applied
                r.append(mk("v_mov_b32", MC[h][qb], t))
                r.append(mk("v_sub_f32", t, 0, t))                           # -reference for the C tuple
                r.append(mk("s_nop", 0))
                for e in range(16):
                    r.append(mk("v_sub_f32", b[16 * h + e], b[16 * h + e], t2))
                for i in range(4):
                    r.append(mk("v_mov_b32", CT16[2 * qb + h][i], t))
                r.append(mk("v_exp_f32", t2, Neg(t2)))                        # factor for everything accumulated at the old reference
                r.append(mk("s_nop", 1))
                if not self.lm:     # (lm: the sum MFMAs of the tile before — old reference — are not all issued yet: LS is rescaled with O, rare_rescale)
                    r.append(mk("v_mul_f32", LS[h][qb], LS[h][qb], t2))
                r.append(mk("v_mov_b32", FS[h][qb], t2))
            if not first:
                r.append(mk("s_or_b32", S_FLAG, S_FLAG, 1 << qb))
            r.append(mk("s_nop", 1))
            r.append(mk("s_branch", Label(lab + "_ret")))
            self.rare.append(r)
            return out
        out.append(mk("v_fma_f32", t, mx[0], A_C, Neg(MCA[qb]), tag="valu"))
        out.append(mk("v_fma_f32", t2, mx[1], A_C, Neg(MCB[qb]), tag="valu"))
        out.append(mk("v_max_f32", t, t, t2, tag="valu"))
        out.append([mk("s_nop", 0, tag="salu"), mk("v_cmp_lt_f32", VCC, THR, t, tag="valu"), mk("s_cbranch_vccnz", Label(lab), tag="branch"),
                    Ins("label", (Label(lab + "_ret"),))])
        # out of line: move the references (scaled units m*c), scale the row sums now, leave the O rescale pending
        r = [Ins("label", (Label(lab),))]
        for h in range(2):
            r.append(mk("v_mul_f32", t, A_C, mx[h]))                     # tile max * c
            r.append(mk("s_nop", 0))
            r.append(mk("v_max_f32", t, t, MC[h][qb]))                   # new reference
            r.append(mk("s_nop", 0))
            r.append(mk("v_sub_f32", t2, MC[h][qb], t))                  # (m_old - m_new) * c  (<= 0; -inf on the first tile)
            r.append(mk("s_nop", 0))
            r.append(mk("v_exp_f32", t2, t2))
            r.append(mk("v_mov_b32", MC[h][qb], t))
            r.append(mk("s_nop", 0))
            if not self.lm:         # (lm: with O, at the phase boundary — rare_rescale)
                r.append(mk("v_mul_f32", LS[h][qb], LS[h][qb], t2))
            r.append(mk("v_mov_b32", FS[h][qb], t2))                     # (1.0 for a row whose reference stayed; one softmax per body: never two pending)
        if not first:               # the q block's first tile: O is still all zeros, nothing to rescale later
            r.append(mk("s_or_b32", S_FLAG, S_FLAG, 1 << qb))
        r.append(mk("s_branch", Label(lab + "_ret")))
        self.rare.append(r)
        return out

    def rare_fix(self, lab, par):
        """Out of line, start of a body (folded scale): rows whose reference moved after the first Q.K^T k-step of THIS body's softmax tile had been
        issued with the old C tuple (rare_sum, `fix`): shift those scores by the pending amount."""
        r = [Ins("label", (Label(lab),))]
        r.append(mk("s_nop", 15))
        r.append(mk("s_nop", 15))
        for qb in range(2):
            skip = self.p.fresh("fix_skip")
            r.append(mk("s_bitcmp1_b32", S_FIX, qb))
            r.append(mk("s_cbranch_scc0", Label(skip)))
            b = SB(qb, par)
            for h in range(2):
                for e in range(16):
                    r.append(mk("v_sub_f32", b[16 * h + e], b[16 * h + e], DSH16[h][qb]))
            r.append(Ins("label", (Label(skip),)))
        r.append(mk("s_mov_b32", S_FIX, 0))
        r.append(mk("s_nop", 1))
        r.append(mk("s_branch", Label(lab + "_ret")))
        return r

    def ct_reader_gaps(self, qb):
        return self.npv + 8 * qb, self.npv + 8 * qb + 7

    def place_pool_kreads(self, load, slots, par):
        """K fragment pool of two k-step slots: k-steps 0, 1 are read during the PV phase, k-step ks >= 2 goes into the slot of ks - 2 as soon as the
        sixteen MFMAs of that k-step are issued (a whole k-step ahead of its own first use)"""
        kr = self.stream_kread(par)                       # order: ks major, kg minor
        self.place(load, slots, kr[:8], self.cfg["kread_ct"][0], self.cfg["kread_ct"][1], 3)
        for ks in range(2, self.NKS16):
            g0 = self.npv + 16 * (ks - 2) + 15
            for kg in range(4):
                it = kr[4 * ks + kg]
                load[g0] += base._weight(it)
                slots[g0].append((g0 + 0.5 + 0.1 * kg, 3, it))

    def rare_rescale(self, lab):
        r = [Ins("label", (Label(lab),))]
        r.append(mk("s_nop", 15))
        r.append(mk("s_nop", 15))
        for qb in range(2):
            skip = self.p.fresh("rr_skip")
            r.append(mk("s_bitcmp1_b32", S_FLAG, qb))
            r.append(mk("s_cbranch_scc0", Label(skip)))
            scr = TMP[:4] if self.lm else EPX      # (lm: v[248:251] hold a constant tuple; the row-max chains in TMP end ahead of the phase boundary)
            for h in range(2):
                if self.lm:
                    # the row sums, with the sum MFMAs of every tile at the old reference landed (s_nop above): checked at full size, then scaled
                    self.lm_fail_check(r, LS[h][qb], TMP[4], TMP[5])
                    r.append(mk("v_mul_f32", LS[h][qb], LS[h][qb], FS[h][qb]))
                for dg in range(self.NDG):
                    acc = self.oacc16(dg, 2 * qb + h)
                    for j in range(4):
                        r.append(mk("v_accvgpr_read_b32", scr[j], acc[j]))
                    r.append(mk("s_nop", 1))
                    for j in range(4):
                        r.append(mk("v_mul_f32", scr[j], scr[j], FS[h][qb]))
                    r.append(mk("s_nop", 1))
                    for j in range(4):
                        r.append(mk("v_accvgpr_write_b32", acc[j], scr[j]))
            r.append(Ins("label", (Label(skip),)))
        r.append(mk("s_mov_b32", S_FLAG, 0))
        r.append(mk("s_nop", 7))
        r.append(mk("s_branch", Label(lab + "_ret")))
        return r

    def stream_kread(self, par):
        g = self.g
        return [mk("ds_read_b128", self.kf16(kg, ks), KR[ks], tag="lds", offset=g.K_SLOT + par * g.SLOT_B + kg * 16 * g.ROWB)
                for ks in range(self.NKS16) for kg in range(4)]

    def stream_vread(self, par):
        """V^T fragments: two transposed reads per (d group, k-step).  A read's lanes 0..31 are the 16-lane groups g = 0, 1: rows 4 g + (n >> 2) — rows r and
        r + 4 of the tile, which the V image (64-byte chunk ^ (row & 3)) keeps in the same banks: a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT
        8.8e6 per c2 launch, profiles/r17_c2_summary.txt).  "ct" kernels (they have the registers) keep a V image of their own in which the 32-byte half
        of a chunk is flipped for rows with (row >> 2) & 1 (fa2_fwd_d128.hip.h: vd0; VD below) and read the odd d groups through a second address set."""
        out = []
        g = self.g
        for kvs in range(2):
            for dg in range(self.NDG):
                off = g.V_BASE + par * g.SLOT_B + 32 * kvs * g.ROWB
                if self.vflip:
                    adr = (self.vro if dg & 1 else VR)[dg >> 1]
                else:
                    adr, off = VR[dg >> 1], off + 32 * (dg & 1)
                out.append(mk("ds_read_b64_tr_b16", self.vf16(dg, kvs).sub(0, 2), adr, tag="lds", offset=off))
                out.append(mk("ds_read_b64_tr_b16", self.vf16(dg, kvs).sub(2, 2), adr, tag="lds", offset=off + 16 * g.ROWB))
        return out

    def seam_q_reads(self):
        g = self.g
        r = []
        two, waited = self.p.fresh("qrd_two"), self.p.fresh("qrd_waited")
        r.append(mk("s_bitcmp1_b32", A_FLAGS, 2))
        r.append(mk("s_cbranch_scc1", Label(two)))
        r.append(mk("s_waitcnt", vmcnt=g.NP))
        r.append(mk("s_branch", Label(waited)))
        r.append(Ins("label", (Label(two),)))
        r.append(mk("s_waitcnt", vmcnt=2 * g.NP))
        r.append(Ins("label", (Label(waited),)))
        for i in range(self.NKS16):
            r.append(mk("v_add_u32", QD[i], S_QSB, KR[i]))
        r.append(mk("s_nop", 0))
        for qg in range(4):
            for ks in range(self.NKS16):
                r.append(mk("ds_read_b128", self.qf16(qg, ks), QD[ks], offset=qg * 16 * g.ROWB))
        return r

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        g = self.g
        bf16_ = self.bf16
        tr = int(self.cfg["trace"][0])
        assert not tr, "the trace builds belong to the 32x32 generator"
        # ---- entry: addresses, the wave's Q through its LDS image, the first tiles
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        nks, ndg = self.NKS16, self.NDG
        nqr = 16 * nks                                                 # registers of the wave's Q fragments
        for ks in range(nks):
            p.emit("v_xor_b32", KR[ks], ks << 6, A_KR0)
        for j in range(ndg // 2):
            p.emit("v_xor_b32", VR[j], j << 6, A_VR0)
        if self.vflip:
            for j in range(ndg // 2):
                p.emit("v_xor_b32", self.vro[j], (j << 6) | 32, A_VR0)      # the other 32-byte half
        if self.lm:
            # the constant A tuples of the row-sum MFMAs: 0.25 (two packed 16-bit values) on the lanes of rows m = lane % 16 with m % 4 == qg
            p.emit("v_lshrrev_b32", TMP[0], g.ROWB.bit_length() - 1, A_KR0)      # A_KR0 = n * ROWB + (swizzled granule << 4), n = lane % 16
            p.emit("v_mov_b32", TMP[1], 0x3e803e80 if bf16_ else 0x34003400)
            p.emit("v_and_b32", TMP[0], 3, TMP[0])
            for qg in range(4):
                p.emit("v_cmp_eq_u32", VCC, qg, TMP[0])
                if self.ct:
                    p.emit("v_cndmask_b32", ONES16[qg][0], 0, TMP[1], VCC)
                    for i in range(1, 4):
                        p.emit("v_mov_b32", ONES16[qg][i], ONES16[qg][0])
                else:
                    p.emit("v_cndmask_b32", TMP[2], 0, TMP[1], VCC)
                    p.emit("s_nop", 0)
                    for i in range(4):
                        p.emit("v_accvgpr_write_b32", self.ones16[qg][i], TMP[2])
        p.emit("s_lshr_b32", S_WAVE, A_LDSW, (g.SLOT_B // 4).bit_length() - 1)
        p.emit("s_mul_i32", S_QSB, S_WAVE, 64 * g.EPI_ROWB)
        p.emit("s_add_u32", S_QSB, S_QSB, g.EPI_BASE)
        p.emit("v_mov_b32", QD[0], A_QD0)
        p.emit("s_lshr_b32", S_TMP2, A_QT16, (16 // g.RPP).bit_length() - 1)
        p.emit("s_sub_u32", S_TMP2, S_TMP2, 1024)
        p.emit("s_mov_b32", S_TMP, 0)
        for i in range(1, g.NP):
            p.emit("s_add_u32", S_TMP, S_TMP, S_TMP2)
            p.emit("v_xor_b32", QD[i], i << 6, A_QD0)
            p.emit("s_nop", 0)
            p.emit("v_add_u32", QD[i], S_TMP, QD[i])
        self.trim_test("q_full")
        self.trim_offsets("k", [QD[i] for i in range(g.NP)], A_QD0, S_TMP2)       # (the Q image of a 16-row group is laid out like a K tile's quarter)
        p.label("q_full")
        p.emit("s_and_b32", S_PF, A_FLAGS, 1)
        p.emit("s_cmp_eq_u32", S_PF, 1)
        p.emit("s_cbranch_scc1", Label("have_q"))
        p.emit("s_mov_b32", S_TMP, A_QW)
        for j in range(4):
            p.emit("s_add_u32", M0, S_QSB, j * g.NP * 1024)
            if j:
                p.emit("s_add_u32", S_TMP, S_TMP, A_QT16)
            else:
                p.emit("s_nop", 0)
            for ins in self.q_group_loads(A_QRS, S_TMP):
                p.ins.append(ins)
        if self.fold:
            # folded scale: Q passes through the (still unused) S banks, is multiplied by c in f32 and rounded back ONCE — the reference oracle's contract
            # `scale * q_frags` (pure_torch_ver.py:61) — then parked in the accumulator file.  A prefetched item's raw fragments come back from there.
            p.emit("s_branch", Label("q_issued"))
            p.label("have_q")
            for i in range(nqr):
                p.emit("v_accvgpr_read_b32", V(VBASE + i), self.qf16(i // (4 * nks), (i % (4 * nks)) // 4)[i % 4])
            p.label("q_issued")
        else:
            p.label("have_q")
        p.emit("v_mov_b32", KD[0], A_KD0)
        p.emit("v_mov_b32", VD[0], A_VD0)
        p.emit("s_mov_b32", S_TMP, 0)
        p.emit("s_mov_b32", S_TMP2, 0)
        for i in range(1, g.NP):
            p.emit("s_add_u32", S_TMP, S_TMP, A_KROW4)
            p.emit("s_add_u32", S_TMP2, S_TMP2, A_VROW4)
            p.emit("v_xor_b32", KD[i], i << 6, A_KD0)
            if self.vflip and ((g.RPP * i) & 4):       # the rows of such a piece have (row >> 2) & 1 set where piece 0's have it clear: their 32-byte halves are
                                                     # flipped in the "ct" V image (head dim 128: the odd pieces; 64: a piece is eight rows, none)
                p.emit("v_xor_b32", VD[i], 32, A_VD0)
                p.emit("s_nop", 0)
                p.emit("v_add_u32", VD[i], S_TMP2, VD[i])
            else:
                p.emit("v_add_u32", VD[i], S_TMP2, A_VD0)
            p.emit("s_nop", 0)
            p.emit("v_add_u32", KD[i], S_TMP, KD[i])
        self.trim_test("kv_full")
        self.trim_offsets("k", [KD[i] for i in range(g.NP)], A_KD0, A_KROW4)
        self.trim_offsets("v", [VD[i] for i in range(g.NP)], A_VD0, A_VROW4)
        p.label("kv_full")
        p.emit("s_mov_b32", S_T, -2)
        p.emit("s_mov_b32", S_FLAG, 0)
        p.emit("s_mov_b32", S_FIX, 0)
        for r in S_SUM:
            p.emit("s_mov_b32", r, 0)
        p.emit("s_mov_b32", S_KOFF, 0)
        p.emit("s_mov_b32", S_NOVM, 0)
        p.emit("s_mov_b32", S_VOFF, 0)
        p.emit("s_cmp_eq_u32", S_PF, 1)
        p.emit("s_cbranch_scc1", Label("staged"))
        p.emit("s_add_u32", M0, A_LDSW, g.K_SLOT)
        p.emit("s_nop", 0)
        for i in range(g.NP):
            p.emit("buffer_load_dwordx4", KD[i], A_KRS, S_KOFF, offen=True, offset=1024 * i, lds=True)
        p.emit("s_add_u32", M0, A_LDSW, g.V_BASE)
        p.emit("s_nop", 0)
        for i in range(g.NP):
            p.emit("buffer_load_dwordx4", VD[i], A_VRS, S_VOFF, offen=True, offset=1024 * i, lds=True)
        p.emit("s_cmp_lt_i32", A_NTWG, 2)
        p.emit("s_cbranch_scc1", Label("no_k1"))
        p.emit("s_add_u32", M0, A_LDSW, g.K_SLOT + g.SLOT_B)
        p.emit("s_nop", 0)
        for i in range(g.NP):
            p.emit("buffer_load_dwordx4", KD[i], A_KRS, A_KTILE, offen=True, offset=1024 * i, lds=True)
        p.label("no_k1")
        # Q (issued first) from the image into the fragment registers; K(0), V(0), K(1) keep flying
        p.emit("s_cmp_lt_i32", A_NTWG, 2)
        p.emit("s_cbranch_scc1", Label("qwait8"))
        p.emit("s_waitcnt", vmcnt=3 * g.NP)
        p.emit("s_branch", Label("qwaited"))
        p.label("qwait8")
        p.emit("s_waitcnt", vmcnt=2 * g.NP)
        p.label("qwaited")
        for ks in range(nks):
            p.emit("v_add_u32", TMP[ks], S_QSB, KR[ks])
        p.emit("s_nop", 0)
        for qg in range(4):
            for ks in range(nks):
                p.emit("ds_read_b128", V(VBASE + 4 * nks * qg + 4 * ks, 4) if self.fold else self.qf16(qg, ks), TMP[ks], offset=qg * 16 * g.ROWB)
        p.label("staged")
        p.emit("s_bitcmp1_b32", A_FLAGS, 1)
        p.emit("s_cselect_b32", S_QH, 0, 8)
        p.emit("s_mov_b32", S_QM0, S_QSB)
        p.emit("s_mov_b32", S_QSOFF, A_NQW)
        p.emit("s_mov_b32", S_KOFF, A_KTILE)
        for qb in range(2):
            for h in range(2):
                p.emit("v_mov_b32", MC[h][qb], 0.0 if self.fold else NEG_INF)
                p.emit("v_mov_b32", LS[h][qb], 0)
                p.emit("v_mov_b32", FS[h][qb], 1.0)
        for i in range(16 * ndg):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        p.emit("s_waitcnt", lgkmcnt=0)
        if self.fold:
            for qg in range(4):
                for i in range(4):
                    p.emit("v_mov_b32", CT16[qg][i], 0)                 # C tuples: the references start at 0
            for i in range(nqr):
                for ins in self.q_prescale_reg(V(VBASE + i), self.qf16(i // (4 * nks), (i % (4 * nks)) // 4)[i % 4], TMP[2 * (i & 1)], TMP[2 * (i & 1) + 1]):
                    p.ins.append(ins)
        p.emit("s_cmp_lt_i32", A_NTWG, 2)
        p.emit("s_cbranch_scc1", Label("wait4"))
        p.emit("s_waitcnt", vmcnt=2 * g.NP)
        p.emit("s_branch", Label("waited"))
        p.label("wait4")
        p.emit("s_waitcnt", vmcnt=g.NP)
        p.label("waited")
        p.emit("s_barrier")

        # ---- head bodies, fast loop, tail dispatch: the base generator's structure
        self.body(0, pv=False, s1=False, s2=True, name="H1", dma=False)
        p.emit("s_cmp_eq_u32", A_NTW, 1)
        p.emit("s_cbranch_scc1", Label("h2b"))
        self.body(1, pv=False, s1=True, s2=True, name="H2", first=True)
        p.emit("s_branch", Label("main"))
        p.label("h2b")
        self.body(1, pv=False, s1=True, s2=False, masked=True, name="H2b", first=True)
        p.label("main")
        p.emit("s_bitcmp1_b32", A_FLAGS, 4)
        p.emit("s_cbranch_scc1", Label("dispatch"))
        p.emit("s_sub_u32", S_NFAST, A_NTW, 3)
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        p.emit("s_cbranch_scc0", Label("dispatch"))
        p.emit("s_nop", 0)
        p.label("fast0")
        self.body(0, guarded=False, name="F0")
        p.emit("s_sub_u32", S_NFAST, S_NFAST, 1)
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        if self.lm:
            # (F0's row-sum chain ended in TS: the max-first bodies behind the loop keep the sums in LSV)
            p.emit("s_cbranch_scc0", Label("f0_exit"))
            self.rare.append([Ins("label", (Label("f0_exit"),))] + [mk("v_mov_b32", LSV[i], TS[i]) for i in range(4)] +
                             [mk("s_nop", 0), mk("s_branch", Label("dispatch"))])
        else:
            p.emit("s_cbranch_scc0", Label("dispatch"))
        self.body(1, guarded=False, name="F1")
        p.emit("s_sub_u32", S_NFAST, S_NFAST, 1)
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        p.emit("s_cbranch_scc1", Label("fast0"))
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
            p.emit("s_cmp_ge_i32", S_D, 3)
            p.emit("s_cbranch_scc1", Label("ta_" + suffix))
            p.emit("s_cmp_eq_u32", S_D, 2)
            p.emit("s_cbranch_scc1", Label("tb_" + suffix))
            p.emit("s_cmp_eq_u32", S_D, 1)
            p.emit("s_cbranch_scc1", Label("tc_" + suffix))
            self.body(par, pv=False, s1=False, s2=False, name="ST%d" % par)
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

        # ---- epilogue: row sums over the row's four lanes, O / l -> 16 bit -> the wave's LDS image (rows of 272 B), LSE out
        p.label("epilogue")
        p.emit("s_nop", 15)
        lse = [TMP[4], TMP[5], TMP[6], TMP[7]]          # LSE of this lane's row of q group qg
        epx = [FS[qg & 1][qg >> 1] for qg in range(4)] if self.lm else EPX      # (lm: v[248:251] hold a constant tuple; no O rescale is pending here)
        for qb in range(2):
            for h in range(2):
                qg = 2 * qb + h
                lt, t = TMP[0], TMP[1]
                if self.lm:
                    blk = []
                    self.lm_fail_check(blk, LS[h][qb], TMP[2], TMP[3])
                    p.ins.extend(blk)
                p.emit("v_mov_b32", lt, LS[h][qb])
                for op in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
                    p.emit("v_mov_b32", t, lt)
                    p.emit("s_nop", 1)
                    p.emit(op, lt, t)
                    p.emit("v_add_f32", lt, lt, t)
                    p.emit("s_nop", 0)
                p.emit("v_rcp_f32", epx[qg], lt)
                p.emit("v_log_f32", t, lt)
                p.emit("s_nop", 0)
                p.emit("v_add_f32", lse[qg], MC[h][qb], t)
        p.emit("s_nop", 0)
        p.emit("s_bitcmp1_b32", A_FLAGS, 3)
        p.emit("s_cbranch_scc1", Label("part_store"))
        for qg in range(4):
            for dg in range(self.NDG):
                acc = self.oacc16(dg, qg)
                for j in range(4):
                    p.emit("v_accvgpr_read_b32", TMP[j], acc[j])
                p.emit("s_nop", 0)
                for j in range(4):
                    p.emit("v_mul_f32", TMP[j], TMP[j], epx[qg])
                p.emit("s_nop", 0)
                p.emit(self.cvt, TMP[0], TMP[0], TMP[1])
                p.emit(self.cvt, TMP[1], TMP[2], TMP[3])
                p.emit("s_nop", 0)
                # 8 bytes: d = 16 dg + 4 g .. + 3 of row 16 qg + n
                p.emit("ds_write_b64", A_EPI, V(TMP[0].idx, 2), offset=16 * qg * g.EPI_ROWB + 32 * dg)
        p.label("lse_out")
        # the LSE leaves in ONE register: lane l = 16 g + n hands over row l of the wave, i.e. q group g
        p.emit("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
        p.emit("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])
        p.emit("s_nop", 0)
        p.emit("v_lshrrev_b32", TMP[0], 4, TMP[0])
        p.emit("s_nop", 0)
        p.emit("v_mov_b32", TMP[1], lse[0])
        for qg in range(1, 4):
            p.emit("v_cmp_eq_u32", VCC, qg, TMP[0])
            p.emit("v_cndmask_b32", TMP[1], TMP[1], lse[qg], VCC)
        p.emit("s_waitcnt", lgkmcnt=0)
        p.emit("v_mov_b32", A_LSE0, TMP[1])
        p.emit("s_branch", Label("end"))
        # ---- a KV-split part (flag bit 3; fa2_fwd_ws): the normalised f32 tile goes straight to the part's workspace tile — float
        # (((dt*4 + g) * 256 + row) * 8 + 4*hi + e) for d = 32 dt + 8 g + 4 hi + e, the layout of every part in this library (fa2_fwd_kernel.hip.h).
        # This lane holds d = 16 dg + 4 g4 + i of row 64 wave + 16 qg + n: dt = dg >> 1, g = 2 (dg & 1) + (g4 >> 1), hi = g4 & 1, e = i.
        p.label("part_store")
        wso, adr = KD[2], KD[3]
        p.emit("v_mbcnt_lo_u32_b32", TMP[0], -1, 0)
        p.emit("v_mbcnt_hi_u32_b32", TMP[0], -1, TMP[0])
        p.emit("s_lshl_b32", S_TMP, S_WAVE, 6)
        p.emit("v_and_b32", TMP[1], 15, TMP[0])                 # n
        p.emit("v_lshrrev_b32", TMP[2], 4, TMP[0])              # g4
        p.emit("v_add_u32", TMP[1], S_TMP, TMP[1])              # 64 wave + n
        p.emit("v_and_b32", TMP[3], 1, TMP[2])                  # hi
        p.emit("v_lshrrev_b32", TMP[2], 1, TMP[2])              # g4 >> 1
        p.emit("v_lshlrev_b32", TMP[1], 5, TMP[1])              # row * 32 bytes
        p.emit("v_lshlrev_b32", TMP[3], 4, TMP[3])
        p.emit("v_lshlrev_b32", TMP[2], 13, TMP[2])             # (g4 >> 1) * 8192
        p.emit("s_nop", 0)
        p.emit("v_add_u32", wso, TMP[1], TMP[3])
        p.emit("s_nop", 0)
        p.emit("v_add_u32", wso, wso, TMP[2])
        for qg in range(4):
            for dg in range(self.NDG):
                acc = self.oacc16(dg, qg)
                for j in range(4):
                    p.emit("v_accvgpr_read_b32", TMP[j], acc[j])
                p.emit("v_add_u32", adr, 512 * qg + 8192 * (4 * (dg >> 1) + 2 * (dg & 1)), wso)
                for j in range(4):
                    p.emit("v_mul_f32", TMP[j], TMP[j], epx[qg])
                p.emit("s_nop", 0)
                p.emit("global_store_dwordx4", adr, V(TMP[0].idx, 4), A_WSB)
                p.emit("s_nop", 3)             # (a store of more than 64 bits: its data registers must not be rewritten right behind it)
        p.emit("s_waitcnt", vmcnt=0)           # the stores, and whatever the item seam prefetched: the next statement counts loads only
        p.emit("s_branch", Label("lse_out"))
        for r in self.rare:
            p.extend(r)
        p.label("end")
        return p


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    ap.add_argument("--opt", default="", help="schedule tunables / options (fwd_d128_gen.parse_opts)")
    ap.add_argument("--probe", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cfg = base.parse_opts(a.opt)
    if base.is_probe(cfg) and not a.probe:
        sys.exit("fwd_m16_gen.py: %r contains timing-probe options; they need --probe" % a.opt)
    # per head dim and dtype: the f32-scale body with the sum check (calls flagged FA2_FLAG_EXACT_SCALE: the LSE a backward pass will consume adds the
    # f32 P), the f32-scale body with the row sums on the matrix pipe ("_lm"), the folded body (row sums on the matrix pipe; opt=nolm: with the sum check)
    # ... and, at head dim 128, the folded body with the sum check and its in-place repair ("_fold_nolm": option "asm" bit 9 clear — fp16 data whose
    # rows outgrow the reference of their first tiles by 16 octaves and more costs the lm bodies a second sweep per item, tools/growth_cliff.py)
    for hd, bf16, kind in ((hd, bf16, kind) for hd in (128, 64) for bf16 in (False, True) for kind in ("", "_lm", "_fold", "_fold_nolm")):
        if kind == "_fold_nolm" and hd == 64:
            continue
        c = dict(cfg)
        c["opt"] = tuple(o for o in cfg.get("opt", ()) if o not in ("ct", "lm", "nolm")) + (("ct",) if kind.startswith("_fold") else ())
        if kind == "_lm" or (kind == "_fold" and "nolm" not in cfg.get("opt", ())):
            c["opt"] += ("lm",)
        prog = Gen16(bf16, hd=hd, **c).build()
        path = os.path.join(a.out, "fa2_fwd_m16_%s%s%s.inc" % ("d64_" if hd == 64 else "", "bf16" if bf16 else "f16", kind))
        base.write_atomic(path, "// GENERATED by csrc/gen/fwd_m16_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + base.render_inline(prog))
        print(path, len(prog.ins), "instructions")


if __name__ == "__main__":
    main()
