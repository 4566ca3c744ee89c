#!/usr/bin/env python3
"""Generator of the hand-scheduled FlashAttention-2 BACKWARD blocks for D = 128 on gfx950.

Replaces, for head dim exactly 128, the compiler-scheduled passes of fa2_bwd_kernel.hip.h (reference counterpart: bwd_kernel,
kernel_fp16.cu:547-740 — one kernel there, with dQ accumulated by unsynchronised global read-modify-writes; here two passes in
which every output element has one owner, fa2_bwd_kernel.hip.h explains the split).  Like the forward generator this emits the
body of ONE inline-asm statement per kernel; the HIP shells (fa2_bwd_d128.hip.h) compute addresses and store the results.

    GenDQ   fa2_bwd_dq_d128_{f16,bf16}.inc    dQ pass: workgroup = 4 waves = 256 Q rows, ONE wave per SIMD (512 registers);
            wave w owns Q rows [64w, 64w+64) as two 32-row blocks qb; KV is swept in tiles of 32 rows.  Also forms
            delta_i = rowsum(dO_i * O_i) for its rows (kernel_fp16.cu:605-631) and hands it to the shell.
    GenDKV  fa2_bwd_dkv_d128_{f16,bf16}.inc   dK/dV pass: see the class.

Products (all "swapped", as in the forward, so that the softmax side is lane-local; X^T tiles are 32 x 32 MFMA results):
    S^T[kv,q]  = K[kv,:] . Q[q,:]       A = K rows   (ds_read_b128 from the K image),      B = Q fragments  (AGPRs, loaded once)
    dP^T[kv,q] = V[kv,:] . dO[q,:]      A = V rows   (ds_read_b128 from the V image),      B = dO fragments (AGPRs, loaded once)
    dQ^T[d,q] += K^T[d,kv] . dS^T[kv,q] A = K^T      (ds_read_b64_tr_b16 from the K image), B = dS packed in place (16-bit)
with P = 2^(S c - L) (L = the forward's log2 LSE: no running max), dS = P (dP - delta); `scale` is applied once, to the finished dQ.

LDS images are the forward's two formats (fa2_fwd_kernel.hip.h): "row" images (read with ds_read_b128) keep 16-byte granule g of
tile row r at r*256 + ((g ^ (r & 15)) << 4); "tr" images (read with ds_read_b64_tr_b16) keep 64-byte chunk c of row r at
r*256 + ((c ^ (r & 3)) << 6).  K is staged twice for the dQ pass (once in each format, one body apart).

Software pipeline of the dQ pass.  Body B(t), t = -2 .. ntiles-1, is 48 MFMAs:
    MFMA  0..15  dQ(t)             ACC[qb][dt] += K^T(t)[dt][ks'] . dS(t)[qb][ks']
    MFMA 16..47  S(t+2), dP(t+2)   k-step by k-step, the four accumulators S[qb], dP[qb] take turns
  between them: the VALU work of tile t+1 (fma, exp2, sub, mul, pack: 72 per q block), the 16 row-fragment reads of tile t+2
  (a pool of 8 slots: k-steps 4..7 reuse the slots of 0..3 as soon as those MFMAs are issued, counted lgkmcnt waits), the 16
  transpose reads of tile t+1 (consumed by the next body's dQ phase), 6 LDS-DMA pieces (K and V rows of tile t+3, the K copy the
  transpose reads of tile t+2 use) and one s_waitcnt + s_barrier.  The wave's last two tiles run masked bodies (causal diagonal
  and ragged Nkv are the same compare against a per-lane limit).  Head / tail bodies are the same generator with streams off.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
import sched  # noqa: E402
from isa import A, S, V, Arg, Ins, Label, M0, Neg, Program, VCC, mk  # noqa: E402

NEG_INF = float("-inf")


def _flat(items):
    out = []
    for x in items:
        out.extend(x if isinstance(x, list) else [x])
    return out


class BodyEmitter:
    """What the bodies of both backward kernels share: fillers placed into MFMA gaps, counted LDS waits, the end-of-body sync."""

    def emit_body(self, p, mfmas, slots, pre=(), boundary=None, post=()):
        """mfmas: list (None = no MFMA in that gap); slots[g]: [(key, stream id, item)]; boundary: {gap: [instructions emitted in
        front of that gap's MFMA]}.  The LDS-wait pass runs over the whole body."""
        start = len(p.ins)
        p.ins.extend(pre)
        for g in range(len(mfmas)):
            if boundary and g in boundary:
                p.ins.extend(boundary[g])
            if mfmas[g] is not None:
                p.ins.append(mfmas[g])
            for (_, _, item) in sorted(slots[g], key=lambda x: (x[0], x[1])):
                p.ins.extend(item if isinstance(item, list) else [item])
        p.ins.extend(post)
        p.ins[start:] = sched.lds_waits(p.ins[start:])


# =====================================================================================================================
#                                                       dQ pass
# =====================================================================================================================
class DQ:
    """Register map and operand list of the dQ statement (order = the operand list in fa2_bwd_d128.hip.h)."""
    A_D0, A_D1 = Arg(0), Arg(1)                    # "=&v" outputs: delta = rowsum(dO * O) of this lane's row in q block 0 / 1
    A_QO0, A_QO1 = Arg(2), Arg(3)                  # byte offset of this lane's 16 Q bytes (k-step 0) in block 0 / 1 from the head base
    A_GO0, A_GO1 = Arg(4), Arg(5)                  # ... dO
    A_OO0, A_OO1 = Arg(6), Arg(7)                  # ... O
    A_LO0, A_LO1 = Arg(8), Arg(9)                  # byte offset of this lane's LSE value
    A_KD0, A_VD0, A_TD0 = Arg(10), Arg(11), Arg(12)   # per-lane LDS-DMA source byte offset (piece 0, tile 0): K rows, V rows, K "tr" copy
    A_KR0, A_VR0 = Arg(13), Arg(14)                # per-lane LDS read offset: row fragment k-step 0 / transposed fragment d-block 0
    A_LIM0, A_LIM1 = Arg(15), Arg(16)              # last-tile mask: kv index (tile-local, minus 4*hi) must be <= this, per q block
    A_EPI = Arg(17)                                # per-lane LDS byte address of the epilogue image: row l31, half hi
    A_QB, A_GB, A_OB, A_LB = Arg(18, "s", 2), Arg(19, "s", 2), Arg(20, "s", 2), Arg(21, "s", 2)    # head bases of Q, dO, O, LSE
    A_KRS, A_VRS = Arg(22, "s", 4), Arg(23, "s", 4)   # buffer descriptors of this head's K / V matrix
    A_C, A_SCALE = Arg(24, "s"), Arg(25, "s")      # scale * log2(e), scale (f32 bits)
    A_NTW, A_NTWG = Arg(26, "s"), Arg(27, "s")     # 32-row KV tiles of this wave / of the workgroup
    A_KTILE, A_VTILE = Arg(28, "s"), Arg(29, "s")  # bytes between consecutive KV tiles in K / V
    A_KROW4, A_VROW4 = Arg(30, "s"), Arg(31, "s")  # 4 * row bytes - 1024: source stride between the two DMA pieces of a wave
    A_LDSW = Arg(32, "s")                          # wave * 2048: this wave's quarter of a tile image
    N_ARGS = 33
    N_VARGS = 18

    VBASE = 24

    @staticmethod
    def SB(par, qb):                               # S^T / P bank (f32, 16 registers): tile parity par, q block qb
        return V(24 + 32 * par + 16 * qb, 16)

    @staticmethod
    def DB(par, qb):                               # dP^T / dS bank; dS is packed in place: k-step ks' -> registers [8ks', 8ks'+4)
        return V(88 + 32 * par + 16 * qb, 16)

    @staticmethod
    def RP(x, ks):                                 # row-fragment pool (x = 0: K, 1: V): 8 slots of 4
        return V(152 + 4 * ((2 * ks + x) % 8), 4)

    @staticmethod
    def TP(dt, ks):                                # K^T fragments of one tile: d block dt, k-step ks' (16 kv rows)
        return V(184 + 16 * ks + 4 * dt, 4)

    KR = [V(216 + i) for i in range(8)]            # row-fragment read addresses, k-step ks
    VR = [V(224 + i) for i in range(4)]            # transposed read addresses, d block dt
    KD = [V(228), V(229)]                          # LDS-DMA source offsets of this wave's 2 pieces of a K tile (row image)
    VD = [V(230), V(231)]                          # ... V tile (row image)
    TD = [V(232), V(233)]                          # ... K tile ("tr" image)
    NL = [V(234), V(235)]                          # -LSE of this lane's row, per q block
    DD = [V(236), V(237)]                          # delta of this lane's row
    TMP = [V(238 + i) for i in range(8)]

    @staticmethod
    def ACC(qb, dt):
        return A(64 * qb + 16 * dt, 16)

    @staticmethod
    def QF(qb, ks):
        return A(128 + 32 * qb + 4 * ks, 4)

    @staticmethod
    def GF(qb, ks):
        return A(192 + 32 * qb + 4 * ks, 4)

    S_T, S_KOFF, S_VOFF, S_TOFF, S_TMP, S_TMP2 = S(60), S(61), S(62), S(63), S(64), S(65)
    S_NFAST, S_D = S(66), S(67)
    CLOBBER_S = list(range(60, 70))

    # LDS: row ring (K | V images of a tile, 2 slots), transposed-read ring (a second K image, 2 slots), epilogue image
    ROW_RING, ROW_SLOT, V_IN_SLOT = 0, 16384, 8192
    TR_RING, TR_SLOT = 32768, 8192
    EPI_BASE, EPI_ROWB = 49152, 272
    LDS_BYTES = 49152 + 4 * 64 * 272               # 118784


class GenDQ(BodyEmitter):
    DEFAULTS = {"valu": (1.0, 47.0), "rowread": (0.0, 15.0), "trread": (17.0, 47.0), "dma": (4.0, 40.0), "opt": (), "abl": ()}

    def __init__(self, bf16=False, **cfg):
        self.cfg = dict(self.DEFAULTS)
        self.cfg.update(cfg)
        self.opt = set(self.cfg["opt"])
        self.bf16 = bf16
        self.mfma = "v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if bf16 else "v_cvt_pk_f16_f32"
        self.dot2 = "v_dot2_f32_bf16" if bf16 else "v_dot2_f32_f16"
        self.p = Program()
        self.rare = []

    # ------------------------------------------------------------------ MFMA lists
    def dq_mfmas(self, par):
        out = []
        for ks in range(2):
            for qb in range(2):
                for dt in range(4):
                    out.append(mk(self.mfma, DQ.ACC(qb, dt), DQ.TP(dt, ks), DQ.DB(par, qb).sub(8 * ks, 4), DQ.ACC(qb, dt), tag="mfma"))
        return out

    def sdp_mfmas(self, par):
        out = []
        for ks in range(8):
            for x, bank, frag in ((0, DQ.SB, DQ.QF), (1, DQ.DB, DQ.GF)):
                for qb in range(2):
                    dst = bank(par, qb)
                    out.append(mk(self.mfma, dst, DQ.RP(x, ks), frag(qb, ks), 0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    def stream_valu(self, qb, par, masked, off):
        """P = 2^(S c - L), dS = P (dP - delta), pairs packed in place in the dP bank.  masked: scores with kv beyond the lane's
        limit become -inf first (tile = the wave's last one for off = 0, the one before it for off = 32)."""
        s, d = DQ.SB(par, qb), DQ.DB(par, qb)
        lim = DQ.A_LIM0 if qb == 0 else DQ.A_LIM1
        out = []
        if masked:
            t2 = DQ.TMP[4 * qb]
            out.append(mk("v_mov_b32", t2, NEG_INF, tag="valu"))
            for r in range(16):
                kvl = (r & 3) + 8 * (r >> 2) - off
                out.append([mk("v_cmp_le_i32", VCC, kvl, lim, tag="valu"), mk("v_cndmask_b32", s[r], t2, s[r], VCC, tag="valu")])
        for k in range(8 + 4):
            F, E, Sb, Mu, C = [], [], [], [], []
            if k < 8:
                for e in (2 * k, 2 * k + 1):
                    F.append(mk("v_fma_f32", s[e], s[e], DQ.A_C, DQ.NL[qb], tag="valu"))
                    Sb.append(mk("v_sub_f32", d[e], d[e], DQ.DD[qb], tag="valu"))
            if 0 <= k - 1 < 8:
                for e in (2 * (k - 1), 2 * (k - 1) + 1):
                    E.append(mk("v_exp_f32", s[e], s[e], tag="trans"))
            if 0 <= k - 2 < 8:
                for e in (2 * (k - 2), 2 * (k - 2) + 1):
                    Mu.append(mk("v_mul_f32", d[e], s[e], d[e], tag="valu"))
            if 0 <= k - 3 < 8:
                e = 2 * (k - 3)
                C.append(mk(self.cvt, d[8 * (e // 8) + (e % 8) // 2], d[e], d[e + 1], tag="valu"))
            out += F + E + Sb + Mu + C
        return out

    def row_read(self, x, ks, par):
        off = DQ.ROW_RING + par * DQ.ROW_SLOT + x * DQ.V_IN_SLOT
        return mk("ds_read_b128", DQ.RP(x, ks), DQ.KR[ks], tag="lds", offset=off)

    def stream_trread(self, par):
        out = []
        for ks in range(2):
            for dt in range(4):
                off = DQ.TR_RING + par * DQ.TR_SLOT + 16 * ks * 256
                out.append(mk("ds_read_b64_tr_b16", DQ.TP(dt, ks).sub(0, 2), DQ.VR[dt], tag="lds", offset=off))
                out.append(mk("ds_read_b64_tr_b16", DQ.TP(dt, ks).sub(2, 2), DQ.VR[dt], tag="lds", offset=off + 8 * 256))
        return out

    def dma_group(self, which, slot_par, guarded, ahead):
        """This wave's 2 pieces of one 32-row image of tile t + ahead: 'k' / 'v' -> the row ring, 't' -> the K copy of the
        transposed-read ring.  Guarded bodies skip tiles past the workgroup's last one."""
        rs, vd, soff = {"k": (DQ.A_KRS, DQ.KD, DQ.S_KOFF), "v": (DQ.A_VRS, DQ.VD, DQ.S_VOFF), "t": (DQ.A_KRS, DQ.TD, DQ.S_TOFF)}[which]
        base = {"k": DQ.ROW_RING + slot_par * DQ.ROW_SLOT, "v": DQ.ROW_RING + slot_par * DQ.ROW_SLOT + DQ.V_IN_SLOT,
                "t": DQ.TR_RING + slot_par * DQ.TR_SLOT}[which]
        out = []
        skip = None
        if guarded:
            skip = self.p.fresh("dma_skip")
            out.append(mk("s_add_u32", DQ.S_TMP2, DQ.S_T, ahead, tag="salu"))
            out.append(mk("s_cmp_lt_i32", DQ.S_TMP2, DQ.A_NTWG, tag="salu"))
            out.append(mk("s_cbranch_scc0", Label(skip), tag="branch"))
        out.append([mk("s_add_u32", M0, DQ.A_LDSW, base, tag="salu"), mk("s_nop", 0, tag="salu")])
        for i in range(2):
            out.append(mk("buffer_load_dwordx4", vd[i], rs, soff, tag="dma", offen=True, offset=1024 * i, lds=True))
        if guarded:
            out.append(Ins("label", (Label(skip),)))
            return [_flat(out)]
        return out

    # ------------------------------------------------------------------ one body
    def body(self, par, dq=True, s1=True, s2=True, masked=False, off=0, guarded=True, dma=True, name="body"):
        """B(t) with t & 1 == par.  dq: dQ(t); s1: the VALU work and the transposed reads of tile t+1 (masked: one of the wave's
        last two tiles, off = 32 for the one before last); s2: the row reads and S / dP of tile t+2."""
        p, cfg = self.p, self.cfg
        abl = set(cfg["abl"]) if name.startswith("F") else set()
        ng = 48
        mf = (self.dq_mfmas(par) if dq else [None] * 16) + (self.sdp_mfmas(par) if s2 else [None] * 32)
        if "mfma" in abl:
            mf = [None] * ng
        load = [0.0] * ng
        slots = [[] for _ in range(ng)]
        pre = []
        if not dq:
            # no dQ MFMAs separate this body's first VALU reads of S / dP from the MFMAs that ended the previous body
            pre += [mk("s_nop", 15), mk("s_nop", 15)]
        if dma and "dma" not in abl:
            grp = self.dma_group("k", par ^ 1, guarded, 3) + self.dma_group("v", par ^ 1, guarded, 3) + self.dma_group("t", par, guarded, 2)
            sched.place(load, slots, grp, cfg["dma"][0], cfg["dma"][1], 2)
        if s2 and "rowread" not in abl:
            first = [self.row_read(x, ks, par) for ks in range(4) for x in range(2)]
            sched.place(load, slots, first, cfg["rowread"][0], cfg["rowread"][1], 3)
            for ks in range(4, 8):
                g = 16 + 4 * (ks - 4) + 3
                for x in range(2):
                    it = self.row_read(x, ks, par)
                    load[g] += sched.weight(it)
                    slots[g].append((g + 0.5 + 0.1 * x, 3, it))
        if s1 and "trread" not in abl:
            sched.place(load, slots, self.stream_trread(par ^ 1), cfg["trread"][0], cfg["trread"][1], 4)
        if s1 and "valu" not in abl:
            w = cfg["valu"]
            sched.place(load, slots, self.stream_valu(0, par ^ 1, masked, off), w[0], w[1] - 1.0, 5)
            sched.place(load, slots, self.stream_valu(1, par ^ 1, masked, off), w[0], w[1], 6)
        self.last_load = load
        boundary = {}
        if s2:
            boundary[16] = [mk("s_nop", 1)]      # (the LDS-wait pass puts the counted wait for the first row fragments here)
        post = [mk("s_add_u32", DQ.S_T, DQ.S_T, 1), mk("s_add_u32", DQ.S_KOFF, DQ.S_KOFF, DQ.A_KTILE),
                mk("s_add_u32", DQ.S_VOFF, DQ.S_VOFF, DQ.A_VTILE), mk("s_add_u32", DQ.S_TOFF, DQ.S_TOFF, DQ.A_KTILE),
                mk("s_waitcnt", vmcnt=0, lgkmcnt=0)]
        if "barrier" not in abl:
            post.append(mk("s_barrier"))
        self.emit_body(p, mf, slots, pre=pre, boundary=boundary, post=post)

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(8):
            p.emit("v_xor_b32", DQ.KR[ks], ks << 5, DQ.A_KR0)
        for dt in range(4):
            p.emit("v_xor_b32", DQ.VR[dt], dt << 6, DQ.A_VR0)
        # loads: Q fragments straight into their AGPRs; dO and O through the (still unused) S / dP banks for delta
        tg, to = V(24, 64), V(88, 64)
        for qb in range(2):
            for ks in range(8):
                p.emit("global_load_dwordx4", DQ.QF(qb, ks), DQ.A_QO0 if qb == 0 else DQ.A_QO1, DQ.A_QB, offset=32 * ks)
        for qb in range(2):
            p.emit("global_load_dword", DQ.NL[qb], DQ.A_LO0 if qb == 0 else DQ.A_LO1, DQ.A_LB)
        for qb in range(2):
            for ks in range(8):
                p.emit("global_load_dwordx4", tg.sub(32 * qb + 4 * ks, 4), DQ.A_GO0 if qb == 0 else DQ.A_GO1, DQ.A_GB, offset=32 * ks)
        for qb in range(2):
            for ks in range(8):
                p.emit("global_load_dwordx4", to.sub(32 * qb + 4 * ks, 4), DQ.A_OO0 if qb == 0 else DQ.A_OO1, DQ.A_OB, offset=32 * ks)
        # DMA source offsets of piece 1: rows 4 further down (row images: the granule swizzle follows the row, xor 4 << 4; "tr" images:
        # the chunk swizzle depends on row & 3 only), and the instruction offset 1024 that selects the LDS piece is taken back out
        p.emit("v_mov_b32", DQ.KD[0], DQ.A_KD0)
        p.emit("v_mov_b32", DQ.VD[0], DQ.A_VD0)
        p.emit("v_mov_b32", DQ.TD[0], DQ.A_TD0)
        p.emit("v_xor_b32", DQ.KD[1], 64, DQ.A_KD0)
        p.emit("v_xor_b32", DQ.VD[1], 64, DQ.A_VD0)
        p.emit("v_add_u32", DQ.TD[1], DQ.A_KROW4, DQ.A_TD0)
        p.emit("s_nop", 0)
        p.emit("v_add_u32", DQ.KD[1], DQ.A_KROW4, DQ.KD[1])
        p.emit("v_add_u32", DQ.VD[1], DQ.A_VROW4, DQ.VD[1])
        p.emit("s_mov_b32", DQ.S_T, -2)
        p.emit("s_mov_b32", DQ.S_KOFF, 0)
        p.emit("s_mov_b32", DQ.S_VOFF, 0)
        p.emit("s_mov_b32", DQ.S_TOFF, 0)
        # K(0), V(0) rows -> row ring slot 0
        for which, rs, vd, base in (("k", DQ.A_KRS, DQ.KD, DQ.ROW_RING), ("v", DQ.A_VRS, DQ.VD, DQ.ROW_RING + DQ.V_IN_SLOT)):
            p.emit("s_add_u32", M0, DQ.A_LDSW, base)
            p.emit("s_nop", 0)
            for i in range(2):
                p.emit("buffer_load_dwordx4", vd[i], rs, DQ.S_KOFF, offen=True, offset=1024 * i, lds=True)
        # the running offsets are those of tile t+3 (rows) / t+2 (transposed copy) of the body that uses them: B(-2) stages 1 / 0
        p.emit("s_mov_b32", DQ.S_KOFF, DQ.A_KTILE)
        p.emit("s_mov_b32", DQ.S_VOFF, DQ.A_VTILE)
        for i in range(128):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        # delta = rowsum(dO * O): the 4 DMA pieces issued last may keep flying
        p.emit("s_waitcnt", vmcnt=4)
        for qb in range(2):
            acc = DQ.DD[qb]
            p.emit("v_mov_b32", acc, 0)
            p.emit("s_nop", 0)
            for i in range(32):
                p.emit(self.dot2, acc, tg[32 * qb + i], to[32 * qb + i], acc)
            p.emit("s_nop", 3)      # a DOT result read by another kind of VALU instruction: 3 wait states, NOT interlocked (measured: the copy below read a stale sum)
            p.emit("v_mov_b32", DQ.TMP[0], acc)
            p.emit("s_nop", 1)
            p.emit("v_permlane32_swap_b32", acc, DQ.TMP[0])
            p.emit("s_nop", 0)
            p.emit("v_add_f32", acc, acc, DQ.TMP[0])
            p.emit("v_sub_f32", DQ.NL[qb], 0, DQ.NL[qb])               # -LSE
        for i in range(64):                                           # dO fragments -> their AGPRs
            p.emit("v_accvgpr_write_b32", A(192 + i), tg[i])
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")

        # ---- head bodies: t = -2 (parity 0): rows + S/dP of tile 0; t = -1 (parity 1): VALU of tile 0, S/dP of tile 1 if it exists
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

        # ---- main: fast bodies while ntw - t >= 4 (tile t+1 is not one of the last two, tiles t+2, t+3 exist)
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
        p.emit("s_sub_u32", DQ.S_D, DQ.A_NTW, DQ.S_T)            # tiles left for this wave, the one whose dQ comes next included
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
            self.body(par, dq=False, s1=False, s2=False, name="ST%d" % par)        # this wave is done: stage + sync only
            p.emit("s_branch", Label("dispatch"))
            p.label("tb3_" + sfx)
            self.body(par, masked=True, off=32, name="TB3%d" % par)                # tile t+1 is the one before the last
            p.emit("s_branch", Label("dispatch"))
            p.label("tb2_" + sfx)
            self.body(par, s2=False, masked=True, off=0, name="TB2%d" % par)       # tile t+1 is the last
            p.emit("s_branch", Label("dispatch"))
            p.label("tc_" + sfx)
            self.body(par, s1=False, s2=False, name="TC%d" % par)
            p.emit("s_branch", Label("dispatch"))

        # ---- epilogue: dQ = acc * scale -> 16 bit -> wave-private LDS image (rows of 272 B); delta out
        p.label("epilogue")
        p.emit("s_nop", 15)
        T = DQ.TMP
        for qb in range(2):
            for dt in range(4):
                acc = DQ.ACC(qb, dt)
                for r4 in (0, 2):
                    for j in range(8):
                        p.emit("v_accvgpr_read_b32", T[j], acc[4 * r4 + j])
                    p.emit("s_nop", 0)
                    for j in range(8):
                        p.emit("v_mul_f32", T[j], DQ.A_SCALE, T[j])
                    p.emit("s_nop", 0)
                    p.emit(self.cvt, T[0], T[0], T[1])
                    p.emit(self.cvt, T[1], T[2], T[3])
                    p.emit(self.cvt, T[2], T[4], T[5])
                    p.emit(self.cvt, T[3], T[6], T[7])
                    p.emit("s_nop", 1)
                    p.emit("v_permlane32_swap_b32", T[0], T[2])
                    p.emit("v_permlane32_swap_b32", T[1], T[3])
                    p.emit("s_nop", 0)
                    # 16 bytes {x0[0], x1[0], x0[1], x1[1]} at row (32qb + l31), column 32dt + 8(r4 + hi)
                    p.emit("ds_write_b128", DQ.A_EPI, V(T[0].idx, 4), offset=32 * qb * DQ.EPI_ROWB + (32 * dt + 8 * r4) * 2)
                    p.emit("s_nop", 1)
        p.emit("s_waitcnt", lgkmcnt=0)
        p.emit("v_mov_b32", DQ.A_D0, DQ.DD[0])
        p.emit("v_mov_b32", DQ.A_D1, DQ.DD[1])
        p.emit("s_branch", Label("end"))
        for r in self.rare:
            p.extend(r)
        p.label("end")
        return p


def clobber_list(vbase, sregs):
    regs = ["v%d" % i for i in range(vbase, 256)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in sregs]
    return ", ".join('"%s"' % r for r in regs + ["vcc", "scc", "memory"])


def render_inline(prog, stem):
    saved = Label.text
    Label.text = lambda self: ".L%s_%s_%%=" % (stem, self.name)
    try:
        return "\n".join('"%s\\n"' % t for t in prog.text_lines()) + "\n"
    finally:
        Label.text = saved


def parse_opts(text):
    cfg = {}
    for item in filter(None, (text or "").split(",")):
        k, _, v = item.partition("=")
        if k in ("abl", "opt"):
            cfg[k] = tuple(x for x in v.split("+") if x)
        else:
            a, _, b = v.partition(":")
            cfg[k] = (float(a), float(b or 0))
    return cfg


def write_atomic(path, text):
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    ap.add_argument("--opt", default="", help="schedule windows / options: 'dq:valu=1:47,dq:abl=dma' (prefix dq: or dkv:)")
    ap.add_argument("--probe", action="store_true", help="allow timing-probe options (abl=...: bodies with wrong results; never for the product build)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    per = {"dq": [], "dkv": []}
    for item in filter(None, a.opt.split(",")):
        k, _, rest = item.partition(":")
        per[k].append(rest)
    cfgs = {k: parse_opts(",".join(v)) for k, v in per.items()}
    if any("abl" in c for c in cfgs.values()) and not a.probe:
        sys.exit("bwd_d128_gen.py: %r contains timing-probe options; they need --probe and must not go into the product build" % a.opt)
    for bf16 in (False, True):
        dt = "bf16" if bf16 else "f16"
        prog = GenDQ(bf16, **cfgs["dq"]).build()
        write_atomic(os.path.join(a.out, "fa2_bwd_dq_d128_%s.inc" % dt),
                     "// GENERATED by csrc/gen/bwd_d128_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + render_inline(prog, "fa2dq"))
        print("fa2_bwd_dq_d128_%s.inc" % dt, len(prog.ins), "instructions")
    write_atomic(os.path.join(a.out, "fa2_bwd_dq_d128_clobbers.inc"),
                 "// GENERATED by csrc/gen/bwd_d128_gen.py — do not edit.\n" + clobber_list(DQ.VBASE, DQ.CLOBBER_S) + "\n")


if __name__ == "__main__":
    main()
