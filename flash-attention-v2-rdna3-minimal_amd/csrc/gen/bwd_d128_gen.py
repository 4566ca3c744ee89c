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


def f_swz(r):
    """granule swizzle of the UNIFIED image format (one image for ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads):
    granule g of row r at r*256 + ((g ^ f(r)) << 4).  The 16 rows of a b128 lane group get 16 different granule slots, the 4 rows x 64 B
    of a transposed read 4 different 64-byte bank quarters."""
    return ((r & 3) << 2) | ((r >> 2) & 3)


def _flat(items):
    out = []
    for x in items:
        out.extend(x if isinstance(x, list) else [x])
    return out


def f_swz16(r):
    """the unified image's granule swizzle under the 16x16x32 bodies (round 6; bwd_dkv_m16_gen.py).  A lane is (n = l % 16, g = l / 16) there: a row
    read's lane group holds g = a for n in {0..3, 12..15} and g = a ^ 1 for n in {4..11}, a transposed read serves rows 4 g + (n >> 2) of two 16-lane
    groups per cycle — under f_swz both kinds hit every bank twice (SQ_LDS_BANK_CONFLICT 3.4e7 of 8.6e7 LDS cycles per c2 launch,
    profiles/r19_bwd_c2_pmc.txt; round 4's 32 x 32 bodies: 2.6e5).  (r & 7) << 1 gives a row read's 16 lanes 16 different slots ((4 ks + g) ^ f: the
    even slots to one half of the group, the odd ones to the other) and the eight rows of a transposed read eight different 32-byte slot pairs."""
    return (r & 7) << 1


class BodyEmitter:
    """What the bodies of both backward kernels share: fillers placed into MFMA gaps, counted LDS waits, the end-of-body sync."""

    def check_running_state(self, body, bookkeeping, name):
        """Legality of a placed body.  The filler streams are written against the running state a body is ENTERED with — the tile counter, the source
        offsets of the LDS-DMA pieces, M0's ring slot, the moving LDS read addresses; the book-keeping instructions that advance that state for the
        next body ride in late gaps.  A schedule window that lets a stream instruction land BEHIND the book-keeping it depends on makes it stage or
        read the next tile's slot while this tile's readers are still on it: wrong gradients, silently (profiles/r09_experiments.txt item 1: every
        LDS-DMA window of the dK / dV pass that reaches the book-keeping gaps).  Raises ValueError for such a schedule."""
        bk = set(id(i) for i in bookkeeping)
        stale = {}                                   # (kind, index) -> the book-keeping instruction that advanced it
        for ins in body:
            if ins.op in ("label", "raw", "s_waitcnt", "s_barrier", "s_nop"):
                continue
            rd, wr = [], []
            for j, o in enumerate(ins.ops):
                if isinstance(o, Neg):
                    o = o.reg
                if hasattr(o, "kind") and hasattr(o, "idx"):
                    rng = [(o.kind, o.idx + t) for t in range(o.n)]
                    writes_first = not (ins.op.startswith("s_cmp") or ins.op.startswith("s_bitcmp") or ins.op.startswith("v_cmp") or
                                        ins.op.startswith("buffer_load") or ins.op.startswith("ds_write") or ins.op.startswith("s_cbranch"))
                    (wr if (j == 0 and writes_first) else rd).extend(rng)
                    if j == 0 and ins.op.startswith("v_permlane"):
                        rd.extend(rng)
            if id(ins) in bk:
                for r in wr:
                    stale[r] = ins
                continue
            for r in rd:
                if r in stale:
                    raise ValueError("%s: illegal schedule — `%s` reads %s%d after the book-keeping `%s` advanced it for the next body (a stream window "
                                     "reaches past the book-keeping gaps)" % (name, ins.text(), r[0], r[1], stale[r].text()))
            for r in wr:
                stale.pop(r, None)

    def emit_body(self, p, mfmas, slots, pre=(), boundary=None, post=(), bookkeeping=(), name="body"):
        """mfmas: list (None = no MFMA in that gap); slots[g]: [(key, stream id, item)]; boundary: {gap: [instructions emitted in
        front of that gap's MFMA]}.  The LDS-wait pass runs over the whole body; bookkeeping: the instructions that advance the running state
        for the next body (check_running_state)."""
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
        if bookkeeping and "nocheck" not in getattr(self, "opt", ()):
            self.check_running_state(p.ins[start:], _flat(bookkeeping), name)
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
    A_DSIGN = Arg(33, "s")                         # +1.0 / -1.0 (f32 bits): the sign delta leaves with (the hand-scheduled dK/dV pass wants -delta)
    N_ARGS = 34
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
    VRB = [V(246 + i) for i in range(4)]           # opt "uni" only: transposed read addresses of rows +8..11 (the unified swizzle differs in one bit)

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
    DEFAULTS = {"valu": (1.0, 47.0), "rowread": (0.0, 15.0), "trread": (17.0, 47.0), "dma": (1.0, 14.0), "opt": (), "abl": ()}

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
            # masked scores must come out of the fma below as -inf: -inf for c >= 0, +inf for a negative scale (found by the randomised sweep:
            # -inf * c = +inf, P = inf, dQ non-finite on every call with a ragged or causal tile and scale < 0)
            t2 = DQ.TMP[4 * qb]
            out.append([mk("v_mov_b32", t2, DQ.A_C, tag="valu"), mk("v_and_b32", t2, 0x80000000, t2, tag="valu"),
                        mk("v_xor_b32", t2, 0xff800000, t2, tag="valu")])
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
                out.append(mk("ds_read_b64_tr_b16", DQ.TP(dt, ks).sub(2, 2), (DQ.VRB if "uni" in self.opt else DQ.VR)[dt], tag="lds", offset=off + 8 * 256))
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
        # the running tile offsets move on behind the DMA pieces that use them (not after the body: that would idle the matrix pipe)
        bk = [mk("s_add_u32", DQ.S_T, DQ.S_T, 1, tag="salu"), mk("s_add_u32", DQ.S_KOFF, DQ.S_KOFF, DQ.A_KTILE, tag="salu"),
              mk("s_add_u32", DQ.S_VOFF, DQ.S_VOFF, DQ.A_VTILE, tag="salu"), mk("s_add_u32", DQ.S_TOFF, DQ.S_TOFF, DQ.A_KTILE, tag="salu")]
        sched.place(load, slots, bk, 44.0, 47.0, 9)
        post = [mk("s_waitcnt", vmcnt=0, lgkmcnt=0)]
        if "barrier" not in abl:
            post.append(mk("s_barrier"))
        self.emit_body(p, mf, slots, pre=pre, boundary=boundary, post=post, bookkeeping=bk, name="dQ body " + name)

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(8):
            p.emit("v_xor_b32", DQ.KR[ks], ks << 5, DQ.A_KR0)
        for dt in range(4):
            p.emit("v_xor_b32", DQ.VR[dt], dt << 6, DQ.A_VR0)
        if "uni" in self.opt:
            p.emit("s_nop", 0)
            for dt in range(4):
                p.emit("v_xor_b32", DQ.VRB[dt], 32, DQ.VR[dt])
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
        uni = "uni" in self.opt     # unified format: rows +4 flip bit 0 of the granule swizzle in every image
        p.emit("v_xor_b32", DQ.KD[1], 16 if uni else 64, DQ.A_KD0)
        p.emit("v_xor_b32", DQ.VD[1], 16 if uni else 64, DQ.A_VD0)
        if uni:
            p.emit("v_xor_b32", DQ.TD[1], 16, DQ.A_TD0)
            p.emit("s_nop", 0)
            p.emit("v_add_u32", DQ.TD[1], DQ.A_KROW4, DQ.TD[1])
        else:
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
        p.emit("v_mul_f32", DQ.A_D0, DQ.A_DSIGN, DQ.DD[0])
        p.emit("v_mul_f32", DQ.A_D1, DQ.A_DSIGN, DQ.DD[1])
        p.emit("s_branch", Label("end"))
        for r in self.rare:
            p.extend(r)
        p.label("end")
        return p


# =====================================================================================================================
#                                                     dK / dV pass
# =====================================================================================================================
class KV:
    """Register map and operand list of the dK/dV statement (order = the operand list in fa2_bwd_d128.hip.h).

    Workgroup = 4 waves = 128 KV rows = two wave PAIRS; pair p owns KV rows [64p, 64p+64) as two 32-row blocks kvb.  The two waves
    of a pair sit on different SIMDs and split the four products (the two KV-owned accumulators do not fit one wave):
        P side   (wave 2p)    S[q,kv] = Q K^T (K rows as B fragments), P = 2^(S c - L), dV^T[d,kv] += dO^T P     accumulates dV
        dS side  (wave 2p+1)  dP[q,kv] = dO V^T - delta (V rows as B fragments; -delta is the C operand of the first k-step),
                              dS = P (dP - delta), dK^T[d,kv] += Q^T dS                                              accumulates dK
    P crosses once per tile through a 4-KiB LDS slot as the packed 16-bit B fragments the dV product consumes (lane for lane: S and
    dP have the same register layout), so S and P are formed once for both products: 4 GEMMs per (q, kv) pair.
    Q is swept in tiles of 32 rows (the launcher requires Nq % 32 == 0: no partial tile, so no row of a tile lies beyond Nq; KV rows
    beyond Nkv are clamped reads whose results are not stored); Q and dO tiles are staged ONCE each, in the unified image format (f_swz) that serves the row
    reads of one wave and the transposed reads of its partner: Q ring 4 slots (a tile lives from body t-3 to body t), dO ring 2.
    L and -delta of a tile (32 floats each) are staged by ONE 4-byte-per-lane LDS-DMA per array and tile (pair 0's waves issue them) and read
    back with four ds_read_b128 per wave in the accumulator's register layout — VMEM instructions are the dearest fillers of these bodies
    (~45 cycles of lost matrix-pipe time each, measured), LDS reads the cheapest (~5).

    Pipelines (body B(t), t = -2 .. n, one barrier each, 32 MFMAs per wave):
        P side    MFMA 0..15 dV(t), 16..31 S(t+2);  VALU: P(t+1) -> packed in place -> written to the pair's slot
        dS side   MFMA 0..15 dK(t-1), 16..31 dP(t+1);  VALU: dS(t) = P(t) * dP'(t), packed in place
      every phase's A fragments are read from LDS one phase earlier; LDS-DMA: Q(t+3), dO(t+2), L / -delta (t+3); LR(t+2) is read from LDS.
    """
    A_FO0, A_FO1 = Arg(0), Arg(1)                  # byte offset of this lane's 16 bytes (k-step 0) of its own K (P side) / V (dS side) row, block 0 / 1
    A_QD0, A_GD0 = Arg(2), Arg(3)                  # per-lane LDS-DMA source byte offset (piece 0, tile 0) in Q / dO
    A_KR0, A_VR0 = Arg(4), Arg(5)                  # per-lane LDS read offset: row fragment k-step 0 / transposed fragment d block 0 (rows +0..3)
    A_LIM0, A_LIM1 = Arg(6), Arg(7)                # P side, causal: this lane's kv row (minus 4*hi, minus the first tile's q0) per block; -big otherwise
    A_PXA = Arg(8)                                 # per-lane LDS byte address in the pair's P slot (parity 0)
    A_LDA = Arg(9)                                 # per-lane LDS byte offset into a tile's 32 L / -delta values: 16 * hi (+ the role's array)
    A_L4 = Arg(10)                                 # lane * 4: source offset of the 4-byte LDS-DMA that stages L / -delta
    A_EPI = Arg(11)                                # per-lane LDS byte address of the epilogue image: row l31, half hi
    A_FB = Arg(12, "s", 2)                         # head base of K (P side) / V (dS side)
    A_QRS, A_GRS, A_LRS = Arg(13, "s", 4), Arg(14, "s", 4), Arg(15, "s", 4)   # buffer descriptors: Q, dO, LSE (P side) / delta workspace (dS side)
    A_C, A_OSCALE = Arg(16, "s"), Arg(17, "s")     # scale * log2(e); factor of the stored accumulator (1.0: dV, scale: dK)
    A_N = Arg(18, "s")                             # number of 32-row Q tiles this workgroup sweeps
    A_QOFF0, A_GOFF0, A_LOFF0 = Arg(19, "s"), Arg(20, "s"), Arg(21, "s")     # byte offset of the first swept tile in Q / dO / the L array
    A_QTILE, A_GTILE = Arg(22, "s"), Arg(23, "s")  # bytes between consecutive tiles in Q / dO
    A_QROW4, A_GROW4 = Arg(24, "s"), Arg(25, "s")  # 4 * row bytes - 1024
    A_LDSW = Arg(26, "s")                          # wave * 2048: this wave's quarter of a tile image
    A_ROLE = Arg(27, "s")                          # 0: P side, 1: dS side
    A_LDM0 = Arg(28, "s")                          # pair 0 only (its waves stage L / -delta for the workgroup): LDS address of the role's array, parity 0; 0 = this wave does not
    A_LDSWQ = Arg(29, "s")                         # LDS offset of the rows of a Q tile this wave stages (= A_LDSW unless the Q pieces are split by role: "qsplit")
    N_ARGS, N_VARGS = 30, 12
    VBASE = 16

    @staticmethod
    def BK(par, kvb):                              # S / P bank (P side) or dP / dS bank (dS side): f32, packed in place
        return V(16 + 32 * par + 16 * kvb, 16)

    @staticmethod
    def LR(par):                                   # L (P side) / -delta (dS side) of a tile in the accumulator's register layout
        return V(80 + 16 * par, 16)

    @staticmethod
    def RP(ks):                                    # row fragments of the next phase B: Q rows (P side) / dO rows (dS side)
        return V(112 + 4 * ks, 4)

    @staticmethod
    def TP(dt, ks):                                # transposed fragments of the next phase A: dO^T (P side) / Q^T (dS side)
        return V(144 + 16 * ks + 4 * dt, 4)

    KR = [V(176 + i) for i in range(8)]            # row read addresses (P side: Q ring, moves every body; dS side: dO ring, static)
    VR = [V(184 + i) for i in range(4)]            # transposed read addresses, rows +0..3 (P side: dO ring, static; dS side: Q ring, moves)
    VRB = [V(188 + i) for i in range(4)]           # ... rows +8..11
    QD = [V(192), V(193), V(223)]                  # LDS-DMA source offsets of this wave's pieces of a Q tile (2; "qsplit": 1 on the P side, 3 on the dS side)
    GD = [V(194), V(195)]                          # ... dO tile
    PR = V(196, 16)                                # dS side: the pair's packed P words of a tile
    LIMT = [V(212), V(213)]                        # P side, masked bodies: the limit relative to the tile
    TMP = [V(214 + i) for i in range(8)]
    XA = V(222)                                    # P slot address of this lane

    @staticmethod
    def ACC(kvb, dt):
        return A(64 * kvb + 16 * dt, 16)

    @staticmethod
    def FF(kvb, ks):                               # the wave's own rows as B fragments: K (P side) / V (dS side)
        return A(128 + 32 * kvb + 4 * ks, 4)

    S_T, S_QOFF, S_GOFF, S_LOFF, S_TMP, S_TMP2 = S(60), S(61), S(62), S(63), S(64), S(65)
    S_NFAST, S_D, S_QSLOT, S_BUMP, S_M0Q = S(66), S(67), S(68), S(69), S(70)
    CLOBBER_S = list(range(60, 72))

    # (the rings were laid out below 64 KiB on the assumption that M0 carries a 16-bit LDS address; tools/ubench/lds_dma_hi.hip, round 4, shows
    #  LDS-DMA landing anywhere in the 160 KiB — the layout stays, the limit is not real; the P slots are written by ds_write and sit above)
    Q_RING, G_RING, LD_BASE, P_SLOTS, SLOT = 0, 32768, 49152, 65536, 8192      # LD_BASE: L[2 parities][64 floats] at +0, -delta likewise at +512
    EPI_ROWB = 272
    LDS_BYTES = 65536 + 16384                      # 81920 (the epilogue image, 4 x 64 rows of 272 B from 0, reuses the rings, which are dead by then)


class GenDKV(BodyEmitter):
    DEFAULTS = {"valu_p": (1.0, 31.0), "valu_s": (1.0, 31.0), "rowread": (0.0, 15.0), "trread": (16.0, 31.0), "dma": (1.0, 12.0), "lread": (20.0, 31.0), "lread_p": (1.0, 12.0), "opt": (), "abl": ()}

    def __init__(self, bf16=False, **cfg):
        self.cfg = dict(self.DEFAULTS)
        self.cfg.update(cfg)
        self.opt = set(self.cfg["opt"])
        self.bf16 = bf16
        self.mfma = "v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if bf16 else "v_cvt_pk_f16_f32"
        self.p = Program()
        # "kfold": the P side folds the scale into its K fragments — K * (-scale*log2e), rounded once to the I/O dtype as the fragments are loaded (the
        # forward's folded-scale contract, on the other operand: pure_torch_ver.py:61 scales q, the product is the same up to one rounding) — and takes
        # the tile's L as the C operand of the first k-step: the matrix pipe delivers L - S c, P = 2^-(that) is ONE v_exp_f32 with a negated source.
        # The 32 v_fma_f32 per body go; the P side is the role every body waits for (it alone carries the 32 transcendentals).
        self.kfold = "kfold" in self.opt
        # "qsplit": of the four 1-KiB pieces of a Q tile a wave pair stages, the P side takes one and the dS side three (dO stays two and two):
        # LDS-DMA pieces are the dearest fillers (~60 issue cycles each) and the P side is the heavier role
        self.qsplit = "qsplit" in self.opt

    # ------------------------------------------------------------------ MFMA lists (shared shapes, role-specific operands)
    def acc_mfmas(self, par):
        """phase A: ACC[kvb][dt] += T(dt, ks') . X(kvb, ks') with X = the packed fragments in bank `par` (P side: P(t), dS side: dS(t-1))"""
        out = []
        for ks in range(2):
            for kvb in range(2):
                for dt in range(4):
                    out.append(mk(self.mfma, KV.ACC(kvb, dt), KV.TP(dt, ks), KV.BK(par, kvb).sub(8 * ks, 4), KV.ACC(kvb, dt), tag="mfma"))
        return out

    def row_mfmas(self, par, cinit):
        """phase B: BK[par][kvb] = R(ks) . F(kvb, ks) over the 8 k-steps (P side: S(t+2); dS side: dP(t+1), starting from -delta)"""
        out = []
        for ks in range(8):
            for kvb in range(2):
                dst = KV.BK(par, kvb)
                c0 = KV.LR(par) if cinit else 0
                out.append(mk(self.mfma, dst, KV.RP(ks), KV.FF(kvb, ks), c0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    def stream_p(self, kvb, par, masked):
        """P side: P = 2^(S c - L[q]) for the tile in bank par, pairs packed in place, then the block's two fragments go to the slot."""
        s, L = KV.BK(par, kvb), KV.LR(par)
        out = []
        if masked:      # causal: q (tile-local: (r&3) + 8(r>>2), + 4*hi folded into the limit) must be >= this lane's kv row
            t2 = KV.TMP[4 * kvb]         # (-inf, or +inf for a negative scale: see GenDQ.stream_valu)
            if self.kfold:               # the bank holds L - S c: masked scores become +inf whatever the sign of the scale, P = 2^-inf = 0
                out.append(mk("v_mov_b32", t2, 0x7f800000, tag="valu"))
            else:
                out.append([mk("v_mov_b32", t2, KV.A_C, tag="valu"), mk("v_and_b32", t2, 0x80000000, t2, tag="valu"),
                            mk("v_xor_b32", t2, 0xff800000, t2, tag="valu")])
            for r in range(16):
                out.append([mk("v_cmp_ge_i32", VCC, (r & 3) + 8 * (r >> 2), KV.LIMT[kvb], tag="valu"),
                            mk("v_cndmask_b32", s[r], t2, s[r], VCC, tag="valu")])
        for k in range(8 + 2):
            F, E, C = [], [], []
            if k < 8 and not self.kfold:
                for e in (2 * k, 2 * k + 1):
                    F.append(mk("v_fma_f32", s[e], s[e], KV.A_C, Neg(L[e]), tag="valu"))
            if 0 <= k - 1 < 8:
                for e in (2 * (k - 1), 2 * (k - 1) + 1):
                    E.append(mk("v_exp_f32", s[e], Neg(s[e]) if self.kfold else s[e], tag="trans"))
            if 0 <= k - 2 < 8:
                e = 2 * (k - 2)
                C.append(mk(self.cvt, s[8 * (e // 8) + (e % 8) // 2], s[e], s[e + 1], tag="valu"))
            out += F + E + C
            if k - 2 in (3, 7):      # a k-step's four words are packed: hand them to the partner wave
                ks = (k - 2) // 4
                out.append(mk("ds_write_b128", KV.XA, s.sub(8 * ks, 4), tag="lds", offset=par * 4096 + 1024 * (2 * kvb + ks)))
        return out

    def stream_pread(self, par):
        return [mk("ds_read_b128", KV.PR.sub(4 * j, 4), KV.XA, tag="lds", offset=par * 4096 + 1024 * j) for j in range(4)]

    def stream_ds(self, kvb, par):
        """dS side: dS = P16 * dP' for the tile in bank par (dP' = dP - delta left the MFMA), pairs packed in place."""
        d = KV.BK(par, kvb)
        out = []
        for k in range(8 + 1):
            M, C = [], []
            if k < 8:
                for e in (2 * k, 2 * k + 1):
                    word = KV.PR[4 * (2 * kvb + e // 8) + (e % 8) // 2]
                    if self.bf16:
                        t = KV.TMP[4 * kvb + (2 * k + (e & 1)) % 4]      # (the two blocks' streams are interleaved: own scratch registers each)
                        M.append(mk("v_and_b32", t, 0xffff0000, word, tag="valu") if e & 1 else mk("v_lshlrev_b32", t, 16, word, tag="valu"))
                        M.append(mk("v_mul_f32", d[e], t, d[e], tag="valu"))
                    else:
                        M.append(mk("v_fma_mix_f32", d[e], word, d[e], 0, tag="valu", op_sel="[%d,0,0]" % (e & 1), op_sel_hi="[1,0,0]"))
            if 0 <= k - 1 < 8:
                e = 2 * (k - 1)
                C.append(mk(self.cvt, d[8 * (e // 8) + (e % 8) // 2], d[e], d[e + 1], tag="valu"))
            out += M + C
        return out

    def stream_rowread(self, base):
        return [mk("ds_read_b128", KV.RP(ks), KV.KR[ks], tag="lds", offset=base) for ks in range(8)]

    def stream_trread(self, base):
        out = []
        for ks in range(2):
            for dt in range(4):
                out.append(mk("ds_read_b64_tr_b16", KV.TP(dt, ks).sub(0, 2), KV.VR[dt], tag="lds", offset=base + 16 * ks * 256))
                out.append(mk("ds_read_b64_tr_b16", KV.TP(dt, ks).sub(2, 2), KV.VRB[dt], tag="lds", offset=base + (16 * ks + 8) * 256))
        return out

    def stream_dma(self, par, P=True):
        """Q(t+3) -> Q ring slot (t+3) % 4 (M0 from the running slot counter), dO(t+2) -> dO ring slot t % 2 = par; pair 0's waves also
        stage the 32 L / -delta values of tile t+3 (one 4-byte-per-lane piece; lanes 32..63 bring the next tile's, which nobody reads)."""
        out = [[mk("s_mov_b32", M0, KV.S_M0Q, tag="salu"), mk("s_nop", 0, tag="salu")]]
        for i in range((1 if P else 3) if self.qsplit else 2):
            out.append(mk("buffer_load_dwordx4", KV.QD[i], KV.A_QRS, KV.S_QOFF, tag="dma", offen=True, offset=1024 * i, lds=True))
        out.append([mk("s_add_u32", M0, KV.A_LDSW, KV.G_RING + par * KV.SLOT, tag="salu"), mk("s_nop", 0, tag="salu")])
        for i in range(2):
            out.append(mk("buffer_load_dwordx4", KV.GD[i], KV.A_GRS, KV.S_GOFF, tag="dma", offen=True, offset=1024 * i, lds=True))
        skip = self.p.fresh("ld_skip")
        out.append([mk("s_cmp_eq_u32", KV.A_LDM0, 0, tag="salu"), mk("s_cbranch_scc1", Label(skip), tag="branch"),
                    mk("s_add_u32", M0, KV.A_LDM0, (par ^ 1) * 256, tag="salu"), mk("s_nop", 0, tag="salu"),
                    mk("buffer_load_dword", KV.A_L4, KV.A_LRS, KV.S_LOFF, tag="dma", offen=True, lds=True),
                    Ins("label", (Label(skip),))])
        return out

    def stream_lread(self, par):
        """L (P side) / -delta (dS side) of tile t+2 -> LR(par), from the parity-par slot of the role's array (A_LDA carries the array)"""
        return [mk("ds_read_b128", KV.LR(par).sub(4 * g, 4), KV.A_LDA, tag="lds", offset=KV.LD_BASE + par * 256 + 32 * g) for g in range(4)]

    # ------------------------------------------------------------------ one body
    def body(self, role, par, acc=True, valu=True, row=True, masked=False, tr=True, rr=True, name="body"):
        """B(t), t & 1 == par, for one role.  acc: phase A (P: dV(t), dS: dK(t-1)); valu: the VALU work (P: tile t+1, bank par^1;
        dS: tile t, bank par); row: phase B (P: S(t+2) -> bank par; dS: dP(t+1) -> bank par^1); tr / rr: the transposed / row
        fragment reads for the NEXT body's phase A / this body's phase B."""
        p, cfg = self.p, self.cfg
        abl = set(cfg["abl"]) if name.startswith("F") else set()
        ng = 32
        P = role == 0
        if P:
            mf = (self.acc_mfmas(par) if acc else [None] * 16) + (self.row_mfmas(par, self.kfold) if row else [None] * 16)
        else:
            mf = (self.acc_mfmas(par ^ 1) if acc else [None] * 16) + (self.row_mfmas(par ^ 1, True) if row else [None] * 16)
        if "mfma" in abl:
            mf = [None] * ng
        load = [0.0] * ng
        slots = [[] for _ in range(ng)]
        pre = []
        if not acc:
            pre += [mk("s_nop", 15), mk("s_nop", 15)]
        if P and masked and valu:
            for kvb in range(2):      # this lane's limit relative to tile t+1: LIM - 32 (t + 1)
                pre.append(mk("s_add_u32", KV.S_TMP, KV.S_T, 1))
                pre.append(mk("s_lshl_b32", KV.S_TMP, KV.S_TMP, 5))
                pre.append(mk("v_subrev_u32", KV.LIMT[kvb], KV.S_TMP, KV.A_LIM0 if kvb == 0 else KV.A_LIM1))
        if "dma" not in abl:
            sched.place(load, slots, self.stream_dma(par, P), cfg["dma"][0], cfg["dma"][1], 2)
            # (kfold, P side: L of tile t+2 is the C operand of this body's first S k-step, MFMA 16 — read it in the first phase)
            lw = cfg["lread_p"] if (P and self.kfold) else cfg["lread"]
            sched.place(load, slots, self.stream_lread(par), lw[0], lw[1], 8)
        if P:
            # Q rows of tile t+2 (Q ring, running address) for phase B; dO^T of tile t+1 (dO ring slot par^1) for the next body's phase A
            if rr and "rowread" not in abl:
                sched.place(load, slots, self.stream_rowread(KV.Q_RING), cfg["rowread"][0], cfg["rowread"][1], 3)
            if tr and "trread" not in abl:
                sched.place(load, slots, self.stream_trread(KV.G_RING + (par ^ 1) * KV.SLOT), cfg["trread"][0], cfg["trread"][1], 4)
            if valu and "valu" not in abl:
                w = cfg["valu_p"]
                sched.place(load, slots, self.stream_p(0, par ^ 1, masked), w[0], w[1] - 1.0, 5)
                sched.place(load, slots, self.stream_p(1, par ^ 1, masked), w[0], w[1], 6)
        else:
            # dO rows of tile t+1 (dO ring slot par^1) for phase B; Q^T of tile t (Q ring, running address) for the next body's phase A
            if valu and "valu" not in abl:
                sched.place(load, slots, self.stream_pread(par), 0.0, 1.0, 1)
                w = cfg["valu_s"]
                sched.place(load, slots, self.stream_ds(0, par), w[0], w[1] - 1.0, 5)
                sched.place(load, slots, self.stream_ds(1, par), w[0], w[1], 6)
            if rr and "rowread" not in abl:
                sched.place(load, slots, self.stream_rowread(KV.G_RING + (par ^ 1) * KV.SLOT), cfg["rowread"][0], cfg["rowread"][1], 3)
            if tr and "trread" not in abl:
                sched.place(load, slots, self.stream_trread(KV.Q_RING), cfg["trread"][0], cfg["trread"][1] - 3.0, 4)
        # book-keeping for the next body — the running tile offsets, the Q ring position of this wave's moving read addresses and of
        # the DMA — rides in the gaps behind the last use of each value in this body (after the body it would idle the matrix pipe)
        bk1 = [mk("s_add_u32", KV.S_T, KV.S_T, 1, tag="salu"), mk("s_add_u32", KV.S_QOFF, KV.S_QOFF, KV.A_QTILE, tag="salu"),
               mk("s_add_u32", KV.S_GOFF, KV.S_GOFF, KV.A_GTILE, tag="salu"), mk("s_add_u32", KV.S_LOFF, KV.S_LOFF, 128, tag="salu"),
               mk("s_add_u32", KV.S_QSLOT, KV.S_QSLOT, 1, tag="salu"), mk("s_and_b32", KV.S_QSLOT, KV.S_QSLOT, 3, tag="salu"),
               # this wave's own ring position wraps when the counter reaches 0 (P side: row reads of tile t+2) / 2 (dS side: tile t)
               [mk("s_cmp_eq_u32", KV.S_QSLOT, 0 if P else 2, tag="salu"), mk("s_cselect_b32", KV.S_BUMP, 4 * KV.SLOT, 0, tag="salu")],
               mk("s_sub_u32", KV.S_BUMP, KV.SLOT, KV.S_BUMP, tag="salu"),
               # the DMA slot runs one ahead of the P side's row-read slot
               mk("s_add_u32", KV.S_TMP2, KV.S_QSLOT, 1, tag="salu"), mk("s_and_b32", KV.S_TMP2, KV.S_TMP2, 3, tag="salu"),
               mk("s_lshl_b32", KV.S_TMP2, KV.S_TMP2, 13, tag="salu"), mk("s_add_u32", KV.S_M0Q, KV.S_TMP2, KV.A_LDSWQ, tag="salu")]
        moving = KV.KR if P else KV.VR + KV.VRB
        bk2 = [mk("v_add_u32", r, KV.S_BUMP, r, tag="valu") for r in moving]
        if "bk" in abl:
            bk1, bk2 = [], []
        sched.place(load, slots, bk1, 20.0, 27.0, 9)
        sched.place(load, slots, bk2, 28.0, 31.0, 9)
        self.last_load = load
        post = [mk("s_waitcnt", vmcnt=0, lgkmcnt=0)]
        if "barrier" not in abl:
            post.append(mk("s_barrier"))
        self.emit_body(p, mf, slots, pre=pre, post=post, bookkeeping=bk1 + bk2, name="dK/dV body %s (%s side)" % (name, "P" if P else "dS"))

    # ------------------------------------------------------------------ one role's sweep
    def role_code(self, role):
        """bodies t = -2 .. n for one role (labels carry the role suffix).  Every path runs exactly n + 3 bodies."""
        p = self.p
        P = role == 0
        sfx = "_p" if P else "_s"
        L = lambda name: Label(name + sfx)  # noqa: E731
        # ---- t = -2 (parity 0): P: S(0);  dS: nothing
        if P:
            self.body(0, 0, acc=False, valu=False, row=True, tr=False, name="H1")
        else:
            self.body(1, 0, acc=False, valu=False, row=False, tr=False, rr=False, name="H1")
        # ---- t = -1 (parity 1): P: P(0) (masked), S(1) if n >= 2;  dS: dP(0)
        if P:
            p.emit("s_cmp_ge_i32", KV.A_N, 2)
            p.emit("s_cbranch_scc1", L("h2"))
            self.body(0, 1, acc=False, valu=True, row=False, masked=True, rr=False, name="H2b")
            p.emit("s_branch", L("loop"))
            p.label("h2" + sfx)
            self.body(0, 1, acc=False, valu=True, row=True, masked=True, name="H2")
        else:
            self.body(1, 1, acc=False, valu=False, row=True, tr=False, name="H2")
        # ---- t >= 0: dispatch on the tiles left, d = n - t (tile t included)
        p.label("loop" + sfx)
        p.emit("s_cmp_gt_i32", KV.S_T, KV.A_N)
        p.emit("s_cbranch_scc1", L("epilogue"))
        p.emit("s_sub_u32", KV.S_D, KV.A_N, KV.S_T)
        p.emit("s_and_b32", KV.S_TMP, KV.S_T, 1)
        p.emit("s_cmp_eq_u32", KV.S_TMP, 1)
        p.emit("s_cbranch_scc1", L("disp_odd"))
        for par, ps in ((0, "e"), (1, "o")):
            if par == 1:
                p.label("disp_odd" + sfx)
            # fast bodies: t >= 3 (past the causal diagonal of every pair) and d >= 3 (tiles t+1, t+2 exist): d - 2 of them in a row
            p.emit("s_cmp_lt_i32", KV.S_D, 3)
            p.emit("s_cbranch_scc1", L("tail_" + ps))
            p.emit("s_cmp_lt_i32", KV.S_T, 3)
            p.emit("s_cbranch_scc1", L("full_" + ps))
            p.emit("s_sub_u32", KV.S_NFAST, KV.S_D, 2)
            p.emit("s_branch", L("fast%d" % par))
            p.label("full_" + ps + sfx)         # like a fast body; the P side masks (tiles 1..3 of the sweep); dS side at t = 0 has no dK yet
            if P:
                self.body(0, par, masked=True, name="G%d" % par)
            else:
                p.emit("s_cmp_eq_u32", KV.S_T, 0)
                p.emit("s_cbranch_scc1", L("first_" + ps))
                self.body(1, par, name="G%d" % par)
                p.emit("s_branch", L("loop"))
                p.label("first_" + ps + sfx)
                self.body(1, par, acc=False, name="G0%d" % par)
            p.emit("s_branch", L("loop"))
            p.label("tail_" + ps + sfx)
            p.emit("s_cmp_eq_u32", KV.S_D, 2)
            p.emit("s_cbranch_scc1", L("d2_" + ps))
            p.emit("s_cmp_eq_u32", KV.S_D, 1)
            p.emit("s_cbranch_scc1", L("d1_" + ps))
            # d == 0 (t == n): P: nothing;  dS: dK(n-1)
            if P:
                self.body(0, par, acc=False, valu=False, row=False, tr=False, rr=False, name="T0%d" % par)
            else:
                self.body(1, par, acc=True, valu=False, row=False, tr=False, rr=False, name="T0%d" % par)
            p.emit("s_branch", L("loop"))
            p.label("d2_" + ps + sfx)           # tiles t, t+1 left.  P: dV(t), P(t+1), no S;  dS: dK(t-1) (if t >= 1), dS(t), dP(t+1)
            if P:
                self.body(0, par, row=False, masked=True, rr=False, name="T2%d" % par)
            else:
                p.emit("s_cmp_eq_u32", KV.S_T, 0)
                p.emit("s_cbranch_scc1", L("d2first_" + ps))
                self.body(1, par, name="T2%d" % par)
                p.emit("s_branch", L("loop"))
                p.label("d2first_" + ps + sfx)
                self.body(1, par, acc=False, name="T20%d" % par)
            p.emit("s_branch", L("loop"))
            p.label("d1_" + ps + sfx)           # tile t is the last.  P: dV(t) only;  dS: dK(t-1) (if t >= 1), dS(t), no dP
            if P:
                self.body(0, par, valu=False, row=False, tr=False, rr=False, name="T1%d" % par)
            else:
                p.emit("s_cmp_eq_u32", KV.S_T, 0)
                p.emit("s_cbranch_scc1", L("d1first_" + ps))
                self.body(1, par, row=False, rr=False, name="T1%d" % par)
                p.emit("s_branch", L("loop"))
                p.label("d1first_" + ps + sfx)
                self.body(1, par, acc=False, row=False, rr=False, name="T10%d" % par)
            p.emit("s_branch", L("loop"))
        # the fast loop: alternating parities until the counter runs out, then back to the dispatch
        for par in (0, 1):
            p.label("fast%d" % par + sfx)
            self.body(role, par, name="F%d" % par)
            p.emit("s_sub_u32", KV.S_NFAST, KV.S_NFAST, 1)
            p.emit("s_cmp_gt_i32", KV.S_NFAST, 0)
            if par == 0:
                p.emit("s_cbranch_scc0", L("loop"))
            else:
                p.emit("s_cbranch_scc1", L("fast0"))
                p.emit("s_branch", L("loop"))
        p.label("epilogue" + sfx)

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(8):
            p.emit("v_xor_b32", KV.KR[ks], ks << 5, KV.A_KR0)
        for dt in range(4):
            p.emit("v_xor_b32", KV.VR[dt], dt << 6, KV.A_VR0)
        p.emit("s_nop", 0)
        for dt in range(4):
            p.emit("v_xor_b32", KV.VRB[dt], 32, KV.VR[dt])
        # own rows (K or V) -> B fragments in AGPRs
        for kvb in range(2):
            for ks in range(8):
                p.emit("global_load_dwordx4", KV.FF(kvb, ks), KV.A_FO0 if kvb == 0 else KV.A_FO1, KV.A_FB, offset=32 * ks)
        # DMA source offsets of piece 1: rows 4 further down flip bit 0 of the unified granule swizzle
        p.emit("v_mov_b32", KV.QD[0], KV.A_QD0)
        p.emit("v_mov_b32", KV.GD[0], KV.A_GD0)
        if self.qsplit:
            # dS side: its Q pieces are the row quads 1, 2, 3 of the pair's 16 rows — the unified swizzle's (row >> 2) & 3 term goes 1 -> 2 -> 3
            p.emit("v_xor_b32", KV.QD[1], 0x30, KV.A_QD0)
            p.emit("v_xor_b32", KV.QD[2], 0x20, KV.A_QD0)
        else:
            p.emit("v_xor_b32", KV.QD[1], 16, KV.A_QD0)
        p.emit("v_xor_b32", KV.GD[1], 16, KV.A_GD0)
        p.emit("v_mov_b32", KV.XA, KV.A_PXA)
        p.emit("v_add_u32", KV.QD[1], KV.A_QROW4, KV.QD[1])
        p.emit("v_add_u32", KV.GD[1], KV.A_GROW4, KV.GD[1])
        if self.qsplit:
            p.emit("v_add_u32", KV.QD[2], KV.A_QROW4, KV.QD[2])
            p.emit("s_nop", 0)
            p.emit("v_add_u32", KV.QD[2], KV.A_QROW4, KV.QD[2])
        p.emit("s_mov_b32", KV.S_T, -2)
        p.emit("s_mov_b32", KV.S_QOFF, KV.A_QOFF0)
        # Q(0) -> Q ring slot 0
        p.emit("s_add_u32", M0, KV.A_LDSWQ, KV.Q_RING)
        p.emit("s_nop", 0)
        if self.qsplit:
            p.emit("buffer_load_dwordx4", KV.QD[0], KV.A_QRS, KV.S_QOFF, offen=True, offset=0, lds=True)
            p.emit("s_cmp_eq_u32", KV.A_ROLE, 1)
            p.emit("s_cbranch_scc0", Label("q0_one"))
            for i in (1, 2):
                p.emit("buffer_load_dwordx4", KV.QD[i], KV.A_QRS, KV.S_QOFF, offen=True, offset=1024 * i, lds=True)
            p.label("q0_one")
        else:
            for i in range(2):
                p.emit("buffer_load_dwordx4", KV.QD[i], KV.A_QRS, KV.S_QOFF, offen=True, offset=1024 * i, lds=True)
        p.emit("s_mov_b32", KV.S_LOFF, KV.A_LOFF0)
        p.emit("s_cmp_eq_u32", KV.A_LDM0, 0)
        p.emit("s_cbranch_scc1", Label("no_ld0"))
        p.emit("s_mov_b32", M0, KV.A_LDM0)
        p.emit("s_nop", 0)
        p.emit("buffer_load_dword", KV.A_L4, KV.A_LRS, KV.S_LOFF, offen=True, lds=True)
        p.label("no_ld0")
        # running state of body -2: it stages Q(1) -> slot 1, dO(0) -> slot 0, L / -delta (1); the P side reads Q rows from slot 0
        # (tile t+2 = 0), the dS side's transposed reads belong to tile t = -2, i.e. slot 2 of the ring
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
        if self.kfold:
            # K fragments * (-scale*log2e), rounded once to the I/O dtype (64 accumulator registers, once per workgroup)
            T = KV.TMP
            for i in range(64):
                areg = A(128 + i)
                t0, t1 = T[2 * (i & 1)], T[2 * (i & 1) + 1]
                p.emit("v_accvgpr_read_b32", t0, areg)
                p.emit("s_nop", 0)
                if self.bf16:
                    p.emit("v_and_b32", t1, 0xffff0000, t0)
                    p.emit("v_lshlrev_b32", t0, 16, t0)
                else:
                    p.emit("v_lshrrev_b32", t1, 16, t0)
                    p.emit("v_cvt_f32_f16", t0, t0)
                    p.emit("v_cvt_f32_f16", t1, t1)
                p.emit("s_nop", 0)
                p.emit("v_mul_f32", t0, Neg(KV.A_C), t0)
                p.emit("v_mul_f32", t1, Neg(KV.A_C), t1)
                p.emit("s_nop", 0)
                p.emit(self.cvt, t0, t0, t1)
                p.emit("s_nop", 0)
                p.emit("v_accvgpr_write_b32", areg, t0)
            p.emit("s_nop", 1)
        p.emit("s_barrier")
        self.role_code(0)
        p.emit("s_branch", Label("epilogue"))
        p.label("role_s")
        for r in KV.VR + KV.VRB:
            p.emit("v_add_u32", r, 2 * KV.SLOT, r)
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")
        self.role_code(1)

        # ---- epilogue (both roles): acc * factor -> 16 bit -> wave-private LDS image (rows of 272 B) over the dead rings
        p.label("epilogue")
        p.emit("s_nop", 15)
        T = KV.TMP
        for kvb in range(2):
            for dt in range(4):
                acc = KV.ACC(kvb, dt)
                for r4 in (0, 2):
                    for j in range(8):
                        p.emit("v_accvgpr_read_b32", T[j], acc[4 * r4 + j])
                    p.emit("s_nop", 0)
                    for j in range(8):
                        p.emit("v_mul_f32", T[j], KV.A_OSCALE, T[j])
                    p.emit("s_nop", 0)
                    p.emit(self.cvt, T[0], T[0], T[1])
                    p.emit(self.cvt, T[1], T[2], T[3])
                    p.emit(self.cvt, T[2], T[4], T[5])
                    p.emit(self.cvt, T[3], T[6], T[7])
                    p.emit("s_nop", 1)
                    p.emit("v_permlane32_swap_b32", T[0], T[2])
                    p.emit("v_permlane32_swap_b32", T[1], T[3])
                    p.emit("s_nop", 0)
                    p.emit("ds_write_b128", KV.A_EPI, V(T[0].idx, 4), offset=32 * kvb * KV.EPI_ROWB + (32 * dt + 8 * r4) * 2)
                    p.emit("s_nop", 1)
        p.emit("s_waitcnt", lgkmcnt=0)
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
        for fold in (False, True):      # two dK / dV bodies per dtype: scale applied to the f32 scores / folded into the K fragments ("kfold"; host: option "fold")
            c = dict(cfgs["dkv"])
            c["opt"] = tuple(o for o in c.get("opt", ()) if o != "kfold") + (("kfold",) if fold else ())
            prog = GenDKV(bf16, **c).build()
            fn = "fa2_bwd_dkv_d128_%s%s.inc" % (dt, "_fold" if fold else "")
            write_atomic(os.path.join(a.out, fn),
                         "// GENERATED by csrc/gen/bwd_d128_gen.py %s — do not edit.  %d instructions.\n" % (a.opt, len(prog.ins)) + render_inline(prog, "fa2dkv"))
            print(fn, len(prog.ins), "instructions")
    write_atomic(os.path.join(a.out, "fa2_bwd_dq_d128_clobbers.inc"),
                 "// GENERATED by csrc/gen/bwd_d128_gen.py — do not edit.\n" + clobber_list(DQ.VBASE, DQ.CLOBBER_S) + "\n")
    write_atomic(os.path.join(a.out, "fa2_bwd_dkv_d128_clobbers.inc"),
                 "// GENERATED by csrc/gen/bwd_d128_gen.py — do not edit.\n" + clobber_list(KV.VBASE, KV.CLOBBER_S) + "\n")


if __name__ == "__main__":
    main()
