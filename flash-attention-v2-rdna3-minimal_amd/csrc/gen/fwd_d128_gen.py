#!/usr/bin/env python3
"""Generator of the hand-scheduled FlashAttention-2 forward main block for D = 128 on gfx950.

Replaces the compiler-scheduled steady state of fa2_fwd_kernel.hip.h for the headline shape class
(reference counterpart: kernel_fp16.cu:381-508, the per-KV-block loop of fwd_kernel).  The output is
the body of ONE inline-asm statement (fa2_fwd_d128_{f16,bf16}.inc) that the HIP kernel
`fwd_d128_kernel` (fa2_fwd_d128.hip.h) wraps: the HIP side computes addresses, the asm block does
everything from the Q load to the normalised O tile staged in LDS, the HIP side stores it.

Shape of the computation (one workgroup = 4 waves = 256 Q rows, ONE wave per SIMD, 512 registers):
    wave w owns Q rows [64w, 64w+64) as two 32-row blocks qb = 0, 1; KV tiles of 64 rows.
    accumulator file:  O[qb][dt]  a[0:127]   (f32, 8 tiles of 32(d) x 32(q))
                       Q[qb][ks]  a[128:191] (MFMA B fragments, loaded once)
                       K[kvb][ks] a[192:255] (MFMA A fragments of the next tile, ds_read_b128 straight into AGPRs)
    arch VGPRs:        S/P banks  v[24:151]  (4 x 32: per q block two banks alternating by tile parity; P is packed IN PLACE)
                       V^T frags  v[152:215] (ds_read_b64_tr_b16)
    both products are "swapped" (S^T = K Q^T, O^T = V^T P^T) exactly as in fa2_fwd_kernel.hip.h, so a lane owns one
    Q row of each block and the softmax is lane-local plus one v_permlane32_swap.

Software pipeline.  Body B(t), t = -2 .. ntiles-1, is 64 MFMAs:
    MFMA  0..31  PV(t)        O[qb] += V(t)^T P(t)^T          (qb 0 then qb 1)
    MFMA 32..63  QK(t+2)      S(t+2)[qb] = K(t+2) Q[qb]^T      (qb 0 then qb 1; kv halves alternate)
  and between them ("gaps") the single-issue work, spread by the scheduler below so that every MFMA gap carries
  about five instructions:
    E0  exp/sum/pack of tile t+1, q block 0      (gaps 0..47)
    M1  row max + rescale decision of tile t+1, q block 1   (gaps 2..)
    E1  exp/sum/pack of tile t+1, q block 1      (after M1 .. 63)
    M0  row max + decision of tile t+2, q block 0  (gaps 50..63)
    K(t+2) fragment reads (gaps 0..31), V(t+1) transpose reads (gaps 33..63), LDS-DMA of K(t+3) and V(t+2)
  One s_barrier per body.  The O rescale of the deferred-max scheme is a rare out-of-line block entered between
  the two MFMA phases (all of PV(t) is in O, nothing of tile t+1 yet), so every value at the old reference is scaled once.
Head / tail bodies are the same generator with streams switched off (and the tail masks switched on).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))
from isa import A, S, V, Arg, Ins, Label, M0, Neg, OFF, Program, VCC, mk  # noqa: E402

THR = 8.0            # deferred rescale threshold, log2 units (FA2_DEFER_THR of the HIP kernel)
NEG_INF = float("-inf")

# ---- inline-asm operands (order = the operand list of the asm statement in fa2_fwd_d128.hip.h)
A_LSE0, A_LSE1 = Arg(0), Arg(1)                    # "=v" outputs: log2-domain LSE of this lane's row in q block 0 / 1
A_Q0, A_Q1 = Arg(2, "v", 2), Arg(3, "v", 2)        # 64-bit global address of this lane's 16 Q bytes (k-step 0) in block 0 / 1
A_KRS, A_VRS = Arg(4, "s", 4), Arg(5, "s", 4)      # buffer descriptors of this head's K / V matrix
A_KD0, A_VD0 = Arg(6), Arg(7)                      # per-lane LDS-DMA source byte offset (piece 0, tile 0), K / V
A_KR0, A_VR0 = Arg(8), Arg(9)                      # per-lane LDS read offset of K fragment k-step 0 / V^T fragment d-block 0
A_LIM0, A_LIM1 = Arg(10), Arg(11)                  # last-tile mask: kv index (local, minus 4*hi) must be <= this, per q block
A_C = Arg(12, "s")                                 # scale * log2(e), f32 bits
A_NTW, A_NTWG = Arg(13, "s"), Arg(14, "s")         # KV tiles of this wave / of the workgroup
A_KTILE, A_VTILE = Arg(15, "s"), Arg(16, "s")      # bytes between consecutive KV tiles in K / V
A_KROW16, A_VROW16 = Arg(17, "s"), Arg(18, "s")    # bytes of 16 rows of K / V (stride between DMA pieces)
A_LDSW = Arg(19, "s")                              # wave * 1024: this wave's slice of a DMA piece
A_EPI = Arg(20)                                    # per-lane LDS byte address of the epilogue image: row l31, half hi
N_ARGS = 21

# ---- fixed registers (everything below is clobbered by the asm statement)
VBASE = 24


def SB(qb, par):                                   # S / P bank of q block qb, tile parity par: 32 VGPRs
    return V(VBASE + 64 * qb + 32 * par, 32)


def VF(dt, ks):                                    # V^T fragment (4 VGPRs)
    return V(152 + 16 * ks + 4 * dt, 4)


KR = [V(216 + i) for i in range(8)]
VR = [V(224 + i) for i in range(4)]
MREF = [V(228), V(229)]                            # running reference max (raw score units)
MC = [V(230), V(231)]                              # MREF * c
LA = [V(232), V(233)]                              # running row sums (two chains per q block)
LB = [V(234), V(235)]
FSC = [V(236), V(237)]                             # pending O rescale factor
MXA = [V(238), V(239)]                             # row-max chains
MXB = [V(240), V(241)]
TMP = [V(242 + i) for i in range(10)]              # v242..v251
NEGINF = V(252)
T2 = [V(253), V(254), V(255)]

S_T, S_KOFF, S_VOFF, S_FLAG, S_TMP, S_TMP2 = S(60), S(61), S(62), S(63), S(64), S(65)
S_KR2, S_KR3, S_VR2, S_VR3, S_NFAST, S_D = S(66), S(67), S(68), S(69), S(70), S(71)
S_T2, S_T3 = S(72), S(73)
CLOBBER_S = list(range(60, 74))
CLOBBER_V = list(range(VBASE, 256))

K_SLOT, V_BASE, SLOT_B = 0, 32768, 16384
EPI_ROWB = 272                                     # bytes per staged O row (256 + 16 pad)
LDS_BYTES = 4 * 64 * EPI_ROWB                      # 69632: the epilogue image is the high-water mark


def OACC(qb, dt):
    return A(64 * qb + 16 * dt, 16)


def QF(qb, ks):
    return A(128 + 32 * qb + 4 * ks, 4)


def KF(kvb, ks):
    return A(192 + 32 * kvb + 4 * ks, 4)


class Gen:
    def __init__(self, bf16=False):
        self.bf16 = bf16
        self.mfma = "v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if bf16 else "v_cvt_pk_f16_f32"
        self.p = Program()
        self.rare = []            # out-of-line blocks appended after the main code
        self.body_id = 0

    # ------------------------------------------------------------------ MFMA lists
    def pv_mfmas(self, par, qb):
        out = []
        for ks in range(4):
            pfrag = SB(qb, par).sub(16 * (ks >> 1) + 8 * (ks & 1), 4)
            for dt in range(4):
                out.append(mk(self.mfma, OACC(qb, dt), VF(dt, ks), pfrag, OACC(qb, dt), tag="mfma"))
        return out

    def qk_mfmas(self, par, qb):
        out = []
        for ks in range(8):
            for kvb in range(2):
                dst = SB(qb, par).sub(16 * kvb, 16)
                out.append(mk(self.mfma, dst, KF(kvb, ks), QF(qb, ks), 0 if ks == 0 else dst, tag="mfma"))
        return out

    # ------------------------------------------------------------------ filler streams
    def stream_exp(self, qb, par):
        """P = 2^(S*c - m*c) in place, two row-sum chains, pack pairs in place (112 instructions), skewed so that
        consecutive instructions of the stream are independent."""
        b = SB(qb, par)
        out = []
        for k in range(32 + 3):
            if k < 32:
                out.append(mk("v_fma_f32", b[k], b[k], A_C, Neg(MC[qb]), tag="valu"))
            if 0 <= k - 1 < 32:
                out.append(mk("v_exp_f32", b[k - 1], b[k - 1], tag="trans"))
            if 0 <= k - 2 < 32:
                e = k - 2
                acc = LA[qb] if (e & 1) == 0 else LB[qb]
                out.append(mk("v_add_f32", acc, acc, b[e], tag="valu"))
            if 0 <= k - 3 < 32 and ((k - 3) & 1) == 1:
                e = k - 4                                     # pair (e, e+1)
                dst = b[8 * (e // 8) + (e % 8) // 2]
                out.append(mk(self.cvt, dst, b[e], b[e + 1], tag="valu"))
        return out

    def stream_max(self, qb, par, masked):
        """mask (tail bodies) -> row max of the 32 scores of this lane -> half-wave exchange -> rescale decision."""
        b = SB(qb, par)
        out = []
        lim = A_LIM0 if qb == 0 else A_LIM1
        if masked:
            # element r of kv half kvb is kv_local = 32*kvb + (r&3) + 8*(r>>2) (+ 4*hi folded into lim): masked iff > lim
            for kvb in range(2):
                for r in range(16):
                    kvl = 32 * kvb + (r & 3) + 8 * (r >> 2)
                    out.append(mk("v_cmp_gt_i32", VCC, kvl, lim, tag="valu"))
                    out.append(mk("v_cndmask_b32", b[16 * kvb + r], b[16 * kvb + r], NEGINF, VCC, tag="valu"))
        for (mx, off) in ((MXA[qb], 0), (MXB[qb], 16)):
            out.append(mk("v_max3_f32", mx, b[off], b[off + 1], b[off + 2], tag="valu"))
        for i in range(6):
            for (mx, off) in ((MXA[qb], 0), (MXB[qb], 16)):
                out.append(mk("v_max3_f32", mx, mx, b[off + 3 + 2 * i], b[off + 4 + 2 * i], tag="valu"))
        out.append(mk("v_max3_f32", MXA[qb], MXA[qb], b[15], b[31], tag="valu"))
        t = TMP[0 + 2 * qb]
        t2 = TMP[1 + 2 * qb]
        out.append(mk("v_max_f32", MXA[qb], MXA[qb], MXB[qb], tag="valu"))
        out.append(mk("v_mov_b32", t, MXA[qb], tag="valu"))
        out.append(mk("s_nop", 1, tag="salu"))
        out.append(mk("v_permlane32_swap_b32", MXA[qb], t, tag="valu"))
        out.append(mk("v_max_f32", MXA[qb], MXA[qb], t, tag="valu"))
        out.append(mk("v_fma_f32", t2, MXA[qb], A_C, Neg(MC[qb]), tag="valu"))
        lab = self.p.fresh("rare_m")
        # (a list inside a stream is an atomic group: the scheduler keeps it contiguous — a branch and its return label)
        out.append([mk("v_cmp_lt_f32", VCC, THR, t2, tag="valu"), mk("s_cbranch_vccnz", Label(lab), tag="branch"),
                    Ins("label", (Label(lab + "_ret"),))])
        # out-of-line: move the reference, scale the row sums now, leave the O rescale pending
        r = []
        r.append(Ins("label", (Label(lab),)))
        r.append(mk("v_max_f32", t, MREF[qb], MXA[qb]))                 # m_new
        r.append(mk("v_mul_f32", t2, A_C, t))                           # m_new * c
        r.append(mk("v_sub_f32", MXB[qb], MC[qb], t2))                  # (m_old - m_new) * c   (<= 0; -inf on the first tile)
        r.append(mk("v_mov_b32", MREF[qb], t))
        r.append(mk("v_exp_f32", MXB[qb], MXB[qb]))
        r.append(mk("v_mov_b32", MC[qb], t2))
        r.append(mk("s_or_b32", S_FLAG, S_FLAG, 1 << qb))
        r.append(mk("s_nop", 0))
        r.append(mk("v_mul_f32", LA[qb], LA[qb], MXB[qb]))
        r.append(mk("v_mul_f32", LB[qb], LB[qb], MXB[qb]))
        r.append(mk("v_mov_b32", FSC[qb], MXB[qb]))
        r.append(mk("s_branch", Label(lab + "_ret")))
        self.rare.append(r)
        return out

    def stream_kread(self, par):
        out = []
        for ks in range(8):
            for kvb in range(2):
                out.append(mk("ds_read_b128", KF(kvb, ks), KR[ks], tag="lds", offset=K_SLOT + par * SLOT_B + kvb * 8192))
        return out

    def stream_vread(self, par):
        out = []
        for ks in range(4):
            for dt in range(4):
                off = V_BASE + par * SLOT_B + 16 * ks * 256
                out.append(mk("ds_read_b64_tr_b16", VF(dt, ks).sub(0, 2), VR[dt], tag="lds", offset=off))
                out.append(mk("ds_read_b64_tr_b16", VF(dt, ks).sub(2, 2), VR[dt], tag="lds", offset=off + 8 * 256))
        return out

    def dma_group(self, which, slot_par, guarded, ahead):
        """4 LDS-DMA pieces of one K or V tile (tile index = t + ahead)."""
        out = []
        rs, vd, soff = (A_KRS, A_KD0, S_KOFF) if which == "k" else (A_VRS, A_VD0, S_VOFF)
        r2, r3, r1 = (S_KR2, S_KR3, A_KROW16) if which == "k" else (S_VR2, S_VR3, A_VROW16)
        base = (K_SLOT if which == "k" else V_BASE) + slot_par * SLOT_B
        skip = None
        if guarded:
            skip = self.p.fresh("dma_skip")
            out.append(mk("s_add_u32", S_TMP2, S_T, ahead, tag="salu"))
            out.append(mk("s_cmp_lt_i32", S_TMP2, A_NTWG, tag="salu"))
            out.append(mk("s_cbranch_scc0", Label(skip), tag="branch"))
        for i in range(4):
            out.append(mk("s_add_u32", M0, A_LDSW, base + i * 4096, tag="salu"))
            if i == 0:
                so = soff
                out.append(mk("s_nop", 0, tag="salu"))
            else:
                out.append(mk("s_add_u32", S_TMP, soff, (r1, r2, r3)[i - 1], tag="salu"))
                so = S_TMP
            out.append(mk("buffer_load_dwordx4", vd, rs, so, tag="dma", offen=True, lds=True))
        if guarded:
            out.append(Ins("label", (Label(skip),)))
            return [out]          # one atomic group: the guard's SCC and branch must not be interleaved with other streams
        return out

    # ------------------------------------------------------------------ one body
    def body(self, par, pv=True, s1=True, s2=True, mask_m0=False, mask_m1=False, guarded=True, name="body"):
        """B(t) with t & 1 == par.  pv: PV(t); s1: tile t+1 work (E0, M1, E1, V(t+1) reads); s2: tile t+2 work
        (K(t+2) reads, QK(t+2), M0).  Returns nothing; appends to self.p."""
        p = self.p
        self.body_id += 1
        mf = []
        mf += self.pv_mfmas(par, 0) if pv else [None] * 16
        mf += self.pv_mfmas(par, 1) if pv else [None] * 16
        mf += self.qk_mfmas(par, 0) if s2 else [None] * 16
        mf += self.qk_mfmas(par, 1) if s2 else [None] * 16
        # filler streams with their gap windows [a, b)
        streams = []
        if s1:
            streams.append((self.stream_exp(0, par ^ 1), 0.0, 48.0))
            m1 = self.stream_max(1, par ^ 1, mask_m1)
            w_m1 = 10.0 if not mask_m1 else 20.0
            streams.append((m1, 2.0, 2.0 + w_m1))
            streams.append((self.stream_exp(1, par ^ 1), 2.5 + w_m1, 64.0))
            streams.append((self.stream_vread(par ^ 1), 33.0, 63.0))
        if s2:
            streams.append((self.stream_kread(par), 0.0, 26.0))
            m0 = self.stream_max(0, par, mask_m0)
            streams.append((m0, 50.0, 63.9))
        dma = self.dma_group("k", par ^ 1, guarded, 3) + self.dma_group("v", par, guarded, 2)
        streams.append((dma, 8.0, 24.0))
        slots = [[] for _ in range(65)]
        for (lst, a, b) in streams:
            n = len(lst)
            for k, ins in enumerate(lst):
                pos = a + (b - a) * (k + 0.5) / n
                slots[int(pos)].append((pos, ins))
        for g in range(64):
            slots[g].sort(key=lambda x: x[0])
            flat = []
            for (pos, item) in slots[g]:
                flat.extend((pos, i) for i in (item if isinstance(item, list) else [item]))
            slots[g] = flat
        # emit: gap g fillers come AFTER mfma g
        for g in range(64):
            if g == 32:
                # phase boundary: all of PV(t) is issued.  Rare O rescale, then K(t+2) fragments must have landed.
                lab = p.fresh("rare_r")
                p.emit("s_cmp_lg_u32", S_FLAG, 0)
                p.emit("s_cbranch_scc1", Label(lab))
                p.label(lab + "_ret")
                self.rare.append(self.rare_rescale(lab))
                if s2:
                    p.emit("s_waitcnt", lgkmcnt=0)
            if mf[g] is not None:
                p.ins.append(mf[g])
            for (_, ins) in slots[g]:
                p.ins.append(ins)
        # end of body: DMA landed, my LDS reads done, then everybody
        p.emit("s_add_u32", S_T, S_T, 1)
        p.emit("s_add_u32", S_KOFF, S_KOFF, A_KTILE)
        p.emit("s_add_u32", S_VOFF, S_VOFF, A_VTILE)
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        p.emit("s_barrier")

    def rare_rescale(self, lab):
        r = [Ins("label", (Label(lab),))]
        r.append(mk("s_nop", 15))
        r.append(mk("s_nop", 15))
        for qb in range(2):
            skip = self.p.fresh("rr_skip")
            r.append(mk("s_bitcmp1_b32", S_FLAG, qb))
            r.append(mk("s_cbranch_scc0", Label(skip)))
            for dt in range(4):
                acc = OACC(qb, dt)
                for i in range(0, 16, 8):
                    for j in range(8):
                        r.append(mk("v_accvgpr_read_b32", TMP[j], acc[i + j]))
                    r.append(mk("s_nop", 1))
                    for j in range(8):
                        r.append(mk("v_mul_f32", TMP[j], TMP[j], FSC[qb]))
                    r.append(mk("s_nop", 1))
                    for j in range(8):
                        r.append(mk("v_accvgpr_write_b32", acc[i + j], TMP[j]))
            r.append(Ins("label", (Label(skip),)))
        r.append(mk("s_mov_b32", S_FLAG, 0))
        r.append(mk("s_nop", 7))
        r.append(mk("s_branch", Label(lab + "_ret")))
        return r

    # ------------------------------------------------------------------ whole block
    def build(self):
        p = self.p
        # ---- entry: constants, state, Q fragments, K(0)
        p.emit("s_waitcnt", vmcnt=0, lgkmcnt=0)
        for ks in range(8):
            p.emit("v_xor_b32", KR[ks], ks << 5, A_KR0)
        for dt in range(4):
            p.emit("v_xor_b32", VR[dt], dt << 6, A_VR0)
        for qb in range(2):
            for ks in range(8):
                p.emit("global_load_dwordx4", QF(qb, ks), A_Q0 if qb == 0 else A_Q1, OFF, offset=32 * ks)
        p.emit("s_mov_b32", S_T, -2)
        p.emit("s_mov_b32", S_FLAG, 0)
        p.emit("s_mov_b32", S_KOFF, 0)
        p.emit("s_lshl_b32", S_KR2, A_KROW16, 1)
        p.emit("s_add_u32", S_KR3, S_KR2, A_KROW16)
        p.emit("s_lshl_b32", S_VR2, A_VROW16, 1)
        p.emit("s_add_u32", S_VR3, S_VR2, A_VROW16)
        # K(0) -> K slot 0 (always exists)
        for i in range(4):
            p.emit("s_add_u32", M0, A_LDSW, K_SLOT + i * 4096)
            if i == 0:
                p.emit("s_nop", 0)
                so = S_KOFF
            else:
                p.emit("s_add_u32", S_TMP, S_KOFF, (A_KROW16, S_KR2, S_KR3)[i - 1])
                so = S_TMP
            p.emit("buffer_load_dwordx4", A_KD0, A_KRS, so, offen=True, lds=True)
        # B(-2) stages K(1) (= t + 3) and V(0) (= t + 2): the running offsets are those of tile t+3 / t+2
        p.emit("s_mov_b32", S_KOFF, A_KTILE)
        p.emit("s_mov_b32", S_VOFF, 0)
        p.emit("v_mov_b32", NEGINF, NEG_INF)
        for qb in range(2):
            p.emit("v_mov_b32", MREF[qb], NEG_INF)
            p.emit("v_mov_b32", MC[qb], NEG_INF)
            p.emit("v_mov_b32", LA[qb], 0)
            p.emit("v_mov_b32", LB[qb], 0)
            p.emit("v_mov_b32", FSC[qb], 1.0)
        for i in range(128):
            p.emit("v_accvgpr_write_b32", A(i), 0)
        p.emit("s_waitcnt", vmcnt=0)
        p.emit("s_barrier")

        # ---- head bodies: t = -2 (parity 0), t = -1 (parity 1)
        p.emit("s_cmp_eq_u32", A_NTW, 1)
        p.emit("s_cbranch_scc1", Label("h1m"))
        self.body(0, pv=False, s1=False, s2=True, name="H1")
        p.emit("s_cmp_eq_u32", A_NTW, 2)
        p.emit("s_cbranch_scc1", Label("h2a"))
        self.body(1, pv=False, s1=True, s2=True, name="H2")
        p.emit("s_branch", Label("main"))
        p.label("h2a")
        self.body(1, pv=False, s1=True, s2=True, mask_m0=True, name="H2a")
        p.emit("s_branch", Label("main"))
        p.label("h1m")
        self.body(0, pv=False, s1=False, s2=True, mask_m0=True, name="H1m")
        self.body(1, pv=False, s1=True, s2=False, mask_m1=True, name="H2b")

        # ---- main: fast bodies while ntw - t >= 4, then the dispatcher
        p.label("main")
        p.emit("s_sub_u32", S_NFAST, A_NTW, 3)            # number of fast bodies (t = 0 .. ntw-4), if positive
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        p.emit("s_cbranch_scc0", Label("dispatch"))
        p.emit("s_nop", 0)
        p.label("fast0")
        self.body(0, guarded=False, name="F0")
        p.emit("s_sub_u32", S_NFAST, S_NFAST, 1)
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        p.emit("s_cbranch_scc0", Label("dispatch"))
        self.body(1, guarded=False, name="F1")
        p.emit("s_sub_u32", S_NFAST, S_NFAST, 1)
        p.emit("s_cmp_gt_i32", S_NFAST, 0)
        p.emit("s_cbranch_scc1", Label("fast0"))

        p.label("dispatch")
        p.emit("s_cmp_ge_i32", S_T, A_NTWG)
        p.emit("s_cbranch_scc1", Label("epilogue"))
        p.emit("s_sub_u32", S_D, A_NTW, S_T)              # tiles left for this wave, the one PV'd next included
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
            self.body(par, pv=False, s1=False, s2=False, name="ST%d" % par)     # this wave is done: stage + sync only
            p.emit("s_branch", Label("dispatch"))
            p.label("ta_" + suffix)
            self.body(par, mask_m0=True, name="TA%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tb_" + suffix)
            self.body(par, s2=False, mask_m1=True, name="TB%d" % par)
            p.emit("s_branch", Label("dispatch"))
            p.label("tc_" + suffix)
            self.body(par, s1=False, s2=False, name="TC%d" % par)
            p.emit("s_branch", Label("dispatch"))

        # ---- epilogue: O / l -> 16 bit -> wave-private LDS image (rows of 272 B); LSE out
        p.label("epilogue")
        p.emit("s_nop", 15)
        for qb in range(2):
            lt, t, inv = TMP[8], TMP[9], T2[0]
            p.emit("v_add_f32", lt, LA[qb], LB[qb])
            p.emit("v_mov_b32", t, lt)
            p.emit("s_nop", 1)
            p.emit("v_permlane32_swap_b32", lt, t)
            p.emit("v_add_f32", lt, lt, t)
            p.emit("s_nop", 0)
            p.emit("v_rcp_f32", inv, lt)
            p.emit("v_log_f32", t, lt)
            p.emit("s_nop", 0)
            p.emit("v_add_f32", A_LSE0 if qb == 0 else A_LSE1, MC[qb], t)
            for dt in range(4):
                acc = OACC(qb, dt)
                for r4 in (0, 2):
                    for j in range(8):
                        p.emit("v_accvgpr_read_b32", TMP[j], acc[4 * r4 + j])
                    p.emit("s_nop", 0)
                    for j in range(8):
                        p.emit("v_mul_f32", TMP[j], TMP[j], inv)
                    p.emit("s_nop", 0)
                    # a0 = TMP0, a1 = TMP1 (regs 4r4+0..3), b0 = TMP2, b1 = TMP3 (regs 4r4+4..7)
                    p.emit(self.cvt, TMP[0], TMP[0], TMP[1])
                    p.emit(self.cvt, TMP[1], TMP[2], TMP[3])
                    p.emit(self.cvt, TMP[2], TMP[4], TMP[5])
                    p.emit(self.cvt, TMP[3], TMP[6], TMP[7])
                    p.emit("s_nop", 1)
                    p.emit("v_permlane32_swap_b32", TMP[0], TMP[2])
                    p.emit("v_permlane32_swap_b32", TMP[1], TMP[3])
                    p.emit("s_nop", 0)
                    # 16 bytes {x0[0], x1[0], x0[1], x1[1]} = {TMP0, TMP1, TMP2, TMP3} at row (32qb + l31), column 32dt + 8(r4 + hi)
                    p.emit("ds_write_b128", A_EPI, V(TMP[0].idx, 4), offset=32 * qb * EPI_ROWB + (32 * dt + 8 * r4) * 2)
                    p.emit("s_nop", 1)
        p.emit("s_waitcnt", lgkmcnt=0)
        # out-of-line blocks
        p.emit("s_branch", Label("end"))
        for r in self.rare:
            p.extend(r)
        p.label("end")
        return p


def render_inline(prog):
    """C string-literal lines for the asm statement; labels get the per-statement unique suffix %=."""
    lines = []
    for t in prog.text_lines():
        lines.append('"%s\\n"' % t)
    return "\n".join(lines) + "\n"


def label_text_inline(name):
    return ".Lfa2d128_%s_%%=" % name


Label.text = lambda self: label_text_inline(self.name)


def clobber_list():
    regs = ["v%d" % i for i in CLOBBER_V] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in CLOBBER_S]
    return ", ".join('"%s"' % r for r in regs + ["vcc", "scc", "memory"])


def main():
    out_dir = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    for bf16 in (False, True):
        g = Gen(bf16)
        prog = g.build()
        path = os.path.join(out_dir, "fa2_fwd_d128_%s.inc" % ("bf16" if bf16 else "f16"))
        with open(path, "w") as f:
            f.write("// GENERATED by csrc/gen/fwd_d128_gen.py — do not edit.  %d instructions.\n" % len(prog.ins))
            f.write(render_inline(prog))
        print(path, len(prog.ins), "instructions")
    with open(os.path.join(out_dir, "fa2_fwd_d128_clobbers.inc"), "w") as f:
        f.write("// GENERATED by csrc/gen/fwd_d128_gen.py — do not edit.\n")
        f.write(clobber_list() + "\n")


if __name__ == "__main__":
    main()
