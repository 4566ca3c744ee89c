"""Run the generated D = 128 backward blocks on the emulator for one workgroup and compare with dense float64 gradients
(TEST INFRASTRUCTURE; the argument set-up mirrors fa2_bwd_d128.hip.h line by line)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
import asm_emu  # noqa: E402
import bwd_d128_gen as gen  # noqa: E402
from asm_emu_harness import from_bits, to_bits  # noqa: E402
from isa import Reg  # noqa: E402

LOG2E = 1.4426950408889634
DQ = gen.DQ
_PROGS = {}
QSPLIT = False          # mirror of the shell's -DFA2_BWD_QSPLIT (programs built with opt "qsplit")
DQ_M16 = False          # the dQ pass built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dq_m16_gen.py) instead of GenDQ
DKV_M16 = False         # the dK / dV pass built on v_mfma_f32_16x16x32 (csrc/gen/bwd_dkv_m16_gen.py) instead of GenDKV


def program(kind, bf16):
    if kind == "dq" and DQ_M16:
        kind = "dq16"
    if kind == "dkv" and DKV_M16:
        kind = "dkv16"
    if (kind, bf16) not in _PROGS:
        if kind == "dq16":
            import bwd_dq_m16_gen
            _PROGS[(kind, bf16)] = bwd_dq_m16_gen.GenDQ16(bf16).build()
        elif kind == "dkv16":
            import bwd_dkv_m16_gen
            _PROGS[(kind, bf16)] = bwd_dkv_m16_gen.GenDKV16(bf16).build()
        else:
            _PROGS[(kind, bf16)] = (gen.GenDQ if kind == "dq" else gen.GenDKV)(bf16).build()
    return _PROGS[(kind, bf16)]


def dense_bwd(q, k, v, do, causal, scale, bf16):
    """float64 attention forward + backward on the rounded inputs: returns o, lse2 (log2 domain), delta, dq, dk, dv."""
    q, k, v, do = (from_bits(to_bits(t, bf16), bf16).astype(np.float64) for t in (q, k, v, do))
    s = (q @ k.T) * scale
    if causal:
        s = np.where(np.arange(k.shape[0])[None, :] > np.arange(q.shape[0])[:, None], -np.inf, s)
    mx = s.max(axis=1, keepdims=True)
    p = np.exp(s - mx)
    l = p.sum(axis=1, keepdims=True)
    p = p / l
    o = p @ v
    lse2 = (mx[:, 0] + np.log(l[:, 0])) * LOG2E
    delta = (do * o).sum(axis=1)
    dp = do @ v.T
    ds = p * (dp - delta[:, None])
    return o, lse2, delta, scale * ds @ k, scale * ds.T @ q, p.T @ do


class Bufs:
    """Global memory of the emulated launch: matrices between NaN guard bands."""

    def __init__(self, bf16):
        self.bf16, self.list, self.addr = bf16, [], 0x10000000

    def add16(self, t):
        pad = np.full(4096, 0x7fc0 if self.bf16 else 0x7e00, dtype=np.uint16)
        arr = np.concatenate([pad, to_bits(t, self.bf16).ravel(), pad]).view(np.uint8)
        return self._put(arr, pad.size * 2)

    def add32(self, t):
        pad = np.full(1024, np.nan, dtype=np.float32)
        arr = np.concatenate([pad, np.asarray(t, dtype=np.float32).ravel(), pad]).view(np.uint8)
        return self._put(arr, pad.size * 4)

    def _put(self, arr, skip):
        base = self.addr
        self.list.append((base, arr))
        self.addr += (arr.size + 0xffff) & ~0xffff
        return base + skip


def _srd(base, n, row_bytes=256):
    return np.array([base & 0xffffffff, base >> 32, (n - 1) * row_bytes + 256, 0x00020000], dtype=np.uint32)


def _pair(base):
    return np.array([base & 0xffffffff, base >> 32], dtype=np.uint32)


def dq_wave_args(w, qblk, Nq, Nkv, causal, scale, bases):
    lane = np.arange(64)
    l31, hi, pp, g1 = lane & 31, lane >> 5, lane & 15, (lane >> 4) & 1
    rb = 256
    q0 = qblk * 256
    qw0 = q0 + 64 * w
    ntiles = (Nkv + 31) // 32
    ntwg, ntw = ntiles, ntiles
    if causal:
        ntwg = min(ntiles, (min(q0 + 256, Nq) - 1) // 32 + 1)
        ntw = min(ntwg, (qw0 + 63) // 32 + 1)
    v = np.zeros((DQ.VBASE, 64), dtype=np.uint32)
    for qb in range(2):
        qrow = qw0 + 32 * qb + l31
        qr = np.minimum(qrow, Nq - 1).astype(np.int64)
        for n0 in (2, 4, 6):
            v[n0 + qb] = (qr * rb + hi * 16).astype(np.uint32)
        v[8 + qb] = (qr * 4).astype(np.uint32)
        lim_c = qrow if causal else np.full(64, 0x3fffffff)
        v[15 + qb] = (np.minimum(lim_c, Nkv - 1) - 32 * (ntw - 1) - 4 * hi).astype(np.int32).view(np.uint32)
    drow, dslot = 8 * w + (lane >> 4), lane & 15
    v[10] = v[11] = (drow * rb + ((dslot ^ (drow & 15)) << 4)).astype(np.uint32)                       # row images (K-style)
    v[12] = (drow * rb + (((((dslot >> 2) ^ (drow & 3)) << 2) | (dslot & 3)) << 4)).astype(np.uint32)  # "tr" image (V-style)
    v[13] = (l31 * 256 + ((hi ^ (l31 & 15)) << 4)).astype(np.uint32)
    v[14] = ((4 * hi + (pp >> 2)) * 256 + ((pp >> 2) << 6) + 32 * g1 + 8 * (pp & 3)).astype(np.uint32)
    v[17] = (DQ.EPI_BASE + w * 64 * DQ.EPI_ROWB + l31 * DQ.EPI_ROWB + hi * 16).astype(np.uint32)
    if DQ_M16:
        # lane = (n = lane % 16, g4 = lane / 16): operands 2..9 carry ROW0, 16 g4, Nq - 1 and the row pitches of Q / dO / O (bwd_dq_m16_gen.py)
        n16, g4 = lane & 15, lane >> 4
        v[2] = (qw0 + n16).astype(np.uint32)
        v[3] = (16 * g4).astype(np.uint32)
        v[4] = np.full(64, Nq - 1, dtype=np.uint32)
        v[5] = v[6] = v[7] = np.full(64, rb, dtype=np.uint32)
        v[8] = v[9] = 0
        v[13] = (n16 * 256 + ((g4 ^ n16) << 4)).astype(np.uint32)
        trow = 4 * g4 + (n16 >> 2)
        # the 16x16x32 body's own "tr" image (round 6): the 32-byte half of a 64-byte chunk is flipped for rows with (row >> 2) & 1 — a transposed read
        # serves rows r and r + 4 in one cycle, which the plain image keeps in the same banks
        v[12] = (drow * rb + (((((dslot >> 2) ^ (drow & 3)) << 2) | ((dslot & 3) ^ (((drow >> 2) & 1) << 1))) << 4)).astype(np.uint32)
        v[14] = (trow * 256 + ((trow & 3) << 6) + 32 * ((trow >> 2) & 1) + 8 * (n16 & 3)).astype(np.uint32)
        l0 = ((qw0 + n16) if causal else np.full(64, 0x3fff0000)) - 32 * (ntw - 1) - 4 * g4
        cap = np.full(64, Nkv - 1) - 32 * (ntw - 1) - 4 * g4
        v[15] = l0.astype(np.int32).view(np.uint32)
        v[16] = cap.astype(np.int32).view(np.uint32)
        v[17] = (DQ.EPI_BASE + w * 64 * DQ.EPI_ROWB + n16 * DQ.EPI_ROWB + g4 * 8).astype(np.uint32)
    args = {n: Reg("v", n) for n in range(DQ.N_VARGS)}
    args[18], args[19], args[20], args[21] = _pair(bases["q"]), _pair(bases["do"]), _pair(bases["o"]), _pair(bases["lse"])
    args[22], args[23] = _srd(bases["k"], Nkv), _srd(bases["v"], Nkv)
    args[24] = int(np.float32(scale * LOG2E).view(np.uint32))
    args[25] = int(np.float32(scale).view(np.uint32))
    args[26], args[27] = ntw, ntwg
    args[28] = args[29] = 32 * rb
    args[30] = args[31] = 4 * rb - 1024
    args[32] = w * 2048
    args[33] = int(np.float32(1.0).view(np.uint32))
    args["vregs"] = v
    return args


def run_dq(q, k, v, do, qblk, causal, scale=None, bf16=False, check_hazards=True):
    """One workgroup of the dQ pass on Q block qblk.  Returns (dq [rows,128] f32, delta [rows] f32, machine, reference dict)."""
    scale = 128 ** -0.5 if scale is None else scale
    Nq, Nkv = q.shape[0], k.shape[0]
    o_ref, lse_ref, delta_ref, dq_ref, dk_ref, dv_ref = dense_bwd(q, k, v, do, causal, scale, bf16)
    b = Bufs(bf16)
    bases = {"q": b.add16(q), "k": b.add16(k), "v": b.add16(v), "do": b.add16(do), "o": b.add16(o_ref), "lse": b.add32(lse_ref)}
    wa = [dq_wave_args(w, qblk, Nq, Nkv, causal, scale, bases) for w in range(4)]
    m = asm_emu.Machine(program("dq", bf16), wa, DQ.LDS_BYTES, b.list, bf16=bf16, check_hazards=check_hazards)
    for w, a in zip(m.waves, wa):
        w.v[:DQ.VBASE] = a["vregs"]
    m.run()
    rows = min(256, Nq - qblk * 256)
    img = m.lds[DQ.EPI_BASE:DQ.EPI_BASE + 4 * 64 * DQ.EPI_ROWB].reshape(256, DQ.EPI_ROWB)[:, :256].copy().view(np.uint16)
    dq = from_bits(img, bf16)[:rows]
    delta = np.empty(256, dtype=np.float32)
    for w in range(4):
        if DQ_M16:          # one output register: lane l hands over row l of the wave
            delta[64 * w:64 * w + 64] = m.waves[w].v[0].view(np.float32)
            continue
        for qb in range(2):
            delta[64 * w + 32 * qb:64 * w + 32 * qb + 32] = m.waves[w].v[qb][:32].view(np.float32)
    r0 = qblk * 256
    o16 = from_bits(to_bits(o_ref, bf16), bf16).astype(np.float64)
    do16 = from_bits(to_bits(do, bf16), bf16).astype(np.float64)
    ref = {"dq": dq_ref[r0:r0 + rows], "delta": (do16 * o16).sum(axis=1)[r0:r0 + rows]}
    return dq, delta[:rows], m, ref


def check_dq(Nq, Nkv, qblk, causal, bf16=False, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    q, k, v, do = (rng.standard_normal((n, 128)) for n in (Nq, Nkv, Nkv, Nq))
    dq, delta, m, ref = run_dq(q, k, v, do, qblk, causal, bf16=bf16)
    err = float(np.abs(dq - ref["dq"]).max())
    derr = float(np.abs(delta - ref["delta"]).max())
    if verbose:
        cyc = np.diff(np.array([0.0] + m.body_cycles))
        print("dQ Nq %d Nkv %d qblk %d causal %d bf16 %d: max|dQ-ref| %.2e (max|ref| %.2f)  max|delta-ref| %.2e  hazards %d  issued/wave %d  body cycles %s"
              % (Nq, Nkv, qblk, causal, bf16, err, np.abs(ref["dq"]).max(), derr, len(m.errors), m.waves[3].n_issued, np.round(cyc[:10]).astype(int)))
        for e in m.errors[:12]:
            print("   !", e)
    return err, derr, m, ref


if __name__ == "__main__":
    check_dq(256, 256, 0, False)


# ---------------------------------------------------------------------------------------------------------------------
KV = gen.KV


def dkv_wave_args(w, kblk, Nq, Nkv, causal, scale, bases):
    lane = np.arange(64)
    l31, hi, pp, g1 = lane & 31, lane >> 5, lane & 15, (lane >> 4) & 1
    rb = 256
    pair, role = w >> 1, w & 1
    kv0 = kblk * 128
    kvw0 = kv0 + 64 * pair
    tile0 = kv0 // 32 if causal else 0
    n = max(1, (Nq + 31) // 32 - tile0)
    v = np.zeros((KV.VBASE, 64), dtype=np.uint32)
    for kvb in range(2):
        kvrow = kvw0 + 32 * kvb + l31
        kr = np.minimum(kvrow, Nkv - 1).astype(np.int64)
        v[kvb] = (kr * rb + hi * 16).astype(np.uint32)
        lim = (kvrow - 32 * tile0 - 4 * hi) if causal else np.full(64, -(1 << 30))
        v[6 + kvb] = lim.astype(np.int32).view(np.uint32)
    drow, dslot = 8 * w + (lane >> 4), lane & 15
    v[2] = v[3] = (drow * rb + ((dslot ^ gen.f_swz(drow)) << 4)).astype(np.uint32)
    qrow0 = 8 * w
    if QSPLIT:            # the Q pieces split by role: the P side stages row quad 0 of the pair's 16 rows, the dS side quads 1..3
        qrow0 = 16 * pair + (4 if role else 0)
        drow = qrow0 + (lane >> 4)
        v[2] = (drow * rb + ((dslot ^ gen.f_swz(drow)) << 4)).astype(np.uint32)
    v[4] = (l31 * 256 + ((hi ^ gen.f_swz(l31)) << 4)).astype(np.uint32)
    i, j = pp >> 2, pp & 3
    trow = 4 * hi + i
    v[5] = (trow * 256 + (((2 * g1 + (j >> 1)) ^ gen.f_swz(trow)) << 4) + 8 * (j & 1)).astype(np.uint32)
    v[8] = (KV.P_SLOTS + pair * 8192 + lane * 16).astype(np.uint32)
    v[9] = (hi * 16 + role * 512).astype(np.uint32)
    v[10] = (lane * 4).astype(np.uint32)
    v[11] = (w * 64 * KV.EPI_ROWB + l31 * KV.EPI_ROWB + hi * 16).astype(np.uint32)
    if DKV_M16:
        # the shell's M16 branch (fa2_bwd_d128.hip.h): lane = (n = lane % 16, g = lane / 16); operands 0, 1, 7, 10 carry the own-row offsets of the
        # four 16-row KV groups, 6 the causal limit of group 0
        n16, g4 = lane & 15, lane >> 4
        for kvg, slot in enumerate((0, 1, 7, 10)):
            kr = np.minimum(kvw0 + 16 * kvg + n16, Nkv - 1).astype(np.int64)
            v[slot] = (kr * rb + 16 * g4).astype(np.uint32)
        lim = (kvw0 + n16 - 32 * tile0 - 4 * g4) if causal else np.full(64, -(1 << 30))
        v[6] = lim.astype(np.int32).view(np.uint32)
        v[2] = v[3] = (drow * rb + ((dslot ^ gen.f_swz16(drow)) << 4)).astype(np.uint32)       # (the 16x16x32 bodies' own granule swizzle: f_swz16)
        v[4] = (n16 * 256 + ((g4 ^ gen.f_swz16(n16)) << 4)).astype(np.uint32)
        tq = 4 * g4 + (n16 >> 2)
        v[5] = (tq * 256 + ((((n16 & 3) >> 1) ^ gen.f_swz16(tq)) << 4) + 8 * (n16 & 1)).astype(np.uint32)
        v[9] = (g4 * 16 + role * 512).astype(np.uint32)
        v[11] = (w * 64 * KV.EPI_ROWB + n16 * KV.EPI_ROWB + g4 * 8).astype(np.uint32)
    args = {k: Reg("v", k) for k in range(KV.N_VARGS)}
    args[12] = _pair(bases["v"] if role else bases["k"])
    args[13], args[14] = _srd(bases["q"], Nq), _srd(bases["do"], Nq)
    lb = bases["ndelta"] if role else bases["lse"]
    args[15] = np.array([lb & 0xffffffff, lb >> 32, Nq * 4, 0x00020000], dtype=np.uint32)
    args[16] = int(np.float32(scale * LOG2E).view(np.uint32))
    args[17] = int(np.float32(scale if role else 1.0).view(np.uint32))
    args[18] = n
    args[19] = args[20] = tile0 * 32 * rb
    args[21] = tile0 * 128
    args[22] = args[23] = 32 * rb
    args[24] = args[25] = 4 * rb - 1024
    args[26] = w * 2048
    args[27] = role
    args[28] = (KV.LD_BASE + role * 512) if pair == 0 else 0
    args[29] = qrow0 * 256
    args["vregs"] = v
    return args


def run_dkv(q, k, v, do, kblk, causal, scale=None, bf16=False, check_hazards=True):
    """One workgroup of the dK/dV pass on KV block kblk (128 rows).  Returns (dk, dv [rows,128] f32, machine, reference dict)."""
    scale = 128 ** -0.5 if scale is None else scale
    Nq, Nkv = q.shape[0], k.shape[0]
    o_ref, lse_ref, delta_ref, dq_ref, dk_ref, dv_ref = dense_bwd(q, k, v, do, causal, scale, bf16)
    o16 = from_bits(to_bits(o_ref, bf16), bf16).astype(np.float64)
    do16 = from_bits(to_bits(do, bf16), bf16).astype(np.float64)
    delta = (do16 * o16).sum(axis=1)
    b = Bufs(bf16)
    bases = {"q": b.add16(q), "k": b.add16(k), "v": b.add16(v), "do": b.add16(do), "lse": b.add32(lse_ref), "ndelta": b.add32(-delta)}
    wa = [dkv_wave_args(w, kblk, Nq, Nkv, causal, scale, bases) for w in range(4)]
    m = asm_emu.Machine(program("dkv", bf16), wa, KV.LDS_BYTES, b.list, bf16=bf16, check_hazards=check_hazards)
    for w, a in zip(m.waves, wa):
        w.v[:KV.VBASE] = a["vregs"]
    m.run()
    rows = min(128, Nkv - kblk * 128)
    img = m.lds[:4 * 64 * KV.EPI_ROWB].reshape(256, KV.EPI_ROWB)[:, :256].copy().view(np.uint16)
    t = from_bits(img, bf16).reshape(2, 2, 64, 128)          # [pair, role, row, d]
    dv = t[:, 0].reshape(128, 128)[:rows]
    dk = t[:, 1].reshape(128, 128)[:rows]
    r0 = kblk * 128
    return dk, dv, m, {"dk": dk_ref[r0:r0 + rows], "dv": dv_ref[r0:r0 + rows]}


def check_dkv(Nq, Nkv, kblk, causal, bf16=False, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    q, k, v, do = (rng.standard_normal((n, 128)) for n in (Nq, Nkv, Nkv, Nq))
    dk, dv, m, ref = run_dkv(q, k, v, do, kblk, causal, bf16=bf16)
    ek, ev = float(np.abs(dk - ref["dk"]).max()), float(np.abs(dv - ref["dv"]).max())
    if verbose:
        cyc = np.diff(np.array([0.0] + m.body_cycles))
        print("dKV Nq %d Nkv %d kblk %d causal %d bf16 %d: max|dK-ref| %.2e (max %.2f)  max|dV-ref| %.2e (max %.2f)  hazards %d  issued/wave %s  body cycles %s"
              % (Nq, Nkv, kblk, causal, bf16, ek, np.abs(ref["dk"]).max(), ev, np.abs(ref["dv"]).max(), len(m.errors),
                 [w.n_issued for w in m.waves[:2]], np.round(cyc[:10]).astype(int)))
        for e in m.errors[:12]:
            print("   !", e)
    return ek, ev, m, ref
