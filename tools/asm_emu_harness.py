"""Run the generated D = 128 forward block on the emulator for one workgroup and compare with dense attention
(TEST INFRASTRUCTURE; the argument set-up mirrors fa2_fwd_d128.hip.h line by line)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
import asm_emu  # noqa: E402
import fwd_d128_gen as gen  # noqa: E402
from isa import Reg  # noqa: E402

LOG2E = 1.4426950408889634
_PROGS = {}
OPT = ()                # generator options of the programs under test (default: the f32-scale body the library ships)


def program(bf16):
    if bf16 not in _PROGS:
        _PROGS[bf16] = gen.Gen(bf16, opt=OPT).build()
    return _PROGS[bf16]


def to_bits(x, bf16):
    x = np.asarray(x, dtype=np.float32)
    return (asm_emu.f32_to_bf16_bits(x) if bf16 else asm_emu.f32_to_f16_bits(x)).astype(np.uint16)


def from_bits(b, bf16):
    return asm_emu.bf16_to_f32(b) if bf16 else asm_emu.f16_to_f32(b)


def wave_args(w, qblk, Nq, Nkv, causal, scale, q_base, k_base, v_base, row_bytes=256):
    """The values fa2_fwd_d128.hip.h hands to the asm statement, for wave w of the workgroup owning Q block qblk."""
    lane = np.arange(64)
    l31, hi = lane & 31, lane >> 5
    pp, g1 = lane & 15, (lane >> 4) & 1
    q0 = qblk * 256
    qw0 = q0 + 64 * w
    ntiles = (Nkv + 63) // 64
    if causal:
        qmax = min(q0 + 256, Nq) - 1
        ntiles = min(ntiles, qmax // 64 + 1)
    ntw = ntiles
    if causal:
        ntw = min(ntiles, (qw0 + 63) // 64 + 1)
    args = {}
    v = np.zeros((24, 64), dtype=np.uint32)
    for qb in range(2):
        qrow = qw0 + 32 * qb + l31
        qr = np.minimum(qrow, Nq - 1)
        addr = q_base + qr.astype(np.int64) * row_bytes + hi * 16
        v[2 + 2 * qb] = (addr & 0xffffffff).astype(np.uint32)
        v[3 + 2 * qb] = (addr >> 32).astype(np.uint32)
        lim_c = qrow if causal else np.full(64, 0x3fffffff)
        lim = np.minimum(lim_c, Nkv - 1) - 64 * (ntw - 1) - 4 * hi
        v[10 + qb] = lim.astype(np.int32).view(np.uint32)
    row = 16 * w + (lane >> 4)
    slot = lane & 15
    gk = slot ^ (row & 15)
    v[6] = (row * row_bytes + gk * 16).astype(np.uint32)
    gv = (((slot >> 2) ^ (row & 3)) << 2) | (slot & 3)
    v[7] = (row * row_bytes + gv * 16).astype(np.uint32)
    v[8] = (l31 * 256 + ((hi ^ (l31 & 15)) << 4)).astype(np.uint32)
    v[9] = ((4 * hi + (pp >> 2)) * 256 + ((pp >> 2) << 6) + 32 * g1 + 8 * (pp & 3)).astype(np.uint32)
    v[12] = (w * 64 * gen.EPI_ROWB + l31 * gen.EPI_ROWB + hi * 16).astype(np.uint32)
    args["vregs"] = v
    args[0], args[1] = Reg("v", 0), Reg("v", 1)
    args[2], args[3] = Reg("v", 2, 2), Reg("v", 4, 2)
    nbytes = ((Nkv - 1) * row_bytes + 256)
    args[4] = np.array([k_base & 0xffffffff, k_base >> 32, nbytes, 0x00020000], dtype=np.uint32)
    args[5] = np.array([v_base & 0xffffffff, v_base >> 32, nbytes, 0x00020000], dtype=np.uint32)
    for n, r in ((6, 6), (7, 7), (8, 8), (9, 9), (10, 10), (11, 11), (20, 12)):
        args[n] = Reg("v", r)
    args[12] = int(np.float32(scale * LOG2E).view(np.uint32))
    args[13], args[14] = ntw, ntiles
    args[15] = args[16] = 64 * row_bytes
    args[17] = args[18] = 4 * row_bytes - 1024
    args[19] = w * 4096
    return args


def run_block(q, k, v, qblk, causal, scale=None, bf16=False, check_hazards=True):
    """q [Nq,128], k/v [Nkv,128] float arrays (rounded to the 16-bit type here).  Returns (o [rows,128] f32,
    lse [rows] f32, machine) for the rows of workgroup qblk that exist."""
    Nq, Nkv = q.shape[0], k.shape[0]
    scale = 128 ** -0.5 if scale is None else scale
    qb_, kb_, vb_ = (to_bits(t, bf16) for t in (q, k, v))
    pad = np.full(4096, 0x7e00 if not bf16 else 0x7fc0, dtype=np.uint16)       # NaN guard bands around every matrix
    bufs, bases = [], []
    addr = 0x10000000
    for b in (qb_, kb_, vb_):
        arr = np.concatenate([pad, b.ravel(), pad]).view(np.uint8)
        bufs.append((addr, arr))
        bases.append(addr + pad.size * 2)
        addr += (arr.size + 0xffff) & ~0xffff
    wa = []
    for w in range(4):
        a = wave_args(w, qblk, Nq, Nkv, causal, scale, bases[0], bases[1], bases[2])
        wa.append(a)
    m = asm_emu.Machine(program(bf16), wa, gen.LDS_BYTES, bufs, bf16=bf16, check_hazards=check_hazards)
    for w, a in zip(m.waves, wa):
        w.v[:24] = a["vregs"]
    m.run()
    rows = min(256, Nq - qblk * 256)
    img = m.lds[:4 * 64 * gen.EPI_ROWB].reshape(256, gen.EPI_ROWB)[:, :256].copy().view(np.uint16)
    o = from_bits(img, bf16)[:rows]
    lse = np.empty(256, dtype=np.float32)
    for w in range(4):
        for qb in range(2):
            lse[64 * w + 32 * qb:64 * w + 32 * qb + 32] = m.waves[w].v[qb][:32].view(np.float32)
    return o, lse[:rows], m


def dense(q, k, v, causal, scale=None, bf16=False, row0=0, pre=False):
    """float64 attention on the rounded inputs, log2-domain LSE.  pre: Q is multiplied by scale*log2(e) and rounded to the
    16-bit type first (the folded-scale contract), so the comparison isolates the kernel from that one rounding."""
    q, k, v = (from_bits(to_bits(t, bf16), bf16).astype(np.float64) for t in (q, k, v))
    scale = 128 ** -0.5 if scale is None else scale
    if pre:
        c = np.float32(scale * LOG2E)
        qs = from_bits(to_bits((q.astype(np.float32) * c).astype(np.float32), bf16), bf16).astype(np.float64)
        s = (qs @ k.T) / LOG2E
    else:
        s = (q @ k.T) * scale
    if causal:
        rows = row0 + np.arange(q.shape[0])
        s = np.where(np.arange(k.shape[0])[None, :] > rows[:, None], -np.inf, s)
    mx = s.max(axis=1, keepdims=True)
    p = np.exp(s - mx)
    l = p.sum(axis=1, keepdims=True)
    return (p / l) @ v, (mx[:, 0] + np.log(l[:, 0])) * LOG2E


def check(Nq, Nkv, qblk, causal, bf16=False, seed=0, kind="randn", spike=False, verbose=True):
    rng = np.random.default_rng(seed)
    mk = (lambda s: rng.standard_normal(s)) if kind == "randn" else (lambda s: rng.random(s))
    q, k, v = mk((Nq, 128)), mk((Nkv, 128)), mk((Nkv, 128))
    if spike:
        q *= 3
        k *= 3
        k[min(Nkv - 1, 200)] = q[5] * 4
        k[min(Nkv - 1, 70)] = q[40] * 2
    o, lse, m = run_block(q, k, v, qblk, causal, bf16=bf16)
    r0 = qblk * 256
    o_ref, lse_ref = dense(q[r0:r0 + o.shape[0]], k, v, causal, bf16=bf16, row0=r0, pre=("pre" in OPT or "ct" in OPT))
    err = float(np.abs(o - o_ref).max())
    lerr = float(np.abs(lse - lse_ref).max())
    if verbose:
        cyc = np.diff(np.array([0.0] + m.body_cycles))
        print("Nq %d Nkv %d qblk %d causal %d bf16 %d: max|O-ref| %.2e  max|LSE-ref| %.2e  hazards %d  issued/wave %d  body cycles %s"
              % (Nq, Nkv, qblk, causal, bf16, err, lerr, len(m.errors), m.waves[3].n_issued, np.round(cyc[:8]).astype(int)))
        for e in m.errors[:12]:
            print("   !", e)
    return err, lerr, m


if __name__ == "__main__":
    check(256, 256, 0, False)
