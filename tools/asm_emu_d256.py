"""Run the generated head-dim-256 forward block (csrc/gen/fwd_m16_d256_gen.py) on the emulator for one workgroup (128 Q rows) and compare with dense
attention.  TEST INFRASTRUCTURE; the argument set-up mirrors fa2_fwd_d256.hip.h line by line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
import asm_emu  # noqa: E402
import asm_emu_harness as h128  # noqa: E402  (converters, dense reference)
import fwd_m16_d256_gen as gen  # noqa: E402
from isa import Reg  # noqa: E402

LOG2E = 1.4426950408889634
D = 256
ROWS = 128               # Q rows per workgroup
DTRIM = None             # a head dim below 256 (the trimmed body, opt=trim): rows of 2 * DTRIM bytes
NKS = None               # ... and the k-steps its body runs (default: ceil(DTRIM / 32), what the launcher picks)
_PROGS = {}


def program(bf16, **cfg):
    if DTRIM is not None:
        cfg = dict(cfg, opt=tuple(cfg.get("opt", ())) + ("trim",), nks=NKS or (DTRIM + 31) // 32)
    key = (bf16, tuple(sorted(cfg.items())))
    if key not in _PROGS:
        _PROGS[key] = gen.Gen256(bf16, **cfg).build()
    return _PROGS[key]


def wave_args(w, qblk, Nq, Nkv, causal, scale, q_base, k_base, v_base, o_base, pitch=None):
    g = gen.Geo256
    d = D if DTRIM is None else DTRIM
    pitch = 2 * d if pitch is None else pitch
    lane = np.arange(64)
    n16, g4 = lane & 15, lane >> 4
    q0 = qblk * ROWS
    qw0 = q0 + 32 * w
    ntiles = (Nkv + 63) // 64
    if causal:
        qmax = min(q0 + ROWS, Nq) - 1
        ntiles = min(ntiles, qmax // 64 + 1)
    ntw = ntiles
    if causal:
        ntw = min(ntiles, (qw0 + 31) // 64 + 1)
    v = np.zeros((16, 64), dtype=np.uint32)
    v[2] = (n16 * pitch + 16 * g4).astype(np.uint32)                                   # A_Q0
    drow, dslot = 16 * w + (lane >> 5), lane & 31
    v[7] = (drow * pitch + ((dslot ^ (drow & 15)) << 4)).astype(np.uint32)              # A_KD0: K image granule ^ (row & 15)
    v[8] = (drow * pitch + (((((dslot >> 2) ^ (drow & 3)) << 2) | ((dslot & 3) ^ (((drow >> 2) & 1) << 1))) << 4)).astype(np.uint32)   # A_VD0
    v[9] = (n16 * g.ROWB + ((g4 ^ (n16 & 15)) << 4)).astype(np.uint32)                 # A_KR0
    trow = 4 * g4 + (n16 >> 2)
    v[10] = (trow * g.ROWB + ((trow & 3) << 6) + 32 * (g4 & 1) + 8 * (n16 & 3)).astype(np.uint32)   # A_VR0
    l0 = ((qw0 + n16) if causal else np.full(64, 0x3fff0000)) - 64 * (ntw - 1) - 4 * g4
    cap = np.full(64, Nkv - 1) - 64 * (ntw - 1) - 4 * g4
    v[11] = l0.astype(np.int32).view(np.uint32)
    v[12] = cap.astype(np.int32).view(np.uint32)
    v[13] = (n16 * pitch + 8 * g4).astype(np.uint32)                                   # A_OO0

    def srd(base, rows):
        return np.array([base & 0xffffffff, base >> 32, (rows - 1) * pitch + 2 * d, 0x00020000], dtype=np.uint32)

    args = {0: Reg("v", 0), 1: Reg("v", 1), 2: Reg("v", 2), 3: qw0 * pitch, 4: srd(q_base, Nq), 5: srd(k_base, Nkv), 6: srd(v_base, Nkv)}
    for n in range(7, 13):
        args[n] = Reg("v", n)
    args[13] = int(np.float32(scale * LOG2E).view(np.uint32))
    args[14], args[15] = ntw, ntiles
    args[16] = args[17] = 64 * pitch
    args[18] = args[19] = 2 * pitch - 1024
    args[20] = w * (g.SLOT_B // 4)
    args[21] = Reg("v", 13)
    args[22] = qw0 * pitch
    args[23] = args[24] = 16 * pitch
    args[25] = srd(o_base, Nq)
    args[26] = d // 8
    args["vregs"] = v
    return args


def run_block(q, k, v, qblk, causal, scale=None, bf16=False, check_hazards=True, **cfg):
    d = D if DTRIM is None else DTRIM
    scale = d ** -0.5 if scale is None else scale
    Nq, Nkv = q.shape[0], k.shape[0]
    pad = np.full(4096, 0x7e00 if not bf16 else 0x7fc0, dtype=np.uint16)
    bufs, bases = [], []
    addr = 0x10000000
    for t in (q, k, v, np.full((Nq, d), np.nan)):
        arr = np.concatenate([pad, h128.to_bits(t, bf16).ravel(), pad]).view(np.uint8).copy()
        bufs.append((addr, arr))
        bases.append(addr + pad.size * 2)
        addr += (arr.size + 0xffff) & ~0xffff
    wa = [wave_args(w, qblk, Nq, Nkv, causal, scale, *bases) for w in range(4)]
    m = asm_emu.Machine(program(bf16, **cfg), wa, gen.Geo256.LDS_BYTES, bufs, bf16=bf16, check_hazards=check_hazards)
    for w, a in zip(m.waves, wa):
        w.v[:16] = a["vregs"]
    m.run()
    rows = min(ROWS, Nq - qblk * ROWS)
    o_bits = bufs[3][1].view(np.uint16)[pad.size:pad.size + Nq * d].reshape(Nq, d)[qblk * ROWS:qblk * ROWS + rows]
    o = h128.from_bits(o_bits.copy(), bf16)
    lse = np.empty(ROWS, dtype=np.float32)
    for w in range(4):
        lse[32 * w:32 * w + 32] = m.waves[w].v[0][:32].view(np.float32)
    # nothing outside the workgroup's rows was written (guard bands and other rows still hold their fill)
    allbits = bufs[3][1].view(np.uint16)
    fill = h128.to_bits(np.array([np.nan]), bf16)[0]
    mask = np.ones(allbits.size, dtype=bool)
    lo = pad.size + qblk * ROWS * d
    mask[lo:lo + rows * d] = False
    assert (allbits[mask] == fill).all(), "a store outside the workgroup's rows"
    return o, lse[:rows].copy(), m


def check(Nq, Nkv, qblk, causal, bf16=False, seed=0, amp=1.0, verbose=True, **cfg):
    rng = np.random.default_rng(seed)
    d = D if DTRIM is None else DTRIM
    q, k, v = rng.standard_normal((Nq, d)) * amp ** 0.5, rng.standard_normal((Nkv, d)) * amp ** 0.5, rng.standard_normal((Nkv, d))
    o, lse, m = run_block(q, k, v, qblk, causal, bf16=bf16, **cfg)
    r0 = qblk * ROWS
    h128_hd = h128.HD
    o_ref, lse_ref = h128.dense(q[r0:r0 + o.shape[0]], k, v, causal, bf16=bf16, row0=r0)
    err = float(np.abs(o - o_ref).max())
    lerr = float(np.abs(lse - lse_ref).max())
    if verbose:
        print("Nq %d Nkv %d qblk %d causal %d bf16 %d: max|O-ref| %.2e  max|LSE-ref| %.2e  hazards %d  issued/wave %d"
              % (Nq, Nkv, qblk, causal, bf16, err, lerr, len(m.errors), m.waves[3].n_issued))
        for e in m.errors[:12]:
            print("   !", e)
    return err, lerr, m


if __name__ == "__main__":
    check(128, 256, 0, False)
