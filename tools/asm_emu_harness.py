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
HD = 128                # head dim of the programs under test (128 or 64)
M16 = False             # the v_mfma_f32_16x16x32 generator (csrc/gen/fwd_m16_gen.py) instead of the 32x32x16 one
DTRIM = None            # a head dim BELOW the body's (rows of 2 * DTRIM bytes; flag bits 8 .. 12 = DTRIM / 8: Gen16.trim_offsets), M16 only


def program(bf16):
    key = (bf16, HD, M16)
    if key not in _PROGS:
        if M16:
            import fwd_m16_gen
            _PROGS[key] = fwd_m16_gen.Gen16(bf16, hd=HD, opt=OPT).build()
        else:
            _PROGS[key] = gen.Gen(bf16, hd=HD, opt=OPT).build()
    return _PROGS[key]


def geo():
    return gen.Geo(HD)


def to_bits(x, bf16):
    x = np.asarray(x, dtype=np.float32)
    return (asm_emu.f32_to_bf16_bits(x) if bf16 else asm_emu.f32_to_f16_bits(x)).astype(np.uint16)


def from_bits(b, bf16):
    return asm_emu.bf16_to_f32(b) if bf16 else asm_emu.f16_to_f32(b)


def wave_args(w, qblk, Nq, Nkv, causal, scale, q_base, k_base, v_base, row_bytes=None, flags=0, nxt=None, ws_base=0):
    """The values the forward shell (fa2_fwd_d128.hip.h) hands to the asm statement, for wave w of the workgroup working on Q block
    qblk.  nxt = (qblk, Nq, q_base, k_base, v_base, Nkv) of the workgroup's next item (flags bit 1) or None."""
    g = geo()
    row_bytes = (g.ROWB if DTRIM is None else 2 * DTRIM) if row_bytes is None else row_bytes
    if DTRIM is not None:
        flags |= (DTRIM // 8) << 8
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

    def q_wave_offset(qblk_, Nq_):          # byte offset of this wave's first Q row (a wave wholly past Nq stages rows 0..63: never stored)
        r0 = qblk_ * 256 + 64 * w
        return (r0 if r0 < Nq_ else 0) * row_bytes

    for qb in range(2):
        qrow = qw0 + 32 * qb + l31
        lim_c = qrow if causal else np.full(64, 0x3fffffff)
        lim = np.minimum(lim_c, Nkv - 1) - 64 * (ntw - 1) - 4 * hi
        v[11 + qb] = lim.astype(np.int32).view(np.uint32)
    # LDS images (fa2_fwd_kernel.hip.h Geo<HD>): K granule ^ ((row / RPB) & KMASK), V 64-byte chunk ^ ((row / RPB) & VMASK)
    gran = g.ROWB // 16                                 # granules per row
    rpb = max(1, 256 // g.ROWB)
    kmask, vmask = min(gran, 16) - 1, min(g.ROWB // 64, 4) - 1
    row = (64 // 4) * w + lane // gran                  # piece 0 of this wave: RPP rows, `gran` lanes per row
    slot = lane % gran
    gk = slot ^ ((row // rpb) & kmask)
    v[7] = (row * row_bytes + gk * 16).astype(np.uint32)
    qrow = lane // gran                                 # Q: piece 0 of a 16-row group, the K image's swizzle
    v[2] = (qrow * row_bytes + ((slot ^ ((qrow // rpb) & kmask)) * 16)).astype(np.uint32)
    gv = (((slot >> 2) ^ ((row // rpb) & vmask)) << 2) | (slot & 3)
    v[8] = (row * row_bytes + gv * 16).astype(np.uint32)
    v[9] = (l31 * g.ROWB + ((hi ^ ((l31 // rpb) & kmask)) << 4)).astype(np.uint32)
    trow = 4 * hi + (pp >> 2)
    v[10] = (trow * g.ROWB + (((trow // rpb) & vmask) << 6) + 32 * g1 + 8 * (pp & 3)).astype(np.uint32)
    v[13] = (g.EPI_BASE + w * 64 * g.EPI_ROWB + l31 * g.EPI_ROWB + hi * 16).astype(np.uint32)
    if M16:
        # lane = (n = lane % 16, g4 = lane / 16).  K / Q fragment of k-step ks: row n, granule 4 ks + g4 of the swizzled row image (the asm xors ks << 6)
        n16, g4 = lane & 15, lane >> 4
        v[9] = (n16 * g.ROWB + ((g4 ^ ((n16 // rpb) & kmask)) << 4)).astype(np.uint32)
        # V^T fragment: the 16-lane group g4 addresses rows 4 g4 + (n >> 2), 8 bytes at column 4 (n & 3) of the 16-column group (chunk 0 swizzled; the asm
        # xors (dg >> 1) << 6 and adds 32 (dg & 1))
        trow = 4 * g4 + (n16 >> 2)
        v[10] = (trow * g.ROWB + (((trow // rpb) & vmask) << 6) + 8 * (n16 & 3)).astype(np.uint32)
        if "ct" in OPT or "lm" in OPT:
            # the folded bodies' own V image: the 32-byte half of a chunk is flipped for rows with (row >> 2) & 1 (conflict-free transposed reads)
            v[10] = v[10] + (32 * (g4 & 1)).astype(np.uint32)
            drow = (64 // 4) * w + lane // gran
            gv2 = (((slot >> 2) ^ ((drow // rpb) & vmask)) << 2) | ((slot & 3) ^ (((drow >> 2) & 1) << 1))
            v[8] = (drow * row_bytes + gv2 * 16).astype(np.uint32)
        # masks of the wave's last tile: row 16 qg + n keeps kv_local = 16 kg + 4 g4 + i  iff  16 kg + i <= min(L0 + 16 qg, cap)
        qrow0 = qw0 + n16
        l0 = (qrow0 if causal else np.full(64, 0x3fff0000)) - 64 * (ntw - 1) - 4 * g4
        cap = np.full(64, Nkv - 1) - 64 * (ntw - 1) - 4 * g4
        v[11] = l0.astype(np.int32).view(np.uint32)
        v[12] = cap.astype(np.int32).view(np.uint32)
        v[13] = (g.EPI_BASE + w * 64 * g.EPI_ROWB + n16 * g.EPI_ROWB + g4 * 8).astype(np.uint32)

    def srd(base, nkv):
        return np.array([base & 0xffffffff, base >> 32, (nkv - 1) * row_bytes + (g.ROWB if DTRIM is None else 2 * DTRIM), 0x00020000], dtype=np.uint32)

    def pair(base):
        return np.array([base & 0xffffffff, base >> 32], dtype=np.uint32)

    args[0], args[1] = Reg("v", 0), Reg("v", 1)
    args[2], args[3] = Reg("v", 2), q_wave_offset(qblk, Nq)
    args[4] = srd(q_base, Nq)
    args[5], args[6] = srd(k_base, Nkv), srd(v_base, Nkv)
    for n in range(7, 13):
        args[n] = Reg("v", n)
    args[13] = int(np.float32(scale * LOG2E).view(np.uint32))
    args[14], args[15] = ntw, ntiles
    args[16] = args[17] = 64 * row_bytes
    args[18] = args[19] = g.RPP * row_bytes - 1024
    args[20] = w * (g.SLOT_B // 4)
    args[21] = Reg("v", 13)
    args[22] = flags
    args[24] = 16 * row_bytes
    if nxt is not None:
        nqblk, nNq, nq_base, nk_base, nv_base, nNkv = nxt
        args[23] = q_wave_offset(nqblk, nNq)
        args[25], args[26], args[27] = srd(nq_base, nNq), srd(nk_base, nNkv), srd(nv_base, nNkv)
    else:
        args[23] = args[3]
        args[25], args[26], args[27] = args[4], args[5], args[6]
    # KV-split parts (flag bit 3): workspace tile of the part, this lane's row 64*w + l31 and half hi
    args[28] = pair(ws_base)
    args["vregs"] = v
    return args


def run_items(items, causal, scale=None, bf16=False, check_hazards=True):
    """One persistent workgroup works through `items` = [(q [Nq,128], k [Nkv,128], v [Nkv,128], qblk), ...]: the asm
    statement runs once per item, registers / LDS / loads in flight carry over, and between two statements this harness
    plays the HIP shell (reads the O tile out of the LDS image, rebinds v0..15).  Returns [(o, lse)], machine."""
    scale = (HD if DTRIM is None else DTRIM) ** -0.5 if scale is None else scale
    pad = np.full(4096, 0x7e00 if not bf16 else 0x7fc0, dtype=np.uint16)       # NaN guard bands around every matrix
    bufs, bases = [], []
    addr = 0x10000000
    for item in items:
        q, k, v = item[:3]
        b3 = []
        for t in (q, k, v):
            arr = np.concatenate([pad, to_bits(t, bf16).ravel(), pad]).view(np.uint8)
            bufs.append((addr, arr))
            b3.append(addr + pad.size * 2)
            addr += (arr.size + 0xffff) & ~0xffff
        bases.append(b3)
    # workspace tiles of KV-split parts (items with a fifth element True): f32 [256 * HD], NaN-filled
    ws_addr = {}
    for it, item in enumerate(items):
        if len(item) > 4 and item[4]:
            arr = np.full(256 * HD, np.nan, dtype=np.float32).view(np.uint8)
            bufs.append((addr, arr))
            ws_addr[it] = addr
            addr += (arr.size + 0xffff) & ~0xffff
    m = None
    outs = []
    sticky = 0           # the shell's bit 5: an item of this workgroup was redone -> the items after it start in safe mode (fa2_fwd_d128.hip.h)
    for it, item in enumerate(items):
        q, k, v, qblk = item[:4]
        Nq, Nkv = q.shape[0], k.shape[0]
        nxt = None
        if it + 1 < len(items):
            nq_, nk_, _, nqblk = items[it + 1][:4]
            nxt = (nqblk, nq_.shape[0], bases[it + 1][0], bases[it + 1][1], bases[it + 1][2], nk_.shape[0])
        flags = (1 if it > 0 else 0) | (2 if nxt is not None else 0) | sticky
        if nxt is not None:
            n_nk = (nxt[5] + 63) // 64
            if causal:
                n_nk = min(n_nk, (min(nxt[0] * 256 + 256, nxt[1]) - 1) // 64 + 1)
            flags |= 4 if n_nk >= 2 else 0
        if it in ws_addr:
            flags |= 8
        wa = [wave_args(w, qblk, Nq, Nkv, causal, scale, bases[it][0], bases[it][1], bases[it][2], flags=flags, nxt=nxt, ws_base=ws_addr.get(it, 0))
              for w in range(4)]
        if m is None:
            m = asm_emu.Machine(program(bf16), wa, geo().LDS_BYTES, bufs, bf16=bf16, check_hazards=check_hazards)
            m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16] = 0           # (the shell clears the workgroup's flag words before the first statement)
            m.redos = 0
            m.repairs = 0
            m.item_flags = []
        else:
            m.reenter(wa)
        m.allow_vm_in_flight = nxt is not None
        m.item_flags.append(flags)
        for w, a in zip(m.waves, wa):
            w.v[:24] = a["vregs"]
        m.run()
        fw = m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16].view(np.uint32).copy()
        if fw.any() and not (fw & 1).any():
            # value 2: a wave of the lm bodies repaired a tile in place (Gen16.lm_repair) — no redo, but the items after this one start in safe mode
            m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16] = 0
            m.repairs += int((fw == 2).sum())
            sticky = 16 | 32
        if m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16].any():
            # a sum-check body met a non-finite P (fwd_d128_gen.py: rare_sum): like the shell, clear the flag words and run the SAME item again in
            # safe mode (flag bit 4), nothing staged (the failed attempt's seam bodies fetched the NEXT item's Q / K / V over this item's)
            m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16] = 0
            m.redos += 1
            sticky = 16 | 32
            flags = (flags & ~1) | sticky
            wa = [wave_args(w, qblk, Nq, Nkv, causal, scale, bases[it][0], bases[it][1], bases[it][2], flags=flags, nxt=nxt, ws_base=ws_addr.get(it, 0))
                  for w in range(4)]
            m.reenter(wa)
            m.allow_vm_in_flight = nxt is not None
            for w, a in zip(m.waves, wa):
                w.v[:24] = a["vregs"]
            m.run()
            assert not m.lds[geo().FAIL_OFF:geo().FAIL_OFF + 16].any(), "the safe-mode redo raised the flag again"
        rows = min(256, Nq - qblk * 256)
        g = geo()
        if it in ws_addr:      # a part: the normalised f32 tile in the workspace, float (((dt*4 + g) * 256 + row) * 8 + 4*hi + e) <-> d = 32dt + 8g + 4hi + e
            arr = [a for (b_, a) in bufs if b_ == ws_addr[it]][0].view(np.float32).reshape(HD // 8, 256, 8)      # [4dt + g, row, 4hi + e]
            o = arr.transpose(1, 0, 2).reshape(256, HD)[:rows].copy()
        else:
            img = m.lds[g.EPI_BASE:g.EPI_BASE + 4 * 64 * g.EPI_ROWB].reshape(256, g.EPI_ROWB)[:, :g.ROWB].copy().view(np.uint16)
            o = from_bits(img, bf16)[:rows]
            if DTRIM is not None:
                assert not np.abs(o[:, DTRIM:]).any(), "the padded columns of O must be exact zeros (V's padded columns are zero-filled)"
                o = o[:, :DTRIM]
        lse = np.empty(256, dtype=np.float32)
        for w in range(4):
            if M16:         # one output register: lane l hands over row l of the wave
                lse[64 * w:64 * w + 64] = m.waves[w].v[0].view(np.float32)
                continue
            for qb in range(2):
                lse[64 * w + 32 * qb:64 * w + 32 * qb + 32] = m.waves[w].v[qb][:32].view(np.float32)
        outs.append((o, lse[:rows].copy()))
    return outs, m


def run_block(q, k, v, qblk, causal, scale=None, bf16=False, check_hazards=True):
    """q [Nq,128], k/v [Nkv,128] float arrays (rounded to the 16-bit type here).  Returns (o [rows,128] f32,
    lse [rows] f32, machine) for the rows of workgroup qblk that exist."""
    outs, m = run_items([(q, k, v, qblk)], causal, scale=scale, bf16=bf16, check_hazards=check_hazards)
    return outs[0][0], outs[0][1], m


def dense(q, k, v, causal, scale=None, bf16=False, row0=0, pre=False):
    """float64 attention on the rounded inputs, log2-domain LSE.  pre: Q is multiplied by scale*log2(e) and rounded to the
    16-bit type first (the folded-scale contract), so the comparison isolates the kernel from that one rounding."""
    q, k, v = (from_bits(to_bits(t, bf16), bf16).astype(np.float64) for t in (q, k, v))
    scale = q.shape[1] ** -0.5 if scale is None else scale
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
    q, k, v = mk((Nq, HD)), mk((Nkv, HD)), mk((Nkv, HD))
    if spike == 3:
        # growth of 121 .. 126.6 octaves over the row's reference (its tile-0 maximum), none past f32's range: the repair factor 2^-d of the sum-check
        # bodies would be 2^-127 for the largest — no normal f32, v_exp_f32 returns 0 for it and the row would be wiped — so the item must be redone
        for row, kv, grow in ((5, 200, 126.6), (40, 300, 126.3), (100, 333, 124.0), (77, 400, 121.0)):
            qr = from_bits(to_bits(q[row], bf16), bf16).astype(np.float64)
            k0 = from_bits(to_bits(k[:64], bf16), bf16).astype(np.float64)
            ref = float((k0 @ qr).max()) * HD ** -0.5 * LOG2E
            k[min(Nkv - 1, kv)] = q[row] * ((ref + grow) / (float((qr ** 2).sum()) * HD ** -0.5 * LOG2E))
    elif spike == 2:
        # a reference that has to move by 15 .. 127 log2 units late in the sweep: the in-place repair of the sum-check bodies (rare_sum), q block 0
        # (row 5) and q block 1 (row 40: the next tile's scores get their shift at the start of the next body, rare_fix), and twice in a row (row 41)
        k[min(Nkv - 1, 200)] = q[5] * 2.7
        k[min(Nkv - 1, 300)] = q[40] * 2.7
        k[min(Nkv - 1, 130)] = q[41] * 1.5
        k[min(Nkv - 1, 330)] = q[41] * 3.0
    elif spike:
        q *= 3
        k *= 3
        k[min(Nkv - 1, 200)] = q[5] * 4
        k[min(Nkv - 1, 70)] = q[40] * 2
    o, lse, m = run_block(q, k, v, qblk, causal, bf16=bf16)
    r0 = qblk * 256
    o_ref, lse_ref = dense(q[r0:r0 + o.shape[0]], k, v, causal, bf16=bf16, row0=r0, pre=("ct" in OPT))
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
