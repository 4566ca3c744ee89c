"""Functional emulator for the instruction subset csrc/gen/fwd_d128_gen.py emits (TEST INFRASTRUCTURE).

It interprets the generator's instruction objects — the very list that is rendered into the shipped inline-asm
body — for the four waves of one workgroup on numpy, so the hand-written stream can be checked against the oracle on
a CPU: register allocation, LDS addressing and swizzles, MFMA fragment bindings, the software pipeline, masks, the
rescale branches.  Besides the values it models what the hardware does NOT interlock, and reports violations:
  * loads are asynchronous: destination registers / LDS bytes change only when an s_waitcnt retires them; reading a
    register with a load in flight is an error;
  * MFMA results are "ready" only ~64 modelled cycles after issue (other MFMAs may chain on them, nothing else may
    read them earlier);
  * VALU -> v_permlane (2 wait states), VALU -> MFMA operand (2), transcendental -> VALU (1), M0 write -> LDS-DMA (1);
  * LDS data races between waves inside one barrier epoch (write vs read by another wave, either order).
A crude issue-cycle model gives a per-body cycle estimate (for comparing schedules, not a prediction).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd", "csrc", "gen"))
from isa import Arg, Label, Neg, Reg, Sym  # noqa: E402

NLANE = 64
LANES = np.arange(NLANE)


class EmuError(Exception):
    pass


def f16_to_f32(bits16):
    return bits16.astype(np.uint16).view(np.float16).astype(np.float32)


def bf16_to_f32(bits16):
    return (bits16.astype(np.uint32) << 16).view(np.float32)


def f32_to_f16_bits(x):
    with np.errstate(over="ignore"):
        return x.astype(np.float16).view(np.uint16).astype(np.uint32)


def f32_to_bf16_bits(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    nan = np.isnan(x)
    r = np.where(nan, 0x7fc0, r)
    return (r & 0xffff).astype(np.uint32)


class Wave:
    def __init__(self, wid, args):
        self.wid = wid
        self.v = np.zeros((256, NLANE), dtype=np.uint32)
        self.a = np.zeros((256, NLANE), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint32)
        self.vcc = np.zeros(NLANE, dtype=bool)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.args = args                    # Arg n -> numpy array(s) / scalar
        self.lgkm = []                      # in-order queue of pending LDS reads / writes: closures
        self.vm = []                        # pending VMEM: closures
        self.inflight = {}                  # (kind, idx) -> count of loads in flight
        self.cycle = 0.0
        self.mfma_free = 0.0                # matrix pipe free at
        self.mfma_ready = {}                # (kind, idx) -> cycle the MFMA result may be read by non-MFMA
        self.issue_idx = 0                  # wait-state counter
        self.last_valu_write = {}           # (kind, idx) -> issue_idx
        self.last_trans_write = {}
        self.last_dot_write = {}            # (kind, idx) -> issue_idx of a v_dot2* write (3 wait states before another kind of VALU may read it)
        self.last_m0_write = -100
        self.at_barrier = False
        self.done = False
        self.n_issued = 0


class Machine:
    def __init__(self, prog, wave_args, lds_bytes, gmem, bf16=False, check_hazards=True):
        self.ins = prog.ins
        self.labels = {}
        for i, ins in enumerate(self.ins):
            if ins.op == "label":
                self.labels[ins.ops[0].name] = i
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.lds_w_epoch = np.full(lds_bytes, -1, dtype=np.int32)
        self.lds_w_wave = np.full(lds_bytes, -1, dtype=np.int8)
        self.lds_r_epoch = np.full(lds_bytes, -1, dtype=np.int32)
        self.lds_r_wave = np.full(lds_bytes, -1, dtype=np.int8)      # -2: several waves
        self.epoch = 0
        self.gmem = gmem                     # list of (base, numpy uint8 array)
        self.waves = [Wave(w, a) for w, a in enumerate(wave_args)]
        self.bf16 = bf16
        self.check = check_hazards
        self.errors = []
        self.body_cycles = []

    # ------------------------------------------------------------ memory
    def _gfind(self, addr, n):
        for base, arr in self.gmem:
            if base <= addr and addr + n <= base + arr.size:
                return arr, addr - base
        raise EmuError("global access out of every buffer: 0x%x (+%d)" % (addr, n))

    def gread(self, addr, n):
        arr, off = self._gfind(addr, n)
        return arr[off:off + n]

    def err(self, w, msg):
        self.errors.append("wave %d pc %d (%s): %s" % (w.wid, w.pc, self.ins[w.pc].text()[:60], msg))
        if len(self.errors) > 30:
            raise EmuError("too many errors:\n" + "\n".join(self.errors))

    # ------------------------------------------------------------ operand access
    def regfile(self, w, kind):
        return w.v if kind == "v" else w.a

    def resolve(self, w, o):
        """Arg -> concrete (Reg or scalar value)."""
        if isinstance(o, Arg):
            return w.args[o.n]
        return o

    def rd32(self, w, o, reader="valu"):
        """32-bit source operand -> uint32[64] (vector) or broadcast scalar."""
        o = self.resolve(w, o)
        if isinstance(o, Neg):
            x = self.rd32(w, o.reg, reader)
            return (x.view(np.float32) * np.float32(-1.0)).view(np.uint32) if False else (x ^ np.uint32(0x80000000))
        if isinstance(o, Reg):
            if o.kind == "s":
                return np.full(NLANE, w.s[o.idx], dtype=np.uint32)
            self.check_read(w, o.kind, o.idx, 1, reader)
            return self.regfile(w, o.kind)[o.idx].copy()
        if isinstance(o, Sym):
            if o.name == "m0":
                return np.full(NLANE, w.m0, dtype=np.uint32)
            raise EmuError("bad symbol source " + o.name)
        if isinstance(o, float):
            return np.full(NLANE, np.float32(o).view(np.uint32), dtype=np.uint32)
        if isinstance(o, (int, np.integer)):
            return np.full(NLANE, np.uint32(int(o) & 0xffffffff), dtype=np.uint32)
        if isinstance(o, np.ndarray):
            return o.astype(np.uint32)
        raise EmuError("bad operand %r" % (o,))

    def rdf(self, w, o, reader="valu"):
        return self.rd32(w, o, reader).view(np.float32)

    def rds(self, w, o):
        o = self.resolve(w, o)
        if isinstance(o, Reg):
            assert o.kind == "s", o
            return int(w.s[o.idx])
        if isinstance(o, Sym) and o.name == "m0":
            return int(w.m0)
        if isinstance(o, float):
            return int(np.float32(o).view(np.uint32))
        return int(o) & 0xffffffff

    def check_read(self, w, kind, idx, n, reader):
        if not self.check:
            return
        for i in range(idx, idx + n):
            key = (kind, i)
            if w.inflight.get(key, 0) > 0:
                self.err(w, "%s%d read while a load into it is in flight" % (kind, i))
            if reader != "mfma_c":
                rdy = w.mfma_ready.get(key)
                if rdy is not None and w.cycle < rdy and reader != "mfma_c":
                    self.err(w, "%s%d read %.0f cycles before the MFMA result is ready (%s)" % (kind, i, rdy - w.cycle, reader))
            if reader == "permlane":
                lw = w.last_valu_write.get(key, -100)
                if w.issue_idx - lw < 3:
                    self.err(w, "v_permlane reads %s%d %d wait states after a VALU write (needs 2)" % (kind, i, w.issue_idx - lw - 1))
            if reader in ("mfma_ab", "mfma_c"):
                lw = w.last_valu_write.get(key, -100)
                if w.issue_idx - lw < 3:
                    self.err(w, "MFMA reads %s%d %d wait states after a VALU write (needs 2)" % (kind, i, w.issue_idx - lw - 1))
            if reader in ("valu", "permlane", "mfma_ab", "mfma_c"):
                lw = w.last_dot_write.get(key, -100)
                if w.issue_idx - lw < 4:
                    self.err(w, "%s%d read %d wait states after a DOT instruction wrote it (needs 3; not interlocked on gfx950)" % (kind, i, w.issue_idx - lw - 1))
            if reader == "valu":
                lw = w.last_trans_write.get(key, -100)
                if w.issue_idx - lw < 2:
                    self.err(w, "VALU reads %s%d right after a transcendental wrote it (needs 1 wait state)" % (kind, i))

    def wr32(self, w, o, val, writer="valu"):
        o = self.resolve(w, o)
        assert isinstance(o, Reg), o
        val = np.asarray(val)
        if val.dtype == np.float32:
            val = val.view(np.uint32)
        if o.kind == "s":
            w.s[o.idx] = np.uint32(int(val) & 0xffffffff)
            return
        key = (o.kind, o.idx)
        if self.check and w.inflight.get(key, 0) > 0:
            self.err(w, "%s%d written while a load into it is in flight" % key)
        self.regfile(w, o.kind)[o.idx] = val.astype(np.uint32)
        w.mfma_ready.pop(key, None)
        w.last_dot_write.pop(key, None)
        if writer == "dot":
            w.last_valu_write[key] = w.issue_idx
            w.last_dot_write[key] = w.issue_idx
        if writer == "valu":
            w.last_valu_write[key] = w.issue_idx
            w.last_trans_write.pop(key, None)
        elif writer == "trans":
            w.last_valu_write[key] = w.issue_idx
            w.last_trans_write[key] = w.issue_idx

    # ------------------------------------------------------------ LDS with race tracking
    def lds_read(self, w, addrs, nbytes):
        """addrs: int array [64]; returns uint8 [64, nbytes]"""
        idx = addrs[:, None].astype(np.int64) + np.arange(nbytes)[None, :]
        if idx.max() >= self.lds.size or idx.min() < 0:
            raise EmuError("LDS read out of range: %d" % idx.max())
        flat = idx.ravel()
        bad = (self.lds_w_epoch[flat] == self.epoch) & (self.lds_w_wave[flat] != w.wid)
        if self.check and bad.any():
            self.err(w, "LDS read of bytes another wave wrote in the same barrier epoch (addr %d)" % flat[np.argmax(bad)])
        same = self.lds_r_epoch[flat] == self.epoch
        self.lds_r_wave[flat] = np.where(same & (self.lds_r_wave[flat] != w.wid), -2, w.wid)
        self.lds_r_epoch[flat] = self.epoch
        return self.lds[idx]

    def lds_write(self, w, addrs, data):
        nbytes = data.shape[1]
        idx = addrs[:, None].astype(np.int64) + np.arange(nbytes)[None, :]
        if idx.max() >= self.lds.size or idx.min() < 0:
            raise EmuError("LDS write out of range: %d" % idx.max())
        flat = idx.ravel()
        bad = (self.lds_r_epoch[flat] == self.epoch) & (self.lds_r_wave[flat] != w.wid)
        if self.check and bad.any():
            self.err(w, "LDS write over bytes another wave read in the same barrier epoch (addr %d)" % flat[np.argmax(bad)])
        bad = (self.lds_w_epoch[flat] == self.epoch) & (self.lds_w_wave[flat] != w.wid)
        if self.check and bad.any():
            self.err(w, "LDS write over bytes another wave wrote in the same barrier epoch (addr %d)" % flat[np.argmax(bad)])
        self.lds[idx] = data
        self.lds_w_epoch[flat] = self.epoch
        self.lds_w_wave[flat] = w.wid

    # ------------------------------------------------------------ helpers
    def mark_inflight(self, w, reg, n, delta):
        for i in range(reg.idx, reg.idx + n):
            key = (reg.kind, i)
            w.inflight[key] = w.inflight.get(key, 0) + delta

    def frag_to_f32(self, w, reg, reader):
        """4 consecutive 32-bit regs -> float32 [64 lanes, 8 elements]"""
        reg = self.resolve(w, reg)
        self.check_read(w, reg.kind, reg.idx, 4, reader)
        rf = self.regfile(w, reg.kind)
        words = rf[reg.idx:reg.idx + 4]                      # [4, 64]
        lo = (words & 0xffff).astype(np.uint16)
        hi = (words >> 16).astype(np.uint16)
        el = np.empty((NLANE, 8), dtype=np.float32)
        conv = bf16_to_f32 if self.bf16 else f16_to_f32
        for i in range(4):
            el[:, 2 * i] = conv(lo[i])
            el[:, 2 * i + 1] = conv(hi[i])
        return el

    # ------------------------------------------------------------ one instruction
    def cost(self, ins):
        t = ins.tag
        op = ins.op
        if op.startswith("v_mfma"):
            return 4.0
        if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32"):
            return 8.0
        if op.startswith("v_"):
            return 4.5
        if op.startswith("ds_"):
            return 4.5
        if op.startswith("buffer_load"):
            return 40.0
        if op.startswith("global_load") or op.startswith("global_store"):
            return 8.0
        if op == "s_nop":
            return 4.0 * (int(ins.ops[0]) + 1)
        return 2.0

    def step(self, w):
        ins = self.ins[w.pc]
        op = ins.op
        nxt = w.pc + 1
        if op == "label" or op == "raw":       # raw = assembler directive (alignment)
            w.pc = nxt
            return
        w.n_issued += 1
        ops = ins.ops
        R = lambda i: self.resolve(w, ops[i])   # noqa: E731
        if op.startswith("buffer_") and not 0 <= ins.mods.get("offset", 0) <= 0xfff:
            raise EmuError("MUBUF offset does not fit 12 bits (the assembler truncates it silently): %s" % ins.text())
        if op.startswith("ds_") and not 0 <= ins.mods.get("offset", 0) <= 0xffff:
            raise EmuError("DS offset does not fit 16 bits: %s" % ins.text())
        if op.startswith("v_mfma_f32_16x16x32"):
            # A[m][k]: lane l holds m = l % 16, k = 8 (l / 16) .. + 7;  B[k][n]: n = l % 16, the same k;  D[m][n]: n = l % 16, m = 4 (l / 16) + i (4 registers)
            start = max(w.cycle, w.mfma_free)
            w.cycle = start
            dst = R(0)
            a = self.frag_to_f32(w, ops[1], "mfma_ab")
            b = self.frag_to_f32(w, ops[2], "mfma_ab")
            Am = np.zeros((16, 32), dtype=np.float32)
            Bm = np.zeros((32, 16), dtype=np.float32)
            for l in range(NLANE):
                Am[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
                Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
            with np.errstate(invalid="ignore", over="ignore"):
                D = Am.astype(np.float64) @ Bm.astype(np.float64)
            c = ops[3]
            rf = self.regfile(w, dst.kind)
            out = np.empty((4, NLANE), dtype=np.float32)
            for r in range(4):
                out[r] = D[4 * (LANES >> 4) + r, LANES & 15].astype(np.float32)
            if not (isinstance(c, int) and c == 0):
                cr = self.resolve(w, c)
                self.check_read(w, cr.kind, cr.idx, 4, "mfma_c")
                with np.errstate(invalid="ignore", over="ignore"):
                    out = (out.astype(np.float64) + self.regfile(w, cr.kind)[cr.idx:cr.idx + 4].view(np.float32).astype(np.float64)).astype(np.float32)
            for r in range(4):
                key = (dst.kind, dst.idx + r)
                if self.check and w.inflight.get(key, 0) > 0:
                    self.err(w, "MFMA writes %s%d while a load into it is in flight" % key)
                rf[dst.idx + r] = out[r].view(np.uint32)
                w.mfma_ready[key] = start + 40.0          # 4 passes: a dependent MFMA / a VALU read this long after the issue
                w.last_valu_write.pop(key, None)
            w.mfma_free = start + 16.0
            w.cycle = start + 4.0
            w.issue_idx += 1
            w.pc = nxt
            return
        if op.startswith("v_mfma"):
            start = max(w.cycle, w.mfma_free)
            w.cycle = start
            dst = R(0)
            a = self.frag_to_f32(w, ops[1], "mfma_ab")       # [lane, 8]: row = lane&31, k = 8*(lane>>5)+j
            b = self.frag_to_f32(w, ops[2], "mfma_ab")
            Am = np.zeros((32, 16), dtype=np.float32)
            Bm = np.zeros((16, 32), dtype=np.float32)
            for l in range(NLANE):
                Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
                Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b[l]
            with np.errstate(invalid="ignore", over="ignore"):
                D = Am.astype(np.float64) @ Bm.astype(np.float64)
            c = ops[3]
            rf = self.regfile(w, dst.kind)
            out = np.empty((16, NLANE), dtype=np.float32)
            n_idx = LANES & 31
            for r in range(16):
                m_idx = (r & 3) + 8 * (r >> 2) + 4 * (LANES >> 5)
                out[r] = D[m_idx, n_idx].astype(np.float32)
            if not (isinstance(c, int) and c == 0):
                cr = self.resolve(w, c)
                self.check_read(w, cr.kind, cr.idx, 16, "mfma_c")
                with np.errstate(invalid="ignore", over="ignore"):
                    out = (out.astype(np.float64) + self.regfile(w, cr.kind)[cr.idx:cr.idx + 16].view(np.float32).astype(np.float64)).astype(np.float32)
            for r in range(16):
                key = (dst.kind, dst.idx + r)
                if self.check and w.inflight.get(key, 0) > 0:
                    self.err(w, "MFMA writes %s%d while a load into it is in flight" % key)
                rf[dst.idx + r] = out[r].view(np.uint32)
                w.mfma_ready[key] = start + 64.0
                w.last_valu_write.pop(key, None)
            w.mfma_free = start + 32.0
            w.cycle = start + 4.0
            w.issue_idx += 1
            w.pc = nxt
            return
        # generic cost / wait-state accounting
        w.cycle += self.cost(ins)
        w.issue_idx += (int(ops[0]) + 1) if op == "s_nop" else 1

        if op == "s_nop":
            pass
        elif op == "s_waitcnt":
            if "lgkmcnt" in ins.mods:
                n = ins.mods["lgkmcnt"]
                while len(w.lgkm) > n:
                    w.lgkm.pop(0)()
                w.cycle += 0.0
            if "vmcnt" in ins.mods:
                n = ins.mods["vmcnt"]
                while len(w.vm) > n:
                    w.vm.pop(0)()
        elif op == "s_barrier":
            w.at_barrier = True
            w.pc = nxt
            return
        elif op in ("s_mov_b32",):
            d = R(0)
            val = self.rds(w, ops[1])
            if isinstance(d, Sym):
                w.m0 = val
                w.last_m0_write = w.issue_idx
            else:
                w.s[d.idx] = val
        elif op == "s_mul_i32":
            d = R(0)
            w.s[d.idx] = np.uint32((self.rds(w, ops[1]) * self.rds(w, ops[2])) & 0xffffffff)
        elif op in ("s_add_u32", "s_sub_u32", "s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
            x, y = self.rds(w, ops[1]), self.rds(w, ops[2])
            if op == "s_add_u32":
                r = x + y
                w.scc = int(r > 0xffffffff)
            elif op == "s_sub_u32":
                r = x - y
                w.scc = int(y > x)
            elif op == "s_lshl_b32":
                r = x << (y & 31)
            elif op == "s_lshr_b32":
                r = x >> (y & 31)
            elif op == "s_and_b32":
                r = x & y
            else:
                r = x | y
            r &= 0xffffffff
            if op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
                w.scc = int(r != 0)
            d = R(0)
            if isinstance(d, Sym):
                assert d.name == "m0"
                w.m0 = r
                w.last_m0_write = w.issue_idx
            else:
                w.s[d.idx] = r
        elif op.startswith("s_cmp_"):
            x, y = self.rds(w, ops[0]), self.rds(w, ops[1])
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            w.scc = int({"s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_lt_i32": sx < sy, "s_cmp_gt_i32": sx > sy,
                         "s_cmp_ge_i32": sx >= sy, "s_cmp_le_i32": sx <= sy, "s_cmp_lt_u32": x < y, "s_cmp_ge_u32": x >= y, "s_cmp_gt_u32": x > y,
                         "s_cmp_le_u32": x <= y}[op])
        elif op == "s_cselect_b32":
            d = R(0)
            w.s[d.idx] = np.uint32(self.rds(w, ops[1]) if w.scc else self.rds(w, ops[2]))
        elif op == "s_bitcmp1_b32":
            w.scc = int((self.rds(w, ops[0]) >> (self.rds(w, ops[1]) & 31)) & 1)
        elif op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccnz", "s_cbranch_vccz"):
            take = {"s_branch": True, "s_cbranch_scc0": w.scc == 0, "s_cbranch_scc1": w.scc == 1,
                    "s_cbranch_vccnz": bool(w.vcc.any()), "s_cbranch_vccz": not w.vcc.any()}[op]
            if take:
                nxt = self.labels[ops[0].name]
                w.cycle += 16
        elif op in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):      # with an all-ones mask: lanes below this one in the low / high half, + src1
            lane = np.arange(64, dtype=np.uint32)
            assert int(self.rd32(w, ops[1])[0]) == 0xffffffff
            cnt = np.minimum(lane, 32) if op.startswith("v_mbcnt_lo") else np.maximum(lane, 32) - 32
            self.wr32(w, ops[0], (cnt + self.rd32(w, ops[2])).astype(np.uint32))
        elif op == "v_mov_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]))
        elif op == "v_subrev_u32":
            self.wr32(w, ops[0], (self.rd32(w, ops[2]).astype(np.int64) - self.rd32(w, ops[1]).astype(np.int64)).astype(np.uint32))
        elif op == "v_fma_mix_f32":
            # src0 is an f16 half (op_sel picks it), src1 f32, src2 the constant 0: the only form the generators emit
            assert ins.mods.get("op_sel_hi") == "[1,0,0]" and isinstance(ops[3], int) and ops[3] == 0
            half = 16 if ins.mods.get("op_sel") == "[1,0,0]" else 0
            x = f16_to_f32(((self.rd32(w, ops[1]) >> half) & 0xffff).astype(np.uint16))
            with np.errstate(invalid="ignore", over="ignore"):
                self.wr32(w, ops[0], (x.astype(np.float64) * self.rdf(w, ops[2]).astype(np.float64)).astype(np.float32))
        elif op in ("v_fma_mixlo_f16", "v_fma_mixhi_f16"):
            # f16 half of src0 (op_sel) x f32 src1 + 0: the exact product (11 x 24 bits fit a double) rounded ONCE to f16 — a fused operation on the hardware
            # (round 6: the emulator rounded via f32 until the GPU disagreed with it by one ulp on rare elements at unlucky scales)
            assert ins.mods.get("op_sel_hi") == "[1,0,0]" and isinstance(ops[3], int) and ops[3] == 0
            half = 16 if ins.mods.get("op_sel") == "[1,0,0]" else 0
            x = f16_to_f32(((self.rd32(w, ops[1]) >> half) & 0xffff).astype(np.uint16))
            with np.errstate(invalid="ignore", over="ignore"):
                r = (x.astype(np.float64) * self.rdf(w, ops[2]).astype(np.float64)).astype(np.float16).view(np.uint16).astype(np.uint32)
            old = self.rd32(w, ops[0])
            self.wr32(w, ops[0], ((old & np.uint32(0xffff0000)) | r) if op == "v_fma_mixlo_f16" else ((old & np.uint32(0x0000ffff)) | (r << 16)))
        elif op == "v_add_u32":
            self.wr32(w, ops[0], (self.rd32(w, ops[1]).astype(np.uint64) + self.rd32(w, ops[2]).astype(np.uint64)).astype(np.uint32))
        elif op == "v_cvt_f32_u32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]).astype(np.float32))
        elif op == "v_lshl_add_u32":
            self.wr32(w, ops[0], ((self.rd32(w, ops[1]).astype(np.uint64) << np.uint64(int(ops[2]) & 31)) + self.rd32(w, ops[3])).astype(np.uint32) & np.uint32(0xffffffff))
        elif op == "v_sub_u32":
            self.wr32(w, ops[0], (self.rd32(w, ops[1]).astype(np.int64) - self.rd32(w, ops[2]).astype(np.int64)).astype(np.uint32))
        elif op == "v_max_u32":
            self.wr32(w, ops[0], np.maximum(self.rd32(w, ops[1]), self.rd32(w, ops[2])))
        elif op == "v_pk_mul_f16":      # two f16 products, each rounded to f16 (denormals kept)
            a_, b_ = self.rd32(w, ops[1]), self.rd32(w, ops[2])
            out = np.zeros(64, dtype=np.uint32)
            for half in (0, 16):
                x = f16_to_f32(((a_ >> half) & 0xffff).astype(np.uint16))
                y = f16_to_f32(((b_ >> half) & 0xffff).astype(np.uint16))
                with np.errstate(invalid="ignore", over="ignore"):
                    out |= f32_to_f16_bits((x.astype(np.float64) * y.astype(np.float64)).astype(np.float32)).astype(np.uint32) << half
            self.wr32(w, ops[0], out)
        elif op in ("v_pk_add_f32", "v_pk_fma_f32"):
            d, a_, b_ = R(0), R(1), R(2)
            neg2 = "1]" in str(ins.mods.get("neg_lo", ""))
            for h in range(2):
                x, y = self.rdf(w, a_[h]), self.rdf(w, b_[h])
                with np.errstate(invalid="ignore", over="ignore"):
                    if op == "v_pk_add_f32":
                        r = x + y
                    else:
                        z = self.rdf(w, R(3)[h])
                        r = (x.astype(np.float64) * y.astype(np.float64) + (-z if neg2 else z).astype(np.float64)).astype(np.float32)
                self.wr32(w, d[h], r.astype(np.float32))
        elif op == "s_memtime":
            d = R(0)
            w.s[d.idx] = np.uint32(int(w.cycle) & 0xffffffff)
            w.s[d.idx + 1] = 0
            w.lgkm.append(lambda: None)
        elif op == "v_and_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]) & self.rd32(w, ops[2]))
        elif op == "v_lshrrev_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[2]) >> (self.rd32(w, ops[1]) & 31))
        elif op == "v_lshlrev_b32":
            self.wr32(w, ops[0], (self.rd32(w, ops[2]).astype(np.uint64) << (self.rd32(w, ops[1]) & 31).astype(np.uint64)).astype(np.uint32))
        elif op == "v_lshl_or_b32":
            x = (self.rd32(w, ops[1]).astype(np.uint64) << (self.rd32(w, ops[2]) & 31).astype(np.uint64)).astype(np.uint32)
            self.wr32(w, ops[0], x | self.rd32(w, ops[3]))
        elif op == "v_cvt_f32_f16":
            self.wr32(w, ops[0], f16_to_f32((self.rd32(w, ops[1]) & 0xffff).astype(np.uint16)))
        elif op == "v_cvt_f16_f32":
            old = self.regfile(w, R(0).kind)[R(0).idx]
            self.wr32(w, ops[0], (old & np.uint32(0xffff0000)) | f32_to_f16_bits(self.rdf(w, ops[1])))
        elif op == "v_cmp_eq_u32":
            w.vcc = self.rd32(w, ops[1]) == self.rd32(w, ops[2])
        elif op == "v_cmp_gt_u32":
            w.vcc = self.rd32(w, ops[1]) > self.rd32(w, ops[2])
        elif op == "v_xor_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]) ^ self.rd32(w, ops[2]))
        elif op in ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32"):
            x, y = self.rdf(w, ops[1]), self.rdf(w, ops[2])
            with np.errstate(invalid="ignore", over="ignore"):
                r = {"v_add_f32": x + y, "v_sub_f32": x - y, "v_mul_f32": x * y}.get(op)
                if op == "v_max_f32":
                    r = np.fmax(x, y)
            self.wr32(w, ops[0], r.astype(np.float32))
        elif op == "v_max3_f32":
            r = np.fmax(np.fmax(self.rdf(w, ops[1]), self.rdf(w, ops[2])), self.rdf(w, ops[3]))
            self.wr32(w, ops[0], r.astype(np.float32))
        elif op == "v_fma_f32":
            x, y, z = self.rdf(w, ops[1]), self.rdf(w, ops[2]), self.rdf(w, ops[3])
            with np.errstate(invalid="ignore", over="ignore"):
                r = (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)
            self.wr32(w, ops[0], r)
        elif op in ("v_exp_f32", "v_log_f32", "v_rcp_f32"):
            x = self.rdf(w, ops[1]).astype(np.float64)
            tiny = float(np.finfo(np.float32).tiny)
            x = np.where(np.abs(x) < tiny, np.copysign(0.0, x), x)          # the transcendental unit takes no denormal inputs ...
            with np.errstate(all="ignore"):
                r = {"v_exp_f32": np.exp2, "v_log_f32": np.log2, "v_rcp_f32": lambda t: 1.0 / t}[op](x)
                r = np.where(np.abs(r) < tiny, np.copysign(0.0, r), r)      # ... and returns none (2^-127 -> 0), whatever the denormal mode
            self.wr32(w, ops[0], r.astype(np.float32), writer="trans")
        elif op in ("v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32"):
            lo, hi = self.rdf(w, ops[1]), self.rdf(w, ops[2])
            cv = f32_to_bf16_bits if op.endswith("bf16_f32") else f32_to_f16_bits
            self.wr32(w, ops[0], (cv(lo) | (cv(hi) << 16)).astype(np.uint32))
        elif op in ("v_cmp_gt_i32", "v_cmp_le_i32", "v_cmp_ge_i32"):
            x, y = self.rd32(w, ops[1]).view(np.int32), self.rd32(w, ops[2]).view(np.int32)
            w.vcc = {"v_cmp_gt_i32": x > y, "v_cmp_le_i32": x <= y, "v_cmp_ge_i32": x >= y}[op]
        elif op in ("v_cmp_lt_f32", "v_cmp_nge_f32", "v_cmp_ngt_f32"):
            x, y = self.rdf(w, ops[1]), self.rdf(w, ops[2])
            with np.errstate(invalid="ignore"):
                w.vcc = {"v_cmp_lt_f32": x < y, "v_cmp_nge_f32": ~(x >= y), "v_cmp_ngt_f32": ~(x > y)}[op]      # (n..: true for NaN operands)
        elif op == "v_ceil_f32":
            with np.errstate(invalid="ignore"):
                self.wr32(w, ops[0], np.ceil(self.rdf(w, ops[1])).astype(np.float32))
        elif op == "ds_write_b32":
            addr = self.rd32(w, ops[0]).astype(np.int64) + ins.mods.get("offset", 0)
            src = R(1)
            self.check_read(w, src.kind, src.idx, 1, "valu")
            data = np.ascontiguousarray(self.regfile(w, src.kind)[src.idx:src.idx + 1].T).view(np.uint8)   # [64, 4]
            self.lds_write(w, addr, data)
            w.lgkm.append(lambda: None)
        elif op == "v_cndmask_b32":
            self.wr32(w, ops[0], np.where(w.vcc, self.rd32(w, ops[2]), self.rd32(w, ops[1])))
        elif op == "v_cmp_eq_u32":
            w.vcc = self.rd32(w, ops[1]) == self.rd32(w, ops[2])
        elif op == "v_permlane32_swap_b32":
            d, s = R(0), R(1)
            x, y = self.rd32(w, d, "permlane"), self.rd32(w, s, "permlane")
            nx, ny = x.copy(), y.copy()
            nx[32:] = y[:32]
            ny[:32] = x[32:]
            self.wr32(w, d, nx)
            self.wr32(w, s, ny)
        elif op == "v_permlane16_swap_b32":
            # within each half of the wave (32 lanes): lanes 16..31 of the first operand <-> lanes 0..15 of the second
            d, s_ = R(0), R(1)
            x, y = self.rd32(w, d, "permlane"), self.rd32(w, s_, "permlane")
            nx, ny = x.copy(), y.copy()
            for h0 in (0, 32):
                nx[h0 + 16:h0 + 32] = y[h0:h0 + 16]
                ny[h0:h0 + 16] = x[h0 + 16:h0 + 32]
            self.wr32(w, d, nx)
            self.wr32(w, s_, ny)
        elif op == "v_min_u32":
            self.wr32(w, ops[0], np.minimum(self.rd32(w, ops[1]), self.rd32(w, ops[2])))
        elif op == "v_mul_lo_u32":
            self.wr32(w, ops[0], (self.rd32(w, ops[1]).astype(np.uint64) * self.rd32(w, ops[2]).astype(np.uint64) & np.uint64(0xffffffff)).astype(np.uint32))
        elif op == "v_min_i32":
            self.wr32(w, ops[0], np.minimum(self.rd32(w, ops[1]).view(np.int32), self.rd32(w, ops[2]).view(np.int32)).view(np.uint32))
        elif op == "ds_write_b64":
            addr = self.rd32(w, ops[0]).astype(np.int64) + ins.mods.get("offset", 0)
            src = R(1)
            self.check_read(w, src.kind, src.idx, 2, "valu")
            data = np.ascontiguousarray(self.regfile(w, src.kind)[src.idx:src.idx + 2].T).view(np.uint8)   # [64, 8]
            self.lds_write(w, addr, data)
            w.lgkm.append(lambda: None)
        elif op == "v_accvgpr_read_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]))
        elif op == "v_accvgpr_write_b32":
            self.wr32(w, ops[0], self.rd32(w, ops[1]))
        elif op == "ds_read_b128":
            dst, addr = R(0), self.rd32(w, ops[1]).astype(np.int64) + ins.mods.get("offset", 0)
            if self.check and (addr & 15).any():
                self.err(w, "ds_read_b128 address not 16-byte aligned")
            data = self.lds_read(w, addr, 16).copy().view(np.uint32)      # [64, 4]
            self.mark_inflight(w, dst, 4, +1)

            def land(dst=dst, data=data):
                self.mark_inflight(w, dst, 4, -1)
                rf = self.regfile(w, dst.kind)
                for i in range(4):
                    rf[dst.idx + i] = data[:, i]
                    w.mfma_ready.pop((dst.kind, dst.idx + i), None)
            w.lgkm.append(land)
        elif op == "ds_read_b64_tr_b16":
            dst, addr = R(0), self.rd32(w, ops[1]).astype(np.int64) + ins.mods.get("offset", 0)
            if self.check and (addr & 7).any():
                self.err(w, "ds_read_b64_tr_b16 address not 8-byte aligned (returns the wrong data on gfx950)")
            raw = self.lds_read(w, addr, 8).copy().view(np.uint16)        # [64 lanes, 4 elems] as addressed
            res = np.empty((NLANE, 4), dtype=np.uint16)
            for l in range(NLANE):
                g, pp = l >> 4, l & 15
                for j in range(4):
                    res[l, j] = raw[16 * g + 4 * j + (pp >> 2), pp & 3]
            words = np.empty((NLANE, 2), dtype=np.uint32)
            words[:, 0] = res[:, 0].astype(np.uint32) | (res[:, 1].astype(np.uint32) << 16)
            words[:, 1] = res[:, 2].astype(np.uint32) | (res[:, 3].astype(np.uint32) << 16)
            self.mark_inflight(w, dst, 2, +1)

            def land(dst=dst, words=words):
                self.mark_inflight(w, dst, 2, -1)
                rf = self.regfile(w, dst.kind)
                for i in range(2):
                    rf[dst.idx + i] = words[:, i]
            w.lgkm.append(land)
        elif op == "ds_write_b128":
            addr = self.rd32(w, ops[0]).astype(np.int64) + ins.mods.get("offset", 0)
            src = R(1)
            self.check_read(w, src.kind, src.idx, 4, "valu")
            data = np.ascontiguousarray(self.regfile(w, src.kind)[src.idx:src.idx + 4].T).view(np.uint8)   # [64, 16]
            self.lds_write(w, addr, data)
            w.lgkm.append(lambda: None)
        elif op == "global_load_dwordx4":
            dst = R(0)
            a = R(1)
            sb = R(2)
            if isinstance(sb, Sym):          # `off`: 64-bit address in a VGPR pair
                lo, hi = self.regfile(w, a.kind)[a.idx].astype(np.uint64), self.regfile(w, a.kind)[a.idx + 1].astype(np.uint64)
                addr = (lo | (hi << np.uint64(32))).astype(np.int64) + ins.mods.get("offset", 0)
            else:                            # saddr form: 64-bit scalar base (numpy uint32[2]) + 32-bit VGPR offset
                self.check_read(w, a.kind, a.idx, 1, "valu")
                base = int(sb[0]) | (int(sb[1]) << 32)
                addr = base + self.regfile(w, a.kind)[a.idx].astype(np.int64) + ins.mods.get("offset", 0)
            data = np.stack([self.gread(int(x), 16).view(np.uint32) for x in addr])       # [64, 4]
            self.mark_inflight(w, dst, 4, +1)

            def land(dst=dst, data=data):
                self.mark_inflight(w, dst, 4, -1)
                rf = self.regfile(w, dst.kind)
                for i in range(4):
                    rf[dst.idx + i] = data[:, i]
            w.vm.append(land)
        elif op == "global_store_dwordx4":           # saddr form: vaddr (32-bit offset), vdata[4], 64-bit scalar base
            a, src, sb = R(0), R(1), R(2)
            self.check_read(w, a.kind, a.idx, 1, "valu")
            self.check_read(w, src.kind, src.idx, 4, "valu")
            base = int(sb[0]) | (int(sb[1]) << 32)
            addr = base + self.regfile(w, a.kind)[a.idx].astype(np.int64) + ins.mods.get("offset", 0)
            data = np.ascontiguousarray(self.regfile(w, src.kind)[src.idx:src.idx + 4].T).view(np.uint8)      # [64, 16]
            for lane in range(64):
                arr, off = self._gfind(int(addr[lane]), 16)
                arr[off:off + 16] = data[lane]
            w.vm.append(lambda: None)
        elif op == "global_load_dword":
            dst, a, sb = R(0), R(1), R(2)
            self.check_read(w, a.kind, a.idx, 1, "valu")
            base = int(sb[0]) | (int(sb[1]) << 32)
            addr = base + self.regfile(w, a.kind)[a.idx].astype(np.int64) + ins.mods.get("offset", 0)
            data = np.array([self.gread(int(x), 4).view(np.uint32)[0] for x in addr], dtype=np.uint32)
            self.mark_inflight(w, dst, 1, +1)

            def land(dst=dst, data=data):
                self.mark_inflight(w, dst, 1, -1)
                self.regfile(w, dst.kind)[dst.idx] = data
            w.vm.append(land)
        elif op in ("v_dot2_f32_f16", "v_dot2_f32_bf16"):
            conv = bf16_to_f32 if op.endswith("bf16") else f16_to_f32
            x, y = self.rd32(w, ops[1]), self.rd32(w, ops[2])
            acc = self.rdf(w, ops[3], "dot_c").astype(np.float64)          # (the same opcode may chain on its accumulator)
            for sh in (0, 16):
                acc = acc + conv(((x >> sh) & 0xffff).astype(np.uint16)).astype(np.float64) * conv(((y >> sh) & 0xffff).astype(np.uint16)).astype(np.float64)
            self.wr32(w, ops[0], acc.astype(np.float32), writer="dot")
        elif op == "buffer_load_dword":
            assert ins.mods.get("lds") and ins.mods.get("offen")
            if self.check and w.issue_idx - w.last_m0_write < 2:
                self.err(w, "LDS-DMA issued right after an M0 write (needs 1 wait state)")
            voff = self.rd32(w, ops[0]).astype(np.int64)
            rs = R(1)
            base = int(rs[0]) | ((int(rs[1]) & 0xffff) << 32)
            nrec = int(rs[2])
            off = voff + self.rds(w, ops[2]) + ins.mods.get("offset", 0)
            data = np.zeros((NLANE, 4), dtype=np.uint8)
            for l in range(NLANE):
                o = int(off[l]) & 0xffffffff
                if o + 4 <= nrec:
                    data[l] = self.gread(base + o, 4)
            dst_addr = (w.m0 & 0x3ffff) + LANES * 4 + ins.mods.get("offset", 0)

            def land(dst_addr=dst_addr, data=data):
                self.lds_write(w, dst_addr, data)
            w.vm.append(land)
        elif op == "buffer_load_dwordx4" and not ins.mods.get("lds"):
            # plain buffer load into VGPRs: dst, voffset, descriptor, soffset (out-of-range reads return 0)
            assert ins.mods.get("offen")
            dst = R(0)
            voff = self.rd32(w, ops[1]).astype(np.int64)
            rs = R(2)
            base = int(rs[0]) | ((int(rs[1]) & 0xffff) << 32)
            nrec = int(rs[2])
            off = voff + self.rds(w, ops[3]) + ins.mods.get("offset", 0)
            data = np.zeros((NLANE, 4), dtype=np.uint32)
            for l in range(NLANE):
                o = int(off[l]) & 0xffffffff
                if o + 16 <= nrec:
                    data[l] = self.gread(base + o, 16).view(np.uint32)
            self.mark_inflight(w, dst, 4, +1)

            def land(dst=dst, data=data):
                self.mark_inflight(w, dst, 4, -1)
                rf = self.regfile(w, dst.kind)
                for i in range(4):
                    rf[dst.idx + i] = data[:, i]
            w.vm.append(land)
        elif op == "buffer_store_dwordx2":
            # vdata[2], voffset, descriptor, soffset: 8 bytes per lane; a lane whose range leaves the descriptor stores nothing
            assert ins.mods.get("offen")
            src = R(0)
            self.check_read(w, src.kind, src.idx, 2, "vmem")
            voff = self.rd32(w, ops[1]).astype(np.int64)
            rs = R(2)
            base = int(rs[0]) | ((int(rs[1]) & 0xffff) << 32)
            nrec = int(rs[2])
            off = voff + self.rds(w, ops[3]) + ins.mods.get("offset", 0)
            rf = self.regfile(w, src.kind)
            for l in range(NLANE):
                o = int(off[l]) & 0xffffffff
                if o + 8 <= nrec:
                    arr, at = self._gfind(base + o, 8)
                    arr[at:at + 8] = np.array([rf[src.idx][l], rf[src.idx + 1][l]], dtype=np.uint32).view(np.uint8)
        elif op == "buffer_load_dwordx4":
            assert ins.mods.get("lds") and ins.mods.get("offen")
            if self.check and w.issue_idx - w.last_m0_write < 2:
                self.err(w, "LDS-DMA issued right after an M0 write (needs 1 wait state)")
            voff = self.rd32(w, ops[0]).astype(np.int64)
            rs = R(1)                                       # numpy uint32[4] descriptor
            soff = self.rds(w, ops[2])
            if self.check and (soff & 0xffffffff) >= 0x80000000:
                self.err(w, "LDS-DMA with a scalar offset that has wrapped (0x%x): out of range for every lane on the hardware" % (soff & 0xffffffff))
            base = int(rs[0]) | ((int(rs[1]) & 0xffff) << 32)
            nrec = int(rs[2])
            off = voff + soff + ins.mods.get("offset", 0)
            data = np.zeros((NLANE, 16), dtype=np.uint8)
            for l in range(NLANE):
                o = int(off[l]) & 0xffffffff
                if o + 16 <= nrec:
                    data[l] = self.gread(base + o, 16)
            dst_addr = (w.m0 & 0x3ffff) + LANES * 16 + ins.mods.get("offset", 0)

            def land(dst_addr=dst_addr, data=data):
                self.lds_write(w, dst_addr, data)
            w.vm.append(land)
        else:
            raise EmuError("unimplemented op " + op)
        w.pc = nxt

    # ------------------------------------------------------------ run the workgroup
    def run(self, max_steps=5_000_000):
        n = len(self.ins)
        steps = 0
        while True:
            progressed = False
            for w in self.waves:
                while not w.done and not w.at_barrier:
                    if w.pc >= n:
                        w.done = True
                        break
                    self.step(w)
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise EmuError("step limit")
            if all(w.done for w in self.waves):
                break
            if all(w.done or w.at_barrier for w in self.waves):
                if any(w.done for w in self.waves) and any(w.at_barrier for w in self.waves):
                    raise EmuError("barrier mismatch: some waves finished while others wait at s_barrier")
                if getattr(self, "on_barrier", None):
                    self.on_barrier(self)
                self.epoch += 1
                c = max(w.cycle for w in self.waves)
                self.body_cycles.append(c)
                for w in self.waves:
                    w.at_barrier = False
                    w.cycle = c
                progressed = True
            if not progressed:
                raise EmuError("deadlock")
        for w in self.waves:
            if w.lgkm or (w.vm and not getattr(self, "allow_vm_in_flight", False)):
                self.errors.append("wave %d ends with loads in flight" % w.wid)
        return self

    def reenter(self, wave_args):
        """The same workgroup runs the statement again with new operands (persistent workgroups: registers, LDS and the
        loads still in flight carry over; compiler-owned registers v0..15 are rebound by the caller)."""
        for w, a in zip(self.waves, wave_args):
            w.args = a
            w.pc = 0
            w.done = False
            w.at_barrier = False
        self.epoch += 1
