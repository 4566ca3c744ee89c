"""Tiny gfx950 program builder used by fwd_d128_gen.py.

A program is a list of `Ins` objects (mnemonic + operand objects + modifiers).  The same list is
(a) rendered to assembler text for the inline-asm body of the HIP kernel and (b) interpreted by the
functional emulator in tools/asm_emu.py (test infrastructure), so what is tested on the CPU is the
instruction stream that ships.

Operand objects
    Reg(kind, index, count)   kind in 'v' (arch VGPR), 'a' (accumulator VGPR), 's' (SGPR)
    Sym(name)                 'vcc', 'exec', 'm0', 'scc', 'off'
    Arg(n)                    compiler-assigned inline-asm operand %n (only in the entry/exit moves)
    int / float               immediates (floats are rendered as hex literals)
    Label(name)               branch target
"""
import struct


class Reg:
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        assert kind in ("v", "a", "s") and idx >= 0 and n >= 1
        self.kind, self.idx, self.n = kind, int(idx), int(n)

    def __getitem__(self, i):              # sub-register
        assert 0 <= i < self.n
        return Reg(self.kind, self.idx + i, 1)

    def sub(self, off, n):
        assert 0 <= off and off + n <= self.n
        return Reg(self.kind, self.idx + off, n)

    def text(self):
        if self.n == 1:
            return "%s%d" % (self.kind, self.idx)
        return "%s[%d:%d]" % (self.kind, self.idx, self.idx + self.n - 1)

    def __repr__(self):
        return self.text()


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


class Sym:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def text(self):
        return self.name

    def __repr__(self):
        return self.name


VCC, EXEC, M0, OFF = Sym("vcc"), Sym("exec"), Sym("m0"), Sym("off")


class Arg:
    """%n operand of the enclosing inline-asm statement; `part` selects a 32-bit half of a 64-bit operand."""
    __slots__ = ("n", "kind", "width")

    def __init__(self, n, kind="v", width=1):
        self.n, self.kind, self.width = n, kind, width

    def text(self):
        return "%%%d" % self.n

    def __repr__(self):
        return "%%%d" % self.n


class Neg:
    """negated VGPR source (VOP3 neg modifier)"""
    __slots__ = ("reg",)

    def __init__(self, reg):
        self.reg = reg

    def text(self):
        return "-" + self.reg.text()


class Label:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def text(self):
        return self.name


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _imm_text(x):
    if isinstance(x, float):
        if x == 0.0:
            return "0"
        for val, txt in ((0.5, "0.5"), (1.0, "1.0"), (2.0, "2.0"), (4.0, "4.0"), (-0.5, "-0.5"), (-1.0, "-1.0"), (-2.0, "-2.0"), (-4.0, "-4.0")):
            if x == val:
                return txt
        return "0x%08x" % f32_bits(x)
    if isinstance(x, int):
        if -16 <= x <= 64:
            return str(x)
        return "0x%08x" % (x & 0xffffffff)
    raise TypeError(x)


class Ins:
    __slots__ = ("op", "ops", "mods", "tag", "comment")

    def __init__(self, op, ops=(), mods=None, tag=None, comment=None):
        self.op, self.ops, self.mods, self.tag, self.comment = op, tuple(ops), dict(mods or {}), tag, comment

    def text(self):
        if self.op == "label":
            return "%s:" % self.ops[0].text()
        if self.op == "raw":
            return self.ops[0]
        parts = []
        for o in self.ops:
            parts.append(o.text() if hasattr(o, "text") else _imm_text(o))
        s = self.op
        if self.op == "s_waitcnt":
            s += " " + " ".join("%s(%d)" % (k, v) for k, v in self.mods.items())
            return s
        if parts:
            s += " " + ", ".join(parts)
        for k, v in self.mods.items():
            if v is True:
                s += " " + k
            elif v is not False and v is not None:
                s += " %s:%s" % (k, v)
        return s

    def __repr__(self):
        return self.text()


class Program:
    def __init__(self):
        self.ins = []
        self._uniq = 0

    def emit(self, op, *ops, tag=None, comment=None, **mods):
        i = Ins(op, ops, mods, tag, comment)
        self.ins.append(i)
        return i

    def extend(self, items):
        self.ins.extend(items)

    def label(self, name):
        self.ins.append(Ins("label", (Label(name),)))

    def fresh(self, stem):
        self._uniq += 1
        return "%s_%d" % (stem, self._uniq)

    def text_lines(self):
        out = []
        for i in self.ins:
            t = i.text()
            if i.comment:
                t += "   ; " + i.comment
            out.append(t)
        return out


def mk(op, *ops, tag=None, **mods):
    return Ins(op, ops, mods, tag)
