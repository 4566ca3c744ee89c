"""Scheduling helpers shared by the asm generators (fwd_d128_gen.py, bwd_d128_gen.py).

A body of a hand-scheduled kernel is a fixed list of MFMAs; everything else (VALU, transcendental, LDS, LDS-DMA, SALU) is a
"filler stream" — an ordered list of instructions or atomic groups (lists) — that `place` spreads over a window of MFMA gaps so
that the weighted issue load of the gaps is flat (water-filling), and `lds_waits` inserts counted `s_waitcnt lgkmcnt(n)` in front
of the first consumer of every LDS read still in flight.
"""
from isa import Neg, mk

# relative issue cost of the instruction classes (the scheduler balances this, not the instruction count)
WEIGHT = {"valu": 1.0, "trans": 1.7, "lds": 1.6, "dma": 3.5, "salu": 0.4, "branch": 0.5, None: 0.0}


def weight(item):
    if isinstance(item, list):
        return sum(weight(i) for i in item)
    return 0.0 if item.op == "label" else WEIGHT.get(item.tag, 1.0)


def set_weights(trans, lds, dma, salu):
    WEIGHT.update({"trans": trans, "lds": lds, "dma": dma, "salu": salu, "branch": salu})


def place(load, slots, items, a, b, sid):
    """Put the ordered `items` of one stream into the MFMA gaps [a, b), filling the least loaded gaps first
    (water-filling on the weighted load) while keeping the stream's order."""
    if not items:
        return
    gaps = list(range(int(a), min(len(load), int(b + 0.999))))
    w = [weight(it) for it in items]
    total = sum(w)
    lo, hi = min(load[g] for g in gaps), max(load[g] for g in gaps) + total + 1.0
    for _ in range(50):                     # water level: sum(max(0, L - load)) == total
        mid = 0.5 * (lo + hi)
        if sum(max(0.0, mid - load[g]) for g in gaps) >= total:
            hi = mid
        else:
            lo = mid
    cap = [max(0.0, hi - load[g]) for g in gaps]
    cum, acc = [], 0.0
    for c in cap:
        acc += c
        cum.append(acc)
    gi, done, counts = 0, 0.0, {}
    for it, wi in zip(items, w):
        centre = done + 0.5 * wi                # the item goes where its centre of weight falls in the free capacity
        while gi < len(gaps) - 1 and cum[gi] < centre:
            gi += 1
        g = gaps[gi]
        done += wi
        load[g] += wi
        counts.setdefault(g, []).append(it)
    for g, lst in counts.items():
        n = len(lst)
        for j, it in enumerate(lst):
            slots[g].append((g + (j + 0.5) / n, sid, it))



def regs(ins):
    """(reads, writes) of an instruction as lists of (kind, lo, hi) register ranges (what the LDS-wait pass needs)."""
    def rng(o):
        if isinstance(o, Neg):
            o = o.reg
        return (o.kind, o.idx, o.idx + o.n) if hasattr(o, "kind") and hasattr(o, "idx") else None
    ops = [rng(o) for o in ins.ops]
    if ins.op.startswith("s_") or ins.op in ("label", "raw"):
        return [], []
    if ins.op.startswith("v_permlane"):
        both = [o for o in ops if o]
        return both, both
    if ins.op.startswith("v_cmp") or ins.op.startswith("buffer_load") or ins.op.startswith("ds_write"):
        return [o for o in ops if o], []
    return [o for o in ops[1:] if o], [o for o in ops[:1] if o]

def lds_waits(items, look=3):
    """Counted `s_waitcnt lgkmcnt(n)` in front of the first instruction that touches the destination of an LDS read still
    in flight (LDS reads return in order, so n = the number of reads issued after the one needed).  One wait also covers
    what the next `look` MFMAs need, so a k-step costs one wait, not four."""
    out, pend = [], []                     # pend: destinations of the reads in flight, oldest first
    def need(ins):
        rd, wr = regs(ins)
        j = -1
        for (k, lo, hi) in rd + wr:
            for i, (pk, plo, phi) in enumerate(pend):
                if pk == k and lo < phi and plo < hi:
                    j = max(j, i)
        return j
    for idx, ins in enumerate(items):
        if ins.op == "s_waitcnt":
            if "lgkmcnt" in ins.mods:
                keep = ins.mods["lgkmcnt"]
                pend = pend[len(pend) - keep:] if keep else []
            out.append(ins)
            continue
        if ins.op == "s_memtime":          # SMEM shares the counter and may return out of order: drain, then count afresh
            if pend:
                out.append(mk("s_waitcnt", lgkmcnt=0))
                pend = []
            out.append(ins)
            out.append(mk("s_waitcnt", lgkmcnt=0))
            continue
        j = need(ins)
        if j >= 0:
            if ins.op.startswith("v_mfma"):
                seen = 0
                for nxt in items[idx + 1:]:
                    if nxt.op.startswith("v_mfma"):
                        j = max(j, need(nxt))
                        seen += 1
                        if seen >= look:
                            break
            n = len(pend) - 1 - j
            out.append(mk("s_waitcnt", lgkmcnt=min(n, 15)))
            pend = pend[j + 1:] if n <= 15 else pend[len(pend) - 15:]
        if ins.op.startswith("ds_read"):
            _, wr = regs(ins)
            pend.append(wr[0])
        out.append(ins)
    return out

