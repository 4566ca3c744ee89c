"""Where does a launch's time go outside its waves?  (developer probe)

A build of the library with -DFA2_STAMP=1 (tools/kbench.py build stamp:-DFA2_STAMP=1) makes every wave of the hand-scheduled forward
kernel record s_memrealtime (100 MHz wall counter) when it starts and after its last store; this tool issues back-to-back launches of
config 2, each with its own LSE buffer (the stamps land in its head), and prints per launch: the spread of the wave starts (dispatch ramp), the
spread of the wave ends (tail imbalance), the busy span first-start -> last-end, and the dead time from the last end of launch i to the
first start of launch i+1 (end-of-kernel fence / cache write-back + the next dispatch).

    python tools/stamp_probe.py [--launches 40] [--variant stamp] [--cfg c2]
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--variant", default="stamp")
    ap.add_argument("--cfg", default="c2")
    a = ap.parse_args()
    var = kbench.Variant(os.path.join(kbench.VAR_DIR, a.variant + ".so"))
    B, H, N, D, dts, causal = kbench.CFGS[a.cfg]
    dt = torch.float16 if dts == "f16" else torch.bfloat16
    dev = torch.device("cuda", 0)
    q, k, v = (torch.rand((B, H, N, D), device=dev, dtype=torch.float32).to(dt) for _ in range(3))
    o = torch.empty_like(q)
    stream = torch.cuda.current_stream().cuda_stream
    warm = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    for _ in range(150):                      # past the clock settling
        var.fwd(q, k, v, o, warm, causal, stream)
    lses = [torch.zeros((B, H, N), dtype=torch.float32, device=dev) for _ in range(a.launches)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for l in lses:
        var.fwd(q, k, v, o, l, causal, stream)
    e1.record()
    torch.cuda.synchronize()
    period_us = e0.elapsed_time(e1) * 1e3 / a.launches
    nwg = min(256, B * H * ((N + 255) // 256) // (2 if causal else 1))
    recs = []
    for l in lses:
        st = l.view(-1)[: nwg * 4 * 8].view(torch.int64).view(nwg * 4, 4).cpu()
        s, e = st[:, 0], st[:, 1]
        ok = (s > 0) & (e > s)
        recs.append((int(s[ok].min()), int(s[ok].max()), int(e[ok].min()), int(e[ok].max()), int(ok.sum())))
        last = st
    tick = 0.01      # us per tick of the 100 MHz counter
    print("%s: %d launches back to back, period %.1f us (events); ticks of 10 ns; waves stamped per launch: %d" % (a.cfg, a.launches, period_us, recs[0][4]))
    print("launch  start-spread  end-spread  busy-span  dead-time-to-next   [us]")
    spans, deads, ramps, tails = [], [], [], []
    for i, (s0, s1, e0_, e1_, _) in enumerate(recs):
        dead = (recs[i + 1][0] - e1_) * tick if i + 1 < len(recs) else float("nan")
        spans.append((e1_ - s0) * tick)
        ramps.append((s1 - s0) * tick)
        tails.append((e1_ - e0_) * tick)
        if i + 1 < len(recs):
            deads.append(dead)
        if i < 6 or i >= len(recs) - 3:
            print("%5d  %11.2f  %10.2f  %9.2f  %10.2f" % (i, ramps[-1], tails[-1], spans[-1], dead))
    print("median: start-spread %.2f  end-spread %.2f  busy-span %.2f  dead %.2f  -> span + dead = %.2f us (period %.1f)" % (
        statistics.median(ramps), statistics.median(tails), statistics.median(spans), statistics.median(deads),
        statistics.median(spans) + statistics.median(deads), period_us))


    # the last launch in detail: per XCD (XCC_ID) and per workgroup — who finishes late, and is it more cycles or a slower clock?
    st = last
    dur = (st[:, 1] - st[:, 0]).double() * tick
    cyc = st[:, 2].double()
    xcc = (st[:, 3] >> 32) & 0xf
    hw = st[:, 3] & 0xffffffff
    cu, se = (hw >> 8) & 0xf, (hw >> 13) & 0x7
    end = (st[:, 1] - st[:, 0].min()).double() * tick
    print("last launch, per XCD: waves, mean wave lifetime us, mean cycles, MHz = cycles / lifetime, mean / max end-time us")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("  xcc %d: %4d  %8.2f  %9.0f  %7.1f  %8.2f %8.2f" % (x, int(m.sum()), float(dur[m].mean()), float(cyc[m].mean()),
                                                                 float((cyc[m] / dur[m]).mean()), float(end[m].mean()), float(end[m].max())))
    q = torch.quantile(end, torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0], dtype=torch.float64))
    print("end-time quantiles (us since the first start): " + " ".join("%.2f" % float(x) for x in q))
    qc = torch.quantile(cyc, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
    print("cycle-count quantiles: " + " ".join("%.0f" % float(x) for x in qc))
    wg_end = end.view(-1, 4).max(dim=1).values
    order = torch.argsort(wg_end)
    print("earliest workgroups (blockIdx: end us, xcc, se, cu): " + ", ".join("%d: %.1f x%d s%d c%d" % (int(i), float(wg_end[i]), int(xcc[4 * i]), int(se[4 * i]), int(cu[4 * i])) for i in order[:8]))
    print("latest workgroups: " + ", ".join("%d: %.1f x%d s%d c%d" % (int(i), float(wg_end[i]), int(xcc[4 * i]), int(se[4 * i]), int(cu[4 * i])) for i in order[-8:]))
    corr = torch.corrcoef(torch.stack([cyc, dur]))[0, 1]
    print("corr(cycles, lifetime) over waves = %.3f" % float(corr))


if __name__ == "__main__":
    main()
