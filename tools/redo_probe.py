"""Sum-check bodies on 3x-amplitude inputs (developer probe): every head of a 66-head call against dense fp32 attention, under several options —
finds rows the in-place repair or the safe-mode redo gets wrong (round 4: the rows that grew by 126 .. 127 octaves; profiles/r12_redo_probe_after_fix.txt).
    python tools/redo_probe.py"""
import sys, os, ctypes
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from rocwmma_fattn import _fa2_lib
import test_parity_gpu as T
dev = torch.device("cuda", 0)
def run(dt, H, N, kind, **opts):
    g = torch.Generator(device="cpu").manual_seed(700 + dt)
    q, k, v = (torch.randn((1, H, N, 128), generator=g) for _ in range(3))
    if kind == "redo":
        q, k = q * 3, k * 3
        k[:, :, 200] = q[:, :, 5] * 4; k[:, :, 70] = q[:, :, 40] * 2; k[:, :, 600] = q[:, :, 800] * 4
    elif kind == "redo1":
        q, k = q * 3, k * 3
        k[:, :, 200] = q[:, :, 5] * 4
    q, k, v = (t.to(T.TORCH_DT[dt]).to(dev) for t in (q, k, v))
    with _fa2_lib.options(**opts):
        plan = T._plan(q, k, False)
        o, lse = T._cabi_forward(q, k, v, False)
    bad = ~torch.isfinite(o.float()).all(dim=-1)[0]          # [H, N]
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * 128 ** -0.5
    ref = torch.matmul(torch.softmax(s, -1), v.float())
    err = (o.float() - ref).abs().amax(dim=-1)[0]
    err = torch.where(bad, torch.zeros_like(err), err)
    heads = bad.any(dim=1).nonzero().flatten().tolist()
    print("dt", dt, "H", H, "N", N, kind, opts, "kernel", plan.kernel, "contract", plan.contract, "| bad rows", int(bad.sum()), "heads", heads[:12], "max finite err %.3g" % float(err.max()))
    if heads:
        h0 = heads[0]
        rows = bad[h0].nonzero().flatten().tolist()
        print("   head", h0, "bad rows", rows[:10], "...", rows[-3:], "count", len(rows), " lse bad", int((~torch.isfinite(lse[0, h0])).sum()))
for dt in (1, 0):
    for kind in ("redo", "redo1", "none"):
        run(dt, 66, 1024, kind)
    run(dt, 66, 1024, "redo", persist=0)
    run(dt, 8, 1024, "redo", rows=256)
run(0, 66, 1024, "redo", fold=0)
run(0, 66, 1024, "redo1", fold=0)
run(1, 66, 1024, "redo", fold=2)
