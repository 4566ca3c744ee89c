"""One planted key per row gives it a log2 score of 20, 24, ... 136 late in the sweep (developer probe): which growth of the reference do the sum-check
bodies of the hand-scheduled forward survive?  Prints, per target, the finite heads and the errors against float64 (found the f32 edge at 126 .. 128).
    python tools/growth_scan.py"""
import sys, os
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from rocwmma_fattn import _fa2_lib
import test_parity_gpu as T
dev = torch.device("cuda", 0)
LOG2E = 1.4426950408889634
for dt in (0, 1):
    g = torch.Generator(device="cpu").manual_seed(5 + dt)
    H, N = 8, 1024
    q, k, v = (torch.randn((1, H, N, 128), generator=g) for _ in range(3))
    # row i of every head: key 300 + i gives it a log2 score of about target[i] (|q_i|^2 * alpha * scale * log2e)
    targets = list(range(20, 140, 4))
    rows = [7 + 9 * j for j in range(len(targets))]        # spread over q blocks 0/1 and both 32-row blocks of waves
    for r, tg in zip(rows, targets):
        n2 = float((q[0, 0, r] ** 2).sum())
        for h in range(H):
            n2 = float((q[0, h, r] ** 2).sum())
            k[0, h, 300 + (r % 600)] = q[0, h, r] * (tg / (n2 * 128 ** -0.5 * LOG2E))
    q, k, v = (t.to(T.TORCH_DT[dt]).to(dev) for t in (q, k, v))
    with _fa2_lib.options(rows=256):
        o, lse = T._cabi_forward(q, k, v, False)
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * 128 ** -0.5
    ref = torch.matmul(torch.softmax(s, -1), v.double())
    lse_ref = torch.logsumexp(s, -1) * LOG2E
    for r, tg in zip(rows, targets):
        oo = o[0, :, r].float()
        fin = torch.isfinite(oo).all(dim=-1)
        err = (oo.double() - ref[0, :, r]).abs().amax(dim=-1)
        lerr = (lse[0, :, r].double() - lse_ref[0, :, r]).abs()
        print("dt", dt, "row", r, "target", tg, "finite heads", int(fin.sum()), "/", H, "max err %.3g" % float(err[fin].max() if fin.any() else float("nan")),
              "lse err %.3g" % float(lerr[fin].max() if fin.any() else float("nan")), "lse sample", [round(float(x), 2) for x in lse[0, :3, r]])
