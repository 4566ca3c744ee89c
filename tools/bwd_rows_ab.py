"""Backward A/B on the GPU (developer tool): FlashAttentionFunction backward with the library option "rows" = 0 (heuristic) / 128 / 256 on
shapes whose dQ grid of 256-row workgroups leaves a partly filled last round.    python tools/bwd_rows_ab.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

SHAPES = [("sdxl-64x64 B2 H10 N4096 D64", 2, 10, 4096, 64), ("H24 N3072 D64", 1, 24, 3072, 64), ("H24 N4096 D64", 1, 24, 4096, 64), ("sd15 B2 H8 N4096 D40", 2, 8, 4096, 40),
          ("sd15 B3 H8 N4096 D40", 3, 8, 4096, 40), ("H24 N4096 D128", 1, 24, 4096, 128), ("B2 H20 N2048 D80", 2, 20, 2048, 80)]
dev = torch.device("cuda", 0)
for name, B, H, N, D in SHAPES:
    q, k, v = (torch.rand((B, H, N, D), device=dev).half().requires_grad_(True) for _ in range(3))
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    go = torch.rand_like(o)
    ts = {0: [], 128: [], 256: []}
    for _ in range(5):
        for rows in ts:
            with _fa2_lib.options(rows=rows):
                for _ in range(3):
                    o.backward(go, retain_graph=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    o.backward(go, retain_graph=True)
                e1.record()
                torch.cuda.synchronize()
                ts[rows].append(e0.elapsed_time(e1) * 1e3 / 20)
    print("%-30s bwd us: heuristic %8.1f   rows=128 %8.1f   rows=256 %8.1f" % (name, *(statistics.median(ts[r]) for r in (0, 128, 256))))
