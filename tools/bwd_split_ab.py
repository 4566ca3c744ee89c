"""Backward split A/B on the GPU (developer tool): FlashAttentionFunction backward with the library option "split" on / off, interleaved.
    python tools/bwd_split_ab.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

SHAPES = [("sdxl-64x64 B2 H10 N4096 D64", 2, 10, 4096, 64), ("sdxl-64x64 B4 H10", 4, 10, 4096, 64), ("H24 N3072 D64", 1, 24, 3072, 64), ("H24 N4096 D64", 1, 24, 4096, 64),
          ("H24 N5632 D64", 1, 24, 5632, 64), ("sd15 B2 H8 N4096 D40", 2, 8, 4096, 40), ("sd15 B3 H8 N4096 D40", 3, 8, 4096, 40), ("B2 H20 N2048 D80", 2, 20, 2048, 80),
          ("c2 B2 H16 N4096 D128", 2, 16, 4096, 128)]
dev = torch.device("cuda", 0)
lib = _fa2_lib.load()
print("%-30s %10s %10s %7s   ws MB" % ("shape", "split us", "plain us", "x"))
for name, B, H, N, D in SHAPES:
    q, k, v = (torch.rand((B, H, N, D), device=dev).half().requires_grad_(True) for _ in range(3))
    with _fa2_lib.options(split=0):
        o = FlashAttentionFunction.apply(q, k, v, None, False)
    go = torch.rand_like(o)
    ts = {1: [], 0: []}
    for _ in range(5):
        for mode in ts:
            with _fa2_lib.options(split=mode):
                for _ in range(3):
                    o.backward(go, retain_graph=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    o.backward(go, retain_graph=True)
                e1.record()
                torch.cuda.synchronize()
                ts[mode].append(e0.elapsed_time(e1) * 1e3 / 20)
    t1, t0 = statistics.median(ts[1]), statistics.median(ts[0])
    print("%-30s %10.1f %10.1f %7.3f   %.1f" % (name, t1, t0, t0 / t1, lib.fa2_bwd_workspace_bytes(0, B, H, N, N, D + (-D % 8), 0) / 2 ** 20))
