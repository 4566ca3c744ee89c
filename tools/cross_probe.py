"""Cross-attention shapes (Nkv = 77) under the forward workgroup-shape switch (developer tool).
Run once per setting: FA2_FWD_ROWS=256|128 python tools/cross_probe.py"""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
SHAPES = [("sdxl_cross64", 2, 10, 4096, 77, 64), ("sdxl_cross32", 2, 20, 1024, 77, 64), ("sd15_cross64", 2, 8, 4096, 77, 40),
          ("sd15_cross32", 2, 8, 1024, 77, 80), ("sd15_cross16", 2, 8, 256, 77, 160), ("b1_cross64", 1, 10, 4096, 77, 64)]
out = []
for name, B, H, N, Nkv, D in SHAPES:
    q = torch.rand((B, H, N, D), device="cuda", dtype=torch.float16)
    k, v = (torch.rand((B, H, Nkv, D), device="cuda", dtype=torch.float16) for _ in range(2))
    res = []
    for f in (lambda: FlashAttentionFunction.apply(q, k, v, None, False), lambda: F.scaled_dot_product_attention(q, k, v)):
        for _ in range(20): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        res.append(best)
    out.append("%s %.1f (sdpa %.1f)" % (name, res[0], res[1]))
print(os.environ.get("FA2_FWD_ROWS", "auto"), " | ".join(out))
