"""A/B of the 256-row and 128-row forward workgroup shapes on small / awkward grids (developer tool).
Run once per setting: FA2_ROWS=256|128 [FA2_ASM=0] python tools/rows_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
SHAPES = [("sd15_16", 2, 8, 256, 160), ("sd15_8", 2, 8, 64, 160), ("d64_n512_h8", 1, 8, 512, 64), ("sdxl64", 2, 10, 4096, 64), ("sdxl32", 2, 20, 1024, 64), ("sd15_64", 2, 8, 4096, 40), ("sd15_32", 2, 8, 1024, 80),
          ("d128_n1k_h16", 1, 16, 1024, 128), ("d128_n2k_h8", 1, 8, 2048, 128), ("d128_n1k_b8", 8, 16, 1024, 128), ("d128_h24", 1, 24, 4096, 128),
          ("d64_h24_3072", 1, 24, 3072, 64), ("c2", 2, 16, 4096, 128)]
out = []
for name, B, H, N, D in SHAPES:
    q, k, v = (torch.rand((B, H, N, D), device="cuda", dtype=torch.float16) for _ in range(3))
    f = lambda: FlashAttentionFunction.apply(q, k, v, None, False)
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    out.append("%s %.1f" % (name, best))
print(os.environ.get("FA2_ROWS", "auto"), os.environ.get("FA2_ASM", "3"), " | ".join(out))
