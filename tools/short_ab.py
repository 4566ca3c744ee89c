"""The short-sweep forward kernel (csrc/fa2_fwd_short.hip.h) against the routing it replaced (option "short" = 0), eager operator, over grid sizes, head
dims, dtypes and sweep lengths 32 .. 128 (developer A/B; profiles/r22_short_ab.txt).
    python tools/short_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
dev = torch.device("cuda", 0)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
print("B H Nq Nkv D dtype: short kernel us / streaming (or hand-scheduled) us")
for (B, H, N) in ((4, 16, 4096), (8, 16, 4096), (2, 8, 1024), (16, 16, 1024), (1, 24, 16384)):
    for D in (64, 128, 40, 80):
        for dt in (torch.float16, torch.bfloat16):
            row = []
            for nkv in (32, 64, 77, 128):
                q = torch.randn((B, H, N, D), device=dev).to(dt); k = torch.randn((B, H, nkv, D), device=dev).to(dt); v = torch.randn_like(k)
                a = t(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
                with _fa2_lib.options(short=0):
                    b = t(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
                row.append("Nkv%d %.1f/%.1f" % (nkv, a, b))
            print("B%d H%d N%d D%d %s: %s" % (B, H, N, D, str(dt)[6:], "  ".join(row)), flush=True)
