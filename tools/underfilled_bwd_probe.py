"""Whole backward of grids that cover at most half of the CUs, through the operator, with the split and with option split = 0 (developer probe, round 6;
FA2_FRONTEND=py FA2_GFX950_LIB=<variant> compares builds)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
dev = torch.device("cuda", 0)
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for (B, H, N, D, dt) in ((1, 8, 4096, 64, torch.float16), (1, 8, 4096, 40, torch.float16), (1, 4, 4096, 64, torch.bfloat16), (1, 4, 2048, 64, torch.float16), (1, 2, 8192, 64, torch.float16), (1, 8, 4096, 80, torch.float16), (1, 16, 2048, 64, torch.float16)):
    q, k, v = (torch.randn((B, H, N, D), device=dev).to(dt).requires_grad_(True) for _ in range(3))
    do = torch.randn((B, H, N, D), device=dev).to(dt)
    o = FlashAttentionFunction.apply(q, k, v, None, False)
    a = t(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    with _fa2_lib.options(split=0):
        b = t(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    print("B%d H%d N%d D%d %s backward: %.1f us (split) / %.1f us (option split = 0)" % (B, H, N, D, str(dt)[6:], a, b), flush=True)
