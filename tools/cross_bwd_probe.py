"""Cross-attention backward (Nkv = 77) on grids that fill the chip: the operator's fa2_bwd against torch SDPA's, eager, forward + backward and
backward alone (developer probe, round 6).  The floor of moving the bytes (Q, O, dO in, dQ out) is printed beside them.
    python tools/cross_bwd_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
import torch.nn.functional as F
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


for (B, H, N, Nkv, D) in ((2, 10, 4096, 77, 64), (8, 16, 4096, 77, 64), (8, 16, 4096, 77, 128), (16, 8, 4096, 77, 40), (16, 20, 1024, 77, 64), (8, 16, 4096, 77, 80), (8, 16, 4096, 128, 64)):
    q = torch.randn((B, H, N, D), device=dev).half().requires_grad_(True)
    k = torch.randn((B, H, Nkv, D), device=dev).half().requires_grad_(True)
    v = torch.randn((B, H, Nkv, D), device=dev).half().requires_grad_(True)
    do = torch.randn((B, H, N, D), device=dev).half()
    res = {}
    for name, f in (("fa2", lambda: FlashAttentionFunction.apply(q, k, v, None, False)), ("sdpa", lambda: F.scaled_dot_product_attention(q, k, v))):
        o = f()
        res[name + " fwd+bwd"] = t(lambda: torch.autograd.grad(f(), (q, k, v), do))
        res[name + " bwd"] = t(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
        if name == "fa2":
            with _fa2_lib.options(short=0):
                res["fa2 bwd short=0"] = t(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    mb = 4 * B * H * N * D * 2 / 1e6
    print("B%d H%d N%d x %d D%d: %s   | Q, O, dO, dQ = %.0f MB: %.1f us at 4.5 TB/s" % (B, H, N, Nkv, D, "  ".join("%s %.1f us" % kv for kv in res.items()), mb, mb / 4.5))
