"""Where the cross-attention calls stand against this box's floors (developer probe, round 6): per call inside a replayed graph of 20 —
a one-element kernel (the launch floor), a copy of Q into O (the floor of moving the call's bytes), torch SDPA, the operator.
    python tools/short_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
import torch.nn.functional as F
dev = torch.device("cuda", 0)


def graphed(fn, n=20, reps=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps / n * 1e3)
    return sorted(ts)[2]


one = torch.zeros(1, device=dev)
print("one-element kernel: %.2f us" % graphed(lambda: one.add_(1.0)))
for name, (B, H, N, Nkv, D) in {"sdxl-cross": (2, 10, 4096, 77, 64), "sd15-cross": (2, 8, 4096, 77, 40), "sdxl-cross-1024": (2, 20, 1024, 77, 64),
                                "sd15-cross-1024": (2, 8, 1024, 77, 80)}.items():
    q = torch.rand((B, H, N, D), device=dev).half(); k = torch.rand((B, H, Nkv, D), device=dev).half(); v = torch.rand_like(k)
    o = torch.empty_like(q)
    res = {"copy Q->O": graphed(lambda: o.copy_(q)), "sdpa": graphed(lambda: F.scaled_dot_product_attention(q, k, v)),
           "fa2": graphed(lambda: FlashAttentionFunction.apply(q, k, v, None, False))}
    with _fa2_lib.options(short=0):
        res["fa2 short=0"] = graphed(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    print(name, " ".join("%s %.2f us" % kv for kv in res.items()))
