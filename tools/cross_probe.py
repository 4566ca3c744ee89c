"""SD cross-attention shapes (Nkv = 77) through the operator under option `rows`, inside a replayed graph of 20 calls, and torch SDPA (developer probe).
Round 4, one box: SDXL B2 H10 N4096 D64 11.1 us at rows 0 / 128, 17.6 at 256, 11.1 in a graph (GPU-bound), SDPA 10.3 / 10.0; SD1.5 D40 7.0 vs 16.0; N1024 7.6 vs 9.3.
    python tools/cross_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
import torch.nn.functional as F
dev = torch.device("cuda", 0)
def bench(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for name, (B, H, N, Nkv, D) in {"sdxl-cross": (2, 10, 4096, 77, 64), "sd15-cross": (2, 8, 4096, 77, 40), "sdxl-cross-1024": (2, 20, 1024, 77, 64)}.items():
    q = torch.rand((B, H, N, D), device=dev).half(); k = torch.rand((B, H, Nkv, D), device=dev).half(); v = torch.rand_like(k)
    res = {}
    for rows in (0, 128, 256):
        with _fa2_lib.options(rows=rows):
            res["rows=%d" % rows] = bench(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    res["sdpa"] = bench(lambda: F.scaled_dot_product_attention(q, k, v))
    g = torch.cuda.CUDAGraph()
    o = FlashAttentionFunction.apply(q, k, v, None, False); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): o = FlashAttentionFunction.apply(q, k, v, None, False)
    res["fa2 in a graph of 20"] = bench(lambda: g.replay(), 100) / 20
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(20): o2 = F.scaled_dot_product_attention(q, k, v)
    res["sdpa in a graph of 20"] = bench(lambda: g2.replay(), 100) / 20
    print(name, " ".join("%s %.2f us" % kv for kv in res.items()))
