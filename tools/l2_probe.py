"""Does K/V L2 residency matter?  Times config 2 with every head reading the SAME K/V (head stride 0: the
working set is 2 MB per batch entry) against the normal layout (developer tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402


def bench(q, k, v, causal, iters=100):
    f = FlashAttentionFunction.apply
    for _ in range(20):
        f(q, k, v, None, causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f(q, k, v, None, causal)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, N, D) in ((2, 16, 4096, 128), (8, 16, 4096, 128), (4, 16, 2048, 128)):
    q, k, v = (torch.randn((B, H, N, D), device="cuda", dtype=torch.float16) for _ in range(3))
    ks, vs = k[:, :1].expand(B, H, N, D), v[:, :1].expand(B, H, N, D)
    for causal in (False, True):
        for rep in range(2):
            t_n = bench(q, k, v, causal)
            t_s = bench(q, ks, vs, causal)
            print("B%d H%d N%d D%d causal=%d: normal %.1f us, shared-KV %.1f us (%.1f%%)" % (B, H, N, D, causal, t_n, t_s, 100 * (t_n - t_s) / t_n))
