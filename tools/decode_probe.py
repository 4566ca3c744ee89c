"""Decode-sized and other underfilled forward grids: the operator against torch SDPA and the byte floor, and with / without the KV-split (developer probe, round 6)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
import torch.nn.functional as F
dev = torch.device("cuda", 0)
def t(fn, n=100):
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
for (B, H, Nq, Nkv, D) in ((1, 32, 1, 8192, 128), (8, 32, 1, 8192, 128), (32, 32, 1, 4096, 128), (1, 32, 16, 32768, 128), (4, 8, 1, 16384, 64)):
    q = torch.randn((B, H, Nq, D), device=dev).half(); k = torch.randn((B, H, Nkv, D), device=dev).half(); v = torch.randn_like(k)
    a = t(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    b = t(lambda: F.scaled_dot_product_attention(q, k, v))
    mb = 2 * B * H * Nkv * D * 2 / 1e6
    print("B%d H%d Nq%d Nkv%d D%d: fa2 %.1f us  sdpa %.1f us   | K, V = %.0f MB: %.1f us at 4.5 TB/s" % (B, H, Nq, Nkv, D, a, b, mb, mb / 4.5), flush=True)
from rocwmma_fattn import _fa2_lib
print("underfilled grids, option split = 1 / 0:")
for (B, H, Nq, Nkv, D, dt) in ((1, 32, 1, 8192, 128, torch.float16), (1, 8, 4096, 4096, 40, torch.float16), (1, 8, 4096, 4096, 128, torch.float16), (1, 8, 4096, 4096, 64, torch.bfloat16),
                               (1, 4, 2048, 2048, 128, torch.float16), (2, 8, 1024, 1024, 80, torch.float16), (1, 16, 512, 8192, 64, torch.float16), (1, 8, 8192, 8192, 128, torch.float16),
                               (1, 10, 4096, 4096, 64, torch.float16), (1, 12, 4096, 4096, 128, torch.float16), (1, 6, 8192, 8192, 128, torch.bfloat16), (2, 20, 1024, 1024, 64, torch.float16), (1, 14, 4096, 4096, 40, torch.float16)):
    q = torch.randn((B, H, Nq, D), device=dev).to(dt); k = torch.randn((B, H, Nkv, D), device=dev).to(dt); v = torch.randn_like(k)
    a = t(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    with _fa2_lib.options(split=0):
        b = t(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    pl = _fa2_lib.fwd_plan(q, k, False, None, workspace_bytes=1 << 30)
    print("B%d H%d Nq%d Nkv%d D%d %s: %.1f / %.1f us   plan nsplit %d items %d kernel %d" % (B, H, Nq, Nkv, D, str(dt)[6:], a, b, pl.nsplit, pl.split_items, pl.kernel), flush=True)
