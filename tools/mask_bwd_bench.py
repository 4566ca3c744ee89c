"""Masked backward on the GPU (developer tool): flash_attention(mask=...) backward against the unmasked backward and torch SDPA's, per call.
    python tools/mask_bwd_bench.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import flash_attention  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [("N4096 D64 dense f16 bias shared", 2, 10, 4096, 4096, 64, torch.float16, "io", (1, 1)), ("N4096 D128 bool shared", 2, 16, 4096, 4096, 128, torch.float16, "bool", (1, 1)),
         ("N1024 D64 dense f16 bias per head", 2, 20, 1024, 1024, 64, torch.float16, "io", (2, 20)), ("sdxl-cross key-padding bool", 2, 10, 4096, 77, 64, torch.float16, "kp", (2, 1))]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("%-36s %10s %10s %10s" % ("case", "masked bwd", "plain bwd", "sdpa bwd"))
for name, B, H, N, Nkv, D, dt, kind, (mb, mh) in CASES:
    q = torch.randn((B, H, N, D), device=dev).to(dt).requires_grad_(True)
    k, v = (torch.randn((B, H, Nkv, D), device=dev).to(dt).requires_grad_(True) for _ in range(2))
    if kind == "bool":
        mask = torch.rand((mb, mh, N, Nkv), device=dev) < 0.8
    elif kind == "kp":
        mask = torch.rand((mb, mh, 1, Nkv), device=dev) < 0.8
    else:
        mask = torch.randn((mb, mh, N, Nkv), device=dev).to(dt)
    do = torch.randn((B, H, N, D), device=dev).to(dt)
    om = flash_attention(q, k, v, mask)
    op = flash_attention(q, k, v, None)
    os_ = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    t = [timeit(lambda o=o: o.backward(do, retain_graph=True)) for o in (om, op, os_)]
    print("%-36s %8.1fus %8.1fus %8.1fus" % (name, *t))
