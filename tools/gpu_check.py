"""Quick on-box sanity + timing of the gfx950 forward kernel (developer tool, not a test).

    python tools/gpu_check.py [--bench]
"""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma  # noqa: E402

LOG2E = 1.4426950408889634


def ref_fp32(q, k, v, causal, scale):
    qf, kf, vf = q.float(), k.float(), v.float()
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if causal:
        nq, nk = s.shape[-2:]
        mask = torch.ones(nq, nk, dtype=torch.bool, device=s.device).triu(1)
        s = s.masked_fill(mask, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vf), lse * LOG2E


def check(B, H, N, D, dtype, causal, nkv=None, seed=0, kind="rand"):
    g = torch.Generator(device="cuda").manual_seed(seed)
    nkv = nkv or N
    mk = (lambda *s: torch.rand(*s, generator=g, device="cuda", dtype=torch.float32)) if kind == "rand" \
        else (lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float32))
    q, k, v = mk(B, H, N, D).to(dtype), mk(B, H, nkv, D).to(dtype), mk(B, H, nkv, D).to(dtype)
    scale = D ** -0.5
    o, _, _, _, _, L = flash_attn_wmma.forward(q, k, v, 64, 128, causal, scale, False)
    torch.cuda.synchronize()
    o_ref, lse_ref = ref_fp32(q, k, v, causal, scale)
    err = (o.float() - o_ref).abs().max().item()
    lerr = (L[:, :, :N] - lse_ref).abs().max().item()
    sd = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal) if nkv == N else None
    sderr = (sd.float() - o_ref).abs().max().item() if sd is not None else float("nan")
    print(f"B{B} H{H} N{N} Nkv{nkv} D{D} {str(dtype)[6:]} causal={int(causal)} {kind}: "
          f"max|O-ref|={err:.3e} (sdpa lowp {sderr:.3e})  max|L-ref|={lerr:.3e}  nan={bool(torch.isnan(o).any())}")
    return err


def bench(B, H, N, D, dtype, causal, iters=50):
    q = torch.rand(B, H, N, D, device="cuda", dtype=dtype)
    k = torch.rand(B, H, N, D, device="cuda", dtype=dtype)
    v = torch.rand(B, H, N, D, device="cuda", dtype=dtype)
    f = FlashAttentionFunction.apply
    for _ in range(10):
        f(q, k, v, None, causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f(q, k, v, None, causal)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
    print(f"bench B{B} H{H} N{N} D{D} {str(dtype)[6:]} causal={int(causal)}: {ms*1e3:.1f} us  "
          f"{flops/ms/1e9:.1f} TFLOPS  ({flops/ms/1e9/2500*100:.1f}% of 2.5 PF)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true")
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0))
    for dtype in (torch.float16, torch.bfloat16):
        for causal in (False, True):
            check(1, 2, 128, 64, dtype, causal)
            check(1, 2, 512, 128, dtype, causal)
            check(2, 3, 1024, 128, dtype, causal, kind="randn")
            check(1, 2, 333, 128, dtype, causal, nkv=None)
            check(1, 2, 300, 64, dtype, causal, nkv=77 if not causal else None)
    if a.bench:
        bench(2, 16, 4096, 128, torch.float16, False)
        bench(2, 16, 4096, 128, torch.bfloat16, True)
        bench(1, 32, 8192, 128, torch.float16, True)
        bench(8, 16, 4096, 128, torch.float16, False)
        bench(2, 16, 4096, 64, torch.float16, False)
