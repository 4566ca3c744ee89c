"""The reference harness's comparison (bench_with_sdpa.py: the attention operator vs torch SDPA), on MI355X:
forward and backward times of FlashAttentionFunction against torch.nn.functional.scaled_dot_product_attention
(whatever fused backend this PyTorch-ROCm build picks) on the BASELINE configs and a few Stable-Diffusion shapes.
Developer tool; prints one line per shape."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

SHAPES = [  # name, B, H, Nq, Nkv, D, dtype, causal
    ("config2", 2, 16, 4096, 4096, 128, torch.float16, False),
    ("config3", 2, 16, 4096, 4096, 128, torch.bfloat16, True),
    ("config4", 1, 32, 8192, 8192, 128, torch.float16, True),
    ("sdxl-64x64", 2, 10, 4096, 4096, 64, torch.float16, False),
    ("sdxl-32x32", 2, 20, 1024, 1024, 64, torch.float16, False),
    ("sdxl-cross", 2, 10, 4096, 77, 64, torch.float16, False),
    ("sd15-64x64", 2, 8, 4096, 4096, 40, torch.float16, False),
    ("sd15-cross", 2, 8, 4096, 77, 40, torch.float16, False),
    ("sd15-16x16", 2, 8, 256, 256, 160, torch.float16, False),
]


def timeit(fn, iters, windows=5):
    """median over `windows` timing windows of `iters` calls (HIP events): a single window of a few milliseconds of host-bound calls — the
    small SD shapes' backward — doubles when the host thread is descheduled once (seen on three boxes: 59 / 117 / 181 us for the same shape)"""
    for _ in range(max(3, iters // 5)):
        fn()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda", 0)
    print("%-12s %-34s %10s %10s %7s %10s %10s %7s" % ("shape", "", "fa2 fwd", "sdpa fwd", "x", "fa2 bwd", "sdpa bwd", "x"))
    for name, B, H, Nq, Nkv, D, dt, causal in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn((B, H, Nq, D), generator=g, device=dev, dtype=torch.float32).to(dt).requires_grad_(True)
        k = torch.randn((B, H, Nkv, D), generator=g, device=dev, dtype=torch.float32).to(dt).requires_grad_(True)
        v = torch.randn((B, H, Nkv, D), generator=g, device=dev, dtype=torch.float32).to(dt).requires_grad_(True)
        do = torch.randn((B, H, Nq, D), generator=g, device=dev, dtype=torch.float32).to(dt)
        iters = 100 if Nq * Nkv <= 4096 * 4096 else 30
        with torch.no_grad():
            t_f = timeit(lambda: FlashAttentionFunction.apply(q, k, v, None, causal), iters)
            t_s = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal), iters)
            err = (FlashAttentionFunction.apply(q, k, v, None, causal).float() - F.scaled_dot_product_attention(q, k, v, is_causal=causal).float()).abs().max().item()
        o_f = FlashAttentionFunction.apply(q, k, v, None, causal)
        o_s = F.scaled_dot_product_attention(q, k, v, is_causal=causal)

        def bwd(o):
            q.grad = k.grad = v.grad = None
            o.backward(do, retain_graph=True)
        t_fb = timeit(lambda: bwd(o_f), max(10, iters // 3))
        t_sb = timeit(lambda: bwd(o_s), max(10, iters // 3))
        desc = "B%d H%d N%d/%d D%d %s%s" % (B, H, Nq, Nkv, D, str(dt)[6:], " causal" if causal else "")
        print("%-12s %-34s %8.1fus %8.1fus %6.2fx %8.1fus %8.1fus %6.2fx   max|fa2-sdpa| %.1e" %
              (name, desc, t_f, t_s, t_s / t_f, t_fb, t_sb, t_sb / t_fb, err))


if __name__ == "__main__":
    main()
