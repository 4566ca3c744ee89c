"""The reference harness's two sweeps, machine-readable (bench_with_sdpa.py:201-224 N-scan, :259-283 D-scan).

    python tools/scan_bench.py > profiles/<round>_scan.json

Protocol of the reference: (B, H) = (1, 24), fp16, non-causal, torch.rand inputs; N-scan: D = 64, N = 512 * i for
i = 1..14; D-scan: N = 4096, D = 16 * i for i = 1..15 (here also 256..512 in steps of 64: the range the gfx950 kernels
add); FLOPs = 4*B*H*N*N*D (x2.5 for the backward, bench_with_sdpa.py:35-41); 10 warm-up + 100 timed calls, wall clock
around a synchronize (bench_with_sdpa.py:13-31) — plus the peak VRAM the reference plots.  Both the operator and
torch SDPA on the same tensors; max |diff| of the two forwards is recorded for every point.
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

B, H = 1, 24
WARMUP, ITERS = 10, 100


def timed(fn):
    for _ in range(WARMUP):
        fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ITERS, torch.cuda.max_memory_allocated() / 2 ** 20


def point(N, D, backward):
    dev = torch.device("cuda", 0)
    q, k, v = (torch.rand((B, H, N, D), dtype=torch.float16, device=dev) for _ in range(3))
    flops = 4.0 * B * H * N * N * D
    rec = {"N": N, "D": D}
    o_fa = FlashAttentionFunction.apply(q, k, v, None, False)
    o_sd = F.scaled_dot_product_attention(q, k, v)
    rec["max_abs_diff"] = round(float((o_fa.float() - o_sd.float()).abs().max()), 6)
    t, mem = timed(lambda: FlashAttentionFunction.apply(q, k, v, None, False))
    rec["fa2_fwd_tflops"], rec["fa2_fwd_vram_mb"] = round(flops / t / 1e12, 1), round(mem, 1)
    t, mem = timed(lambda: F.scaled_dot_product_attention(q, k, v))
    rec["sdpa_fwd_tflops"], rec["sdpa_fwd_vram_mb"] = round(flops / t / 1e12, 1), round(mem, 1)
    if backward:
        do = torch.rand_like(q)
        for name, fn in (("fa2", lambda a, b, c: FlashAttentionFunction.apply(a, b, c, None, False)),
                         ("sdpa", lambda a, b, c: F.scaled_dot_product_attention(a, b, c))):
            qg, kg, vg = (t_.detach().requires_grad_(True) for t_ in (q, k, v))
            og = fn(qg, kg, vg)

            def bwd():
                qg.grad = kg.grad = vg.grad = None
                og.backward(do, retain_graph=True)
            t, _ = timed(bwd)
            rec[name + "_bwd_tflops"] = round(2.5 * flops / t / 1e12, 1)
    return rec


def main():
    out = {"_comment": "reference sweeps (bench_with_sdpa.py:201-283) on %s: B=%d H=%d fp16 non-causal, %d warm-up + %d timed calls, wall clock; "
                       "TFLOPS = 4*B*H*N*N*D / t (backward x2.5)" % (torch.cuda.get_device_name(0), B, H, WARMUP, ITERS)}
    out["n_scan_d64"] = [point(512 * i, 64, True) for i in range(1, 15)]
    d_list = [16 * i for i in range(1, 16)] + [256, 320, 384, 448, 512]
    out["d_scan_n4096"] = [point(4096, d, d <= 256) for d in d_list]
    out["n_scan_d128"] = [point(n, 128, False) for n in (512, 1024, 2048, 4096, 8192, 16384)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
