"""The reference harness's two sweeps, machine-readable (bench_with_sdpa.py:201-224 N-scan, :259-283 D-scan).

    python tools/scan_bench.py [--dtype f16|bf16] [--layout bhnd|bnhd] [--quick] > profiles/<round>_scan.json

--dtype bf16 is the protocol of bench_with_sdpa_bf16.py (:53-58: the same sweeps on torch.bfloat16 inputs), --layout bnhd that of
bench_with_sdpa_BNHD.py (:106 `wmma_fttn(q, k, v, None, causal, None, True)` on [B, N, H, D] tensors, N = 512 * i, :115-124; torch SDPA gets
the transposed views, as the reference's sdp_pt does); both together are bench_with_sdpa_bf16_BNHD.py.

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
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, workspace_pool_bytes  # noqa: E402

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


DTYPE, BNHD = torch.float16, False


def point(N, D, backward):
    dev = torch.device("cuda", 0)
    shape = (B, N, H, D) if BNHD else (B, H, N, D)
    q, k, v = (torch.rand(shape, dtype=DTYPE, device=dev) for _ in range(3))
    flops = 4.0 * B * H * N * N * D
    rec = {"N": N, "D": D}
    fa2 = lambda a, b, c: FlashAttentionFunction.apply(a, b, c, None, False, None, BNHD)  # noqa: E731
    if BNHD:      # torch SDPA on the transposed views (bench_with_sdpa_BNHD.py's sdp_pt), result back in [B, N, H, D]
        sdpa = lambda a, b, c: F.scaled_dot_product_attention(a.transpose(1, 2), b.transpose(1, 2), c.transpose(1, 2)).transpose(1, 2)  # noqa: E731
    else:
        sdpa = lambda a, b, c: F.scaled_dot_product_attention(a, b, c)  # noqa: E731
    o_fa = fa2(q, k, v)
    o_sd = sdpa(q, k, v)
    rec["max_abs_diff"] = round(float((o_fa.float() - o_sd.float()).abs().max()), 6)
    # peak memory as the reference records it (bench_with_sdpa.py:34: max_memory_allocated over the timed calls, inputs included).  The operator's
    # split scratch is one block per stream that stays allocated between calls (FlashAttn.py: _workspace): it is part of the operator's peak,
    # reported separately too, and taken out of torch SDPA's figure (it is not SDPA's memory, it just is still there when SDPA runs)
    t, mem = timed(lambda: fa2(q, k, v))
    pool = workspace_pool_bytes() / 2 ** 20
    rec["fa2_fwd_tflops"], rec["fa2_fwd_vram_mb"], rec["fa2_ws_pool_mb"] = round(flops / t / 1e12, 1), round(mem, 1), round(pool, 1)
    rec["fa2_fwd_vram_excl_pool_mb"] = round(mem - pool, 1)
    t, mem = timed(lambda: sdpa(q, k, v))
    rec["sdpa_fwd_tflops"], rec["sdpa_fwd_vram_mb"] = round(flops / t / 1e12, 1), round(mem - pool, 1)
    if backward:
        do = torch.rand_like(q)
        for name, fn in (("fa2", fa2), ("sdpa", sdpa)):
            qg, kg, vg = (t_.detach().requires_grad_(True) for t_ in (q, k, v))
            og = fn(qg, kg, vg)

            def bwd():
                qg.grad = kg.grad = vg.grad = None
                og.backward(do, retain_graph=True)
            t, _ = timed(bwd)
            rec[name + "_bwd_tflops"] = round(2.5 * flops / t / 1e12, 1)
    return rec


def main():
    global DTYPE, BNHD, WARMUP, ITERS
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=["f16", "bf16"], default="f16")
    ap.add_argument("--layout", choices=["bhnd", "bnhd"], default="bhnd")
    ap.add_argument("--quick", action="store_true", help="3 warm-up + 20 timed calls and every other point (a short lease)")
    a = ap.parse_args()
    DTYPE = torch.float16 if a.dtype == "f16" else torch.bfloat16
    BNHD = a.layout == "bnhd"
    if a.quick:
        WARMUP, ITERS = 3, 20
    src = {("f16", "bhnd"): "bench_with_sdpa.py:201-283", ("bf16", "bhnd"): "bench_with_sdpa_bf16.py:53-58 (+ the sweeps of bench_with_sdpa.py)",
           ("f16", "bnhd"): "bench_with_sdpa_BNHD.py:95-124", ("bf16", "bnhd"): "bench_with_sdpa_bf16_BNHD.py"}[(a.dtype, a.layout)]
    out = {"_comment": "reference sweeps (%s) on %s: B=%d H=%d %s %s non-causal, %d warm-up + %d timed calls, wall clock; "
                       "TFLOPS = 4*B*H*N*N*D / t (backward x2.5)" % (src, torch.cuda.get_device_name(0), B, H, a.dtype, a.layout.upper(), WARMUP, ITERS),
           "dtype": a.dtype, "layout": a.layout}
    step = 2 if a.quick else 1
    out["n_scan_d64"] = [point(512 * i, 64, True) for i in range(1, 15, step)]
    d_list = [16 * i for i in range(1, 16, step)] + [256, 320, 384, 448, 512][::step]
    out["d_scan_n4096"] = [point(4096, d, d <= 256) for d in d_list]
    out["n_scan_d128"] = [point(n, 128, False) for n in (512, 1024, 2048, 4096, 8192, 16384)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
