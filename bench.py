#!/usr/bin/env python3
"""Forward-attention benchmark of the gfx950 FlashAttention-2 path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE forward call FlashAttentionFunction.apply(q, k, v, None, causal) — the same call the
reference harness times (bench_with_sdpa.py:99, :136) — on synthetic torch.rand U[0,1) inputs
(bench_with_sdpa.py:118-120) that are resident in HBM before the timed region starts.
FLOPs = 4*B*H*N*N*D, x0.5 when causal (bench_with_sdpa.py:35-41).

Workloads (BASELINE.json configs):
  c2 (default)  B2 H16 N4096 D128 fp16 non-causal — the configuration the metric is quoted on
  c3            B2 H16 N4096 D128 bf16 causal
  c4            B1 H32 N8192 D128 fp16 causal
  c5            B64 H16 N4096 D128 fp16 non-causal, batch split over the ranks (strong scaling)
With N > 1 ranks every rank runs the per-GPU workload on its own shard (c2/c3/c4: weak scaling,
global batch = N x per-GPU batch; c5: strong).  Attention is independent per (batch, head), so no
collective is on the data path; the only RCCL calls are the barriers and the max-over-ranks of the
elapsed time.

Rank 0 prints ONE JSON line.  `value` = whole-job TFLOPS (all ranks' FLOPs / max-over-ranks wall
time of the K steps, bracketed by barrier + synchronize).  `roofline` prices the kernel against the
2.5 PFLOP/s dense fp16/bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md) using the average
launch duration measured with HIP events on the launch stream.  `cpu_baseline` times the C port of
the same algorithm (oracle/, test infrastructure) on a bounded sample of the workload on the host
cores; `cpu_sdpa` times torch's CPU scaled_dot_product_attention — the comparator call of the
reference harness (bench_with_sdpa.py:65-70) moved to device="cpu" — on the same sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

WORKLOADS = {
    #       B   H   N     D    dtype            causal  scaling
    "c2": (2, 16, 4096, 128, torch.float16, False, "weak"),
    "c3": (2, 16, 4096, 128, torch.bfloat16, True, "weak"),
    "c4": (1, 32, 8192, 128, torch.float16, True, "weak"),
    "c5": (64, 16, 4096, 128, torch.float16, False, "strong"),
}


def attention_flops(B, H, Nq, Nkv, D, causal):
    return 4.0 * B * H * Nq * Nkv * D * (0.5 if causal else 1.0)


def cpu_baseline(H_total, N, D, causal, dtype, seed, budget_s=12.0):
    """Time the C oracle (a port of the reference algorithm, oracle/fa2_oracle.c) and torch CPU SDPA
    on a bounded sample: whole heads of the workload, as many as fit ~budget_s of CPU time."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import fa2_oracle as fo
    cores = os.cpu_count() or 1
    threads = min(cores, fo.max_threads())
    dt_code = fo.DTYPE_F16 if dtype == torch.float16 else fo.DTYPE_BF16
    g = torch.Generator(device="cpu").manual_seed(seed)

    def mk(h):
        return [torch.rand((1, h, N, D), generator=g, dtype=torch.float32).to(dtype) for _ in range(3)]

    def bits(t):
        return t.view(torch.int16).numpy().view(np.uint16)

    q, k, v = mk(1)
    fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)       # warm-up (thread pool, pages)
    t0 = time.perf_counter()
    fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t_head = time.perf_counter() - t0
    heads = int(max(1, min(H_total, budget_s / max(t_head, 1e-6))))
    q, k, v = mk(heads)
    t0 = time.perf_counter()
    fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t_pass = time.perf_counter() - t0
    reps = int(max(1, min(40, budget_s / max(t_pass, 1e-6))))  # many-core hosts finish a pass in ~1 s: repeat it
    t0 = time.perf_counter()
    for _ in range(reps):
        fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t_port = time.perf_counter() - t0
    flops = attention_flops(1, heads, N, N, D, causal) * reps
    out = {"value": round(flops / t_port / 1e12, 5), "unit": "TFLOPS", "cores": threads, "kind": "port",
           "sample": "%d of %d heads of the workload (N=%d D=%d) x %d passes, %.2f s, oracle/fa2_oracle.c Br=32 Bc=64, OpenMP"
                     % (heads, H_total, N, D, reps, t_port)}
    # torch CPU SDPA on the same sample
    torch.set_num_threads(cores)
    sd_heads = min(heads, 32)
    qs, ks, vs = q[:, :sd_heads], k[:, :sd_heads], v[:, :sd_heads]
    torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=causal)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=causal)
    t_sd = (time.perf_counter() - t0) / reps
    sdpa = {"value": round(attention_flops(1, sd_heads, N, N, D, causal) / t_sd / 1e12, 5), "unit": "TFLOPS",
            "cores": cores, "kind": "torch.nn.functional.scaled_dot_product_attention on device=cpu",
            "sample": "%d heads (N=%d D=%d), %.3f s per call" % (sd_heads, N, D, t_sd)}
    return out, sdpa


def read_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)   # past the DVFS ramp-up after idle
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU: the attention operator has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from rocwmma_fattn.FlashAttn import FlashAttentionFunction
    from rocwmma_fattn.shard import shard_bounds

    B, H, N, D, dtype, causal, scaling = WORKLOADS[args.workload]
    if scaling == "strong":
        lo, hi = shard_bounds(B, world, rank)
        B_local, B_global = hi - lo, B
    else:
        B_local, B_global = B, B * world
    cfg_idx = sorted(WORKLOADS).index(args.workload) + 1
    g = torch.Generator(device=device).manual_seed(1234 + cfg_idx + rank)
    q, k, v = (torch.rand((B_local, H, N, D), generator=g, device=device, dtype=torch.float32).to(dtype)
               for _ in range(3))
    attn = FlashAttentionFunction.apply

    for _ in range(args.warmup):
        o = attn(q, k, v, None, causal)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # torch's current stream == the stream the kernel is launched on
    for _ in range(args.steps):
        o = attn(q, k, v, None, causal)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    assert torch.isfinite(o.float()).all(), "non-finite attention output"

    # Informational, outside the timed region and never part of `value`: the backward of the same workload
    # through autograd, timed the way the reference harness does (O.backward(dO, retain_graph=True) in a loop,
    # bench_with_sdpa.py:78-88) — grads are dropped instead of zeroed so no accumulation kernels are counted.
    bwd = None
    if world == 1:
        qg, kg, vg = (t.detach().requires_grad_(True) for t in (q, k, v))
        do = torch.rand_like(q)
        og = attn(qg, kg, vg, None, causal)

        def one_backward():
            qg.grad = kg.grad = vg.grad = None
            og.backward(do, retain_graph=True)

        for _ in range(5):
            one_backward()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_b = 30
        b0.record()
        for _ in range(n_b):
            one_backward()
        b1.record()
        torch.cuda.synchronize()
        bwd_ms = b0.elapsed_time(b1) / n_b
        assert all(torch.isfinite(t.grad.float()).all() for t in (qg, kg, vg)), "non-finite gradient"
        bwd = {"bwd_ms": round(bwd_ms, 4),
               "bwd_tflops": round(2.5 * attention_flops(B_local, H, N, N, D, causal) / (bwd_ms * 1e-3) / 1e12, 1),
               "note": "bwd FLOPs = 2.5 x fwd (bench_with_sdpa.py:35-41); launches: dQ (+ delta), dV, dK (dK+dV fused at D <= 64)"}

    times = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(times[0]), float(times[1])

    if rank == 0:
        flops_global = attention_flops(B_global, H, N, N, D, causal)
        flops_local = attention_flops(B_local, H, N, N, D, causal)
        value = flops_global * args.steps / elapsed / 1e12
        achieved = flops_local / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "fwd attention TFLOPS (and % MFMA roofline) at B2 H16 N4096 D128 fp16",
            "value": round(value, 2), "unit": "TFLOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": "%s: B%d H%d N%d D%d %s causal=%s per GPU, BHND, torch.rand U[0,1)"
                                   % (args.workload, B_local, H, N, D, str(dtype)[6:], causal),
                       "global_batch": B_global, "parallelism": "batch-shard x%d (no data-path collective)" % world},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": read_traffic(args.workload),
                         "kernel_ms": round(kernel_ms, 5), "flops_per_launch": flops_local},
            "pct_of_mfma_roofline": round(100.0 * value / (MFMA_PEAK_TFLOPS * world), 2),
        }
        if bwd is not None:
            line["backward"] = bwd
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["cpu_sdpa"] = cpu_baseline(B * H, N, D, causal, dtype, 1234 + cfg_idx)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
