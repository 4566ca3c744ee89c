#!/usr/bin/env python3
"""Forward-attention benchmark of the gfx950 FlashAttention-2 path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--no-cpu-baseline]
                    [--steady-launches S] [--collectives] [--backend nccl|gloo] [--same-device]
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
elapsed time.  `--collectives` additionally times, OUTSIDE the timed region and never folded into
`value`, the edge transfers a caller with the full tensors on rank 0 would pay (SURVEY §8e timing
rule): scatter_batch of q/k/v and gather_batch of o (rocwmma_fattn/shard.py).
`--backend gloo --same-device` runs the multi-rank control flow with every rank on cuda:0 (NCCL
refuses duplicate devices): the dry run the 1-GPU test suite uses; its numbers are not a scaling claim.

Default workload: c2 with one rank; with N > 1 ranks and no --workload the run measures c5 — BASELINE.json's multi-GPU
configuration (B = 64 split over the ranks, strong scaling) — as `value`, and then the c2 weak-scaling figure (every rank its own
B2 shard) under the separate key `weak_c2`.

Rank 0 prints ONE JSON line.
  value        whole-job TFLOPS: all ranks' FLOPs / max-over-ranks wall time of the K steps, bracketed
               by barrier + synchronize (contract).  Measured AFTER the settle phase (below); `warmup` is the W of the command line,
               `warmup_effective` every launch of the operator that preceded the timed region (gate + cold region + settle + W).
  cold         the same W warm-up + K timed steps run straight after the parity gate, BEFORE any settling — the protocol of rounds 1-3
               (and of a harness that only knows W): {"value", "ms_per_step", "kernel_ms"}; compare rounds on this key or on `steady`.
  sequence     what the process does, in order (also in the JSON line): parity gate -> settle -> W warm-up + K
               timed steps (the contract) -> K launches with per-launch events -> steady re-timing ->
               informational backward -> CPU baseline.  The timed region carries no event between launches (one costs
               1.6 %, tools/event_overhead.py).
  settle       before the warm-up the same call is launched in chunks of 25 for at least 150 ms and until six consecutive
               chunks agree within 1 % (at most `--settle N` launches, default 2000; checked bit-identical to the gate's output): after idle
               the chip boosts, overshoots its power budget, throttles and recovers over 100 .. 400 launches
               (profiles/r03_clock_settling.txt) — five warm-up launches measure that transient, not the kernel
               (round 3: 7 % under `steady`).  `--settle 0` reproduces the cold-start number.
  launch_ms    min / median / max / first / last of K per-launch durations measured straight after the timed
               region (HIP events on the launch stream, one event between consecutive launches).
  steady       the same loop re-timed for --steady-launches launches AFTER the contractual region
               (extra evidence, not the metric): the rate the kernel sustains once the clock has settled.
  sustained    (N = 1) the same loop for --sustain-seconds of GPU time (default 5): launches, mean launch time, first /
               last / slowest chunk of 500 — long enough for an outside utilisation sampler to see the device busy
               and for clock and temperature to reach their steady state (round 6; evidence, not the metric).
  roofline     prices the kernel against the 2.5 PFLOP/s dense fp16/bf16 MFMA peak
               (/opt/skills/guides/MI355X_MICROARCH.md) using the average launch duration of the timed
               region; `sustained_peak` / `frac_sustained` use the MFMA-only micro-benchmark measured on
               this chip under load (profiles/mfma_peak.json, tools/ubench/mfma_peak.hip) when present.
  check        parity gate before the warm-up (every head of one call, or 64 of them, against dense fp32 attention on
               the GPU) and one head of the timed output again after the timed region.
  cpu_baseline torch CPU scaled_dot_product_attention — the comparator call of the reference harness
               (bench_with_sdpa.py:65-70) on device="cpu", i.e. the reference's own CPU path — on a bounded
               sample of the workload on all host cores; `cpu_port` = the C port of the algorithm
               (oracle/, test infrastructure) on the same sample with the same thread count.
"""
import argparse
import json
import os
import statistics
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
    """torch CPU SDPA (the reference's CPU path) and the C oracle (a port of the reference algorithm,
    oracle/fa2_oracle.c) on a bounded sample: whole heads of the workload, ~budget_s of CPU time each,
    both on the same number of threads."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import fa2_oracle as fo
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, fo.max_threads()))
    torch.set_num_threads(threads)
    g = torch.Generator(device="cpu").manual_seed(seed)
    heads = int(min(H_total, 32))
    q, k, v = (torch.rand((1, heads, N, D), generator=g, dtype=torch.float32).to(dtype) for _ in range(3))
    sdpa = torch.nn.functional.scaled_dot_product_attention
    sdpa(q, k, v, is_causal=causal)                      # warm-up (thread pool, pages)
    t0 = time.perf_counter()
    sdpa(q, k, v, is_causal=causal)
    t_call = time.perf_counter() - t0
    reps = int(max(3, min(200, budget_s / max(t_call, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        sdpa(q, k, v, is_causal=causal)
    t_sd = time.perf_counter() - t0
    flops = attention_flops(1, heads, N, N, D, causal)
    ref = {"value": round(flops * reps / t_sd / 1e12, 5), "unit": "TFLOPS", "cores": threads, "kind": "reference",
           "sample": "torch.nn.functional.scaled_dot_product_attention on device=cpu (the reference harness's comparator "
                     "call, bench_with_sdpa.py:65-70): %d of %d heads of the workload (N=%d D=%d %s causal=%s) x %d calls, %.2f s"
                     % (heads, H_total, N, D, str(dtype)[6:], causal, reps, t_sd)}

    dt_code = fo.DTYPE_F16 if dtype == torch.float16 else fo.DTYPE_BF16

    def bits(t):
        return t.view(torch.int16).numpy().view(np.uint16)

    fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t0 = time.perf_counter()
    fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t_pass = time.perf_counter() - t0
    reps = int(max(1, min(40, budget_s / max(t_pass, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fo.fwd_c(bits(q), bits(k), bits(v), dt_code, causal, nthreads=threads)
    t_port = time.perf_counter() - t0
    port = {"value": round(flops * reps / t_port / 1e12, 5), "unit": "TFLOPS", "cores": threads, "kind": "port",
            "sample": "oracle/fa2_oracle.c (Br=32 Bc=64, OpenMP) on the same %d heads x %d passes, %.2f s" % (heads, reps, t_port)}
    return ref, port


def read_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def dense_head_check(q, k, v, o, causal, b, h):
    """max |o - dense fp32 attention| on one (batch, head) of the timed output."""
    qf, kf, vf = q[b, h].float(), k[b, h].float(), v[b, h].float()
    s = (qf @ kf.t()) * (q.shape[-1] ** -0.5)
    if causal:
        n, m = s.shape
        s = s.masked_fill(torch.ones(n, m, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    ref = torch.softmax(s, -1) @ vf
    return float((o[b, h].float() - ref).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)   # past the DVFS ramp-up after idle
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: c2 with one rank; c5 (+ the c2 weak-scaling figure under `weak_c2`) with --gpus > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-launches", type=int, default=400, help="launches of the post-region steady re-timing (0 = skip)")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="GPU seconds of the same loop after the steady re-timing (evidence: utilisation samplers, thermal steady state; 0 = skip)")
    ap.add_argument("--collectives", action="store_true", help="also time scatter_batch / gather_batch (reported separately)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--same-device", action="store_true", help="every rank on cuda:0 (dry run of the multi-rank path on one GPU)")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--settle", type=int, default=2000,
                    help="upper bound of the launches of the operator issued (and checked bit-identical) BEFORE the W warm-up steps: chunks of 25 for at least "
                         "150 ms and until six consecutive chunks agree within 1 %%, i.e. until the chip's power management has settled on this kernel's "
                         "load (0 = none: the cold-start number)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group, barriers, reductions and --collectives even with ONE rank (executes the RCCL path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` with no launcher around it: become the launcher.  One rank per GPU under
        # torch.distributed.run on a free loopback port; rank 0 of the children prints the JSON line, this process is replaced.
        if not args.same_device:
            have = torch.cuda.device_count()
            if have < args.gpus:
                sys.exit("bench.py --gpus %d: only %d GPU(s) visible (use --backend gloo --same-device for a one-GPU dry run)" % (args.gpus, have))
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.realpath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU: the attention operator has no CPU path")
    dev_index = 0 if args.same_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    host_sync = args.backend == "gloo"     # gloo: barriers / reductions on CPU tensors
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if use_dist:
            if host_sync:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])

    def max_over_ranks(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if host_sync else device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    from rocwmma_fattn.FlashAttn import FlashAttentionFunction
    from rocwmma_fattn.shard import gather_batch, scatter_batch, shard_bounds
    attn = FlashAttentionFunction.apply

    default_multi = args.workload is None and world > 1      # N > 1 and no --workload: c5 (BASELINE.json configs[4]) is the line, c2 weak rides along
    if args.workload is None:
        args.workload = "c5" if world > 1 else "c2"

    def timed_region(q, k, v, causal):
        """The contract: W untimed steps, then exactly K timed steps between barrier + synchronize.  Returns (wall seconds, kernel ms per step
        by HIP events on the launch stream, last output)."""
        o = None
        for _ in range(args.warmup):
            o = attn(q, k, v, None, causal)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                      # torch's current stream == the stream the kernel is launched on
        for _ in range(args.steps):
            o = attn(q, k, v, None, causal)
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / args.steps, o

    def measure(workload, extras):
        """gate -> cold region -> settle -> contractual region (-> per-launch events, steady: `extras`) of one workload on this rank's shard."""
        B, H, N, D, dtype, causal, scaling = WORKLOADS[workload]
        if scaling == "strong":
            lo, hi = shard_bounds(B, world, rank)
            B_local, B_global = hi - lo, B
        else:
            B_local, B_global = B, B * world
        cfg_idx = sorted(WORKLOADS).index(workload) + 1
        g = torch.Generator(device=device).manual_seed(1234 + cfg_idx + rank)
        q, k, v = (torch.rand((B_local, H, N, D), generator=g, device=device, dtype=torch.float32).to(dtype)
                   for _ in range(3))
        r = {"workload": workload, "B": B, "H": H, "N": N, "D": D, "dtype": dtype, "causal": causal, "scaling": scaling,
             "B_local": B_local, "B_global": B_global, "cfg_idx": cfg_idx, "q": q, "k": k, "v": v}

        # ---- parity gate BEFORE anything is timed: every (batch, head) of one forward call (a strided sample of 64 when there are
        #      more) against dense fp32 attention on the GPU.
        check_tol = 2e-3 if dtype == torch.float16 else 1.6e-2
        o = attn(q, k, v, None, causal)
        n_heads = B_local * H
        picks = list(range(n_heads)) if n_heads <= 64 else [int(i * n_heads / 64) for i in range(64)]
        gate_err = max(dense_head_check(q, k, v, o, causal, i // H, i % H) for i in picks)
        assert gate_err <= check_tol, "output differs from dense fp32 attention before timing: %g" % gate_err
        o_gate = o
        launches = 1

        # ---- cold region: the contract's W + K straight after the gate, nothing settled — what rounds 1-3 reported as `value`, and what a harness
        #      that only knows W warm-up launches sees.  Reported under `cold`; it also is the first stretch of load the settling below builds on.
        cold_elapsed, cold_kernel_ms, _ = timed_region(q, k, v, causal)
        launches += args.warmup + args.steps

        # ---- settling, stated plainly: after idle the chip boosts, overshoots its power budget, throttles and needs ~100 launches (25 ms) of THIS load to
        #      find its steady clock (profiles/r03_clock_settling.txt: the first 30 launches of a process run 15 % slower than the 100th; on other boxes
        #      the dip comes later: profiles/r12_bench_driver_args_fixed150.json — 150 launches ahead put the 20 timed ones INTO it, 0.2506 ms against
        #      0.2099 steady; an idle gap of <= 1 ms does not restart the transient, 5 ms does: tools/settle_probe.py).  Five warm-up launches cannot cover
        #      that, so the operator is first run in chunks of 25 for at least 150 ms AND until six consecutive chunks agree within 1 % (at most `--settle`
        #      launches) — doubling as a determinism gate: the last result must equal the first bit for bit.
        #      `--settle 0` skips it (then `value` is a second cold region); the line reports what was done.
        settle_info = {"max_launches": args.settle, "launches": 0}
        o_settled = None
        if args.settle > 0:
            chunk, times = 25, []
            while settle_info["launches"] < args.settle:
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(chunk):
                    o_s = attn(q, k, v, None, causal)
                c1.record()
                torch.cuda.synchronize()
                times.append(c0.elapsed_time(c1) / chunk)
                settle_info["launches"] += chunk
                # settled: at least 150 ms under this load (the throttle episode comes 35 .. 50 ms after the load starts and lasts 10 .. 20 ms: a plateau
                # before it fooled a pure convergence test, profiles/r12_bench_driver_args_settle_converged_early.json) AND the last six chunks agree
                # within 1 % (the transient — boost, throttle, recovery — moves the launch time by 5 .. 18 %)
                if sum(times) * chunk >= 150.0 and max(times[-6:]) <= 1.01 * min(times[-6:]):
                    break
            # (the bit-identity check of o_s against the gate's output waits until the timed region is over: the FIRST launch of a kernel the process has not
            #  run yet — torch.equal's compare — loads its code object, the GPU idles for milliseconds and the next ~40 attention launches run 10 .. 18 %
            #  slower, tools/settle_probe.py: nothing that is new to the process may sit between here and the timed region)
            o_settled = o_s
            settle_info.update({"chunk": chunk, "first_chunk_ms": round(times[0], 5), "slowest_chunk_ms": round(max(times), 5),
                                "last_chunk_ms": round(times[-1], 5), "gpu_ms": round(sum(times) * chunk, 1),
                                "converged": sum(times) * chunk >= 150.0 and max(times[-6:]) <= 1.01 * min(times[-6:])})
            launches += settle_info["launches"]
            del o_s

        # ---- contractual region: W untimed warm-up steps, then exactly K timed steps between barrier + synchronize
        elapsed, kernel_ms, o = timed_region(q, k, v, causal)
        launches += args.warmup
        r.update({"elapsed": elapsed, "kernel_ms": kernel_ms, "cold_elapsed": cold_elapsed, "cold_kernel_ms": cold_kernel_ms,
                  "settle": settle_info, "warmup_effective": launches, "gate_err": gate_err, "n_picks": len(picks), "n_heads": n_heads,
                  "check_tol": check_tol})
        assert torch.equal(o_gate, o), "the operator is not deterministic run to run (gate vs timed output)"
        if o_settled is not None:
            assert torch.equal(o_settled, o), "the operator is not deterministic run to run (settling phase)"
            settle_info["bit_identical_to_timed_output"] = True
        del o_settled, o_gate
        if not extras:
            assert torch.isfinite(o.float()).all(), "non-finite attention output"
            r["check_err"] = dense_head_check(q, k, v, o, causal, B_local - 1, H - 1)
            assert r["check_err"] <= check_tol, "timed output differs from dense fp32 attention: %g" % r["check_err"]
            return r
        # per-launch durations: K further launches straight after the region, one event between consecutive launches (kept out
        # of the region itself: an event between two launches costs 1.6 % — tools/event_overhead.py)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            o2 = attn(q, k, v, None, causal)
            evs[i + 1].record()
        torch.cuda.synchronize()
        r["per_launch"] = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        assert torch.equal(o, o2), "the operator is not deterministic run to run"
        assert torch.isfinite(o.float()).all(), "non-finite attention output"
        r["check_err"] = dense_head_check(q, k, v, o, causal, B_local - 1, H - 1)
        assert r["check_err"] <= check_tol, "timed output differs from dense fp32 attention: %g" % r["check_err"]

        # ---- steady re-timing (evidence, not the metric): same loop, after the region, clock settled
        r["steady"] = None
        if args.steady_launches > 0:
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(args.steady_launches):
                o = attn(q, k, v, None, causal)
            s1.record()
            torch.cuda.synchronize()
            steady_ms = s0.elapsed_time(s1) / args.steady_launches
            r["steady"] = {"launches": args.steady_launches, "kernel_ms": round(steady_ms, 5),
                           "tflops": round(attention_flops(B_local, H, N, N, D, causal) / (steady_ms * 1e-3) / 1e12, 2)}
        # ---- sustained load (evidence, not the metric): the same loop for --sustain-seconds of GPU time, in chunks — long enough for an outside
        #      utilisation sampler (rocm-smi every few seconds) to see the device busy, and for the clock / temperature to reach their steady state
        r["sustained"] = None
        if args.sustain_seconds > 0 and extras and world == 1:
            chunk, chunks, total_ms = 500, [], 0.0
            while total_ms < args.sustain_seconds * 1e3 and len(chunks) < 200:
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(chunk):
                    o = attn(q, k, v, None, causal)
                s1.record()
                torch.cuda.synchronize()
                chunks.append(s0.elapsed_time(s1) / chunk)
                total_ms += chunks[-1] * chunk
            mean_ms = total_ms / (chunk * len(chunks))
            r["sustained"] = {"launches": chunk * len(chunks), "gpu_seconds": round(total_ms * 1e-3, 2), "kernel_ms": round(mean_ms, 5),
                              "tflops": round(attention_flops(B_local, H, N, N, D, causal) / (mean_ms * 1e-3) / 1e12, 2),
                              "first_chunk_ms": round(chunks[0], 5), "last_chunk_ms": round(chunks[-1], 5), "slowest_chunk_ms": round(max(chunks), 5)}
        return r

    m = measure(args.workload, True)
    B, H, N, D, dtype, causal, scaling = (m[x] for x in ("B", "H", "N", "D", "dtype", "causal", "scaling"))
    B_local, B_global, cfg_idx = m["B_local"], m["B_global"], m["cfg_idx"]
    q, k, v = m["q"], m["k"], m["v"]
    elapsed, kernel_ms, per_launch, steady, settle_info = m["elapsed"], m["kernel_ms"], m["per_launch"], m["steady"], m["settle"]
    check_err, check_tol, gate_err = m["check_err"], m["check_tol"], m["gate_err"]

    # ---- edge transfers (SURVEY §8e timing rule): reported separately, never part of `value`
    coll = None
    if args.collectives and use_dist:
        shape = (B_global, H, N, D)
        # gloo (the dry runs) moves CPU tensors: the slabs are staged through host memory there; RCCL moves device tensors
        cdev = torch.device("cpu") if host_sync else device
        full = [torch.rand(shape, device=device, dtype=torch.float32).to(dtype).to(cdev) for _ in range(3)] if rank == 0 else [None] * 3
        for _ in range(2):            # first pass = warm-up (communicator setup)
            torch.cuda.synchronize(); barrier(); t0c = time.perf_counter()
            slabs = [scatter_batch(t, shape, dtype, cdev, src=0, always_collective=True).to(device) for t in full]
            torch.cuda.synchronize(); barrier(); t_sc = time.perf_counter() - t0c
            o_loc = attn(*slabs, None, causal)
            torch.cuda.synchronize(); barrier(); t0c = time.perf_counter()
            o_full = gather_batch(o_loc.to(cdev), B_global, always_collective=True)
            torch.cuda.synchronize(); barrier(); t_ga = time.perf_counter() - t0c
        lo_c, hi_c = shard_bounds(B_global, world, rank)
        assert torch.equal(o_full[lo_c:hi_c].to(device), o_loc), "gathered output differs from the local slab"
        if rank == 0:
            assert all(torch.equal(sl, t[lo_c:hi_c].to(device)) for sl, t in zip(slabs, full)), "scattered slab differs from the root's slice"
        t_sc, t_ga = max_over_ranks([t_sc, t_ga])
        nbytes = B_global * H * N * D * 2
        coll = {"scatter_qkv_ms": round(t_sc * 1e3, 3), "gather_o_ms": round(t_ga * 1e3, 3),
                "scatter_bytes": 3 * nbytes, "gather_bytes": nbytes, "backend": args.backend, "world": world,
                "note": "root-held [B_global,H,N,D] tensors: dist.scatter x3, all_gather_into_tensor x1 (%s); not part of value"
                        % ("host-staged over gloo" if host_sync else "device tensors over RCCL")}
        del full, slabs, o_full

    # ---- informational backward (reference harness style, bench_with_sdpa.py:78-88), outside the timed region
    bwd = None
    if world == 1 and not args.no_backward:
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
        fwd_flops = attention_flops(B_local, H, N, N, D, causal)
        useful = 2.5 * fwd_flops / (bwd_ms * 1e-3) / 1e12           # the reference's convention: backward = 2.5 x forward = 5 GEMMs
        executed = 3.5 * fwd_flops / (bwd_ms * 1e-3) / 1e12         # what the two passes execute: 3 (dQ pass) + 4 (dK/dV pass) = 7 GEMMs
        bwd = {"bwd_ms": round(bwd_ms, 4), "bwd_tflops": round(useful, 1),
               "roofline": {"bound": "mfma", "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "useful": round(useful, 1), "frac_useful": round(useful / MFMA_PEAK_TFLOPS, 4),
                            "executed": round(executed, 1), "frac_executed": round(executed / MFMA_PEAK_TFLOPS, 4),
                            "gemms_executed": 7, "gemms_counted": 5},
               "note": "bwd FLOPs = 2.5 x fwd (bench_with_sdpa.py:35-41); two deterministic passes (dQ: S, dP, dQ; dK/dV: S, dP, dV, dK) "
                       "recompute S and dP, so 7 GEMM-equivalents execute; time = wall of 30 autograd backward calls / 30 (HIP events)"}

    elapsed, kernel_ms, cold_elapsed, cold_kernel_ms = max_over_ranks([elapsed, kernel_ms, m["cold_elapsed"], m["cold_kernel_ms"]])
    # ---- N > 1 with no --workload: the c2 weak-scaling figure (every rank its own B2 H16 N4096 D128 shard) next to the c5 line
    weak = None
    if default_multi:
        del q, k, v
        m.pop("q"), m.pop("k"), m.pop("v")
        torch.cuda.empty_cache()
        w = measure("c2", False)
        w_el, w_km, w_cel = max_over_ranks([w["elapsed"], w["kernel_ms"], w["cold_elapsed"]])
        w_flops = attention_flops(w["B_global"], w["H"], w["N"], w["N"], w["D"], w["causal"])
        weak = {"workload": "c2: B2 H16 N4096 D128 f16 causal=False per GPU", "scaling": "weak", "global_batch": w["B_global"],
                "value": round(w_flops * args.steps / w_el / 1e12, 2), "unit": "TFLOPS", "ms_per_step": round(w_el / args.steps * 1e3, 5),
                "kernel_ms": round(w_km, 5), "cold_value": round(w_flops * args.steps / w_cel / 1e12, 2),
                "pct_of_mfma_roofline": round(100.0 * w_flops * args.steps / w_el / 1e12 / (MFMA_PEAK_TFLOPS * world), 2),
                "warmup_effective": w["warmup_effective"]}
    rank_info = None
    if use_dist:
        # which device every rank ran on, as the process group saw it (evidence that N ranks on N GPUs took part)
        me = {"rank": rank, "device": "cuda:%d" % dev_index, "name": torch.cuda.get_device_name(dev_index),
              "uuid": str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", ""))}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, me)
        rank_info = {"backend": args.backend, "world_size": dist.get_world_size(), "ranks": gathered}

    if rank == 0:
        flops_global = attention_flops(B_global, H, N, N, D, causal)
        flops_local = attention_flops(B_local, H, N, N, D, causal)
        value = flops_global * args.steps / elapsed / 1e12
        achieved = flops_local / (kernel_ms * 1e-3) / 1e12
        peak_rec = read_json("mfma_peak.json")
        # (fp16 head-dim-128 launches run the 16x16x32 MFMA since round 5: the MFMA-only rate of that instruction where it was measured)
        sustained = (peak_rec.get("sustained_tflops_f16_16x16x32") or peak_rec.get("sustained_tflops_f16")) if dtype == torch.float16 else peak_rec.get("sustained_tflops_bf16")
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": read_json("hbm_traffic.json").get(args.workload, {}).get("hbm_bytes_per_launch"),
                "traffic_source": "profiles/hbm_traffic.json: the committed rocprofv3 --pmc passes of this workload (counters cannot be read inside this run)",
                "kernel_ms": round(kernel_ms, 5), "flops_per_launch": flops_local}
        if sustained:
            roof["sustained_peak"] = sustained
            roof["frac_sustained"] = round(achieved / sustained, 4)
        line = {
            "metric": "fwd attention TFLOPS (and % MFMA roofline) at B2 H16 N4096 D128 fp16",
            "value": round(value, 2), "unit": "TFLOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_effective": m["warmup_effective"],
            "cold": {"value": round(flops_global * args.steps / cold_elapsed / 1e12, 2), "ms_per_step": round(cold_elapsed / args.steps * 1e3, 5),
                     "kernel_ms": round(cold_kernel_ms, 5), "frac": round(flops_local / (cold_kernel_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                     "note": "the same W + K region straight after the parity gate, before the settle phase: the protocol of rounds 1-3"},
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": "%s: B%d H%d N%d D%d %s causal=%s per GPU, BHND, torch.rand U[0,1)"
                                   % (args.workload, B_local, H, N, D, str(dtype)[6:], causal),
                       "global_batch": B_global, "parallelism": "batch-shard x%d (no data-path collective)" % world},
            "roofline": roof,
            "sequence": ["parity_gate", "cold_warmup", "cold_timed", "settle", "warmup", "timed", "per_launch_events", "steady", "sustained", "backward_info", "cpu_baseline"],
            "settle": settle_info,
            "pct_of_mfma_roofline": round(100.0 * value / (MFMA_PEAK_TFLOPS * world), 2),
            "launch_ms": {"min": round(min(per_launch), 5), "median": round(statistics.median(per_launch), 5),
                          "max": round(max(per_launch), 5), "first": round(per_launch[0], 5), "last": round(per_launch[-1], 5)},
            "check": {"max_abs_err_vs_dense_fp32": round(check_err, 6), "tol": check_tol, "head": [B_local - 1, H - 1],
                      "gate_before_timing": {"heads_checked": m["n_picks"], "of": m["n_heads"], "max_abs_err": round(gate_err, 6)}},
        }
        if steady is not None:
            line["steady"] = steady
        if m.get("sustained") is not None:
            line["sustained"] = m["sustained"]
        if weak is not None:
            line["weak_c2"] = weak
        if rank_info is not None:
            line["dist"] = rank_info
            if args.force_dist and world == 1:
                line["dist"]["note"] = "process group, barriers and reductions executed with one rank (--force-dist)"
        if args.backend != "nccl" or args.same_device:
            line["dry_run"] = "backend=%s same_device=%s: control-flow check of the multi-rank path, not a scaling measurement" % (
                args.backend, args.same_device)
        if coll is not None:
            line["collectives"] = coll
        if bwd is not None:
            line["backward"] = bwd
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["cpu_port"] = cpu_baseline(B * H, N, D, causal, dtype, 1234 + cfg_idx)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
