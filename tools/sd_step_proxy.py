"""Attention-only proxy of one Stable-Diffusion UNet denoising step.

The reference's only published numbers are ComfyUI it/s tables (README.md:114-154); neither ComfyUI nor model weights
exist in this environment, so this tool replays what the attention hook would see during ONE UNet step — the sequence
of self- and cross-attention calls of SDXL-base at 1024x1024 and of SD1.5 at 512x512, classifier-free-guidance batch 2,
in the [B, N, heads*dim_head] layout both front ends use — through `rocwmma_fattn.sd_hook.attention_bnhd` (eager and
as a captured HIP graph) and through the same hosts' default path, torch SDPA on [B, H, N, D] views.  It reports the
attention time per step; everything else in a UNet step (convolutions, GEMMs, norms) is outside the hot path.

    python tools/sd_step_proxy.py [--iters 20] > profiles/r02_sd_step_proxy.json

Call lists (transformer blocks per resolution, each = 1 self-attention + 1 cross-attention on 77 text tokens):
  SDXL-base 1024^2 (latent 128^2): 64x64 tokens, 10 heads x 64: 2*2 (down) + 3*2 (up) = 10 blocks;
                                   32x32 tokens, 20 heads x 64: 2*10 (down) + 10 (mid) + 3*10 (up) = 60 blocks
  SD1.5 512^2 (latent 64^2):       64x64, 8 x 40: 2 + 3 = 5;  32x32, 8 x 80: 5;  16x16, 8 x 160: 5;  8x8, 8 x 160: 1 (mid)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.sd_hook import attention_bnhd  # noqa: E402

MODELS = {
    # name: [(tokens, heads, dim_head, blocks)], text tokens
    "sdxl_base_1024": ([(4096, 10, 64, 10), (1024, 20, 64, 60)], 77),
    "sd15_512": ([(4096, 8, 40, 5), (1024, 8, 80, 5), (256, 8, 160, 5), (64, 8, 160, 1)], 77),
}


def call_list(model, batch=2):
    levels, text = MODELS[model]
    calls = []
    for (n, h, d, blocks) in levels:
        for _ in range(blocks):
            calls.append((batch, n, n, h, d))        # self-attention
            calls.append((batch, n, text, h, d))     # cross-attention
    return calls


def sdpa_host_path(q, k, v, heads):
    """What ComfyUI's attention_pytorch does with the same inputs: view as [B, H, N, D], SDPA, back to [B, N, H*D]."""
    b, nq, inner = q.shape
    d = inner // heads
    q4, k4, v4 = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q4, k4, v4)
    return o.transpose(1, 2).reshape(b, nq, inner)


def time_step(fn_step, iters):
    for _ in range(3):
        fn_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn_step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {"_comment": "attention calls of ONE UNet step (CFG batch 2, fp16), tools/sd_step_proxy.py on %s; ms per step" % torch.cuda.get_device_name(0)}
    for model in MODELS:
        calls = call_list(model)
        g = torch.Generator(device=dev).manual_seed(11)
        bufs = {}
        for (b, nq, nkv, h, d) in calls:       # one set of tensors per distinct call shape (activations of a real step differ, the shapes do not)
            if (nq, nkv, h, d) not in bufs:
                bufs[(nq, nkv, h, d)] = (torch.randn((b, nq, h * d), generator=g, device=dev, dtype=torch.float16),
                                         torch.randn((b, nkv, h * d), generator=g, device=dev, dtype=torch.float16),
                                         torch.randn((b, nkv, h * d), generator=g, device=dev, dtype=torch.float16))
        flops = sum(4.0 * b * h * nq * nkv * d for (b, nq, nkv, h, d) in calls)

        def step_fa2():
            o = None
            for (b, nq, nkv, h, d) in calls:
                q, k, v = bufs[(nq, nkv, h, d)]
                o = attention_bnhd(q, k, v, h)
            return o

        def step_sdpa():
            o = None
            for (b, nq, nkv, h, d) in calls:
                q, k, v = bufs[(nq, nkv, h, d)]
                o = sdpa_host_path(q, k, v, h)
            return o

        # parity of every distinct call shape before timing
        worst = 0.0
        for (nq, nkv, h, d), (q, k, v) in bufs.items():
            worst = max(worst, float((attention_bnhd(q, k, v, h).float() - sdpa_host_path(q.float(), k.float(), v.float(), h)).abs().max()))
        assert worst <= 5e-3, worst
        fa2_gpu, fa2_wall = time_step(step_fa2, args.iters)
        sd_gpu, sd_wall = time_step(step_sdpa, args.iters)
        # the same call sequence as one captured HIP graph: the operator only enqueues on the caller's stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            step_fa2()
            with torch.cuda.graph(graph, stream=side):
                step_fa2()
        torch.cuda.current_stream().wait_stream(side)
        gr_gpu, gr_wall = time_step(graph.replay, args.iters)
        sgraph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            step_sdpa()
            with torch.cuda.graph(sgraph, stream=side):
                step_sdpa()
        torch.cuda.current_stream().wait_stream(side)
        sg_gpu, sg_wall = time_step(sgraph.replay, args.iters)
        out[model] = {
            "attention_calls_per_step": len(calls), "attention_gflop_per_step": round(flops / 1e9, 1),
            "fa2_eager_ms": round(fa2_gpu, 3), "fa2_eager_host_wall_ms": round(fa2_wall, 3),
            "fa2_graph_ms": round(gr_gpu, 3),
            "sdpa_eager_ms": round(sd_gpu, 3), "sdpa_eager_host_wall_ms": round(sd_wall, 3),
            "sdpa_graph_ms": round(sg_gpu, 3),
            "speedup_eager": round(sd_gpu / fa2_gpu, 3), "speedup_graph": round(sg_gpu / gr_gpu, 3),
            "fa2_graph_tflops": round(flops / (gr_gpu * 1e-3) / 1e12, 1),
            "max_abs_diff_vs_sdpa_fp32": round(worst, 6),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
