"""KV-split tail A/B on the GPU (developer tool): FlashAttentionFunction.apply with the library option "split" on and off, interleaved in
one process, on shapes whose grid of 256-row workgroups leaves a partly filled last round — the reference harness's own sweep
(B1 H24 D64, bench_with_sdpa.py:201-224), Stable-Diffusion shapes, D = 128 — and a few that must not change.
    python tools/split_ab.py [--rounds 5] [--iters 50]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

SHAPES = [("sdxl-64x64 B2 H10 N4096 D64", 2, 10, 4096, 4096, 64), ("sdxl-64x64 B4 H10", 4, 10, 4096, 4096, 64), ("sd15-64x64 B2 H8 D40", 2, 8, 4096, 4096, 40),
          ("sd15-64x64 B3 H8 D40", 3, 8, 4096, 4096, 40), ("H24 N4096 D128", 1, 24, 4096, 4096, 128), ("H24 N8192 D128", 1, 24, 8192, 8192, 128),
          ("H40 N2048 D128", 1, 40, 2048, 2048, 128), ("c2 B2 H16 N4096 D128", 2, 16, 4096, 4096, 128), ("B2 H20 N2048 D80", 2, 20, 2048, 2048, 80)]
SHAPES += [("harness H24 D64 N%d" % n, 1, 24, n, n, 64) for n in (2048, 2560, 3072, 3584, 4096, 4608, 5120, 5632, 6144, 6656, 7168)]


def timed(fn, iters):
    for _ in range(max(3, iters // 5)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    lib = _fa2_lib.load()
    dev = torch.device("cuda", 0)
    print("%-32s %6s %9s %9s %7s %9s %9s   ws MB" % ("shape", "items", "split us", "plain us", "x", "split TF", "plain TF"))
    for name, B, H, N, Nkv, D in SHAPES:
        g = torch.Generator(device=dev).manual_seed(3)
        q = torch.rand((B, H, N, D), generator=g, device=dev).half()
        k, v = (torch.rand((B, H, Nkv, D), generator=g, device=dev).half() for _ in range(2))
        ts = {1: [], 0: []}
        for _ in range(a.rounds):
            for mode in (1, 0):
                with _fa2_lib.options(split=mode):
                    ts[mode].append(timed(lambda: FlashAttentionFunction.apply(q, k, v, None, False), a.iters))
        t1, t0 = statistics.median(ts[1]), statistics.median(ts[0])
        fl = 4.0 * B * H * N * Nkv * D
        need = lib.fa2_fwd_workspace_bytes(0, B, H, N, Nkv, D + (-D % 8), 0)
        print("%-32s %6d %9.1f %9.1f %7.3f %9.1f %9.1f   %.1f" % (name, B * H * ((N + 255) // 256), t1, t0, t0 / t1, fl / t1 / 1e6, fl / t0 / 1e6, need / 2 ** 20))


if __name__ == "__main__":
    main()
