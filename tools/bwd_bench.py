"""Backward A/B on the GPU (developer tool): times fa2_bwd through the C-ABI, interleaved in one process, for
  * several settings of the library option "asm" (bit 1: hand-scheduled backward, bits 2 / 3: ... without its dQ / dK-dV pass), and / or
  * several builds of the library (tools/kbench.py build NAME:bgen=...  ->  tools/variants/NAME.so),
one pass at a time if asked (option "bwd_parts": 1 = dQ pass only, 2 = dK/dV pass only).
    python tools/bwd_bench.py [--cfg c2,c3,c4] [--rounds 5] [--iters 10] [--modes 3,1] [--parts 3] [--libs base,var1]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402

CFGS = {"c2": (2, 16, 4096, 128, torch.float16, False), "c3": (2, 16, 4096, 128, torch.bfloat16, True),
        "c4": (1, 32, 8192, 128, torch.float16, True), "b8": (8, 16, 4096, 128, torch.float16, False),
        "n1k": (2, 16, 1024, 128, torch.float16, False), "n2kc": (1, 8, 2048, 128, torch.float16, True),
        "n2k": (2, 16, 2048, 128, torch.float16, False), "n512": (2, 16, 512, 128, torch.float16, False),
        "h8n4k": (1, 8, 4096, 128, torch.float16, False), "h8n8k": (1, 8, 8192, 128, torch.float16, False),
        "d64": (2, 16, 4096, 64, torch.float16, False), "d64c": (2, 16, 4096, 64, torch.bfloat16, True), "sdxl": (2, 10, 4096, 64, torch.float16, False),
        "sd15": (2, 8, 4096, 40, torch.float16, False), "d64n8k": (1, 24, 8192, 64, torch.float16, False), "sdxl32": (2, 20, 1024, 64, torch.float16, False)}
GEMMS = {1: 3, 2: 4, 3: 7}      # GEMM-equivalents executed by the passes (a forward is 2)


def load(name):
    if name == "base":
        return _fa2_lib.load()
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
    for sym, (restype, argtypes) in _fa2_lib.SYMBOLS.items():
        fn = getattr(lib, sym)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c2,c3,c4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--modes", default="3,1")
    ap.add_argument("--parts", type=int, default=3)
    ap.add_argument("--libs", default="base")
    ap.add_argument("--folds", default="", help="settings of option 'fold' to compare, e.g. 1,0 (default: leave the option alone)")
    ap.add_argument("--fill", default="rand", choices=["rand", "randn"])
    a = ap.parse_args()
    libs = {n: load(n) for n in a.libs.split(",")}
    modes = [int(x) for x in a.modes.split(",")]
    folds = [int(x) for x in a.folds.split(",") if x] or [None]
    arms = [(ln, m, f) for ln in libs for m in modes for f in folds]
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mk = torch.rand if a.fill == "rand" else torch.randn
    for cname in a.cfg.split(","):
        B, H, N, D, dt, causal = CFGS[cname]
        q, k, v, do = (mk((B, H, N, D), device=dev, dtype=torch.float32).to(dt) for _ in range(4))
        o = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        delta = torch.empty_like(lse)
        s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
        s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
        code = 0 if dt == torch.float16 else 1
        sc = float(D ** -0.5)
        base = libs[a.libs.split(",")[0]]
        _fa2_lib.check(base.fa2_fwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                                    s3(q), s3(k), s3(v), s3(o), s2, sc, int(causal), stream))
        outs = {}

        def bwd(arm, parts):
            ln, mode, fold = arm
            lib = libs[ln]
            if fold is not None:
                lib.fa2_set_option(b"fold", fold)
            dq, dk, dv = outs.setdefault(arm, tuple(torch.zeros_like(q) for _ in range(3)))
            lib.fa2_set_option(b"asm", mode)
            lib.fa2_set_option(b"bwd_parts", parts)
            _fa2_lib.check(lib.fa2_bwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                       dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, N, N, D,
                                       s3(q), s3(k), s3(v), s3(o), s3(do), s3(dq), s3(dk), s3(dv), s2, sc, int(causal), stream))

        times = {arm: [] for arm in arms}
        for arm in arms:
            bwd(arm, 3)                       # a full call first: fills delta for parts = 2 runs and the outputs for the comparison
            for _ in range(2):
                bwd(arm, a.parts)
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for arm in arms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    bwd(arm, a.parts)
                e1.record()
                torch.cuda.synchronize()
                times[arm].append(e0.elapsed_time(e1) / a.iters)
        for lib in libs.values():
            if folds != [None]:
                lib.fa2_set_option(b"fold", 1)
            lib.fa2_set_option(b"asm", 963)
            lib.fa2_set_option(b"bwd_parts", 3)
        fwd_flops = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
        print("-- %s: B%d H%d N%d D%d %s causal=%d  parts=%d" % (cname, B, H, N, D, str(dt)[6:], causal, a.parts))
        ref = outs[arms[-1]]
        for arm in arms:
            med = statistics.median(times[arm])
            diff = max(float((x.float() - y.float()).abs().max()) for x, y in zip(outs[arm], ref))
            print("   %-10s asm=%-2d fold=%-4s median %8.1f us  best %8.1f us  executed %6.1f TF  (reference convention, 2.5x fwd: %6.1f TF)   max|grad - grad[last arm]| %.2e"
                  % (arm[0], arm[1], arm[2], med * 1e3, min(times[arm]) * 1e3, GEMMS[a.parts] / 2 * fwd_flops / med / 1e9,
                     2.5 * fwd_flops / med / 1e9 if a.parts == 3 else float("nan"), diff))


if __name__ == "__main__":
    main()
