"""A/B of the TRIMMED kernels (head dims below the kernel's HD run only the k-steps / column blocks that hold real columns:
fa2_fwd_kernel.hip.h KSQ / DTN / RTD, fa2_bwd_kernel.hip.h KSN / DTN) against a build without them (developer tool):

    python tools/kbench.py build notrim:-DFA2_TRIM=0             (CPU container)
    python tools/trim_ab.py [--bwd] [--rounds 5] [--iters 20]    (GPU box)

Every shape is checked against dense fp32 attention (forward: O and LSE; --bwd: the autograd gradients) on both builds, then timed
interleaved through the C-ABI (fa2_fwd / fa2_bwd)."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402

# (label, B, H, Nq, Nkv, D, dtype, causal)
SHAPES = [("ref D-scan", 1, 24, 4096, 4096, d, torch.float16, False) for d in (16, 32, 40, 48, 80, 96, 144, 160, 176, 192, 208, 224)] + [
    ("sd15-64x64", 2, 8, 4096, 4096, 40, torch.float16, False),
    ("sd15-32x32", 2, 8, 1024, 1024, 80, torch.float16, False),
    ("bf16 causal", 2, 16, 4096, 4096, 72, torch.bfloat16, True),
    ("sd15-16x16", 2, 8, 256, 256, 160, torch.float16, False),
    ("sd15-cross", 2, 8, 4096, 77, 40, torch.float16, False),
    ("sd21-64x64", 2, 5, 4096, 4096, 64, torch.float16, False),
    ("bf16 causal", 2, 16, 4096, 4096, 96, torch.bfloat16, True),
    ("bf16 causal", 2, 16, 2048, 2048, 192, torch.bfloat16, True),
    ("f16 causal", 4, 16, 2048, 2048, 40, torch.float16, True),
    ("ragged", 3, 5, 1000, 777, 88, torch.float16, False),
    ("ragged", 3, 5, 1000, 777, 152, torch.bfloat16, True),
    ("ragged", 3, 5, 333, 1111, 24, torch.float16, False),
    ("f16 causal", 2, 16, 2048, 2048, 160, torch.float16, True),
    ("small grid", 1, 8, 1024, 1024, 160, torch.float16, False),
    ("small grid", 1, 8, 1024, 1024, 192, torch.bfloat16, True),
    ("d-scan bf16", 1, 24, 4096, 4096, 136, torch.bfloat16, False),
    ("ref D-scan", 1, 24, 4096, 4096, 240, torch.float16, False),
    ("ref D-scan", 1, 24, 4096, 4096, 256, torch.float16, False),
    ("bf16 causal", 2, 16, 4096, 4096, 256, torch.bfloat16, True),
    ("b8 d256", 8, 16, 2048, 2048, 256, torch.float16, False),
    ("d-scan >256", 1, 24, 4096, 4096, 264, torch.float16, False),
    ("d-scan >256", 1, 24, 4096, 4096, 320, torch.float16, False),
    ("d-scan >256", 1, 24, 4096, 4096, 384, torch.float16, False),
    ("d-scan >256", 1, 24, 4096, 4096, 448, torch.float16, False),
    ("d-scan >256", 1, 24, 4096, 4096, 512, torch.float16, False),
    ("bf16 causal", 2, 8, 2048, 2048, 320, torch.bfloat16, True),
    ("ragged", 2, 3, 700, 900, 392, torch.bfloat16, True),
    ("sd vae", 1, 1, 4096, 4096, 512, torch.float16, False),
]


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
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--libs", default="base,notrim")
    ap.add_argument("--dmin", type=int, default=0, help="only shapes with D >= this")
    ap.add_argument("--dmax", type=int, default=1 << 20, help="only shapes with D <= this")
    ap.add_argument("--bwd", action="store_true", help="time fa2_bwd (forward once, then the backward) instead of the forward")
    a = ap.parse_args()
    libs = {n: load(n) for n in a.libs.split(",")}
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(torch.cuda.get_device_name(0), "builds:", list(libs))
    for label, B, H, Nq, Nkv, D, dt, causal in SHAPES:
        if D < a.dmin or D > a.dmax:
            continue
        g = torch.Generator(device=dev).manual_seed(7)
        q = torch.rand((B, H, Nq, D), generator=g, device=dev, dtype=torch.float32).to(dt)
        k, v = (torch.rand((B, H, Nkv, D), generator=g, device=dev, dtype=torch.float32).to(dt) for _ in range(2))
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
        s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
        s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
        code = 0 if dt == torch.float16 else 1
        sc = float(D ** -0.5)
        outs = {n: torch.full_like(q, float("nan")) for n in libs}

        do = torch.rand((B, H, Nq, D), generator=g, device=dev, dtype=torch.float32).to(dt) if a.bwd else None
        delta = torch.empty_like(lse)
        grads = {n: tuple(torch.full_like(t, float("nan")) for t in (q, k, v)) for n in libs} if a.bwd else {}

        def bwd(n):
            dq, dk, dv = grads[n]
            _fa2_lib.check(libs[n].fa2_bwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[n].data_ptr(), do.data_ptr(), lse.data_ptr(),
                                           dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nkv, D,
                                           s3(q), s3(k), s3(v), s3(outs[n]), s3(do), s3(dq), s3(dk), s3(dv), s2, sc, int(causal), stream))

        def fwd(n):
            _fa2_lib.check(libs[n].fa2_fwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[n].data_ptr(), lse.data_ptr(), B, H, Nq, Nkv, D,
                                           s3(q), s3(k), s3(v), s3(outs[n]), s2, sc, int(causal), stream))

        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * sc
        if causal:
            s = s.masked_fill(torch.ones(Nq, Nkv, dtype=torch.bool, device=dev).triu(1), float("-inf"))
        ref = torch.matmul(torch.softmax(s, -1), v.float())
        lref = torch.logsumexp(s, -1) * 1.4426950408889634
        del s
        errs = {}
        for n in libs:
            lse.fill_(float("nan"))
            fwd(n)
            torch.cuda.synchronize()
            errs[n] = (float((outs[n].float() - ref).abs().max()), float((lse - lref).abs().max()))
        if a.bwd:      # gradients of dense fp32 attention (autograd) as the reference of both builds
            qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
            sr = torch.matmul(qr, kr.transpose(-1, -2)) * sc
            if causal:
                sr = sr.masked_fill(torch.ones(Nq, Nkv, dtype=torch.bool, device=dev).triu(1), float("-inf"))
            torch.matmul(torch.softmax(sr, -1), vr).backward(do.float())
            del sr
            for n in libs:
                fwd(n)
                bwd(n)
                torch.cuda.synchronize()
                errs[n] = tuple(float((gg.float() - r.grad).abs().max() / max(1.0, float(r.grad.abs().max()))) for gg, r in zip(grads[n], (qr, kr, vr)))
        run = bwd if a.bwd else fwd
        times = {n: [] for n in libs}
        for n in libs:
            for _ in range(3):
                run(n)
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for n in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(n)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / a.iters)
        flops = 4.0 * B * H * Nq * Nkv * D * (0.5 if causal else 1.0) * (2.5 if a.bwd else 1.0)
        line = "%-11s B%d H%-2d N%4d/%-4d D%-3d %-8s c%d |" % (label, B, H, Nq, Nkv, D, str(dt)[6:], causal)
        for n in libs:
            med = statistics.median(times[n])
            if a.bwd:
                line += "  %s %8.1f us %7.1f TF  rel |dq dk dv - ref| %.1e %.1e %.1e |" % ((n, med * 1e3, flops / med / 1e9) + errs[n])
            else:
                line += "  %s %8.1f us %7.1f TF  |O-ref| %.1e |L-ref| %.1e |" % (n, med * 1e3, flops / med / 1e9, errs[n][0], errs[n][1])
        print(line)


if __name__ == "__main__":
    main()
