"""Head dims below the body's on the hand-scheduled 16x16x32 forward bodies (round 5; csrc/gen/fwd_m16_gen.py: trim_offsets; host.cpp: plan_range)
against the trimmed compiler-scheduled kernels (option asm bit 6 clear), interleaved through the C-ABI; every result checked against dense fp32.
    python tools/trim_asm_ab.py [--rounds 7] [--iters 20]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402

SHAPES = [("harness D-scan", 1, 24, 4096, d, torch.float16, False) for d in (40, 48, 56, 96, 104, 112, 120)] + [
    ("B2 H16 D-scan", 2, 16, 4096, d, torch.float16, False) for d in (24, 32, 40, 48, 56, 72, 80, 88, 96)] + [
    ("sd15-64x64", 2, 8, 4096, 40, torch.float16, False), ("sd15-64x64 B4", 4, 8, 4096, 40, torch.float16, False),
    ("causal", 2, 16, 4096, 48, torch.float16, True), ("causal", 2, 16, 4096, 112, torch.float16, True), ("bf16", 2, 16, 4096, 112, torch.bfloat16, False),
    ("bf16 causal", 2, 16, 4096, 96, torch.bfloat16, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = _fa2_lib.load()
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = lib.fa2_get_option(b"asm")
    for name, B, H, N, D, dt, causal in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(D)
        q, k, v = (torch.randn((B, H, N, D), generator=g).to(dt).to(dev) for _ in range(3))
        o = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
        s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
        code = 0 if dt == torch.float16 else 1

        def call():
            _fa2_lib.check(lib.fa2_fwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                                       s3(q), s3(k), s3(v), s3(o), s2, float(D ** -0.5), int(causal), stream))
        s = torch.matmul(q[:1, :2].float(), k[:1, :2].float().transpose(-1, -2)) * D ** -0.5
        if causal:
            s = s.masked_fill(torch.ones(N, N, dtype=torch.bool, device=dev).triu(1), float("-inf"))
        ref = torch.matmul(torch.softmax(s, -1), v[:1, :2].float())
        res, errs, kern = {}, {}, {}
        times = {"asm": [], "hip": []}
        for arm, mask in (("asm", full), ("hip", full & ~64)):
            lib.fa2_set_option(b"asm", mask)
            kern[arm] = _fa2_lib.fwd_plan(q, k, causal).kernel
            o.zero_()
            call()
            torch.cuda.synchronize()
            errs[arm] = float((o[:1, :2].float() - ref).abs().max())
        for _ in range(a.rounds):
            for arm, mask in (("asm", full), ("hip", full & ~64)):
                lib.fa2_set_option(b"asm", mask)
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                times[arm].append(e0.elapsed_time(e1) / a.iters * 1e3)
        lib.fa2_set_option(b"asm", full)
        ta, th = statistics.median(times["asm"]), statistics.median(times["hip"])
        fl = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
        print("%-16s B%d H%d N%d D%-3d %s causal=%d: 16x16 body (kernel %d) %7.1f us %6.0f TF | bit 6 clear (kernel %d) %7.1f us %6.0f TF | x%.3f | max err %.1e / %.1e"
              % (name, B, H, N, D, str(dt)[6:], causal, kern["asm"], ta, fl / ta / 1e6, kern["hip"], th, fl / th / 1e6, th / ta, errs["asm"], errs["hip"]))


if __name__ == "__main__":
    main()
