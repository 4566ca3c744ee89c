"""Head dim 256: the hand-scheduled 128-row kernel (option "asm" bit 10) against the compiler-scheduled kernels, same process, interleaved (developer tool).
    python tools/d256_ab.py"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402

SHAPES = [(1, 24, 4096, 4096, 0, False), (1, 24, 4096, 4096, 1, False), (2, 16, 2048, 2048, 0, False), (2, 16, 2048, 2048, 1, True), (2, 16, 4096, 4096, 0, True),
          (1, 32, 8192, 8192, 0, True), (8, 16, 1024, 1024, 0, False), (2, 8, 1024, 1024, 0, False), (1, 16, 16384, 16384, 0, False)]


def main():
    lib = _fa2_lib.load()
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = lib.fa2_get_option(b"asm")
    dims = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256]
    for D in dims:
      for (B, H, N, Nkv, dt, causal) in (SHAPES if D == 256 else SHAPES[:1] + SHAPES[3:5]):
       if True:
        tdt = torch.float16 if dt == 0 else torch.bfloat16
        q, k, v = (torch.randn((B, H, n, D), device=dev).to(tdt) for n in (N, Nkv, Nkv))
        o = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731

        def call():
            _fa2_lib.check(lib.fa2_fwd(dt, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, Nkv, D,
                                       s3(q), s3(k), s3(v), s3(o), _fa2_lib.strides2(lse.stride(0), lse.stride(1)), float(D ** -0.5), int(causal), stream))
        for _ in range(100):
            call()
        torch.cuda.synchronize()
        ts = {"asm": [], "hip": []}
        for _ in range(5):
            for name, mask in (("asm", full | 1024), ("hip", full & ~1024)):
                lib.fa2_set_option(b"asm", mask)
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ts[name].append(e0.elapsed_time(e1) / 10 * 1e3)
        lib.fa2_set_option(b"asm", full)
        flops = 4.0 * B * H * N * Nkv * D * (0.5 if causal else 1.0)
        a, h_ = statistics.median(ts["asm"]), statistics.median(ts["hip"])
        print("B%d H%d N%d D%d %s causal=%d: hand-scheduled %.1f us (%.0f TF, %.3f of peak), compiler-scheduled %.1f us (%.0f TF); %.2fx"
              % (B, H, N, D, "f16" if dt == 0 else "bf16", causal, a, flops / a / 1e6, flops / a / 1e6 / 2500, h_, flops / h_ / 1e6, h_ / a))


if __name__ == "__main__":
    main()
