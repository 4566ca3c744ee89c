"""Forward A/B on the GPU (developer tool): times fa2_fwd through the C-ABI for several settings of one library option, interleaved.
    python tools/fwd_ab.py --opt persist --values 1,0 [--cfg c3,c4] [--rounds 7] [--iters 20]"""
import argparse, ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402
CFGS = {"c2": (2, 16, 4096, 128, torch.float16, False), "c3": (2, 16, 4096, 128, torch.bfloat16, True), "c4": (1, 32, 8192, 128, torch.float16, True),
        "b8c": (8, 16, 4096, 128, torch.float16, True), "c2k": (4, 16, 2048, 128, torch.float16, True), "c16k": (1, 8, 16384, 128, torch.float16, True),
        "d64": (2, 16, 4096, 64, torch.float16, False), "d64c": (2, 16, 4096, 64, torch.bfloat16, True), "d64n8k": (1, 24, 8192, 64, torch.float16, False),
        "sdxl": (2, 10, 4096, 64, torch.float16, False), "d64n1k": (2, 20, 1024, 64, torch.float16, False)}
ap = argparse.ArgumentParser()
ap.add_argument("--opt", default="persist"); ap.add_argument("--values", default="1,0"); ap.add_argument("--cfg", default="c3,c4")
ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = _fa2_lib.load()
vals = [int(x) for x in a.values.split(",")]
dev = torch.device("cuda", 0)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for cname in a.cfg.split(","):
    B, H, N, D, dt, causal = CFGS[cname]
    q, k, v = (torch.rand((B, H, N, D), device=dev, dtype=torch.float32).to(dt) for _ in range(3))
    o = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))
    s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))
    def fwd(val):
        lib.fa2_set_option(a.opt.encode(), val)
        _fa2_lib.check(lib.fa2_fwd(0 if dt == torch.float16 else 1, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                                   s3(q), s3(k), s3(v), s3(o), s2, float(D ** -0.5), int(causal), stream))
    times = {x: [] for x in vals}
    for x in vals:
        for _ in range(5): fwd(x)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for x in vals:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): fwd(x)
            e1.record(); torch.cuda.synchronize()
            times[x].append(e0.elapsed_time(e1) / a.iters)
    flops = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
    print("-- %s: B%d H%d N%d D%d %s causal=%d" % (cname, B, H, N, D, str(dt)[6:], causal))
    for x in vals:
        med = statistics.median(times[x])
        print("   %s=%-3d median %8.1f us  %7.1f TF   best %8.1f us" % (a.opt, x, med * 1e3, flops / med / 1e9, min(times[x]) * 1e3))
lib.fa2_set_option(a.opt.encode(), vals[0])
