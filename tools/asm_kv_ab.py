"""Forward, hand-scheduled vs HIP kernel over the KV length (developer tool): the item head / tail of the hand-scheduled body is a fixed
cost per item, so short KV (cross-attention) is where the plain HIP kernel can win.   python tools/asm_kv_ab.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
from rocwmma_fattn import _fa2_lib
dev = torch.device("cuda", 0)
for dt in (torch.float16, torch.bfloat16):
  for D in (64, 128):
    for (B, H, N) in ((4, 16, 4096), (2, 16, 4096), (8, 16, 1024)):
        for Nkv in (77, 256, 512, 768, 1024, 1536, 2048):
            q = torch.rand((B, H, N, D), device=dev).to(dt); k, v = (torch.rand((B, H, Nkv, D), device=dev).to(dt) for _ in range(2))
            res = {}
            for rnd in range(5):
                full = _fa2_lib.load().fa2_get_option(b"asm")          # (bit 5: the hand-scheduled forward also on short KV sweeps; bit 0 clear: HIP kernels)
                for name, opts in (("asm", dict(asm=full | 32)), ("hip", dict(asm=full & ~1)), ("default", {})):
                    with _fa2_lib.options(**opts):
                        for _ in range(10): FlashAttentionFunction.apply(q, k, v, None, False)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(60): FlashAttentionFunction.apply(q, k, v, None, False)
                        e1.record(); torch.cuda.synchronize()
                        res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 60)
            a, h, d = (statistics.median(res[n]) for n in ("asm", "hip", "default"))
            print("%s D%d B%d H%d N%d/%d: asm %6.1f hip %6.1f default %6.1f  hip/asm %.3f" % (str(dt)[6:], D, B, H, N, Nkv, a, h, d, h / a), flush=True)
for dt in (torch.float16, torch.bfloat16):          # causal self-attention: the mean sweep is half the sequence
  for D in (64, 128):
    for (B, H) in ((8, 16), (2, 16)):
        for N in (512, 768, 1024, 1536, 2048):
            q, k, v = (torch.rand((B, H, N, D), device=dev).to(dt) for _ in range(3))
            res = {}
            for rnd in range(5):
                full = _fa2_lib.load().fa2_get_option(b"asm")
                for name, opts in (("asm", dict(asm=full | 32)), ("hip", dict(asm=full & ~1))):
                    with _fa2_lib.options(**opts):
                        for _ in range(10): FlashAttentionFunction.apply(q, k, v, None, True)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(60): FlashAttentionFunction.apply(q, k, v, None, True)
                        e1.record(); torch.cuda.synchronize()
                        res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 60)
            a, h = statistics.median(res["asm"]), statistics.median(res["hip"])
            print("causal %s D%d B%d H%d N%d: asm %6.1f hip %6.1f  hip/asm %.3f" % (str(dt)[6:], D, B, H, N, a, h, h / a), flush=True)
