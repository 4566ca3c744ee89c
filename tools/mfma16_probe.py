"""A/B of the 8-wave forward kernel on v_mfma_f32_16x16x32 (FA2_MFMA16=1) against the 32x32x16 form (developer tool).
Run once per setting: [FA2_MFMA16=1] [FA2_FWD_D128=hip] python tools/mfma16_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
SHAPES = [("c2", 2, 16, 4096, 128, torch.float16, False), ("c3", 2, 16, 4096, 128, torch.bfloat16, True), ("b8", 8, 16, 4096, 128, torch.float16, False),
          ("sdxl64", 2, 10, 4096, 64, torch.float16, False), ("d64_h32", 2, 16, 4096, 64, torch.float16, False), ("d64_h32_causal", 2, 16, 4096, 64, torch.bfloat16, True),
          ("sd15_64", 2, 8, 4096, 40, torch.float16, False), ("sdxl32", 2, 20, 1024, 64, torch.float16, False)]
out = []
for name, B, H, N, D, dt, causal in SHAPES:
    q, k, v = (torch.rand((B, H, N, D), device="cuda").to(dt) for _ in range(3))
    f = lambda: FlashAttentionFunction.apply(q, k, v, None, causal)
    for _ in range(60): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(60): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 60 * 1e3)
    fl = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
    out.append("%s %.1fus %.0fTF" % (name, best, fl / best / 1e6))
print("mfma16=%s d128=%s | " % (os.environ.get("FA2_MFMA16", "0"), os.environ.get("FA2_FWD_D128", "asm")) + " | ".join(out))
