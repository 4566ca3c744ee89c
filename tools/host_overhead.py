"""Host-side cost of one operator call (developer tool): tiny problem, so the GPU time is negligible."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma
q, k, v = (torch.rand((1, 2, 128, 128), device="cuda", dtype=torch.float16) for _ in range(3))
f = FlashAttentionFunction.apply
for name, fn in (("FlashAttentionFunction.apply", lambda: f(q, k, v, None, False)),
                 ("flash_attn_wmma.forward", lambda: flash_attn_wmma.forward(q, k, v, 64, 128, False, 0.088, False))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-32s issue %.1f us/call, incl. drain %.1f us/call" % (name, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
