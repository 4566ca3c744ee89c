"""SDXL / SD 1.5 self-attention calls in a loop, for rocprofv3 --kernel-trace (developer tool; tools/kernel_medians.py reads the trace)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
dev = torch.device("cuda", 0)
for (B, H, N, D) in ((2, 10, 4096, 64), (2, 8, 4096, 40), (2, 20, 1024, 64)):
    q, k, v = (torch.randn((B, H, N, D), device=dev).half() for _ in range(3))
    for _ in range(40):
        FlashAttentionFunction.apply(q, k, v, None, False)
    torch.cuda.synchronize()
