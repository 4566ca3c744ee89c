"""Forward + backward of two cross-attention shapes in a loop, for rocprofv3 --kernel-trace (developer tool; tools/kernel_medians.py reads the trace)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
dev = torch.device("cuda", 0)
for (B, H, N, Nkv, D) in ((8, 16, 4096, 77, 64), (8, 16, 4096, 77, 128)):
    q = torch.randn((B, H, N, D), device=dev).half().requires_grad_(True)
    k = torch.randn((B, H, Nkv, D), device=dev).half().requires_grad_(True)
    v = torch.randn((B, H, Nkv, D), device=dev).half().requires_grad_(True)
    do = torch.randn((B, H, N, D), device=dev).half()
    for _ in range(30):
        torch.autograd.grad(FlashAttentionFunction.apply(q, k, v, None, False), (q, k, v), do)
    torch.cuda.synchronize()
