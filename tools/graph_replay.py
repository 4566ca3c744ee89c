"""Host-bound small calls under HIP-graph replay (developer tool): per-call time of the SDXL cross-attention shape
eagerly and as a captured graph of 20 calls."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402

f = FlashAttentionFunction.apply
q = torch.randn((2, 10, 4096, 64), device="cuda", dtype=torch.float16)
k = torch.randn((2, 10, 77, 64), device="cuda", dtype=torch.float16)
v = torch.randn((2, 10, 77, 64), device="cuda", dtype=torch.float16)
NCALL = 20


def eager():
    for _ in range(NCALL):
        f(q, k, v, None, False)


for _ in range(5):
    eager()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        eager()
torch.cuda.current_stream().wait_stream(side)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / NCALL * 1e3


print("SDXL cross-attention B2 H10 N4096 Nkv77 D64: eager %.1f us/call, HIP-graph replay %.1f us/call" % (timeit(eager), timeit(graph.replay)))
