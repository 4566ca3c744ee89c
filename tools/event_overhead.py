"""Does a HIP event between consecutive launches cost time?  (developer probe for bench.py's per-launch record)"""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
attn = FlashAttentionFunction.apply
dev = torch.device("cuda", 0)
q, k, v = (torch.rand((2, 16, 4096, 128), device=dev, dtype=torch.float32).half() for _ in range(3))
def loop(n, events):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    e[0].record()
    for i in range(n):
        attn(q, k, v, None, False)
        if events or i == n - 1:
            e[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return wall, e[0].elapsed_time(e[n]) / n
# cold behaviour: chunks of 10 launches from the first launch of the process
for c in range(12):
    w, g = loop(10, False)
    print("chunk %2d (10 launches, no events): wall %.4f ms  gpu %.4f ms" % (c, w, g))
for rep in range(4):
    for ev in (True, False):
        w, g = loop(20, ev)
        print("20 launches, per-launch events=%s: wall %.4f ms/launch  gpu %.4f" % (ev, w, g))
w, g = loop(400, False)
print("400 launches: wall %.4f gpu %.4f" % (w, g))
time.sleep(1.0)
for c in range(6):
    w, g = loop(5, False)
    print("after 1 s idle, chunk %d (5 launches): wall %.4f gpu %.4f" % (c, w, g))
