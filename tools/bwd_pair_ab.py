"""A/B of the backward variants (developer tool): config-2/3/4 shapes, one child process per setting (the switches are read once).

    python tools/bwd_pair_ab.py            # runs the children
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
def run(B, H, N, D, dtype, causal):
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v = (torch.rand((B, H, N, D), generator=g, device="cuda").to(dtype).requires_grad_(True) for _ in range(3))
    o = FlashAttentionFunction.apply(q, k, v, None, causal)
    go = torch.rand(o.shape, generator=g, device="cuda").to(dtype)
    for _ in range(5):
        o.backward(go, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        o.backward(go, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3
r = [run(2, 16, 4096, 128, torch.float16, False), run(2, 16, 4096, 128, torch.bfloat16, True), run(1, 32, 8192, 128, torch.float16, True),
     run(8, 16, 4096, 128, torch.float16, False), run(2, 16, 1024, 128, torch.float16, False), run(2, 8, 4096, 80, torch.float16, False),
     run(2, 8, 1024, 80, torch.float16, False), run(1, 8, 2048, 128, torch.float16, True)]
print(" ".join("%%8.1f" %% x for x in r))
''' % ROOT

def main():
    print("%-22s %8s %8s %8s %8s %8s %8s %8s %8s   (us per backward)" % ("setting", "c2", "c3", "c4", "B8", "N1024", "D80", "sd15_32", "N2k_c"))
    settings = [("separate passes", {"FA2_BWD_PAIR": "0"})] + [("wave-pair dK+dV pass", {"FA2_BWD_PAIR": "1"}), ("  + dQ pinned 256-row", {"FA2_BWD_PAIR": "1", "FA2_BWD_DQ_ROWS": "256"})]
    for rep in range(2):
        for name, env in settings:
            e = dict(os.environ); e.update(env)
            res = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
            out = res.stdout.strip().splitlines()
            print("%-22s %s" % (name, out[-1] if out else "ERR " + res.stderr[-300:]), flush=True)

if __name__ == "__main__":
    main()
