"""Quick on-box check of the gfx950 backward kernels against torch autograd in fp32 (developer tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction  # noqa: E402


def ref(q, k, v, do, causal):
    qf, kf, vf = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qf, kf.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    o = torch.matmul(torch.softmax(s, -1), vf)
    o.backward(do.float())
    return qf.grad, kf.grad, vf.grad


def check(B, H, N, D, dtype, causal, nkv=None, kind="randn", time_it=False):
    g = torch.Generator(device="cuda").manual_seed(3)
    nkv = nkv or N
    mk = torch.randn if kind == "randn" else torch.rand
    q, k, v = (mk(s, generator=g, device="cuda", dtype=torch.float32).to(dtype).requires_grad_(True)
               for s in ((B, H, N, D), (B, H, nkv, D), (B, H, nkv, D)))
    do = torch.randn((B, H, N, D), generator=g, device="cuda", dtype=torch.float32).to(dtype)
    o = FlashAttentionFunction.apply(q, k, v, None, causal)
    o.backward(do)
    torch.cuda.synchronize()
    rq, rk, rv = ref(q, k, v, do, causal)
    errs = [(a.float() - b).abs().max().item() for a, b in ((q.grad, rq), (k.grad, rk), (v.grad, rv))]
    mags = [b.abs().max().item() for b in (rq, rk, rv)]
    msg = ""
    if time_it:
        for t in (q, k, v):
            t.grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o = FlashAttentionFunction.apply(q, k, v, None, causal)
        for _ in range(3):
            o.backward(do, retain_graph=True)
        e0.record()
        for _ in range(10):
            o.backward(do, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 4.0 * B * H * N * nkv * D * (0.5 if causal else 1.0) * 2.5
        msg = "  bwd %.1f us  %.0f TFLOPS (2.5x fwd FLOPs convention)" % (ms * 1e3, fl / ms / 1e9)
    print(f"B{B} H{H} N{N} Nkv{nkv} D{D} {str(dtype)[6:]} causal={int(causal)} {kind}: max err dq/dk/dv "
          + " ".join("%.2e" % e for e in errs) + "  (|ref| " + " ".join("%.1f" % m for m in mags) + ")"
          + ("  NaN!" if any(torch.isnan(t.grad).any() for t in (q, k, v)) else "") + msg)


if __name__ == "__main__":
    for dtype in (torch.float16, torch.bfloat16):
        for causal in (False, True):
            check(1, 2, 128, 64, dtype, causal)
            check(1, 2, 384, 128, dtype, causal)
            check(2, 3, 777, 128, dtype, causal)
            check(1, 2, 300, 64, dtype, causal, nkv=None if causal else 77)
            check(1, 2, 200, 40, dtype, causal, kind="rand")
    check(2, 16, 4096, 128, torch.float16, False, time_it=True)
    check(2, 16, 4096, 128, torch.bfloat16, True, time_it=True)
