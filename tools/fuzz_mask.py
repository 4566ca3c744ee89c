"""Randomised parity sweep of the attention bias / mask path on the GPU (developer tool): flash_attention(q, k, v, mask, causal, scale) — C-ABI
fa2_fwd_bias / fa2_bwd_bias — against dense float64 attention with the same mask on the device.  Draws shapes, head dims (8..256: the masked
backward stops there), dtypes, the mask kind (bool keep-mask, additive in the I/O dtype, additive f32), every broadcast pattern over [B, H, Nq, Nkv],
aligned and unaligned Nkv (the three load forms of the kernels), fully masked rows, causal on top.  Bounds: the test-suite's.
    python tools/fuzz_mask.py --cases 300 --seed 1 [--bwd-every 2]"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import flash_attention  # noqa: E402

FLOOR = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
GRAD_TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def dense64(q, k, v, mask, causal, scale):
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else s + mask.double()
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    dead = torch.isinf(s).all(dim=-1, keepdim=True)                      # fully masked rows: O = 0 (the kernels' convention)
    p = torch.softmax(s.masked_fill(dead, 0.0), dim=-1).masked_fill(dead, 0.0)
    return torch.matmul(p, v.double())


def one_case(i, rng, gen, want_bwd):
    dtype = rng.choice([torch.float16, torch.bfloat16])
    D = rng.choice([8, 16, 40, 64, 64, 80, 96, 128, 128, 160, 256])
    pick = lambda: rng.choice([1, 31, 32, 33, 64, 65, 77, 128, 129, 255, 256, 257, 512]) if rng.random() < 0.4 else rng.randint(1, 900 if D <= 128 else 400)  # noqa: E731
    Nq, Nkv = pick(), pick()
    if rng.random() < 0.3:
        Nkv = Nq
    if rng.random() < 0.5:
        Nkv = max(16, Nkv // 16 * 16)                                     # the aligned forms (groups of four, whole 16-byte granules)
    B, H = rng.randint(1, 3), rng.randint(1, 6)
    if rng.random() < 0.2:
        # grids that fill the chip: dense per-row masks then run the 8-wave kernels with the tile staged by LDS-DMA (fa2_fwd_kernel.hip.h, BIAS = 2)
        D = rng.choice([40, 64, 64, 96, 128, 128])
        Nq = rng.choice([512, 700, 1024, 1500, 2048])
        Nkv = rng.choice([256, 512, 1000, 1024, 1536, 2040]) // 16 * 16
        heads = rng.randint(100 // ((Nq + 255) // 256) + 1, 300 // ((Nq + 255) // 256) + 2)
        B = rng.choice([b for b in (1, 2, 3) if heads % b == 0])
        H = heads // B
    causal = rng.random() < 0.25
    scale = D ** -0.5 * rng.choice([1.0, 1.0, 1.0, 0.5, 2.0])
    mk = lambda *s: torch.randn(*s, generator=gen, device="cuda").to(dtype)  # noqa: E731
    q, k, v = mk(B, H, Nq, D), mk(B, H, Nkv, D), mk(B, H, Nkv, D)
    kind = rng.choice(["bool", "io", "f32"])
    shape = [rng.choice([1, B]), rng.choice([1, H]), rng.choice([1, Nq]), Nkv]
    nd = rng.choice([4, 4, 3, 2])
    if nd == 3:
        shape = [rng.choice([1, H]), rng.choice([1, Nq]), Nkv]
    elif nd == 2:
        shape = [rng.choice([1, Nq]), Nkv]
    if kind == "bool":
        mask = torch.rand(*shape, generator=gen, device="cuda") < rng.choice([0.3, 0.7, 0.95])
        if rng.random() < 0.3 and shape[-2] > 1:
            mask[..., rng.randrange(shape[-2]), :] = False                 # a fully masked row
    else:
        mask = (torch.randn(*shape, generator=gen, device="cuda") * rng.choice([0.5, 2.0])).to(dtype if kind == "io" else torch.float32)
    desc = dict(i=i, B=B, H=H, Nq=Nq, Nkv=Nkv, D=D, dtype=str(dtype)[6:], causal=causal, kind=kind, mask_shape=shape, bwd=want_bwd)
    fails = []
    o_true = dense64(q, k, v, mask, causal, scale)
    if want_bwd:
        qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = flash_attention(qg, kg, vg, mask, causal, scale)
        go = torch.randn(o.shape, generator=gen, device="cuda").to(dtype)
        o.backward(go)
        q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
        dense64(q64, k64, v64, mask, causal, scale).backward(go.double())
        for name, g, g64 in (("dq", qg.grad, q64.grad), ("dk", kg.grad, k64.grad), ("dv", vg.grad, v64.grad)):
            if not torch.isfinite(g.float()).all():
                fails.append("%s non-finite" % name)
                continue
            err = (g.double() - g64).abs().max().item()
            lim = 2 * GRAD_TOL[dtype] * max(1.0, g64.abs().max().item())
            if err > lim:
                fails.append("%s err %.3e > %.3e" % (name, err, lim))
        o = o.detach()
    else:
        o = flash_attention(q, k, v, mask, causal, scale)
    torch.cuda.synchronize()
    if not torch.isfinite(o.float()).all():
        fails.append("O non-finite")
    else:
        err = (o.double() - o_true).abs().max().item()
        lim = 2 * FLOOR[dtype] * max(1.0, v.float().abs().max().item())
        if err > lim:
            fails.append("O err %.3e > %.3e" % (err, lim))
        desc["o_err"] = err
    desc["fails"] = fails
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--bwd-every", type=int, default=2)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    gen = torch.Generator(device="cuda").manual_seed(a.seed)
    bad, worst, n_bwd = [], 0.0, 0
    for i in range(a.cases):
        want_bwd = a.bwd_every > 0 and i % a.bwd_every == 0
        try:
            d = one_case(i, rng, gen, want_bwd)
        except Exception as e:
            d = dict(i=i, fails=["exception: %r" % (e,)])
        n_bwd += int(want_bwd)
        worst = max(worst, d.get("o_err", 0.0))
        if d["fails"]:
            bad.append(d)
            print("FAIL", json.dumps(d), flush=True)
    line = json.dumps(dict(cases=a.cases, backward_cases=n_bwd, seed=a.seed, failures=len(bad), worst_o_err=worst, device=torch.cuda.get_device_name(0), failing=bad))
    print(line)
    if a.out:
        open(a.out, "w").write(line + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
