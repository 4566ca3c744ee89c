"""Randomised parity sweep on the GPU (developer tool; the committed output lives under profiles/).

Draws shapes, head dims, dtypes, layouts (BHND / BNHD views, padded row strides, head slices of a larger tensor), scales
and the causal flag at random, runs the operator (forward and backward) and compares with dense float64
attention computed by torch on the same device.  Bounds are the test suite's (tests/conftest.py): FLOOR for O, 1e-3 for the
log2 LSE (scaled by the logit magnitude for large |scale|), GRAD_TOL * max(1, max|g|) for gradients.  Also checks, case by
case, that memory around the outputs is untouched (canaries) and that a second run is bit-identical.

    python tools/fuzz_parity.py --cases 400 --seed 1 [--bwd-every 2]
"""
import argparse
import json
import math
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma  # noqa: E402

LOG2E = 1.4426950408889634
FLOOR = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
GRAD_TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def dense64(q, k, v, causal, scale):
    qf, kf, vf = q.double(), k.double(), v.double()
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vf), lse * LOG2E


def make(shape_bhnd, dtype, layout, rng, gen, dist):
    """A [B,H,N,D] tensor (as the operator sees it when BNHD_fmt is False) in one of several memory layouts."""
    B, H, N, D = shape_bhnd
    draw = (lambda *s: torch.randn(*s, generator=gen, device="cuda")) if dist == "randn" else \
        (lambda *s: torch.rand(*s, generator=gen, device="cuda"))
    if layout == "contig":
        return draw(B, H, N, D).to(dtype)
    if layout == "bnhd_view":          # stored [B,N,H,D], viewed as BHND (head and row strides exchanged)
        return draw(B, N, H, D).to(dtype).permute(0, 2, 1, 3)
    if layout == "rowpad":             # rows padded to a longer pitch (a column slice of a wider matrix)
        pad = 8 * rng.randint(1, 4)
        return draw(B, H, N, D + pad).to(dtype)[..., :D]
    if layout == "headslice":          # a head range of a tensor with more heads
        extra = rng.randint(1, 3)
        return draw(B, H + extra, N, D).to(dtype)[:, extra // 2: extra // 2 + H]
    raise ValueError(layout)


def one_case(i, rng, gen, want_bwd, force=None):
    """force: None (draw everything) | 'wide' (a grid of more 256-row workgroups than CUs) | 'longcausal' (persistent causal pair units) | 'd256' (head dims 136 .. 256 on the
    32-rows-per-wave kernel)."""
    dtype = rng.choice([torch.float16, torch.bfloat16])
    dmax = 512
    D = rng.choice([8, 16, 24, 32, 40, 48, 64, 72, 80, 96, 104, 112, 120, 128, 128, 128, 136, 144, 160, 176, 192, 208, 224, 232, 256, 320, 328, 384, 448, 512])
    while D > dmax:
        D = rng.choice([32, 40, 64, 80, 96, 128, 160, 192, 224, 256])
    big = rng.random() < 0.15
    nmax = 2300 if big and D <= 128 else 700 if D <= 256 else 300
    pick_n = lambda: rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 77, 127, 128, 129, 255, 256, 257, 511, 512, 513]) \
        if rng.random() < 0.35 else rng.randint(1, nmax)  # noqa: E731
    Nq, Nkv = pick_n(), pick_n()
    if rng.random() < 0.4:
        Nkv = Nq
    B, H = rng.randint(1, 3), rng.randint(1, 5)
    if big:
        B, H = 1, rng.randint(1, 3)
    causal = rng.random() < 0.4
    if force == "wide" or (force is None and rng.random() < 0.12):
        # wide grids: more 256-row workgroups than CUs, so that the persistent hand-scheduled kernels (several items per workgroup, the folded fp16
        # bodies) and the split of a partly filled last round (forward and backward, through the operator's own workspace) are drawn too
        D = rng.choice([40, 64, 64, 80, 128, 128])
        Nq = rng.choice([1024, 1536, 2048, 2304, 2560, 3072]) - rng.choice([0, 0, 0, 7, 100])
        Nkv = Nq if rng.random() < 0.7 else rng.choice([1024, 2048, 3072]) - rng.choice([0, 0, 13])
        items_per_head = (Nq + 255) // 256
        heads = rng.randint(256 // items_per_head + 1, 640 // items_per_head)
        B = rng.choice([b for b in (1, 2, 3, 4) if heads % b == 0])
        H = heads // B
        causal = rng.random() < 0.25
    elif force == "longcausal" or (force is None and rng.random() < 0.04):
        # long causal sequences: >= 32 q blocks per head, or more pair units than CUs — the persistent causal launches (pairs of q blocks through the item seam)
        D = rng.choice([64, 128])
        if rng.random() < 0.5:
            Nq = Nkv = rng.choice([8192, 8448, 9000])
            B, H = 1, rng.randint(1, 3)
        else:
            Nq = Nkv = rng.choice([2048, 2304, 3000])
            heads = rng.randint(70, 130)
            B = rng.choice([b for b in (1, 2) if heads % b == 0])
            H = heads // B
        causal = True
    elif force == "d256" or (force is None and rng.random() < 0.08):
        # head dims 136 .. 256 over sweeps long enough for the hand-scheduled 32-rows-per-wave kernel (csrc/gen/fwd_m16_d256_gen.py: >= 512 keys, causal
        # 1024) and its unguarded main-loop bodies
        D = rng.choice([136, 144, 152, 160, 176, 192, 200, 224, 240, 248, 256, 256, 256])
        Nq = rng.choice([128, 256, 640, 1024]) - rng.choice([0, 0, 5, 77]) if rng.random() < 0.5 else rng.randint(1, 1500)
        Nkv = Nq if rng.random() < 0.4 and Nq >= 512 else rng.randint(512, 2600)
        B, H = rng.randint(1, 2), rng.randint(1, 3)
        causal = rng.random() < 0.3
        if causal:
            Nq = Nkv = max(Nq, Nkv, 1024)
    dist = rng.choice(["rand", "randn"])
    scale = D ** -0.5
    r = rng.random()
    if r < 0.1:
        scale = -scale
    elif r < 0.2:
        scale = scale * rng.choice([0.25, 3.0])
    layouts = ["contig", "bnhd_view", "rowpad", "headslice"]
    lq, lk, lv = rng.choice(layouts), rng.choice(layouts), rng.choice(layouts)
    q = make((B, H, Nq, D), dtype, lq, rng, gen, dist)
    k = make((B, H, Nkv, D), dtype, lk, rng, gen, dist)
    v = make((B, H, Nkv, D), dtype, lv, rng, gen, dist)
    # round 6: a third of the N(0,1) cases get logits of N(0, amp^2), amp 3 .. 8 — rows that outgrow the reference of their first tile by 16 octaves and
    # more: the in-place repair of the lm forward bodies (csrc/gen/fwd_m16_gen.py: lm_repair), the reference moves of the max-first bodies behind it
    amp = 1.0
    if dist == "randn" and scale > 0 and rng.random() < 0.34:
        amp = rng.choice([3.0, 4.0, 6.0, 8.0])
        q = (q.float() * amp ** 0.5).to(dtype)
        k = (k.float() * amp ** 0.5).to(dtype)
    desc = dict(i=i, B=B, H=H, Nq=Nq, Nkv=Nkv, D=D, dtype=str(dtype)[6:], causal=causal, scale=round(scale, 5), dist=dist, amp=amp,
                layouts=[lq, lk, lv], bwd=want_bwd)
    o_true, lse_true = dense64(q, k, v, causal, scale)
    o_exact = o_true         # (a differentiated call never folds the scale: FA2_FLAG_EXACT_SCALE — its O is held to this one, whatever the plan below says)
    # Launches of the hand-scheduled bodies that fold scale * log2(e) into Q, rounded once to the I/O dtype (FA2_CONTRACT_PRESCALE_Q; the reference
    # oracle's contract, pure_torch_ver.py:61), are held to the truth that applies that rounding and nothing else — per head range, as fa2_fwd_plan
    # names the contract of the launch that served it (the plan is the one the library executes: include/fa2_gfx950.h).
    ret0 = flash_attn_wmma.forward(q, k, v, 64, 128, causal, scale, False)
    qk, kk = ret0[1], ret0[2]                       # what the C-ABI was handed (D padded to a multiple of 8, strides made kernel-ready)
    ws = 0 if causal else _fa2_lib.load().fa2_fwd_workspace_bytes(0 if dtype == torch.float16 else 1, B, H, Nq, Nkv, qk.shape[3], 0)
    plan = _fa2_lib.fwd_plan(qk, kk, causal, scale, workspace_bytes=ws)
    desc["plan"] = [plan.kernel, plan.contract, plan.heads_main, plan.kernel_tail, plan.contract_tail, plan.nsplit]
    pre_main, pre_tail = plan.contract & _fa2_lib.FA2_CONTRACT_PRESCALE_Q, plan.contract_tail & _fa2_lib.FA2_CONTRACT_PRESCALE_Q
    if pre_main or pre_tail:
        qs = (q.float() * (scale * LOG2E)).to(dtype)
        o_alt, lse_alt = dense64(qs, k, v, causal, 1.0 / LOG2E)
        folded = torch.zeros(B * H, dtype=torch.bool, device=q.device)
        if pre_main:
            folded[:plan.heads_main] = True
        if pre_tail:
            folded[plan.heads_main:] = True
        folded = folded.view(B, H)
        o_true = torch.where(folded[:, :, None, None], o_alt, o_true)
        lse_true = torch.where(folded[:, :, None], lse_alt, lse_true)
    fails = []

    if want_bwd:
        qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))   # clone keeps the strides of dense views only
        o = FlashAttentionFunction.apply(qg, kg, vg, None, causal, scale)
        go = torch.randn(o.shape, generator=gen, device="cuda").to(dtype)
        o.backward(go)
        q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
        o64, _ = dense64(q64, k64, v64, causal, scale)
        o64.backward(go.double())
        for name, g, g64 in (("dq", qg.grad, q64.grad), ("dk", kg.grad, k64.grad), ("dv", vg.grad, v64.grad)):
            if not torch.isfinite(g.float()).all():
                fails.append("%s non-finite" % name)
                continue
            lim = GRAD_TOL[dtype] * max(1.0, g64.abs().max().item())
            # rounding of the 16-bit result itself: half an ulp at the gradient's own magnitude
            err = (g.double() - g64).abs().max().item()
            if err > 2 * lim:
                fails.append("%s err %.3e > %.3e" % (name, err, 2 * lim))
        o = o.detach()
        o_true = o_exact
        lse = flash_attn_wmma.forward(q, k, v, 64, 128, causal, scale, False)[5][:, :, :Nq]     # LSE is checked in backward cases too
    else:
        o = FlashAttentionFunction.apply(q, k, v, None, causal, scale)
        ret = flash_attn_wmma.forward(q, k, v, 64, 128, causal, scale, False)
        lse = ret[5][:, :, :Nq]
        if not torch.equal(o, ret[0]):
            fails.append("operator and extension forward differ bit-wise")
    torch.cuda.synchronize()
    if not torch.isfinite(o.float()).all():
        fails.append("O non-finite")
    else:
        err = (o.double() - o_true).abs().max().item()
        vmax = max(1.0, v.float().abs().max().item())
        if err > 2 * FLOOR[dtype] * vmax:
            fails.append("O err %.3e > %.3e" % (err, 2 * FLOOR[dtype] * vmax))
        desc["o_err"] = err
    if lse is not None:
        # the f32 logit itself carries ~2^-24 relative rounding per accumulate step: bound scales with the logit magnitude
        smag = (q.float().abs().max() * k.float().abs().max() * D * abs(scale) * LOG2E).item()
        lim = max(1e-3, 4e-6 * smag)
        p16 = (plan.contract | plan.contract_tail) & _fa2_lib.FA2_CONTRACT_LSUM_P16
        if (D == 64 or p16) and dtype == torch.bfloat16:
            # launches whose row sums add the ROUNDED P (FA2_CONTRACT_LSUM_P16: the 16x16x32 bodies, on the matrix pipe; head dim 64's causal / wide
            # launches did before): when one key dominates a row (large logits) the sum carries that single term's bf16 rounding, log2(1 + 2^-9) =
            # 2.8e-3, on top of the f32 bound (tests hold it to LSE_TOL_P16_BF16 against the oracle under the same contract; against float64 truth,
            # with logits scaled 3x: 4.4e-3 seen)
            lim = max(lim, 2.8e-3 + lim, 6e-3)          # (a dominant key whose P sits just above a power of two: log2(1 + 2^-8) = 5.6e-3; tests/conftest.py LSE_TOL_P16_BF16)
        elif p16:
            lim = max(lim, 3.6e-4 + lim)          # fp16: log2(1 + 2^-12)
        lerr = (lse.double() - lse_true).abs().max().item()
        if not lerr <= lim:
            fails.append("LSE err %.3e > %.3e" % (lerr, lim))
        desc["lse_err"] = lerr
    desc["fails"] = fails
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--bwd-every", type=int, default=3, help="every n-th case also runs the backward (0 = never)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--force", default=None, choices=["wide", "longcausal", "d256"], help="every case of one family (one_case)")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    gen = torch.Generator(device="cuda").manual_seed(args.seed)
    bad, worst_o, worst_l, n_bwd = [], 0.0, 0.0, 0
    for i in range(args.cases):
        want_bwd = args.bwd_every > 0 and i % args.bwd_every == 0
        try:
            d = one_case(i, rng, gen, want_bwd, force=args.force)
        except Exception as e:                      # a refused shape or a launch error is a finding too
            d = dict(i=i, fails=["exception: %r" % (e,)])
        n_bwd += int(want_bwd)
        worst_o = max(worst_o, d.get("o_err", 0.0))
        worst_l = max(worst_l, d.get("lse_err", 0.0))
        if d["fails"]:
            bad.append(d)
            print("FAIL", json.dumps(d), flush=True)
    summary = dict(cases=args.cases, backward_cases=n_bwd, seed=args.seed, failures=len(bad), worst_o_err=worst_o,
                   worst_lse_err=worst_l, device=torch.cuda.get_device_name(0), failing=bad)
    line = json.dumps(summary)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
