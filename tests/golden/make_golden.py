"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own oracle.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

For every case it records
  * the seeded inputs q, k, v as 16-bit patterns (fp16 or bf16) — torch.rand U[0,1) like the
    reference harness (bench_with_sdpa.py:118-120), or torch.randn for the signed-input case;
  * `o_ref`  : output of the reference oracle pure_torch_ver.FlashAttentionFunction.forward
               (pure_torch_ver.py:22-90; tiled FA2 forward in the input dtype, Br=64, Bc=256),
               as 16-bit patterns;
  * `l_ref`  : its saved L (natural-log LSE, float32, padded length; pure_torch_ver.py:84-87);
  * `o_true` / `lse2_true`: dense float64 attention (torch) rounded to float32 — the ground truth the
               tolerance is stated against; lse2 is the log2-domain LSE the kernels store
               (kernel_fp16.cu:541-542);
  * backward (aligned self-attention cases only — the reference oracle's -100 padding is only valid there):
    `do` seeded upstream gradient (16-bit), `dq_ref/dk_ref/dv_ref` = pure_torch_ver backward
    (pure_torch_ver.py:92-153), the float64 truth is recomputed by the tests with torch autograd (not stored).
Nothing of the reference's source text is stored: fixtures are inputs and outputs only.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.realpath(__file__))
REFERENCE = "/root/reference/pure_torch_ver.py"
LOG2E = 1.4426950408889634

# name, (B,H,Nq,D), Nkv, dtype, input kind, seed
CASES = [
    ("c1_f16", (1, 2, 128, 64), 128, torch.float16, "rand", 1234),      # BASELINE config 1
    ("c1_bf16", (1, 2, 128, 64), 128, torch.bfloat16, "rand", 1234),
    ("mt_f16", (1, 1, 384, 128), 384, torch.float16, "rand", 1235),     # multi-tile, D=128
    ("mt_bf16", (1, 1, 384, 128), 384, torch.bfloat16, "rand", 1236),
    ("ragged_f16", (1, 2, 200, 64), 200, torch.float16, "rand", 1237),  # Nq, Nkv not block multiples
    ("cross_f16", (1, 2, 200, 64), 77, torch.float16, "rand", 1238),    # SD cross-attention Nkv=77
    ("signed_f16", (1, 2, 256, 64), 256, torch.float16, "randn", 1239),  # signed inputs, aligned
    ("signed_bf16", (1, 2, 256, 64), 256, torch.bfloat16, "randn", 1240),
]


def load_reference():
    spec = importlib.util.spec_from_file_location("pure_torch_ver", REFERENCE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # module level only flips torch.backends flags (pure_torch_ver.py:5-7)
    return mod


class _Ctx:
    """Minimal stand-in for the autograd ctx so forward() can be called directly and L read back."""

    def save_for_backward(self, *t):
        self.saved = t

    @property
    def saved_tensors(self):
        return self.saved


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def dense_truth(q, k, v, causal, do=None):
    qd, kd, vd = (t.double().requires_grad_(do is not None) for t in (q, k, v))
    s = torch.matmul(qd, kd.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool).triu(1), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    o = torch.matmul(torch.softmax(s, dim=-1), vd)
    grads = None
    if do is not None:
        o.backward(do.double())
        grads = tuple(t.grad.float().numpy() for t in (qd, kd, vd))
    return o.detach().float().numpy(), (lse.detach() * LOG2E).float().numpy(), grads


def main():
    ref = load_reference()
    fwd = ref.FlashAttentionFunction.forward
    total = 0
    for name, (B, H, N, D), Nkv, dtype, kind, seed in CASES:
        g = torch.Generator(device="cpu").manual_seed(seed)
        gen = torch.rand if kind == "rand" else torch.randn
        q = gen((B, H, N, D), generator=g, dtype=torch.float32).to(dtype)
        k = gen((B, H, Nkv, D), generator=g, dtype=torch.float32).to(dtype)
        v = gen((B, H, Nkv, D), generator=g, dtype=torch.float32).to(dtype)
        out = {"q": bits(q), "k": bits(k), "v": bits(v),
               "meta": np.array([B, H, N, Nkv, D, 0 if dtype == torch.float16 else 1, seed], dtype=np.int64)}
        for causal in (False, True):
            if causal and Nkv != N:
                continue  # the reference only defines top-left causal for self-attention shapes
            ctx = _Ctx()
            o_ref = fwd(ctx, q.clone(), k.clone(), v.clone(), None, causal)
            l_ref = ctx.saved[4]
            tag = "c" if causal else "nc"
            with_bwd = (Nkv == N) and (N % 64 == 0)
            do = None
            if with_bwd:
                do = torch.randn((B, H, N, D), generator=g, dtype=torch.float32).to(dtype)
                dq_ref, dk_ref, dv_ref = ref.FlashAttentionFunction.backward(ctx, do.clone())[:3]
                out["do_" + tag] = bits(do)
                out["dq_ref_" + tag], out["dk_ref_" + tag], out["dv_ref_" + tag] = bits(dq_ref), bits(dk_ref), bits(dv_ref)
            o_true, lse2_true, grads = dense_truth(q, k, v, causal, do)
            if with_bwd:   # float64 truth is NOT stored: tests recompute it with torch autograd
                print("%-12s causal=%d reference-oracle bwd max err dq/dk/dv: %s" % (name, causal, " ".join(
                    "%.2e" % np.abs(r.float().numpy() - t).max() for r, t in zip((dq_ref, dk_ref, dv_ref), grads))))
            out["o_ref_" + tag] = bits(o_ref)
            out["l_ref_" + tag] = l_ref.float().numpy()
            out["o_true_" + tag] = o_true
            out["lse2_true_" + tag] = lse2_true
            err = (o_ref.float().numpy() - o_true)
            print(f"{name:12s} causal={int(causal)} reference-oracle max|O-true|={np.abs(err).max():.3e}")
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        total += os.path.getsize(path)
    print("wrote %d fixtures, %.1f KiB" % (len(CASES), total / 1024))


if __name__ == "__main__":
    if not os.path.exists(REFERENCE):
        sys.exit("reference tree not present: fixtures can only be regenerated in the build container")
    main()
