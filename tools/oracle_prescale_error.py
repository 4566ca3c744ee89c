"""CPU companion of tools/fold_evidence.py: how far is the REFERENCE ORACLE'S OWN scaling contract — pure_torch_ver.py:61 `scale * q_frags`, Q pre-scaled in
the I/O dtype: oracle flag PRESCALE_Q — from float64 attention as the logits grow, next to the reference kernel's contract (the f32 product scaled,
kernel_fp16.cu:164: flag 0)?  Uses oracle/ (test infrastructure); no GPU.

    python tools/oracle_prescale_error.py > profiles/r16_oracle_prescale_error.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fa2_oracle as fo  # noqa: E402

rng = np.random.default_rng(3)
N, D = 1024, 128
print("C oracle (oracle/fa2_oracle.c) against float64 attention on the rounded inputs, B1 H2 N%d D%d non-causal, N(0,1) x amplitude; max |O err| / max |LSE err| (log2 units)" % (N, D))
for dt, name in ((fo.DTYPE_F16, "f16"), (fo.DTYPE_BF16, "bf16")):
    for amp in (1.0, 2.0, 3.0):
        q, k = ((amp * rng.standard_normal((1, 2, N, D))).astype(np.float32) for _ in range(2))
        v = rng.standard_normal((1, 2, N, D)).astype(np.float32)
        qb, kb, vb = (fo.f32_to_bits(x, dt) for x in (q, k, v))
        qf, kf, vf = (fo.bits_to_f32(x, dt).astype(np.float64) for x in (qb, kb, vb))
        s = np.einsum("bhqd,bhkd->bhqk", qf, kf) * D ** -0.5
        mx = s.max(-1, keepdims=True)
        p = np.exp(s - mx)
        l = p.sum(-1, keepdims=True)
        o_t = np.einsum("bhqk,bhkd->bhqd", p / l, vf)
        lse_t = (mx[..., 0] + np.log(l[..., 0])) * fo.LOG2E
        out = []
        for tag, flags in (("kernel contract, f32 product scaled", 0), ("reference oracle's contract, Q pre-scaled in the I/O dtype", fo.PRESCALE_Q)):
            ob, lse = fo.fwd_c(qb, kb, vb, dt, False, flags=flags)
            out.append("%s: %.2e / %.2e" % (tag, np.abs(fo.bits_to_f32(ob, dt) - o_t).max(), np.abs(lse - lse_t).max()))
        print("%-4s x%.0f  max|logit| %5.1f   %s" % (name, amp, np.abs(s).max(), "   ".join(out)))
