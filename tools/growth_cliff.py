"""Where does the safe-mode redo of the `lm` forward bodies start to cost?  (developer probe, round 5)

The 16x16x32 bodies with the row sums on the matrix pipe (csrc/gen/fwd_m16_gen.py, opt=lm) have no in-place repair: a row whose scores outgrow the
reference of its first tiles by 16 octaves (fp16: a P beyond 65504) sends its ITEM through the safe-mode redo.  The sum-check bodies (option asm bit 9
clear: 16x16x32; bit 6 clear: 32x32x16) repair such rows in place.  This times config 2's shape on N(0,1) inputs scaled by `amp` (logits ~ N(0, amp^2) after the 1/sqrt(D) scale) under
both, same process, interleaved.
    python tools/growth_cliff.py"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn import _fa2_lib  # noqa: E402


def main():
    lib = _fa2_lib.load()
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, H, N, D = 2, 16, 4096, 128
    full = lib.fa2_get_option(b"asm")
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
        for causal in (0, 1):
            for amp in (1.0, 2.0, 3.0, 4.0, 6.0, 8.0):
                g = torch.Generator(device="cpu").manual_seed(5)
                q, k, v = ((amp ** 0.5 if i < 2 else 1.0) * torch.randn((B, H, N, D), generator=g) for i in range(3))
                q, k, v = (t.to(dt).to(dev) for t in (q, k, v))
                o = torch.empty_like(q)
                lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
                s3 = lambda t: _fa2_lib.strides3(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
                s2 = _fa2_lib.strides2(lse.stride(0), lse.stride(1))

                def call():
                    _fa2_lib.check(lib.fa2_fwd(code, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                                               s3(q), s3(k), s3(v), s3(o), s2, float(D ** -0.5), causal, stream))
                arms = (("16x16 lm", full), ("16x16 repair", full & ~512), ("32x32 repair", full & ~64))
                # settle the clock on this shape first (after idle the chip boosts, overshoots its power budget and throttles for tens of ms: the first
                # arm timed would pay for it), then time the arms INTERLEAVED, round by round
                for _ in range(400):
                    call()
                torch.cuda.synchronize()
                ts = {name: [] for name, _ in arms}
                for _ in range(7):
                    for name, mask in arms:
                        lib.fa2_set_option(b"asm", mask)
                        for _ in range(5):
                            call()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(20):
                            call()
                        e1.record()
                        torch.cuda.synchronize()
                        ts[name].append(e0.elapsed_time(e1) / 20 * 1e3)
                res = {name: statistics.median(v) for name, v in ts.items()}
                lib.fa2_set_option(b"asm", full)
                smax = float((q[0, 0].float() @ k[0, 0].float().T).abs().max()) * D ** -0.5
                print("%s causal=%d amp %.0f (max |logit| of head 0: %.0f): 16x16 lm %.1f us, 16x16 sum check (asm bit 9 clear) %.1f us, 32x32 sum check %.1f us; lm / 16x16 sum check %.3f"
                      % (str(dt)[6:], causal, amp, smax, res["16x16 lm"], res["16x16 repair"], res["32x32 repair"], res["16x16 lm"] / res["16x16 repair"]))


if __name__ == "__main__":
    main()
