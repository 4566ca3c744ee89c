"""How long an idle gap does the chip forgive?  (developer probe for bench.py's settle phase)
After a settled loop of config-2 launches: pause (host sleep / a small compare kernel / nothing), then time 20 launches."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd"))
from rocwmma_fattn.FlashAttn import FlashAttentionFunction
attn = FlashAttentionFunction.apply
dev = torch.device("cuda", 0)
q, k, v = (torch.rand((2, 16, 4096, 128), device=dev, dtype=torch.float32).half() for _ in range(3))
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        o = attn(q, k, v, None, False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, o
def settle(n=400):
    for _ in range(n // 25):
        t, o = timed(25)
    return t, o
o_keep = attn(q, k, v, None, False)
def real_equal(o):
    assert torch.equal(o, o_keep)
def bench_like(o):
    assert torch.equal(o, o_keep)
    for _ in range(5):
        attn(q, k, v, None, False)
    torch.cuda.synchronize(); torch.cuda.synchronize()
def dense_gate(o):
    qf, kf, vf = q[0, 0].float(), k[0, 0].float(), v[0, 0].float()
    for _ in range(4):
        ref = torch.softmax((qf @ kf.t()) * 128 ** -0.5, -1) @ vf
    float((o[0, 0].float() - ref).abs().max())
for name, pause in (("real torch.equal", real_equal), ("bench-like (equal, 5 warm-ups, 2 syncs)", bench_like), ("4 dense fp32 heads + .item()", dense_gate), ("nothing", lambda o: None), ("sync only", lambda o: torch.cuda.synchronize()), ("sleep 50us", lambda o: time.sleep(50e-6)),
                    ("sleep 200us", lambda o: time.sleep(200e-6)), ("sleep 1ms", lambda o: time.sleep(1e-3)), ("sleep 5ms", lambda o: time.sleep(5e-3)),
                    ("sleep 50ms", lambda o: time.sleep(50e-3)), ("torch.equal", lambda o: torch.equal(o, o)),
                    ("equal + 5 warmups + sync", lambda o: (torch.equal(o, o), [attn(q, k, v, None, False) for _ in range(5)], torch.cuda.synchronize())),
                    ("5 warmups + sync", lambda o: ([attn(q, k, v, None, False) for _ in range(5)], torch.cuda.synchronize()))):
    res = []
    for rep in range(3):
        ts, o = settle()
        pause(o)
        t20, _ = timed(20)
        t20b, _ = timed(20)
        res.append((ts, t20, t20b))
    print("%-28s settled %.4f | next 20: %s | the 20 after: %s" % (name, res[-1][0], " ".join("%.4f" % r[1] for r in res), " ".join("%.4f" % r[2] for r in res)))
