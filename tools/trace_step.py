"""Per-slot timeline of the hand-placed steady-state step (developer tool).
   build:  python tools/kbench.py build trace:-DFA2_HANDSCHED=1,-DFA2_TRACE=1
   run:    python tools/trace_step.py [variant]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "trace"
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
B, H, N, D = 2, 16, 4096, 128
q, k, v = (torch.randn((B, H, N, D), device="cuda", dtype=torch.float16) for _ in range(3))
o = torch.empty_like(q)
lse = torch.empty((B, H, N), device="cuda", dtype=torch.float32)
s3 = (ctypes.c_int64 * 3)(H * N * D, N * D, D)
s2 = (ctypes.c_int64 * 2)(H * N, N)
lib.fa2_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int64)] * 5 + \
    [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
for _ in range(5):
    rc = lib.fa2_fwd(0, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                     s3, s3, s3, s3, s2, D ** -0.5, 0, None)
    assert rc == 0, rc
torch.cuda.synchronize()
n = 8 * 2 * 16
buf = (ctypes.c_longlong * n)()
assert lib.fa2_debug_read_trace(buf, n) == 0
labels = ["start"] + ["slot%d" % (4 * i + 3) for i in range(8)] + ["-", "-", "-", "pre-barrier", "post-barrier", "end"]
for w in range(8):
    for par in range(2):
        t = [buf[(w * 2 + par) * 16 + i] for i in range(16)]
        order = sorted(range(15), key=lambda i: t[i] if t[i] else 1 << 62)
        order = [i for i in order if t[i]]
        print("wave %d step %d: total %5d | " % (w, par, t[14] - t[0]) +
              " ".join("%d:+%d" % (order[i + 1], t[order[i + 1]] - t[order[i]]) for i in range(len(order) - 1)) +
              " | start offset vs wave0 %d" % (t[0] - buf[par * 16]))
# stamps: 12 = after the last slot, 13 = after finish (or after the barrier when FA2_HS_FINISH_AFTER), 14 = end of step
