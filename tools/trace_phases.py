"""Phase timeline of the ping-pong kernel (developer tool; needs a -DFA2_TRACE=1 variant).
    python tools/kbench.py build trace:-DFA2_TRACE=1      (CPU container)
    python tools/trace_phases.py [variant]                 (GPU box)
Prints, for workgroup 0, per wave the mean shader-clock cycles spent in each phase over tiles 8..55.
Stamps: 0 tile start, 6 after QK^T MFMAs, 1 end of first phase, 2 after barrier 1, 3 end of second phase,
4 after LDS writes, 5 after barrier 2 (group A: first = QK^T+softmax, second = P.V; group B rotated)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "trace"
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
i64p = ctypes.POINTER(ctypes.c_int64)
lib.fa2_fwd.restype = ctypes.c_int
lib.fa2_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [i64p] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
lib.fa2_debug_set_trace.argtypes = [ctypes.c_void_p]
B, H, N, D = 2, 16, 4096, 128
dev = torch.device("cuda", 0)
q, k, v = (torch.rand((B, H, N, D), device=dev).half() for _ in range(3))
o = torch.empty_like(q)
lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
trace = torch.zeros((8, 64, 8), dtype=torch.int64, device=dev)
lib.fa2_debug_set_trace(trace.data_ptr())
s3 = lambda t: (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
for _ in range(3):
    rc = lib.fa2_fwd(0, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                     s3(q), s3(k), s3(v), s3(o), (ctypes.c_int64 * 2)(lse.stride(0), lse.stride(1)), D ** -0.5, 0, None)
    assert rc == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.fa2_fwd(0, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
            s3(q), s3(k), s3(v), s3(o), (ctypes.c_int64 * 2)(lse.stride(0), lse.stride(1)), D ** -0.5, 0, None)
e1.record()
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
print("kernel %.1f us" % (e0.elapsed_time(e1) * 1e3))
tiles = slice(8, 56)
print("whole tile (stamp0 -> next stamp0), cycles: ", [int(np.diff(t[w, 8:57, 0]).mean()) for w in range(8)])
span = t[0, 56, 0] - t[0, 8, 0]
print("wave0 tiles 8..56: %d cycles total -> %.1f cycles/tile" % (span, span / 48.0))
hdr = ["first:mfma/qk", "first:rest", "bar1 wait", "second phase", "lds writes", "bar2 wait"]
for w in range(8):
    a = t[w, tiles]
    grp = "A" if w < 4 else "B"
    if grp == "A":   # 0 -> 6 (QK) -> 1 (softmax done) -> 2 (bar) -> 3 (PV) -> 4 (writes) -> 5 (bar)
        seg = [a[:, 6] - a[:, 0], a[:, 1] - a[:, 6], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4]]
        names = ["QK", "softmax", "bar1", "PV", "writes", "bar2"]
    else:            # 0 -> 1 (PV prev) -> 2 (bar) -> 6 (QK) -> 3 (softmax) -> 4 (writes) -> 5 (bar)
        seg = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 6] - a[:, 2], a[:, 3] - a[:, 6], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4]]
        names = ["PV", "bar1", "QK", "softmax", "writes", "bar2"]
    print("wave %d (%s): " % (w, grp) + "  ".join("%s %5d" % (n, int(x.mean())) for n, x in zip(names, seg)))
# offsets between the two waves of SIMD 0 at the tile start
print("stamp0 offset wave4 - wave0 (cycles):", int((t[4, tiles, 0] - t[0, tiles, 0]).mean()))
