"""Phase timeline of the 3-phase ping-pong kernel (needs a -DFA2_TRACE=1 -DFA2_PIPE=3 variant)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "trace3"
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
i64p = ctypes.POINTER(ctypes.c_int64)
lib.fa2_fwd.restype = ctypes.c_int
lib.fa2_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [i64p] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
lib.fa2_debug_set_trace.argtypes = [ctypes.c_void_p]
B, H, N, D = 2, 16, 4096, 128
dev = torch.device("cuda", 0)
q, k, v = (torch.rand((B, H, N, D), device=dev).half() for _ in range(3))
o = torch.empty_like(q); lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
trace = torch.zeros((8, 64, 8), dtype=torch.int64, device=dev)
lib.fa2_debug_set_trace(trace.data_ptr())
s3 = lambda t: (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))
def run():
    return lib.fa2_fwd(0, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, N, D,
                       s3(q), s3(k), s3(v), s3(o), (ctypes.c_int64 * 2)(lse.stride(0), lse.stride(1)), D ** -0.5, 0, None)
for _ in range(3): assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
print("kernel %.1f us; cycles/tile wave0: %.0f" % (e0.elapsed_time(e1) * 1e3, (t[0, 56, 0] - t[0, 8, 0]) / 48.0))
tl = slice(8, 56)
for w in range(8):
    a = t[w, tl]
    names = ["QK", "bar", "SM", "bar", "PV", "bar"] if w < 4 else ["PV", "bar", "QK", "bar", "SM", "bar"]
    seg = [a[:, i + 1] - a[:, i] for i in range(6)]
    print("wave %d (%s): " % (w, "A" if w < 4 else "B") + "  ".join("%s %5d" % (n, int(x.mean())) for n, x in zip(names, seg)))
