"""ctypes binding of libfa2_gfx950.so — the C-ABI declared in include/fa2_gfx950.h.

There is deliberately NO fallback: if the shared library is missing and cannot be built, or a call is
made with tensors that are not on a ROCm device, this module raises.  (The CPU restatement under
oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes
import importlib.util
import os

# PyTorch-ROCm must be loaded first: it brings its own libamdhip64.so.7, and the kernel library has
# to bind to that SAME runtime instance (it is handed torch's streams and device pointers).  Loading
# libfa2_gfx950.so before torch would pull in the system runtime under the same soname instead.
import torch  # noqa: F401

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
LIB_PATH = os.path.join(_PKG_DIR, "libfa2_gfx950.so")
_OVERRIDE = os.environ.get("FA2_GFX950_LIB")      # developer A/B: load this build of the library instead (never built here)

FA2_DTYPE_F16 = 0
FA2_DTYPE_BF16 = 1
FA2_BIAS_NONE, FA2_BIAS_IO_DTYPE, FA2_BIAS_F32, FA2_BIAS_BOOL = 0, 1, 2, 3     # bias_kind of fa2_fwd_bias

FA2_KERNEL_HIP_256, FA2_KERNEL_HIP_128, FA2_KERNEL_ASM, FA2_KERNEL_HIP_BIAS = 1, 2, 3, 4              # fa2_fwd_plan_t.kernel
FA2_CONTRACT_PRESCALE_Q, FA2_CONTRACT_LSUM_P16 = 1, 2                                                   # fa2_fwd_plan_t.contract bits

_i64p = ctypes.POINTER(ctypes.c_int64)


class FwdPlan(ctypes.Structure):
    """fa2_fwd_plan_t (include/fa2_gfx950.h)."""
    _fields_ = [(n, ctypes.c_int) for n in ("kernel", "contract", "rows", "heads_main", "kernel_tail", "contract_tail", "rows_tail",
                                             "nsplit", "split_items")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_FWD_ARGTYPES = [
    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,  # q k v o lse
    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,                  # B H Nq Nkv D
    _i64p, _i64p, _i64p, _i64p, _i64p,                                                     # strides
    ctypes.c_float, ctypes.c_int, ctypes.c_void_p,                                         # scale causal stream
]

_FWD_BIAS_ARGTYPES = [ctypes.c_int] + _FWD_ARGTYPES[:-1] + [ctypes.c_void_p, ctypes.c_int, _i64p, ctypes.c_void_p]  # ... bias kind strides stream

_BWD_ARGTYPES = [ctypes.c_void_p] * 10 + [ctypes.c_int] * 5 + [_i64p] * 9 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]

# every symbol include/fa2_gfx950.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "fa2_bwd_f16": (ctypes.c_int, _BWD_ARGTYPES),
    "fa2_bwd_bf16": (ctypes.c_int, _BWD_ARGTYPES),
    "fa2_bwd": (ctypes.c_int, [ctypes.c_int] + _BWD_ARGTYPES),
    "fa2_fwd_f16": (ctypes.c_int, _FWD_ARGTYPES),
    "fa2_fwd_bf16": (ctypes.c_int, _FWD_ARGTYPES),
    "fa2_fwd": (ctypes.c_int, [ctypes.c_int] + _FWD_ARGTYPES),
    "fa2_fwd_bias": (ctypes.c_int, _FWD_BIAS_ARGTYPES),
    "fa2_fwd_ws": (ctypes.c_int, [ctypes.c_int] + _FWD_ARGTYPES[:-1] + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),   # ... workspace bytes stream
    "fa2_fwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "fa2_bwd_ws": (ctypes.c_int, [ctypes.c_int] + _BWD_ARGTYPES[:-1] + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fa2_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "fa2_bwd_bias": (ctypes.c_int, [ctypes.c_int] + _BWD_ARGTYPES[:-1] + [ctypes.c_void_p, ctypes.c_int, _i64p, ctypes.c_void_p]),
    "fa2_bwd_bias_ws": (ctypes.c_int, [ctypes.c_int] + _BWD_ARGTYPES[:-1] + [ctypes.c_void_p, ctypes.c_int, _i64p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fa2_bwd_bias_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "fa2_supported_head_dims": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "fa2_padded_head_dim": (ctypes.c_int, [ctypes.c_int]),
    "fa2_tile_rows": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "fa2_fwd_prescales_q": (ctypes.c_int, [ctypes.c_int, ctypes.c_float]),
    "fa2_fwd_plan": (ctypes.c_int, [ctypes.c_int] * 6 + [_i64p, _i64p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(FwdPlan)]),
    "fa2_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "fa2_get_option": (ctypes.c_int, [ctypes.c_char_p]),
    "fa2_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "fa2_version": (ctypes.c_char_p, []),
}

_lib = None
WS_CACHE = {}        # FlashAttn.py: shape -> workspace bytes, valid for the current option values (dropped whenever an option is set here)


def _build_module():
    spec = importlib.util.spec_from_file_location("_fa2_build", os.path.join(_PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load(build_if_missing=True):
    """dlopen the in-tree library (building it with hipcc first when absent) and type its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    # build() compiles only when the library is missing or older than its sources (stamp = digest of sources + flags),
    # so a stale .so is never used silently after a kernel edit.
    if _OVERRIDE:
        lib_path = _OVERRIDE
    elif build_if_missing:
        lib_path = _build_module().build()
    else:
        lib_path = LIB_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError("fa2: %s is missing; run `python %s`" % (LIB_PATH, os.path.join(_PKG_DIR, "build.py")))
    lib = ctypes.CDLL(lib_path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def error_string(code):
    return load().fa2_error_string(int(code)).decode()


def check(code):
    if code != 0:
        raise RuntimeError("fa2 call failed (%d): %s" % (code, error_string(code)))


def set_option(name, value):
    """fa2_set_option, and forget what was planned under the old value."""
    check(load().fa2_set_option(name.encode(), int(value)))
    WS_CACHE.clear()


class options:
    """with _fa2_lib.options(rows=256, persist=0): ... — set tuning switches of the library (fa2_set_option) and restore
    them on exit.  Process-wide: for A/B measurements and tests, not for concurrent use."""

    def __init__(self, **kw):
        self.kw = kw
        self.saved = {}

    def __enter__(self):
        lib = load()
        for k, v in self.kw.items():
            old = lib.fa2_get_option(k.encode())
            if old < 0:
                raise ValueError("fa2: unknown option %r" % k)
            self.saved[k] = old
            check(lib.fa2_set_option(k.encode(), int(v)))
        WS_CACHE.clear()
        return self

    def __exit__(self, *exc):
        lib = load()
        for k, v in self.saved.items():
            lib.fa2_set_option(k.encode(), v)
        WS_CACHE.clear()
        return False


FA2_FLAG_CAUSAL, FA2_FLAG_EXACT_SCALE = 1, 2       # bits of the `causal` argument (include/fa2_gfx950.h)


def call_flags(causal):
    """The `causal` argument of the C-ABI from what a caller of the operator passed: a bool (the reference's argument), or the flags themselves —
    bit 1, FA2_FLAG_EXACT_SCALE, marks the forward of a call that will be differentiated."""
    if isinstance(causal, bool) or causal is None:
        return FA2_FLAG_CAUSAL if causal else 0
    c = int(causal)
    if c & ~(FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE):
        raise ValueError("fa2: `causal` is a bool or the flag word FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE (0 .. 3), got %r" % (causal,))
    return c


def fwd_plan(q, k, causal, scale=None, bias_kind=FA2_BIAS_NONE, workspace_bytes=0):
    """fa2_fwd_plan for the call fa2_fwd*(q, k, ...) would be: which kernel(s) serve it and under which numerical contract.  `causal`: bool, or the
    call's flags (FA2_FLAG_CAUSAL | FA2_FLAG_EXACT_SCALE)."""
    B, H, Nq, D = q.shape
    dt = FA2_DTYPE_F16 if q.dtype == torch.float16 else FA2_DTYPE_BF16
    plan = FwdPlan()
    check(load().fa2_fwd_plan(dt, B, H, Nq, k.shape[2], D, strides3(q.stride(0), q.stride(1), q.stride(2)),
                              strides3(k.stride(0), k.stride(1), k.stride(2)), float(D ** -0.5 if scale is None else scale),
                              call_flags(causal), int(bias_kind), int(workspace_bytes), ctypes.byref(plan)))
    return plan


def strides3(a, b, c):
    return (ctypes.c_int64 * 3)(a, b, c)


def strides2(a, b):
    return (ctypes.c_int64 * 2)(a, b)
