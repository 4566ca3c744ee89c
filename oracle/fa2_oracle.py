"""Python face of the CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/fa2_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (flash-attention-v2-rdna3-minimal_amd/) never does.

Two restatements of the reference's forward algorithm live here:
  * `fwd_c(...)`      — ctypes call into oracle/_build/libfa2_oracle.so (fa2_oracle.c), the tiled
                        online-softmax loop of rocwmma_fattn/kernel_fp16.cu:381-543, OpenMP over
                        (batch*head, row-block).  Used for parity at test sizes and as bench.py's
                        cpu_baseline ("port").
  * `fwd_numpy(...)`  — a short numpy float64 statement of the same math (dense softmax per row,
                        no tiling), used to cross-check the C code on small cases; and
    `fwd_numpy_tiled` — numpy statement of the tiled recurrence in the I/O dtype exactly as the
                        reference's pure_torch_ver.py:22-90 writes it (fp16 only: numpy has no bf16).

Parity pinned: tests/test_oracle.py compares all of them with tests/golden/*.npz, produced by
importing the reference's pure_torch_ver.py (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.realpath(__file__))
_LIB_PATH = os.path.join(_DIR, "_build", "libfa2_oracle.so")
_SRC_PATH = os.path.join(_DIR, "fa2_oracle.c")

DTYPE_F16 = 0
DTYPE_BF16 = 1
ROUND_S = 1
ROUND_O = 2
BF16_TRUNC = 4
PRESCALE_Q = 8      # scale*log2e folded into Q in the I/O dtype (pure_torch_ver.py:61)
PRESCALE_FUSED = 32  # with PRESCALE_Q, fp16: the product is rounded to fp16 ONCE from the exact product (v_fma_mixlo_f16), not via f32 (pure_torch_ver.py:61)
LSUM_P16 = 16       # the row sum adds the rounded P (what the P.V product consumes): the head-dim-64 asm body's row sums ride the matrix pipe
LOG2E = 1.4426950408889634

_lib = None


def build(force=False):
    """gcc-compile the C oracle (oracle/Makefile) unless the .so is newer than the source."""
    if not force and os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(_SRC_PATH):
        return _LIB_PATH
    res = subprocess.run(["make", "-B", "-C", _DIR], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        lib.fa2_oracle_fwd.restype = ctypes.c_int
        lib.fa2_oracle_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [i64p] * 5 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.fa2_oracle_fwd_bias.restype = ctypes.c_int
        lib.fa2_oracle_fwd_bias.argtypes = lib.fa2_oracle_fwd.argtypes + [ctypes.c_void_p, i64p]
        lib.fa2_oracle_bwd.restype = ctypes.c_int
        lib.fa2_oracle_bwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [i64p] * 9 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.fa2_oracle_bwd_bias.restype = ctypes.c_int
        lib.fa2_oracle_bwd_bias.argtypes = lib.fa2_oracle_bwd.argtypes + [ctypes.c_void_p, i64p]
        lib.fa2_oracle_max_threads.restype = ctypes.c_int
        lib.fa2_oracle_f32_to_f16.restype = ctypes.c_uint16
        lib.fa2_oracle_f32_to_f16.argtypes = [ctypes.c_float]
        lib.fa2_oracle_f16_to_f32.restype = ctypes.c_float
        lib.fa2_oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
        lib.fa2_oracle_f32_to_bf16.restype = ctypes.c_uint16
        lib.fa2_oracle_f32_to_bf16.argtypes = [ctypes.c_float, ctypes.c_int]
        lib.fa2_oracle_bf16_to_f32.restype = ctypes.c_float
        lib.fa2_oracle_bf16_to_f32.argtypes = [ctypes.c_uint16]
        _lib = lib
    return _lib


def max_threads():
    return _load().fa2_oracle_max_threads()


# ---------------------------------------------------------------- 16-bit float helpers (numpy)

def bf16_bits_to_f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x, trunc=False):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    if not trunc:
        u = u + (np.uint32(0x7FFF) + ((u >> 16) & 1))
    return (u >> 16).astype(np.uint16)


def bits_to_f32(bits, dtype):
    bits = np.asarray(bits, dtype=np.uint16)
    return bits.view(np.float16).astype(np.float32) if dtype == DTYPE_F16 else bf16_bits_to_f32(bits)


def f32_to_bits(x, dtype):
    x = np.asarray(x, dtype=np.float32)
    return x.astype(np.float16).view(np.uint16) if dtype == DTYPE_F16 else f32_to_bf16_bits(x)


# ---------------------------------------------------------------- C oracle

def fwd_c(q_bits, k_bits, v_bits, dtype, causal=False, scale=None, Br=32, Bc=64, flags=0, nthreads=0, bias=None):
    """q_bits/k_bits/v_bits: uint16 arrays [B,H,N,D] holding fp16 or bf16 bit patterns (C-contiguous).
    bias: None, or a float array broadcastable to [B,H,Nq,Nkv] added to the scaled scores (-inf masks a position).
    Returns (o_bits uint16 [B,H,Nq,D], lse float32 [B,H,Nq] in the log2 domain)."""
    lib = _load()
    q = np.ascontiguousarray(q_bits, dtype=np.uint16)
    k = np.ascontiguousarray(k_bits, dtype=np.uint16)
    v = np.ascontiguousarray(v_bits, dtype=np.uint16)
    B, H, Nq, D = q.shape
    Nkv = k.shape[2]
    assert k.shape == (B, H, Nkv, D) and v.shape == (B, H, Nkv, D)
    if scale is None:
        scale = D ** -0.5
    o = np.empty((B, H, Nq, D), dtype=np.uint16)
    lse = np.empty((B, H, Nq), dtype=np.float32)

    def s3(n):
        return (ctypes.c_int64 * 3)(H * n * D, n * D, D)

    args = (dtype, q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data, lse.ctypes.data,
            B, H, Nq, Nkv, D, s3(Nq), s3(Nkv), s3(Nkv), s3(Nq), (ctypes.c_int64 * 2)(H * Nq, Nq),
            float(scale), int(bool(causal)), int(Br), int(Bc), int(flags), int(nthreads))
    if bias is None:
        rc = lib.fa2_oracle_fwd(*args)
    else:
        bf = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, dtype=np.float32), (B, H, Nq, Nkv)))
        rc = lib.fa2_oracle_fwd_bias(*args, bf.ctypes.data, (ctypes.c_int64 * 3)(H * Nq * Nkv, Nq * Nkv, Nkv))
    if rc != 0:
        raise RuntimeError("fa2_oracle_fwd failed (%d)" % rc)
    return o, lse


def bwd_c(q_bits, k_bits, v_bits, o_bits, do_bits, lse, dtype, causal=False, scale=None, flags=0, nthreads=0, bias=None):
    """Backward oracle (fa2_oracle.c: fa2_oracle_bwd / fa2_oracle_bwd_bias).  All *_bits are uint16 [B,H,N,D]; lse float32 [B,H,>=Nq]
    in the log2 domain (the forward's output); bias: None or a float array broadcastable to [B,H,Nq,Nkv] (the biased forward's).
    Returns (dq_bits, dk_bits, dv_bits)."""
    lib = _load()
    q, k, v, o, do = (np.ascontiguousarray(t, dtype=np.uint16) for t in (q_bits, k_bits, v_bits, o_bits, do_bits))
    lse = np.ascontiguousarray(lse, dtype=np.float32)
    B, H, Nq, D = q.shape
    Nkv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    dq, dk, dv = np.empty_like(q), np.empty_like(k), np.empty_like(v)

    def s3(n):
        return (ctypes.c_int64 * 3)(H * n * D, n * D, D)

    ls = (ctypes.c_int64 * 2)(H * lse.shape[2], lse.shape[2])
    args = (dtype, q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data, do.ctypes.data,
            lse.ctypes.data, dq.ctypes.data, dk.ctypes.data, dv.ctypes.data, B, H, Nq, Nkv, D,
            s3(Nq), s3(Nkv), s3(Nkv), s3(Nq), s3(Nq), ls, s3(Nq), s3(Nkv), s3(Nkv),
            float(scale), int(bool(causal)), int(flags), int(nthreads))
    if bias is None:
        rc = lib.fa2_oracle_bwd(*args)
    else:
        bf = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, dtype=np.float32), (B, H, Nq, Nkv)))
        rc = lib.fa2_oracle_bwd_bias(*args, bf.ctypes.data, (ctypes.c_int64 * 3)(H * Nq * Nkv, Nq * Nkv, Nkv))
    if rc != 0:
        raise RuntimeError("fa2_oracle_bwd failed (%d)" % rc)
    return dq, dk, dv


def bwd_numpy(q, k, v, do, causal=False, scale=None, bias=None):
    """Dense float64 gradients of O = softmax(Q K^T scale [+ bias] [+mask]) V contracted with dO (pure_torch_ver.py:92-153
    without the tiling): returns (dq, dk, dv).  Fully masked rows (bias only) contribute nothing."""
    q, k, v, do = (np.asarray(t, dtype=np.float64) for t in (q, k, v, do))
    D = q.shape[-1]
    if scale is None:
        scale = D ** -0.5
    s = np.einsum("bhid,bhjd->bhij", q, k) * scale
    if bias is not None:
        s = s + np.asarray(bias, dtype=np.float64)
    if causal:
        nq, nk = s.shape[-2:]
        s = np.where(np.triu(np.ones((nq, nk), dtype=bool), 1), -np.inf, s)
    mx = s.max(-1, keepdims=True)
    dead = ~np.isfinite(mx)
    with np.errstate(invalid="ignore"):
        p = np.where(dead, 0.0, np.exp(s - np.where(dead, 0.0, mx)))
    p = p / np.where(dead, 1.0, p.sum(-1, keepdims=True))
    o = np.einsum("bhij,bhjd->bhid", p, v)
    dv = np.einsum("bhij,bhid->bhjd", p, do)
    dp = np.einsum("bhid,bhjd->bhij", do, v)
    delta = (do * o).sum(-1, keepdims=True)
    ds = p * (dp - delta) * scale
    dq = np.einsum("bhij,bhjd->bhid", ds, k)
    dk = np.einsum("bhij,bhid->bhjd", ds, q)
    return dq, dk, dv


# ---------------------------------------------------------------- numpy restatements

def fwd_numpy(q, k, v, causal=False, scale=None, bias=None):
    """Dense float64 attention: softmax(Q K^T * scale [+ bias] [+ causal mask]) V and the log2-domain LSE
    (= the quantity kernel_fp16.cu:541-542 stores).  q,k,v: float arrays [B,H,N,D]; bias broadcastable to [B,H,Nq,Nkv]
    (-inf masks; fully masked rows return O = 0, LSE = -inf)."""
    q, k, v = (np.asarray(t, dtype=np.float64) for t in (q, k, v))
    D = q.shape[-1]
    if scale is None:
        scale = D ** -0.5
    s = np.einsum("bhid,bhjd->bhij", q, k) * (scale * LOG2E)
    if bias is not None:
        s = s + np.asarray(bias, dtype=np.float64) * LOG2E
    if causal:
        nq, nk = s.shape[-2:]
        s = np.where(np.triu(np.ones((nq, nk), dtype=bool), 1), -np.inf, s)  # column > row masked
    m = s.max(-1, keepdims=True)
    dead = ~np.isfinite(m)                                                   # fully masked rows (bias only)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = np.exp2(s - np.where(dead, 0.0, m))
        l = p.sum(-1, keepdims=True)
        o = np.einsum("bhij,bhjd->bhid", p / np.where(dead, 1.0, l), v)
        lse = (m + np.log2(l))[..., 0]
    return o, lse


def fwd_numpy_tiled(q, k, v, causal=False, Br=64, Bc=256, dtype=np.float16):
    """The reference oracle's own recurrence (pure_torch_ver.py:22-90) in numpy: running max, sum
    and O kept in the INPUT dtype, natural-exp domain, -65500 causal fill, Q/K padded with -100 and
    V with 0 to block multiples.  Returns (O [B,H,N,D] dtype, L [B,H,N_padded] float32 natural-log)."""
    q, k, v = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    B, H, N, D = q.shape
    scale = dtype(D ** -0.5)

    def pad(t, mult, val):
        r = t.shape[2] % mult
        if r == 0:
            return t
        extra = np.full(t.shape[:2] + (mult - r, t.shape[3]), val, dtype=dtype)
        return np.concatenate([t, extra], axis=2)

    q, k, v = pad(q, Br, -100), pad(k, Bc, -100), pad(v, Bc, 0)   # pure_torch_ver.py:34-36
    Np = q.shape[2]
    o = np.zeros_like(q)
    L = np.zeros((B, H, Np), dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        for r0 in range(0, Np, Br):
            qi = (scale * q[:, :, r0:r0 + Br]).astype(dtype)            # :61 scale applied to Q first
            m_old = np.full((B, H, Br), -np.inf, dtype=dtype)           # :54-56
            l_i = np.zeros((B, H, Br), dtype=dtype)                     # :57
            Oi = np.zeros((B, H, Br, D), dtype=dtype)                   # :58
            for c0 in range(0, k.shape[2], Bc):
                kj, vj = k[:, :, c0:c0 + Bc], v[:, :, c0:c0 + Bc]
                S = np.einsum("bhrd,bhcd->bhrc", qi.astype(np.float32), kj.astype(np.float32)).astype(dtype)  # :60-62
                if causal and r0 < c0 + Bc - 1:                         # :64-69
                    mask = np.triu(np.ones((Br, kj.shape[2]), dtype=bool), r0 - c0 + 1)
                    S = np.where(mask, dtype(-65500.0), S)
                m_new = np.maximum(S.max(-1), m_old)                    # :71-72
                P = np.exp((S - m_new[..., None]).astype(dtype)).astype(dtype)   # :73
                rowsum = P.astype(np.float32).sum(-1).astype(dtype)     # :74
                diff = (m_old - m_new).astype(dtype)                    # :75
                e = np.exp(diff).astype(dtype)
                l_i = (l_i * e).astype(dtype) + rowsum                  # :76
                Oi = (Oi * e[..., None]).astype(dtype)                  # :77
                Oi = Oi + np.einsum("bhrc,bhcd->bhrd", P.astype(np.float32), vj.astype(np.float32)).astype(dtype)  # :78
                m_old = m_new
            o[:, :, r0:r0 + Br] = (Oi / l_i[..., None]).astype(dtype)  # :81-82
            L[:, :, r0:r0 + Br] = (m_old + np.log(l_i).astype(dtype)).astype(np.float32)  # :84-85
    return o[:, :, :N], L
