"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/fa2_gfx950.h declares, argument validation returns the documented codes before anything
touches a GPU, and the Python operator mirrors the reference's rocwmma_fattn/FlashAttn.py surface."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import PKG, ROOT
from rocwmma_fattn import _fa2_lib
from rocwmma_fattn.FlashAttn import FlashAttentionFunction, flash_attn_wmma

HEADER = os.path.join(ROOT, "include", "fa2_gfx950.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fa2_\w+)\s*\(", text)))


def _codes():
    text = open(HEADER).read()
    return {m[0]: int(m[1]) for m in re.findall(r"#define\s+(FA2_\w+)\s+(-?\d+)", text)}


def test_library_is_in_tree_and_exports_every_declared_symbol():
    lib = _fa2_lib.load()
    assert os.path.dirname(_fa2_lib.LIB_PATH) == PKG
    declared = _declared_symbols()
    assert "fa2_fwd_f16" in declared and "fa2_fwd_bf16" in declared and "fa2_fwd" in declared
    assert "fa2_bwd_f16" in declared and "fa2_bwd_bf16" in declared and "fa2_bwd" in declared
    assert set(declared) == set(_fa2_lib.SYMBOLS), "ctypes binding and header disagree"
    for name in declared:
        assert getattr(lib, name) is not None


def test_informational_entry_points():
    lib = _fa2_lib.load()
    dims = (ctypes.c_int * 8)()
    n = lib.fa2_supported_head_dims(dims, 8)
    got = list(dims[:n])
    assert got == sorted(got) and 64 in got and 128 in got
    assert lib.fa2_padded_head_dim(40) == 64 and lib.fa2_padded_head_dim(64) == 64
    assert lib.fa2_padded_head_dim(111) == 128 and lib.fa2_padded_head_dim(0) == -1
    assert lib.fa2_padded_head_dim(max(got) + 1) == -1
    qr, kr = ctypes.c_int(), ctypes.c_int()
    assert lib.fa2_tile_rows(128, ctypes.byref(qr), ctypes.byref(kr)) == 0
    assert qr.value % 32 == 0 and kr.value % 32 == 0
    assert lib.fa2_version().decode().startswith("fa2_gfx950")
    assert _fa2_lib.error_string(0) == "ok"


def test_validation_codes_without_a_gpu():
    """Every rejected call returns before the launch, so this runs on a machine with no GPU."""
    lib = _fa2_lib.load()
    c = _codes()
    buf = ctypes.create_string_buffer(4096 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    s3 = _fa2_lib.strides3(2 * 16 * 64, 16 * 64, 64)
    s2 = _fa2_lib.strides2(32, 16)

    def call(dtype=0, q=p, k=p, v=p, o=p, lse=p, B=1, H=2, Nq=16, Nkv=16, D=64, qs=s3, scale=0.125):
        return lib.fa2_fwd(dtype, q, k, v, o, lse, B, H, Nq, Nkv, D, qs, s3, s3, s3, s2, scale, 0, None)

    assert call(q=None) == c["FA2_ERR_NULL_POINTER"]
    assert call(dtype=7) == c["FA2_ERR_DTYPE"]
    assert call(Nkv=0) == c["FA2_ERR_BAD_SHAPE"]
    assert call(D=44) == c["FA2_ERR_HEAD_DIM"]          # head dims are masked in-kernel in steps of 8 columns
    assert call(D=4096) == c["FA2_ERR_HEAD_DIM"] and call(D=520) == c["FA2_ERR_HEAD_DIM"]      # the forward reaches 512
    assert call(q=p + 2) == c["FA2_ERR_ALIGNMENT"]
    assert call(qs=_fa2_lib.strides3(2048, 1024, 68)) == c["FA2_ERR_ALIGNMENT"]
    assert call(scale=float("nan")) == c["FA2_ERR_SCALE"]
    for code in c.values():
        assert _fa2_lib.error_string(code)
    with pytest.raises(RuntimeError, match="fa2 call failed"):
        _fa2_lib.check(c["FA2_ERR_HEAD_DIM"])
    assert lib.fa2_fwd_f16(None, p, p, p, p, 1, 1, 1, 1, 64, s3, s3, s3, s3, s2, 1.0, 0, None) == c["FA2_ERR_NULL_POINTER"]
    assert lib.fa2_fwd_bf16(p, p, p, p, p, 1, 1, 1, 1, 52, s3, s3, s3, s3, s2, 1.0, 0, None) == c["FA2_ERR_HEAD_DIM"]
    # backward: same validation, ten pointers and eight stride triples
    bwd = lambda **kw: lib.fa2_bwd(kw.get("dtype", 0), kw.get("q", p), p, p, p, p, p, p, p, p, kw.get("ws", p), 1, 2, 16,  # noqa: E731
                                   kw.get("Nkv", 16), kw.get("D", 64), s3, s3, s3, s3, kw.get("dos", s3), s3, s3, s3, s2,
                                   kw.get("scale", 0.125), 0, None)
    assert bwd(q=None) == c["FA2_ERR_NULL_POINTER"] and bwd(ws=None) == c["FA2_ERR_NULL_POINTER"]
    assert bwd(dtype=3) == c["FA2_ERR_DTYPE"] and bwd(Nkv=0) == c["FA2_ERR_BAD_SHAPE"]
    assert bwd(D=520) == c["FA2_ERR_HEAD_DIM"] and bwd(D=100) == c["FA2_ERR_HEAD_DIM"] and bwd(scale=float("inf")) == c["FA2_ERR_SCALE"]      # the backward reaches 512 too
    assert bwd(dos=_fa2_lib.strides3(2048, 1024, 66)) == c["FA2_ERR_ALIGNMENT"]
    assert lib.fa2_bwd_f16(p, p, p, p, p, p, p, p, p, None, 1, 1, 1, 1, 64, s3, s3, s3, s3, s3, s3, s3, s3, s2, 1.0, 0, None) \
        == c["FA2_ERR_NULL_POINTER"]
    assert lib.fa2_bwd_bf16(p, p, p, p, p, p, p, p, p, p, 1, 1, 1, 1, 7, s3, s3, s3, s3, s3, s3, s3, s3, s2, 1.0, 0, None) \
        == c["FA2_ERR_HEAD_DIM"]


def test_backward_bias_entry_validation_codes_without_a_gpu():
    """fa2_bwd_bias rejects bad bias arguments before any launch, like fa2_fwd_bias."""
    lib = _fa2_lib.load()
    c = _codes()
    buf = ctypes.create_string_buffer(4096 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    s3 = _fa2_lib.strides3(2 * 16 * 64, 16 * 64, 64)
    s2 = _fa2_lib.strides2(32, 16)
    bs = _fa2_lib.strides3(0, 0, 16)

    def call(bias=p, kind=_fa2_lib.FA2_BIAS_F32, strides=bs, D=64):
        return lib.fa2_bwd_bias(0, p, p, p, p, p, p, p, p, p, p, 1, 2, 16, 16, D, s3, s3, s3, s3, s3, s3, s3, s3, s2, 0.125, 0,
                                bias, kind, strides, None)

    assert call(bias=None) == c["FA2_ERR_NULL_POINTER"] and call(strides=None) == c["FA2_ERR_NULL_POINTER"]
    assert call(kind=9) == c["FA2_ERR_BIAS"] and call(strides=_fa2_lib.strides3(-1, 0, 16)) == c["FA2_ERR_BIAS"]
    assert call(bias=p + 2) == c["FA2_ERR_ALIGNMENT"]
    assert call(D=520) == c["FA2_ERR_HEAD_DIM"]


def test_bias_entry_validation_codes_without_a_gpu():
    """fa2_fwd_bias (the attention bias / mask extension) rejects bad bias arguments before any launch."""
    lib = _fa2_lib.load()
    c = _codes()
    buf = ctypes.create_string_buffer(8192 + 16)
    p = (ctypes.addressof(buf) + 15) & ~15
    s3 = _fa2_lib.strides3(2 * 16 * 64, 16 * 64, 64)
    s2 = _fa2_lib.strides2(32, 16)
    b3 = _fa2_lib.strides3(0, 0, 16)

    def call(bias=p, kind=c["FA2_BIAS_F32"], bs=b3, q=p, D=64):
        return lib.fa2_fwd_bias(0, q, p, p, p, p, 1, 2, 16, 16, D, s3, s3, s3, s3, s2, 0.125, 0, bias, kind, bs, None)

    assert call(kind=9) == c["FA2_ERR_BIAS"]
    assert call(bias=None) == c["FA2_ERR_NULL_POINTER"] and call(bs=None) == c["FA2_ERR_NULL_POINTER"]
    assert call(bs=_fa2_lib.strides3(0, -16, 16)) == c["FA2_ERR_BIAS"]
    assert call(bias=p + 2) == c["FA2_ERR_ALIGNMENT"]                                   # f32 bias on a 2-byte boundary
    assert call(bias=p + 1, kind=c["FA2_BIAS_IO_DTYPE"]) == c["FA2_ERR_ALIGNMENT"]
    assert call(q=None) == c["FA2_ERR_NULL_POINTER"] and call(D=44) == c["FA2_ERR_HEAD_DIM"]   # the fa2_fwd checks still apply
    assert (c["FA2_BIAS_NONE"], c["FA2_BIAS_IO_DTYPE"], c["FA2_BIAS_F32"], c["FA2_BIAS_BOOL"]) == (
        _fa2_lib.FA2_BIAS_NONE, _fa2_lib.FA2_BIAS_IO_DTYPE, _fa2_lib.FA2_BIAS_F32, _fa2_lib.FA2_BIAS_BOOL)


def test_mask_preparation_is_broadcast_without_copies():
    """Host logic of flash_attention(mask=...): right-aligned broadcasting as in torch SDPA, broadcast dimensions become
    stride 0 (the tensor is not expanded in memory), bool -> one byte per element, foreign float dtypes -> the I/O dtype."""
    from rocwmma_fattn.FlashAttn import _prepare_bias, flash_attention
    B, H, Nq, Nkv = 2, 3, 10, 77
    dev = torch.device("cpu")
    m = torch.zeros((B, 1, Nq, Nkv), dtype=torch.bool)
    t, kind, st = _prepare_bias(m, B, H, Nq, Nkv, torch.float16, dev)
    assert kind == _fa2_lib.FA2_BIAS_BOOL and t.dtype == torch.uint8 and t.data_ptr() == m.data_ptr() and st == (Nq * Nkv, 0, Nkv)
    m = torch.zeros((Nq, Nkv), dtype=torch.float32)
    t, kind, st = _prepare_bias(m, B, H, Nq, Nkv, torch.bfloat16, dev)
    assert kind == _fa2_lib.FA2_BIAS_F32 and t.data_ptr() == m.data_ptr() and st == (0, 0, Nkv)
    m = torch.zeros((H, 1, Nkv), dtype=torch.float16)                                   # 3-D: [H, 1, Nkv] against [B, H, Nq, Nkv]
    t, kind, st = _prepare_bias(m, B, H, Nq, Nkv, torch.float16, dev)
    assert kind == _fa2_lib.FA2_BIAS_IO_DTYPE and t.data_ptr() == m.data_ptr() and st == (0, Nkv, 0)
    t, kind, st = _prepare_bias(torch.zeros((B, H, Nq, Nkv), dtype=torch.float64), B, H, Nq, Nkv, torch.bfloat16, dev)
    assert kind == _fa2_lib.FA2_BIAS_IO_DTYPE and t.dtype == torch.bfloat16 and st == (H * Nq * Nkv, Nq * Nkv, Nkv)
    t, kind, st = _prepare_bias(torch.zeros((B, H, Nq, 1)), B, H, Nq, Nkv, torch.float16, dev)   # broadcast over Nkv: expanded
    assert t.shape == (B, H, Nq, Nkv) and t.stride(3) == 1
    t, kind, st = _prepare_bias(torch.zeros((B, H, Nq, 2 * Nkv))[..., ::2], B, H, Nq, Nkv, torch.float16, dev)
    assert t.stride(3) == 1
    for bad in (torch.zeros((B, H, Nq, Nkv + 1)), torch.zeros((5, Nq, Nkv)), torch.zeros(Nkv), torch.zeros((1, 1, 1, 1, Nkv))):
        with pytest.raises(RuntimeError):
            _prepare_bias(bad, B, H, Nq, Nkv, torch.float16, dev)
    q = torch.rand(1, 2, 16, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="ROCm device"):                              # no CPU path for the masked call either
        flash_attention(q, q, q, torch.zeros(16, 16, dtype=torch.bool))


def test_operator_surface_matches_reference():
    # reference: rocwmma_fattn/FlashAttn.py:45-49 forward(ctx, q, k, v, mask=None, causal=None, scale=None, BNHD_fmt=False, *args, **kwargs)
    sig = inspect.signature(FlashAttentionFunction.forward)
    names = list(sig.parameters)
    assert names[:8] == ["ctx", "q", "k", "v", "mask", "causal", "scale", "BNHD_fmt"]
    assert sig.parameters["mask"].default is None and sig.parameters["causal"].default is None
    assert sig.parameters["scale"].default is None and sig.parameters["BNHD_fmt"].default is False
    assert issubclass(FlashAttentionFunction, torch.autograd.Function)
    # reference: host.cpp:3-8 forward(q,k,v,Br,Bc,causal,scale,permute_NH)
    fsig = inspect.signature(flash_attn_wmma.forward)
    assert list(fsig.parameters) == ["q", "k", "v", "Br", "Bc", "causal", "scale", "permute_NH"]
    assert hasattr(flash_attn_wmma, "backward")


def test_operator_refuses_cpu_tensors_loudly():
    q = torch.rand(1, 2, 16, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="ROCm device"):
        FlashAttentionFunction.apply(q, q, q, None, False)
    with pytest.raises(RuntimeError, match="4-D"):
        flash_attn_wmma.forward(q[0], q[0], q[0], 64, 128, False, 1.0, False)
    with pytest.raises(RuntimeError, match="ROCm device"):
        flash_attn_wmma.backward(q, q, q, q, q, torch.zeros(1, 2, 16), 16, 16, 64, 128, 128, False, 0.125, False)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import or link it."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), (dirpath, f)
                assert "fa2_oracle" not in text and "libfa2_oracle" not in text, (dirpath, f)


def _meta_plan(B, H, N, Nkv, D, dt=None, causal=False, scale=None, ws=0, kpad=0):
    import torch
    dt = dt or torch.float16
    q = torch.empty((B, H, N, D), dtype=dt, device="meta")
    k = torch.empty((B, H, Nkv, D + kpad), dtype=dt, device="meta")[..., :D]
    return _fa2_lib.fwd_plan(q, k, causal, scale, workspace_bytes=ws)


def test_forward_plan_names_the_kernel_and_the_contract_of_every_baseline_config():
    """fa2_fwd_plan (host logic, no GPU needed: the CU count defaults to 256) pins WHICH kernel serves a call and under WHICH numerical contract —
    the launch code executes the same plan (csrc/host.cpp: plan_fwd), so a launch that silently fell from the hand-scheduled body to the HIP
    kernel (or the reverse) changes these answers.  The GPU parity tests compare against the oracle under exactly the contract named here."""
    import torch
    K, C = _fa2_lib, _fa2_lib
    # BASELINE.json configs 2, 3, 4 and config 5's per-rank shard: the hand-scheduled body, one launch; fp16 folds the scale (and, on the bodies built on
    # v_mfma_f32_16x16x32, adds the rounded P into the row sums — on the matrix pipe), bf16 does not
    FOLD128 = C.FA2_CONTRACT_PRESCALE_Q | C.FA2_CONTRACT_LSUM_P16
    for shape, dt, causal, contract in (((2, 16, 4096, 4096, 128), torch.float16, False, FOLD128),
                                        ((2, 16, 4096, 4096, 128), torch.bfloat16, True, C.FA2_CONTRACT_LSUM_P16),
                                        ((1, 32, 8192, 8192, 128), torch.float16, True, FOLD128),
                                        ((8, 16, 4096, 4096, 128), torch.float16, False, FOLD128)):
        p = _meta_plan(*shape, dt=dt, causal=causal)
        assert (p.kernel, p.contract, p.rows, p.heads_main, p.kernel_tail, p.nsplit) == (K.FA2_KERNEL_ASM, contract, 256, shape[0] * shape[1], 0, 0), p.as_dict()
    # config 1 (B1 H2 N128 D64): a grid this small runs 128-row workgroups of the HIP kernel, f32 scale
    p = _meta_plan(1, 2, 128, 128, 64)
    assert (p.kernel, p.contract, p.rows) == (K.FA2_KERNEL_HIP_128, 0, 128)
    # head dim 64: fp16 -> the hand-scheduled body (folded scale, row sums of the rounded P on the matrix pipe: the 16x16x32 body); bf16 non-causal ->
    # 8-wave HIP kernel; bf16 causal -> the 32x32x16 body, f32 scale
    p = _meta_plan(2, 16, 4096, 4096, 64)
    assert (p.kernel, p.contract) == (K.FA2_KERNEL_ASM, FOLD128)
    # (bf16, f32 scale: the 16x16x32 body with the row sums on the matrix pipe; option asm bit 9 clear: the 8-wave HIP kernel non-causal, the 32x32x16 body causal)
    p = _meta_plan(2, 16, 4096, 4096, 64, dt=torch.bfloat16)
    assert (p.kernel, p.contract) == (K.FA2_KERNEL_ASM, C.FA2_CONTRACT_LSUM_P16)
    p = _meta_plan(2, 16, 4096, 4096, 64, dt=torch.bfloat16, causal=True)
    assert (p.kernel, p.contract) == (K.FA2_KERNEL_ASM, C.FA2_CONTRACT_LSUM_P16)
    with _fa2_lib.options(asm=451):
        assert (_meta_plan(2, 16, 4096, 4096, 64, dt=torch.bfloat16).kernel, _meta_plan(2, 16, 4096, 4096, 64, dt=torch.bfloat16, causal=True).contract) == (K.FA2_KERNEL_HIP_256, 0)
    # head dims 136 .. 256 (round 6): the hand-scheduled 128-row kernel, f32 scale, row sums of the rounded P — from 512 keys on (causal: 1024), D = 256 with
    # K row pitches that are multiples of 512 bytes; differentiated calls, short sweeps, padded K at D = 256 and option "asm" bit 10
    # clear keep the compiler-scheduled kernels
    for D in (136, 160, 176, 224, 256):
        p = _meta_plan(1, 24, 4096, 4096, D)
        assert (p.kernel, p.contract, p.rows, p.heads_main) == (K.FA2_KERNEL_ASM, C.FA2_CONTRACT_LSUM_P16, 128, 24), (D, p.as_dict())
    assert _meta_plan(2, 16, 2048, 2048, 256, dt=torch.bfloat16, causal=True).kernel == K.FA2_KERNEL_ASM
    assert _meta_plan(1, 24, 4096, 4096, 256, causal=_fa2_lib.FA2_FLAG_EXACT_SCALE).kernel != K.FA2_KERNEL_ASM
    assert _meta_plan(1, 24, 4096, 256, 256).kernel != K.FA2_KERNEL_ASM and _meta_plan(2, 16, 512, 512, 256, causal=True).kernel != K.FA2_KERNEL_ASM
    assert _meta_plan(1, 24, 4096, 4096, 256, kpad=8).kernel != K.FA2_KERNEL_ASM and _meta_plan(1, 24, 4096, 4096, 224, kpad=8).kernel == K.FA2_KERNEL_ASM
    assert _meta_plan(1, 24, 4096, 4096, 128 + 8).kernel == K.FA2_KERNEL_ASM and _meta_plan(1, 24, 4096, 4096, 264).kernel != K.FA2_KERNEL_ASM
    with _fa2_lib.options(asm=963):
        assert _meta_plan(1, 24, 4096, 4096, 256).kernel != K.FA2_KERNEL_ASM
    # two launches: 544 workgroups = two full rounds of the body + the last two heads as 128-row workgroups of the HIP kernel
    p = _meta_plan(2, 17, 4096, 4096, 64)
    assert (p.heads_main, p.kernel, p.kernel_tail, p.contract_tail, p.rows_tail) == (32, K.FA2_KERNEL_ASM, K.FA2_KERNEL_HIP_128, 0, 128)
    # what the bodies cannot serve goes to the HIP kernels: other head dims, short KV sweeps, a K row pitch that is not a multiple of a tile row, a negative scale
    assert _meta_plan(2, 16, 4096, 4096, 80).kernel == K.FA2_KERNEL_HIP_256
    # ... except the head dims just below a body's, which the 16x16x32 bodies serve with zero-filled padded columns (round 5): 40 .. 56 and 88 .. 120 (fp16,
    # folded scale), 104 .. 120 (f32 scale)
    for D, dt, want in ((96, torch.float16, K.FA2_KERNEL_ASM), (40, torch.float16, K.FA2_KERNEL_ASM), (120, torch.bfloat16, K.FA2_KERNEL_ASM),
                        (96, torch.bfloat16, K.FA2_KERNEL_HIP_256), (32, torch.float16, K.FA2_KERNEL_HIP_256)):
        assert _meta_plan(2, 16, 4096, 4096, D, dt=dt).kernel == want, (D, dt)
    with _fa2_lib.options(asm=_fa2_lib.load(build_if_missing=False).fa2_get_option(b"asm") & ~64):
        assert _meta_plan(2, 16, 4096, 4096, 96).kernel == K.FA2_KERNEL_HIP_256
    # (round 6) non-causal sweeps of at most two KV tiles, no bias, head dims <= 128: the single-pass 128-row kernel (csrc/fa2_fwd_short.hip.h), f32 scale;
    # option "short" = 0 and option "rows" keep the streaming kernels
    for (nkv, D, dt) in ((77, 128, torch.float16), (77, 64, torch.float16), (128, 40, torch.bfloat16), (1, 80, torch.float16)):
        p = _meta_plan(2, 16, 4096, nkv, D, dt=dt)
        assert (p.kernel, p.contract, p.rows, p.nsplit, p.kernel_tail) == (K.FA2_KERNEL_HIP_128, 0, 128, 0, 0), (nkv, D, p.as_dict())
    assert _meta_plan(2, 16, 4096, 129, 64).kernel != K.FA2_KERNEL_HIP_128 and _meta_plan(2, 16, 4096, 77, 160).kernel != K.FA2_KERNEL_HIP_128
    with _fa2_lib.options(short=0):
        assert _meta_plan(2, 16, 4096, 77, 128).kernel == K.FA2_KERNEL_HIP_256
    assert _meta_plan(2, 16, 2048, 2048, 128, kpad=8).kernel == K.FA2_KERNEL_HIP_256
    assert _meta_plan(2, 16, 4096, 4096, 128, scale=-0.1).kernel == K.FA2_KERNEL_HIP_256
    # the fold needs scale * log2(e) <= 1 (the prescaled Q must stay inside fp16's range): a larger scale runs the f32-scale body of the same schedule
    p = _meta_plan(2, 16, 4096, 4096, 128, scale=0.8)
    assert (p.kernel, p.contract) == (K.FA2_KERNEL_ASM, C.FA2_CONTRACT_LSUM_P16)
    p = _meta_plan(2, 16, 4096, 4096, 64, scale=0.8)           # head dim 64 without the fold: the f32-scale 16x16x32 body too
    assert (p.kernel, p.contract) == (K.FA2_KERNEL_ASM, C.FA2_CONTRACT_LSUM_P16)
    # the KV-split of a partly filled last round needs the caller's workspace (fa2_fwd_ws); whole items and parts share one kernel and one contract
    lib = _fa2_lib.load(build_if_missing=False)
    need = lib.fa2_fwd_workspace_bytes(_fa2_lib.FA2_DTYPE_F16, 2, 10, 4096, 4096, 64, 0)
    assert need > 0
    p = _meta_plan(2, 10, 4096, 4096, 64, ws=need)
    assert (p.nsplit, p.split_items, p.kernel, p.kernel_tail) == (4, 64, K.FA2_KERNEL_ASM, 0)
    assert _meta_plan(2, 10, 4096, 4096, 64, ws=need - 1).nsplit == 0 and _meta_plan(2, 10, 4096, 4096, 64).nsplit == 0
    # (round 6) a grid that covers at most half of the CUs over a long sweep — a decode-sized call — splits EVERY item, on the 8-wave kernel
    need = lib.fa2_fwd_workspace_bytes(_fa2_lib.FA2_DTYPE_F16, 1, 32, 1, 8192, 128, 0)
    p = _meta_plan(1, 32, 1, 8192, 128, ws=need)
    assert need > 0 and (p.nsplit, p.split_items, p.kernel, p.rows) == (8, 32, K.FA2_KERNEL_HIP_256, 256), p.as_dict()
    assert _meta_plan(1, 32, 1, 8192, 128).nsplit == 0 and lib.fa2_fwd_workspace_bytes(_fa2_lib.FA2_DTYPE_F16, 1, 32, 1, 512, 128, 0) == 0
    # a masked call: the BIAS kernels
    q = torch.empty((2, 10, 1024, 64), dtype=torch.float16, device="meta")
    assert _fa2_lib.fwd_plan(q, q, False, bias_kind=_fa2_lib.FA2_BIAS_BOOL).kernel == K.FA2_KERNEL_HIP_BIAS
    # validation is the call's own
    bad = _fa2_lib.FwdPlan()
    import ctypes
    assert lib.fa2_fwd_plan(0, 1, 1, 16, 16, 12, None, None, 0.1, 0, 0, 0, ctypes.byref(bad)) == -3      # FA2_ERR_HEAD_DIM
    assert lib.fa2_fwd_plan(0, 1, 1, 16, 16, 64, None, None, 0.1, 0, 0, 0, None) == -1                   # FA2_ERR_NULL_POINTER


def test_option_fold_switches_the_contract_and_nothing_else():
    """Option "fold": 0 = every launch scales the f32 product like the reference kernel (kernel_fp16.cu:164), 1 (default) = fp16 launches of the
    hand-scheduled bodies fold the scale into Q, 2 = bf16 launches too.  Setting any option bumps the "epoch" cached plans are keyed on."""
    import torch
    lib = _fa2_lib.load(build_if_missing=False)
    assert lib.fa2_get_option(b"fold") == 1
    e0 = lib.fa2_get_option(b"epoch")
    with _fa2_lib.options(fold=0):
        assert lib.fa2_get_option(b"epoch") > e0
        p = _meta_plan(2, 16, 4096, 4096, 128)                 # f32 scale; the row sums of the 16x16x32 bodies stay on the matrix pipe
        assert (p.kernel, p.contract) == (_fa2_lib.FA2_KERNEL_ASM, _fa2_lib.FA2_CONTRACT_LSUM_P16)
        with _fa2_lib.options(asm=3):                          # ... the 32x32x16 bodies add the f32 P
            assert _meta_plan(2, 16, 4096, 4096, 128).contract == 0
        p = _meta_plan(2, 16, 4096, 4096, 64)                  # head dim 64 fp16 without the fold: the f32-scale 16x16x32 body (row sums on the matrix pipe)
        assert (p.kernel, p.contract) == (_fa2_lib.FA2_KERNEL_ASM, _fa2_lib.FA2_CONTRACT_LSUM_P16)
        with _fa2_lib.options(asm=451):                        # ... bit 9 clear: back on the 8-wave kernel (non-causal)
            assert (_meta_plan(2, 16, 4096, 4096, 64).kernel, _meta_plan(2, 16, 4096, 4096, 64).contract) == (_fa2_lib.FA2_KERNEL_HIP_256, 0)
        assert lib.fa2_fwd_prescales_q(128, 0.1) == 0
    fold128 = _fa2_lib.FA2_CONTRACT_PRESCALE_Q | _fa2_lib.FA2_CONTRACT_LSUM_P16
    with _fa2_lib.options(fold=2):
        p = _meta_plan(2, 16, 4096, 4096, 128, dt=torch.bfloat16, causal=True)
        assert (p.kernel, p.contract) == (_fa2_lib.FA2_KERNEL_ASM, fold128)
    assert lib.fa2_get_option(b"fold") == 1 and lib.fa2_set_option(b"fold", 3) < 0
    p = _meta_plan(2, 16, 4096, 4096, 128)
    assert p.contract == fold128
    # the rounded-P row sums belong to the folded bodies built on v_mfma_f32_16x16x32 (option "asm" bit 6); the 32x32x16 bodies add the f32 P
    # (option "asm" bits 6 and 9; bit 9 clear: the 16x16x32 bodies with the sum check and its in-place repair — for fp16 data with very peaky rows)
    for mask in (3, 67, 451):
        with _fa2_lib.options(asm=mask):
            assert _meta_plan(2, 16, 4096, 4096, 128).contract == _fa2_lib.FA2_CONTRACT_PRESCALE_Q


def test_plan_struct_in_the_header_matches_the_ctypes_structure():
    """fa2_fwd_plan_t is the one struct of the C-ABI: its fields, their order and the FA2_KERNEL_* / FA2_CONTRACT_* values in include/fa2_gfx950.h
    must be what rocwmma_fattn/_fa2_lib.py binds (all fields are `int`)."""
    import re
    text = open(HEADER).read()
    body = re.search(r"typedef struct fa2_fwd_plan_t \{(.*?)\} fa2_fwd_plan_t;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            assert decl.startswith("int "), decl
            fields += [f.strip() for f in decl[4:].split(",")]
    assert fields == [n for n, _ in _fa2_lib.FwdPlan._fields_]
    consts = dict(re.findall(r"#define\s+(FA2_(?:KERNEL|CONTRACT)_\w+)\s+(\d+)", text))
    for name, val in consts.items():
        assert getattr(_fa2_lib, name) == int(val), name
    assert set(consts) == {"FA2_KERNEL_HIP_256", "FA2_KERNEL_HIP_128", "FA2_KERNEL_ASM", "FA2_KERNEL_HIP_BIAS", "FA2_CONTRACT_PRESCALE_Q", "FA2_CONTRACT_LSUM_P16"}
