"""Kernel-variant A/B harness (developer tool).

  build (CPU container):  python tools/kbench.py build NAME[:-DFOO=1,-DBAR=2] ...
  run   (GPU box):        python tools/kbench.py run [--rounds R] [--iters I] [--cfg c2,c3,c4,b8] [NAME ...]

Variants are separate copies of libfa2_gfx950.so compiled with extra -D flags into tools/variants/
(git-ignored, but they travel with gpurun).  `run` checks every variant against an fp32 torch
reference, then times them INTERLEAVED in one process (cdna guide §5.4 rule 24) on random data and
prints median / min per variant.
"""
import argparse
import ctypes
import glob
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
PKG = os.path.join(ROOT, "flash-attention-v2-rdna3-minimal_amd")
VAR_DIR = os.path.join(ROOT, "tools", "variants")
sys.path.insert(0, PKG)

CFGS = {
    "c2": (2, 16, 4096, 128, "f16", False),
    "c3": (2, 16, 4096, 128, "bf16", True),
    "c4": (1, 32, 8192, 128, "f16", True),
    "b8": (8, 16, 4096, 128, "f16", False),
    "c2b": (2, 16, 4096, 128, "bf16", False),
    "b8b": (8, 16, 4096, 128, "bf16", False),
    "c4b": (1, 32, 8192, 128, "bf16", True),
    "d64": (2, 16, 4096, 64, "f16", False),
    "d64c": (2, 16, 4096, 64, "bf16", True),
    "d64b": (2, 16, 4096, 64, "bf16", False),
    "d64h": (1, 24, 5120, 64, "f16", False),
    "d256": (2, 16, 2048, 256, "f16", False),
    "d256c": (2, 16, 2048, 256, "bf16", True),
    "d256h": (1, 24, 4096, 256, "f16", False),
    "d256l": (2, 16, 4096, 256, "f16", True),
    "n2k": (4, 16, 2048, 128, "f16", False),
    "n1k": (8, 16, 1024, 128, "f16", False),
    "n512": (16, 16, 512, 128, "f16", False),
    "n8k": (1, 16, 8192, 128, "f16", False),
    "n16k": (1, 8, 16384, 128, "f16", False),
    "c4k_b8": (8, 16, 4096, 128, "f16", True),
    "c4k_b1": (1, 16, 4096, 128, "f16", True),
    "c4k_h32": (1, 32, 4096, 128, "f16", True),
    "c2k": (4, 16, 2048, 128, "f16", True),
    "c4k": (2, 16, 4096, 128, "f16", True),
    "c8k": (1, 16, 8192, 128, "f16", True),
    "c16k": (1, 8, 16384, 128, "f16", True),
}


def build(specs):
    """NAME[:flag,flag,...]: flags are -D... compiler flags, gen=<options of csrc/gen/fwd_d128_gen.py> and bgen=<options of
    bwd_d128_gen.py> (';' between options, e.g. gen=e=10:64;abl=exp+dma).  Generator options go through --probe into a private
    directory of the variant — the product build (build.py) never sees them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    os.makedirs(VAR_DIR, exist_ok=True)
    ok = True
    for s in specs:
        name, _, flags = s.partition(":")
        fl = [f for f in flags.split(",") if f]
        extra = [f for f in fl if not f.startswith(("gen=", "bgen=", "m16gen=", "d256gen=", "dq16gen=", "dkv16gen=", "only=", "opts="))]
        only = None if extra else ["fwd_asm", "bwd_asm"]          # generator-only variants: recompile just the units that include the bodies
        for f in fl:
            if f.startswith("only="):                            # only=fwd_asm+host: -D flags that matter to these units alone
                only = f[5:].split("+")
        opts = {}
        for f in fl:
            if f.startswith("gen="):
                opts["fwd_d128_gen.py"] = f[4:].replace(";", ",")
            if f.startswith("bgen="):
                opts["bwd_d128_gen.py"] = f[5:].replace(";", ",")
            if f.startswith("m16gen="):      # options of csrc/gen/fwd_m16_gen.py (the 16x16x32 body)
                opts["fwd_m16_gen.py"] = f[7:].replace(";", ",")
            if f.startswith("d256gen="):     # options of csrc/gen/fwd_m16_d256_gen.py (the head-dim-256 body)
                opts["fwd_m16_d256_gen.py"] = f[8:].replace(";", ",")
            if f.startswith("dq16gen="):     # ... of the 16x16x32 backward passes (csrc/gen/bwd_dq_m16_gen.py, bwd_dkv_m16_gen.py; timed by tools/bwd_bench.py --libs)
                opts["bwd_dq_m16_gen.py"] = f[8:].replace(";", ",")
            if f.startswith("dkv16gen="):
                opts["bwd_dkv_m16_gen.py"] = f[9:].replace(";", ",")
        gdir = os.path.join(VAR_DIR, name + "_gen")
        os.makedirs(gdir, exist_ok=True)
        b.generate(gdir, opts, probe=True)
        out = os.path.join(VAR_DIR, name + ".so")
        for f in fl:
            if f.startswith("opts="):        # opts=fold=0;asm=67 -> <name>.opts: fa2_set_option on this copy of the library at load (Variant)
                with open(os.path.join(VAR_DIR, name + ".opts"), "w") as fo:
                    fo.write(f[5:].replace(";", " ") + "\n")
        try:
            log = b.compile_library(out, extra_flags=extra, inc_dir=gdir, verbose=True, only=only)
        except RuntimeError as e:
            ok = False
            print("== %s: BUILD FAILED\n%s" % (name, str(e)[-3000:]))
            continue
        lines = log.splitlines()
        for i, l in enumerate(lines):
            if "Function Name" in l and ("d128" in l or "d64" in l) and "Lb0ELb0E" in l:
                info = []
                for m in lines[i + 1:i + 12]:
                    for key in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "SGPRs:", "VGPRs Spill", "LDS Size"):
                        if key in m:
                            info.append(m.split("remark: ")[-1].strip())
                print("== %s: %s: %s" % (name, l.split("Function Name: ")[-1][:60], "; ".join(x.split("    ")[-1] for x in info)))
    return ok


class Variant:
    def __init__(self, path):
        import torch  # noqa: F401  (torch's HIP runtime first)
        self.name = os.path.basename(path)[:-3]
        self.lib = ctypes.CDLL(path)
        # <name>.opts next to the library: "asm=67 fold=0" -> fa2_set_option on THIS copy of the library (options are per loaded library)
        if os.path.exists(path[:-3] + ".opts"):
            self.lib.fa2_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
            for kv in open(path[:-3] + ".opts").read().split():
                k, _, v = kv.partition("=")
                assert self.lib.fa2_set_option(k.encode(), int(v)) == 0, kv
        i64p = ctypes.POINTER(ctypes.c_int64)
        self.lib.fa2_fwd.restype = ctypes.c_int
        self.lib.fa2_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [i64p] * 5 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]

    def fwd(self, q, k, v, o, lse, causal, stream):
        import torch
        B, H, N, D = q.shape
        s3 = lambda t: (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
        rc = self.lib.fa2_fwd(0 if q.dtype == torch.float16 else 1, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                              o.data_ptr(), lse.data_ptr(), B, H, N, k.shape[2], D, s3(q), s3(k), s3(v), s3(o),
                              (ctypes.c_int64 * 2)(lse.stride(0), lse.stride(1)), float(D ** -0.5), int(causal), stream)
        if rc != 0:
            raise RuntimeError("%s: fa2_fwd rc=%d" % (self.name, rc))


def ref_fp32(q, k, v, causal):
    import torch
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    return torch.matmul(torch.softmax(s, -1), v.float()), torch.logsumexp(s, -1) * 1.4426950408889634


def run(names, rounds, iters, cfgs, fill="rand"):
    import torch
    paths = sorted(glob.glob(os.path.join(VAR_DIR, "*.so")))
    if names:
        paths = [p for p in paths if os.path.basename(p)[:-3] in names]
    variants = [Variant(p) for p in paths]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    print(torch.cuda.get_device_name(0), "variants:", [v.name for v in variants])
    # correctness
    checks = [((2, 3, 777, 128), torch.float16, False, "randn"), ((2, 3, 777, 128), torch.bfloat16, True, "randn"),
              ((1, 2, 1024, 128), torch.float16, True, "rand"), ((1, 2, 300, 64), torch.float16, False, "rand"),
              ((1, 4, 2048, 128), torch.float16, False, "rand")]
    for var in variants:
        msgs = []
        for shape, dt, causal, kind in checks:
            g = torch.Generator(device=dev).manual_seed(1)
            mk = torch.randn if kind == "randn" else torch.rand
            q, k, v = (mk(shape, generator=g, device=dev, dtype=torch.float32).to(dt) for _ in range(3))
            o = torch.empty_like(q)
            lse = torch.empty(shape[:3], dtype=torch.float32, device=dev)
            var.fwd(q, k, v, o, lse, causal, stream)
            torch.cuda.synchronize()
            o_ref, lse_ref = ref_fp32(q, k, v, causal)
            msgs.append("nan" if not torch.isfinite(o.float()).all() else"%.1e/%.0e" % ((o.float() - o_ref).abs().max().item(), (lse - lse_ref).abs().max().item()))
        print("check %-14s max|O-ref|/max|L-ref|: %s" % (var.name, "  ".join(msgs)))
    # timing
    for cname in cfgs:
        B, H, N, D, dts, causal = CFGS[cname]
        dt = torch.float16 if dts == "f16" else torch.bfloat16
        mk = {"rand": torch.rand, "randn": torch.randn, "zeros": torch.zeros}[fill]   # zeros: no data toggling -> the chip is not power-limited
        q, k, v = (mk((B, H, N, D), device=dev, dtype=torch.float32).to(dt) for _ in range(3))
        o = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        flops = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)
        times = {v_.name: [] for v_ in variants}
        for var in variants:            # warm-up
            for _ in range(5):
                var.fwd(q, k, v, o, lse, causal, stream)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for var in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    var.fwd(q, k, v, o, lse, causal, stream)
                e1.record()
                torch.cuda.synchronize()
                times[var.name].append(e0.elapsed_time(e1) / iters)
        print("-- %s: B%d H%d N%d D%d %s causal=%d" % (cname, B, H, N, D, dts, causal))
        for var in variants:
            if "trace" in var.name:     # developer builds of the d128 kernel that return cycle sums in the LSE tensor
                var.fwd(q, k, v, o, lse, causal, stream)
                torch.cuda.synchronize()
                for (bb, hh) in ((0, 0), (B - 1, H - 1)):      # (c2: the heads of batch 1 are the workgroups' second items)
                    x = lse[bb, hh].reshape(-1, 64)      # [q block * 4 + wave, 64]: first 32 = output 0, next 32 = output 1
                    print("   %s: head (%d,%d) per-wave sums (out0, out1) of q block 0: %s  last q block: %s" % (
                        var.name, bb, hh, [(float(x[w, 0]), float(x[w, 32])) for w in range(4)], [(float(x[-4 + w, 0]), float(x[-4 + w, 32])) for w in range(4)]))
        for name, ts in times.items():
            med, mn = statistics.median(ts), min(ts)
            print("   %-16s median %8.1f us  %7.1f TF (%4.1f%%)   best %8.1f us  %7.1f TF"
                  % (name, med * 1e3, flops / med / 1e9, flops / med / 1e9 / 25, mn * 1e3, flops / mn / 1e9))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("names", nargs="*")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfg", default="c2")
    ap.add_argument("--fill", default="rand", choices=["rand", "randn", "zeros"])
    a = ap.parse_args()
    if a.cmd == "build":
        sys.exit(0 if build(a.names) else 1)
    run(a.names, a.rounds, a.iters, a.cfg.split(","), a.fill)
