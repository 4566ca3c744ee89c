"""Ahead-of-time build of the gfx950 FlashAttention-2 library (libfa2_gfx950.so).

The reference JIT-builds its extension at import time with the arch pinned to gfx1100
(rocwmma_fattn/FlashAttn.py:16-41).  Here the library is a plain C-ABI shared object compiled once,
in-tree, with `hipcc --offload-arch=gfx950`; hipcc cross-compiles without a GPU, so the same command
runs in CI containers and on the MI355X box.

    python flash-attention-v2-rdna3-minimal_amd/build.py [--force] [--verbose]

The build is hermetic: the generated asm bodies are produced with no options (timing-probe options of the
generators need --probe and a separate output directory, tools/kbench.py), the stamp covers the sources,
the flags AND the generated files, the whole build runs under a file lock (ranks of a multi-process launch
that all find the library stale build it once), and every output is written to a temporary name and renamed.
"""
import argparse
import concurrent.futures
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

PKG_DIR = os.path.dirname(os.path.realpath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_NAME = "libfa2_gfx950.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
STAMP_PATH = LIB_PATH + ".stamp"
LOCK_PATH = LIB_PATH + ".lock"

# translation units: (object name, source, extra flags) — compiled in parallel, linked into one shared object
UNITS = [
    ("host", "host.cpp", []),
    ("fwd_hip_f16", "fwd_hip.cpp", ["-DFA2_TU_BF16=0"]),
    ("fwd_hip_bf16", "fwd_hip.cpp", ["-DFA2_TU_BF16=1"]),
    ("fwd_hip_trim_f16", "fwd_hip.cpp", ["-DFA2_TU_BF16=0", "-DFA2_TU_TRIM=1"]),
    ("fwd_hip_trim_bf16", "fwd_hip.cpp", ["-DFA2_TU_BF16=1", "-DFA2_TU_TRIM=1"]),
    ("fwd_asm", "fwd_asm.cpp", []),
    ("bwd_hip_f16", "bwd_hip.cpp", ["-DFA2_TU_BF16=0"]),
    ("bwd_hip_bf16", "bwd_hip.cpp", ["-DFA2_TU_BF16=1"]),
    ("bwd_hip_trim_f16", "bwd_hip.cpp", ["-DFA2_TU_BF16=0", "-DFA2_TU_TRIM=1"]),
    ("bwd_hip_trim_bf16", "bwd_hip.cpp", ["-DFA2_TU_BF16=1", "-DFA2_TU_TRIM=1"]),
    ("bwd_asm", "bwd_asm.cpp", []),
    ("bwd_bias_hip_f16", "bwd_bias_hip.cpp", ["-DFA2_TU_BF16=0"]),
    ("bwd_bias_hip_bf16", "bwd_bias_hip.cpp", ["-DFA2_TU_BF16=1"]),
]
FRONTEND_SRC = "frontend.cpp"                      # optional compiled front end of the operator (host-only C++, g++)
FRONTEND_PATH = os.path.join(PKG_DIR, "rocwmma_fattn", "_fa2_frontend.so")
GENERATORS = [os.path.join("gen", "fwd_d128_gen.py"), os.path.join("gen", "fwd_m16_gen.py"), os.path.join("gen", "fwd_m16_d256_gen.py"), os.path.join("gen", "bwd_d128_gen.py"),
              os.path.join("gen", "bwd_dq_m16_gen.py"), os.path.join("gen", "bwd_dkv_m16_gen.py")]
HEADERS = ["fa2_launch.h", "fa2_fwd_kernel.hip.h", "fa2_fwd_short.hip.h", "fa2_fwd_d128.hip.h", "fa2_fwd_d256.hip.h", "fa2_bwd_kernel.hip.h", "fa2_bwd_short.hip.h", "fa2_bwd_d128.hip.h",
           os.path.join(INCLUDE, "fa2_gfx950.h"), os.path.join("gen", "isa.py"), os.path.join("gen", "sched.py")] + GENERATORS

HIPCC_FLAGS = [
    "-x", "hip",
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fvisibility-inlines-hidden",
    "-fno-honor-nans",          # no canonicalising v_max before fmaxf on MFMA outputs; +-inf still honoured
    "--offload-compress",       # the gfx950 code objects are stored compressed in the library (the HIP runtime inflates them at load): 6.8 -> 2.x MB
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); the gfx950 library cannot be built")


def _existing(names):
    out = []
    for name in names:
        path = name if os.path.isabs(name) else os.path.join(CSRC, name)
        if os.path.exists(path):
            out.append(path)
    return out


def _source_digest():
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for (obj, src, flags) in UNITS:
        h.update(("%s %s %s" % (obj, src, " ".join(flags))).encode())
    for path in _existing(sorted(set(u[1] for u in UNITS)) + HEADERS + [FRONTEND_SRC]):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _dir_digest(d):
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(".inc"):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def generate(out_dir=None, opts=None, probe=False):
    """Run the asm generators: the hand-scheduled bodies are emitted as .inc files (default: next to the kernels; every file is
    written under a temporary name and renamed).  `opts` = {generator file name: option string} (developer variants only)."""
    out_dir = out_dir or CSRC
    for g in GENERATORS:
        path = os.path.join(CSRC, g)
        if not os.path.exists(path):
            continue
        cmd = [sys.executable, path, "--out", out_dir]
        o = (opts or {}).get(os.path.basename(g))
        if o:
            cmd += ["--opt", o]
        if probe:
            cmd.append("--probe")
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("%s failed:\n%s\n%s" % (g, res.stdout, res.stderr))
    return out_dir


OBJ_DIR = os.path.join(PKG_DIR, "build", "obj")      # objects of the last product build (developer variants relink against them)


def compile_library(out_path, extra_flags=(), inc_dir=None, verbose=False, jobs=None, only=None, keep_objects=False):
    """hipcc every translation unit (in parallel) and link them into `out_path`.  inc_dir: where the generated .inc files are
    (default csrc/).  only: names of the units to recompile (developer variants: the others are taken from the objects the last
    product build kept in build/obj).  Returns the concatenated compiler output (resource usage remarks with verbose=True)."""
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u[1]))]
    tmpdir = tempfile.mkdtemp(prefix="fa2_build_")
    log = []
    try:
        def one(u):
            obj, src, flags = u
            cmd = [_hipcc()] + HIPCC_FLAGS + list(flags) + list(extra_flags) + ["-I", INCLUDE, "-I", CSRC]
            if inc_dir:
                cmd += ["-I", inc_dir, "-DFA2_D128_INC_DIR=%s" % inc_dir]
            if verbose:
                cmd.append("-Rpass-analysis=kernel-resource-usage")
            cmd += ["-c", os.path.join(CSRC, src), "-o", os.path.join(tmpdir, obj + ".o")]
            res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
            return obj, res
        todo = [u for u in units if only is None or u[0] in only]
        for u in units:
            if u not in todo:
                cached = os.path.join(OBJ_DIR, u[0] + ".o")
                if not os.path.exists(cached):
                    raise RuntimeError("no cached object for %s: run build.py --force first" % u[0])
                shutil.copy(cached, os.path.join(tmpdir, u[0] + ".o"))
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 4)) as ex:
            for obj, res in ex.map(one, todo):
                log.append(res.stdout + res.stderr)
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed on %s (%d):\n%s\n%s" % (obj, res.returncode, res.stdout[-4000:], res.stderr[-8000:]))
        tmp = out_path + ".tmp.%d" % os.getpid()
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(tmpdir, u[0] + ".o") for u in units] + ["-o", tmp]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed (%d):\n%s\n%s" % (res.returncode, res.stdout, res.stderr))
        os.replace(tmp, out_path)
        if keep_objects:
            os.makedirs(OBJ_DIR, exist_ok=True)
            for u in units:
                shutil.copy(os.path.join(tmpdir, u[0] + ".o"), os.path.join(OBJ_DIR, u[0] + ".o"))
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return "\n".join(log)


def build_frontend(verbose=False):
    """rocwmma_fattn/_fa2_frontend.so: the forward of the reference's pybind module in C++ over the C-ABI (csrc/frontend.cpp).
    Optional — the Python implementation in FlashAttn.py is used when it is absent — so a failure here is reported, not raised,
    and an existing module is left alone (peers may have it loaded)."""
    try:
        import sysconfig
        import torch
        from torch.utils import cpp_extension as ce
        gxx = shutil.which("g++") or shutil.which("c++")
        if not gxx:
            raise RuntimeError("no g++")
        tlib = ce.library_paths()[0]
        cmd = [gxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_fa2_frontend",
               "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
        for inc in ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include", INCLUDE]:
            cmd += ["-I", inc]
        tmp = FRONTEND_PATH + ".tmp.%d" % os.getpid()
        cmd += [os.path.join(CSRC, FRONTEND_SRC), "-o", tmp, "-L", tlib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
                "-ltorch_python", "-L", PKG_DIR, "-lfa2_gfx950", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + tlib]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(res.stderr[-2000:])
        os.replace(tmp, FRONTEND_PATH)
        return FRONTEND_PATH
    except Exception as e:   # noqa: BLE001 - optional component
        print("fa2 build: compiled front end not built (%s); the Python front end will be used" % str(e)[:500], file=sys.stderr)
        return None


def _read_stamp():
    try:
        with open(STAMP_PATH) as f:
            return f.read().split()
    except OSError:
        return []


def is_current():
    """The library exists, was built from these sources and flags, and the generated bodies next to the kernels are the ones
    it was built from."""
    if not os.path.exists(LIB_PATH):
        return False
    st = _read_stamp()
    return len(st) >= 2 and st[0] == _source_digest() and st[1] == _dir_digest(CSRC)


def build(force=False, verbose=False):
    """Compile libfa2_gfx950.so in-tree unless it is already up to date.  Returns its path."""
    if not force and is_current():
        return LIB_PATH
    with open(LOCK_PATH, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # one builder at a time; the others wait and then find the library current
        try:
            if not force and is_current():
                return LIB_PATH
            digest = _source_digest()
            generate()
            log = compile_library(LIB_PATH, verbose=verbose, keep_objects=True)
            if verbose:
                print(log, file=sys.stderr)
            ok = build_frontend(verbose) is not None
            tmp = STAMP_PATH + ".tmp.%d" % os.getpid()
            with open(tmp, "w") as f:
                f.write("%s %s %s\n" % (digest, _dir_digest(CSRC), "frontend" if ok else "no-frontend"))
            os.replace(tmp, STAMP_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
