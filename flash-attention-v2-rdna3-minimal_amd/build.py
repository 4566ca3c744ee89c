"""Ahead-of-time build of the gfx950 FlashAttention-2 forward library (libfa2_gfx950.so).

The reference JIT-builds its extension at import time with the arch pinned to gfx1100
(rocwmma_fattn/FlashAttn.py:16-41).  Here the library is a plain C-ABI shared object compiled once,
in-tree, with `hipcc --offload-arch=gfx950`; hipcc cross-compiles without a GPU, so the same command
runs in CI containers and on the MI355X box.

    python flash-attention-v2-rdna3-minimal_amd/build.py [--force] [--verbose]
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.realpath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_NAME = "libfa2_gfx950.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
STAMP_PATH = LIB_PATH + ".stamp"

SOURCES = ["host.cpp"]
FRONTEND_SRC = "frontend.cpp"                      # optional compiled front end of the operator (host-only C++, g++)
FRONTEND_PATH = os.path.join(PKG_DIR, "rocwmma_fattn", "_fa2_frontend.so")
HEADERS = ["fa2_fwd_kernel.hip.h", "fa2_fwd_kernel16.hip.h", "fa2_fwd_d128.hip.h", "fa2_bwd_kernel.hip.h", os.path.join(INCLUDE, "fa2_gfx950.h"),
           os.path.join("gen", "isa.py"), os.path.join("gen", "fwd_d128_gen.py")]
GENERATED = ["fa2_fwd_d128_f16.inc", "fa2_fwd_d128_bf16.inc", "fa2_fwd_d128_f16_fold.inc", "fa2_fwd_d128_bf16_fold.inc",
             "fa2_fwd_d128_clobbers.inc"]   # written by gen/fwd_d128_gen.py

HIPCC_FLAGS = [
    "-x", "hip",
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-fno-honor-nans",          # no canonicalising v_max before fmaxf on MFMA outputs; +-inf still honoured
] + os.environ.get("FA2_EXTRA_HIPCC_FLAGS", "").split()    # e.g. "-DFA2_PRESCALE_MAX_HD=64" (kernel knobs, see the headers)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); the gfx950 library cannot be built")


def _source_digest():
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for name in SOURCES + HEADERS + [FRONTEND_SRC]:
        path = name if os.path.isabs(name) else os.path.join(CSRC, name)
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def generate():
    """Run the asm generator: the hand-scheduled D = 128 forward body is emitted as .inc files next to the kernels."""
    res = subprocess.run([sys.executable, os.path.join(CSRC, "gen", "fwd_d128_gen.py")], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("fwd_d128_gen.py failed:\n%s\n%s" % (res.stdout, res.stderr))


def build_frontend(verbose=False):
    """rocwmma_fattn/_fa2_frontend.so: the forward of the reference's pybind module in C++ over the C-ABI (csrc/frontend.cpp).
    Optional — the Python implementation in FlashAttn.py is used when it is absent — so a failure here is reported, not raised."""
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
        if os.path.exists(FRONTEND_PATH):
            os.remove(FRONTEND_PATH)
        print("fa2 build: compiled front end not built (%s); the Python front end will be used" % str(e)[:500], file=sys.stderr)
        return None


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == _source_digest()


def build(force=False, verbose=False):
    """Compile libfa2_gfx950.so in-tree unless it is already up to date.  Returns its path."""
    digest = _source_digest()
    if not force and is_current():
        return LIB_PATH
    generate()
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", INCLUDE, "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd += ["-o", tmp]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, cwd=CSRC, capture_output=not verbose, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed (%d):\n%s\n%s" % (res.returncode, res.stdout or "", res.stderr or ""))
    os.replace(tmp, LIB_PATH)
    build_frontend(verbose)
    with open(STAMP_PATH, "w") as f:
        f.write(digest + "\n")
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
