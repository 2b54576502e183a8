"""Build recipe for the native library (hipcc, gfx950 only, in-tree output).

``python -m pyfaceanalysis_amd.build`` compiles ``csrc/*.cpp|*.hip`` into
``pyfaceanalysis_amd/libhigsfa.so``.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhigsfa.so")
SOURCES = ["hg_tree.cpp", "hg_capi.cpp", "hg_hostpack.cpp", "hg_generic.hip", "hg_fused.hip", "hg_fused_front.hip", "hg_fused_igsfa.hip", "hg_fused_prod.hip", "hg_fused_tail.hip",
           "hg_gauss.hip", "hg_extract.hip", "hg_cascade.hip", "hg_train.hip"]
HEADERS = ["hg_common.hpp", os.path.join("..", "..", "include", "higsfa.h")]
HOST_ONLY = {"hg_hostpack.cpp"}      # no HIP in them: built with g++ (function multiversioning, which the device pass rejects)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result"]
if os.environ.get("HIGSFA_CXXFLAGS"):     # experiments: extra compiler flags (e.g. -DHG_BRANCHY_EXPANSION=0)
    FLAGS += os.environ["HIGSFA_CXXFLAGS"].split()
if os.environ.get("HIGSFA_DIAG"):        # diagnostic build: kernel instantiations with s_memtime stamps (tools/stamp_stages.sh)
    FLAGS.append("-DHIGSFA_DIAG")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _header_paths():
    hdr = [os.path.join(CSRC, h) for h in HEADERS]
    return hdr + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".hpp", ".h", ".inc"))]


def is_stale():
    """True when build() would compile or link something (pure mtime checks, starts no process)."""
    if not os.path.exists(LIB):
        return True
    hdr = _header_paths()
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(CSRC, "_obj", src + ".o")
        if _newer(sp, op) or any(_newer(h, op) for h in hdr) or _newer(op, LIB):
            return True
    return False


def oracle_is_stale(oracle_dir):
    """Same for the oracle's C restatements (oracle/Makefile): a .c newer than the .so built from it."""
    pairs = (("ref_c.c", "libref_c.so"), ("fast_cpu.c", "libfast_cpu.so"), ("fast_cpu_pow.c", "libfast_cpu.so"))
    return any(os.path.exists(os.path.join(oracle_dir, c)) and _newer(os.path.join(oracle_dir, c), os.path.join(oracle_dir, so))
               for c, so in pairs)


def build(force=False, verbose=False):
    """Compile every translation unit (objects cached under csrc/_obj) and link the library."""
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    hdr_paths += [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".hpp", ".h", ".inc"))]
    objs, relink = [], force or not os.path.exists(LIB)
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(h, op) for h in hdr_paths):
            if src in HOST_ONLY:
                cmd = [CXX, "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", sp, "-o", op]
            else:
                cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            relink = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    if relink:
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrocsolver", "-lrocblas"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
