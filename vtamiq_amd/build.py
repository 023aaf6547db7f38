"""Builds libvtamiq_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vtamiq_amd.build [--force]            the product library
    python -m vtamiq_amd.build --fp8 [--force]      libvtamiq_hip_fp8.so: the same sources with -DVTQ_WITH_FP8, i.e. with the fp8 experiment
                                                    (include/vtamiq_hip_fp8.h, vtamiq_amd/experimental_fp8.py), loaded through its own handle (_lib.load_fp8())
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libvtamiq_hip.so")
LIB_FP8 = os.path.join(HERE, "libvtamiq_hip_fp8.so")
SOURCES = ["gemm.hip", "gemm_st.hip", "gemm_rowln.hip", "attention.hip", "elementwise.hip", "head.hip", "skinny.hip", "cls_tail.hip", "patches.hip", "metrics.hip", "mfma_stream.hip", "engine.hip"]
DEPS = ["dev_common.h", "kernels.h", os.path.join("..", "..", "include", "vtamiq_hip.h"), os.path.join("..", "..", "include", "vtamiq_hip_fp8.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (no v_accvgpr_read/write shuffles around the softmax)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, fp8: bool = False) -> str:
    hipcc = _hipcc()
    OBJ = os.path.join(CSRC, "_obj_fp8" if fp8 else "_obj")
    LIB = LIB_FP8 if fp8 else globals()["LIB"]
    extra = ["-DVTQ_WITH_FP8"] if fp8 else []
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, d) for d in DEPS] + [os.path.abspath(__file__)]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + extra + os.environ.get("VTQ_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        # -Bsymbolic: every internal call binds inside this library (the product and the fp8 build share symbol names and coexist in a process)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, fp8="--fp8" in sys.argv))
