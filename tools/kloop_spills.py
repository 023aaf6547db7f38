#!/usr/bin/env python3
"""are the GEMM kernels' SGPR spills (v_readlane / v_writelane, scratch) inside a K loop?  Compiles csrc/gemm.hip and
csrc/gemm_rowln.hip to assembly and lists, per kernel instantiation, every loop that issues >= 48 MFMAs with the number of spill /
lane-spill instructions inside it.  CPU only (hipcc cross-compiles)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for src in ("gemm.hip", "gemm_rowln.hip"):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
                        "-I" + os.path.join(ROOT, "vtamiq_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-S", "-o", out,
                        os.path.join(ROOT, "vtamiq_amd", "csrc", src), "--cuda-device-only"], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    funcs = re.split(r"\n(?=_ZN3vtq[^\n]*kernel[^\n:]*:[ \t]*;[^\n]*\n)", txt)
    print(f"# {src}")
    for f in funcs[1:]:
        name = f.split(":")[0]
        body = f.split(".Lfunc_end")[0]
        nsg = re.search(r"; SGPRSpill|sgpr_spill", f)
        lines = body.split("\n")
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = lines[labels[m.group(1)]:i]
                nm = sum("v_mfma" in s for s in seg)
                if nm >= 48:
                    loops.append((nm, sum(("v_readlane" in s or "v_writelane" in s or "scratch_" in s) for s in seg)))
        meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n.*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S)
        short = re.sub(r"^_ZN3vtq\d+_GLOBAL__N_1", "", name)
        # innermost loops are the ones with the fewest MFMAs that still hold a K tile
        print(f"  {short[:60]:60s} sgpr/vgpr spills {meta.group(1) if meta else '?':>3}/{meta.group(2) if meta else '?':<2}  loops with MFMAs (mfmas, spill ops inside): {sorted(set(loops))}")
