#!/usr/bin/env python3
"""Builds the round-4 evidence files under profiles/ from the raw collection merged back into gpurun_out/r04p by
tools/collect_profiles_r04.sh.  (The one-off measurements of the round -- anatomy and A/B of the whole-row GEMM, the parity tail, the fp8
study, the K-loop spill check -- were written to profiles/ directly from their tools.)   Run in the repo after the gpurun call:
python tools/make_profiles_r04.py"""
import collections, csv, glob, io, json, os, sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r04p")
P = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_prof as SP  # noqa: E402


def cap(fn, *a):
    b = io.StringIO()
    with redirect_stdout(b):
        fn(*a)
    return b.getvalue()


def clean(name, keep=None):
    out = []
    for ln in open(os.path.join(G, name), errors="replace"):
        if "amdgpu.ids" in ln or "UserWarning" in ln or "warnings.warn" in ln:
            continue
        if keep is None or keep(ln):
            out.append(ln.rstrip("\n"))
    return "\n".join(out) + "\n"


def write(name, text):
    open(os.path.join(P, name), "w").write(text)
    print("wrote profiles/" + name, len(text), "bytes")


def pmc_table(d, pat):
    f = glob.glob(os.path.join(G, d) + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = SP.short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    return {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in agg.items()}


line = [l for l in open(os.path.join(G, "bench_line.json")) if l.startswith('{"metric"')][-1]
write("r04_bench_line.json", line)
lh = [l for l in open(os.path.join(G, "bench_line_headline_profiled.json")) if l.startswith('{"metric"')][-1]
dh = json.loads(lh)
write("r04_bench_line_headline_profiled.json", lh)
write("r04_bench_headline_kernel_stats.txt",
      "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-second-mode --no-north-star --no-fidelity --no-sustained --no-live-traffic\n"
      "#          --no-e2e --no-secondary --no-collective-check      (only the headline mode at B = 32 runs: 5 warm-up + 20 timed forwards)\n"
      f"# the same process printed roofline.avg_launch_ms = {dh['roofline']['avg_launch_ms']:.4f} ms for gemm_pp2_kernel<f16, 3, 1> (fc1, HIP events on the\n"
      "# launch stream inside the timed region); the rocprofv3 average below covers warm-up + timed launches of the same kernel.\n"
      + cap(SP.stats, os.path.join(G, "stats")))

t1 = pmc_table("gemm_pmc", "gemm")
txt = ["# command: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace",
       "#          -- python3 tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3 fp16",
       "# fc1 GEMM of BASELINE configs[1]: M = 32256 (64 sequences x 501 rows, padded), N = 3072, K = 768, GELU epilogue; mean per dispatch.",
       "# GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* are summed over the 256 CUs (x4 SIMDs for the per-SIMD busy counters).",
       "# The kernel is round 3's (round 4 did not touch its main loop or epilogue; its measurement knobs are compiled out of the product build).", "",
       cap(SP.pmc, os.path.join(G, "gemm_pmc"), "gemm"),
       "# derived (per kernel): cycles per XCD = GRBM_GUI_ACTIVE / 8; MFMA-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles per XCD)"]
for k, v in t1.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    txt.append(f"#   {k:34s} cycles/XCD {cyc:9.0f}   MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %   MFMA insts {v['SQ_INSTS_MFMA']:.3e}   "
               f"LDS bank-conflict cycles/CU {v['SQ_LDS_BANK_CONFLICT'] / 256:8.0f}")
txt += ["", "# Round 3 (profiles/r03_gemm_fc1_pmc.txt): fp16x3 668780 cycles/XCD, MFMA busy 65.4 %."]
write("r04_gemm_fc1_pmc.txt", "\n".join(txt))

f = pmc_table("fetch", "gemm"); w = pmc_table("write", "gemm")
names = {"fp16x3": "gemm_pp2_kernel<f16, 3, 1>", "fp16": "gemm_pp2_kernel<f16, 1, 1>"}
M, N, K = 32256, 3072, 768
alg = {"fp16x3": (M * K * 4 + N * K * 4, M * N * 4), "fp16": (M * K * 2 + N * K * 2, M * N * 2)}
js = {"kernel": "gemm_pp2_kernel<T, TERMS, GELU> (fc1), M=32256 N=3072 K=768 (B=32 pairs)",
      "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gemm_bench.py --only fc1; see r04_gemm_fc1_traffic.txt",
      "note": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; L2-miss side bytes, "
              "Infinity-Cache hits included"}
tt = ["# commands: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3 fp16",
      "#           rocprofv3 --pmc WRITE_SIZE --kernel-trace -- (same)          separate passes; units KiB; mean per dispatch", "",
      cap(SP.pmc, os.path.join(G, "fetch"), "gemm"), cap(SP.pmc, os.path.join(G, "write"), "gemm"),
      "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B -> x2; WRITE_SIZE exact.  Both are the L2's",
      "# memory-side request counters: reads served by the 256 MiB Infinity Cache are INCLUDED (they are L2 misses, not HBM reads)."]
for mode, kn in names.items():
    if kn not in f:
        continue
    rd, wr = 2 * f[kn]["FETCH_SIZE"] * 1024, w[kn]["WRITE_SIZE"] * 1024
    a_r, a_w = alg[mode]
    js[mode] = {"FETCH_SIZE_KiB": f[kn]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[kn]["WRITE_SIZE"], "bytes_per_launch": rd + wr,
                "algorithmic_bytes_per_launch": a_r + a_w}
    tt.append(f"# {mode:7s}: read 2 x {f[kn]['FETCH_SIZE']:.0f} KiB = {rd / 1e6:6.1f} MB (algorithmic A + W {a_r / 1e6:6.1f} MB), write {wr / 1e6:6.1f} MB "
              f"(algorithmic {a_w / 1e6:6.1f} MB) -> {(rd + wr) / 1e6:6.1f} MB per launch")
write("r04_gemm_fc1_traffic.txt", "\n".join(tt))
write("r04_gemm_fc1_traffic.json", json.dumps(js, indent=1))

# the whole-row kernel next to the two launches it replaces: counters and memory-side traffic
tr = pmc_table("rowln_pmc", "")
fr, wr_ = pmc_table("rowln_fetch", ""), pmc_table("rowln_write", "")
rl = ["# commands: rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/rowln_bench.py --M 32256 --rounds 1     (three passes: the SQ counters, FETCH_SIZE, WRITE_SIZE)",
      "# out-proj (K = 768) and fc2 (K = 3072) at B = 32, fp16x3: gemm_rowln_kernel<f16, LN> (whole-row tile, LayerNorm in the epilogue; LN = false: the x update only)",
      "# beside gemm_pp2_kernel<f16, 3, 2> (256 x 256 residual GEMM) + layernorm_kernel; mean per dispatch over both shapes' launches of a kernel name", "",
      cap(SP.pmc, os.path.join(G, "rowln_pmc"), ""), "# derived: cycles/XCD = GRBM_GUI_ACTIVE / 8; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)"]
for k, v in tr.items():
    if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0 and ("gemm" in k or "layernorm" in k):
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        rl.append(f"#   {k:44s} cycles/XCD {cyc:9.0f}  MFMA busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc * 100:5.1f} %  LDS bank-conflict cycles/CU {v.get('SQ_LDS_BANK_CONFLICT', 0) / 256:8.0f}")
rl.append("# memory side (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), MB per dispatch, mean over the out-proj and the fc2 launches of a kernel name:")
for k in fr:
    if k in wr_ and ("gemm" in k or "layernorm" in k):
        rl.append(f"#   {k:44s} read {2 * fr[k]['FETCH_SIZE'] * 1024 / 1e6:7.1f} MB   write {wr_[k]['WRITE_SIZE'] * 1024 / 1e6:7.1f} MB")
write("r04_rowln_pmc.txt", "\n".join(rl) + "\n")

ta = pmc_table("attn_pmc", "attention")
at = ["# command: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES",
      "#          SQ_WAIT_INST_ANY --kernel-trace -- python3 tools/attn_bench.py --fmt fp16x3 fp16   (64 sequences x 501 tokens x 12 heads x 64: the encoder shape at B=32)",
      "# The attention kernels are round 3's (unchanged in round 4: DESIGN.md section 4.2 / section 9 say why the 64-rows-per-wave form was not built).", "",
      cap(SP.pmc, os.path.join(G, "attn_pmc"), "attention"), "# derived: cycles/XCD = GRBM_GUI_ACTIVE / 8; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)"]
for k, v in ta.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    at.append(f"#   {k:52s} cycles/XCD {cyc:9.0f}  MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %  VALU insts (incl. MFMA) per MFMA inst "
              f"{v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']:5.1f}  LDS bank conflicts {v['SQ_LDS_BANK_CONFLICT']:.0f}")
write("r04_attention_pmc.txt", "\n".join(at) + "\n")

write("r04_class_profile.txt", "# python3 tools/class_profile.py --precision fp16x3 fp16 ; --precision fp16x3 --fused-ln ; --refdefault --batch 16 --patches 512\n"
      "# (HIP events on the launch stream around every kernel class; BASELINE configs[1], B=32, N=500, and the reference-default topology)\n"
      + clean("class_profile.txt") + clean("class_profile_fused.txt") + clean("class_profile_refdefault.txt"))
write("r04_configs.txt", "# python3 tools/run_config.py ...: the other BASELINE / reference shapes end to end on one MI355X (timing + one pair against the oracle)\n"
      "# configs[3]: --variant ViT-L16 --batch 16 --patches 1024 --scales 3\n" + clean("config3_vitl.txt")
      + "# reference default topology (train_config.py:169-194): --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault\n" + clean("refdefault.txt"))
write("r04_golden_errors.txt", "# python3 tools/golden_errors.py   (GPU box): every golden case x mode, raw relative error against the reference's scores\n" + clean("golden_errors.txt"))
write("r04_fuzz_parity.txt", "# python3 tools/fuzz_parity.py --cases 100 --precision fp16x3   (GPU box): random topologies / shapes / inputs in the parity mode against the oracle\n"
      + clean("fuzz.txt"))
write("r04_pytest_gpu.txt", "# python3 -m pytest tests -m gpu -q   and   __graft_entry__.smoke()   (GPU box, the tree of this commit)\n"
      + clean("pytest_gpu.txt", keep=lambda l: "passed" in l or "failed" in l or "FAILED" in l or "skipped" in l or "error" in l.lower()) + clean("smoke.txt"))
