#!/usr/bin/env python3
"""Builds the round-2 evidence files under profiles/ from the raw collections merged back into gpurun_out/ by
tools/collect_profiles.sh (r02p), tools/collect_profiles2.sh (r02q) and the earlier measurement runs of this round
(tools/runs/: r02c gemm ablation, r02d CU-count sweep, r02e column-group sweep, r02f clock evidence, r02i fp8 stage parity,
r02ab same-box comparison with the round-1 tree).
Run in the repo after the gpurun calls:  python tools/make_profiles.py"""
import io, json, os, re, subprocess, sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_prof as SP  # noqa: E402


def cap(fn, *a):
    b = io.StringIO()
    with redirect_stdout(b):
        fn(*a)
    return b.getvalue()


def clean(path, keep=None):
    out = []
    for ln in open(os.path.join(G, path), errors="replace"):
        if "amdgpu.ids" in ln or "UserWarning" in ln or "warnings.warn" in ln:
            continue
        if keep is None or keep(ln):
            out.append(ln.rstrip("\n"))
    return "\n".join(out) + "\n"


def write(name, text):
    open(os.path.join(P, name), "w").write(text)
    print("wrote profiles/" + name, len(text), "bytes")


def pmc_table(d, pat):
    """{kernel: {counter: mean per dispatch}}"""
    import collections, csv, glob
    f = glob.glob(os.path.join(G, d) + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = SP.short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    return {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in agg.items()}


# ---- bench lines and kernel stats ----------------------------------------------------------------------------------------
line = [l for l in open(os.path.join(G, "r02p/bench_line.json")) if l.startswith('{"metric"')][-1]
write("r02_bench_line.json", line)
lp = [l for l in open(os.path.join(G, "r02p/bench_line_profiled.json")) if l.startswith('{"metric"')][-1]
write("r02_bench_line_profiled.json", lp)
write("r02_bench_kernel_stats.txt",
      "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (default run: headline fp16x3 at B=32, then the\n"
      "# other modes at B=32, then every mode at B=64: averages of one kernel name MIX B=32 and B=64 launches; the headline-only run,\n"
      "# whose fc1 average is the one the bench line's roofline reports, is r02_bench_headline_kernel_stats.txt)\n"
      + cap(SP.stats, os.path.join(G, "r02p/stats")))
lh = [l for l in open(os.path.join(G, "r02q/bench_line_headline_profiled.json")) if l.startswith('{"metric"')][-1]
dh = json.loads(lh)
write("r02_bench_line_headline_profiled.json", lh)
write("r02_bench_headline_kernel_stats.txt",
      "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-second-mode --no-north-star\n"
      f"# the same process printed roofline.avg_launch_ms = {dh['roofline']['avg_launch_ms']:.4f} ms for gemm_pp2_kernel<f16, 3, 1> (fc1, HIP events on the\n"
      "# launch stream inside the timed region); the rocprofv3 average below covers warm-up + timed launches of the same kernel.\n"
      + cap(SP.stats, os.path.join(G, "r02q/stats")))

# ---- fc1 GEMM: counters, clock evidence, traffic ---------------------------------------------------------------------------
t1, t2 = pmc_table("r02p/gemm_pmc", "gemm"), pmc_table("r02p/gemm_pmc2", "gemm")
txt = ["# command: rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3 fp16x2 fp16 bf16x3 fp8",
       "# fc1 GEMM of BASELINE configs[1]: M = 32256 (64 sequences x 501 rows, padded), N = 3072, K = 768, GELU epilogue; mean per dispatch.",
       "# GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* are summed over the 256 CUs (x4 SIMDs for the per-SIMD busy counters).", ""]
txt.append(cap(SP.pmc, os.path.join(G, "r02p/gemm_pmc"), "gemm"))
txt.append(cap(SP.pmc, os.path.join(G, "r02p/gemm_pmc2"), "gemm"))
txt.append("# derived (per kernel): cycles per XCD = GRBM_GUI_ACTIVE / 8; MFMA-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles per XCD);")
txt.append("# VALU issue = SQ_ACTIVE_INST_VALU / 4 / (1024 x cycles) [quad-cycles -> cycles as in the guide's counter table]")
for k, v in t1.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    extra = t2.get(k, {})
    txt.append(f"#   {k:34s} cycles/XCD {cyc:9.0f}   MFMA busy {busy*100:5.1f} %   MFMA insts {v['SQ_INSTS_MFMA']:.3e}   LDS bank-conflict cycles/CU "
               f"{v['SQ_LDS_BANK_CONFLICT']/256:8.0f}   VALU insts {extra.get('SQ_INSTS_VALU', 0):.3e}")
txt += ["", "# ---- clock under load (r02f): the same fc1 kernel on 8 CUs per XCD (M = 8192, VTQ_GEMM_CUS=8) and on all 32 (M = 32768): same work per CU",
        "# command: VTQ_GEMM_CUS=<n> rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... -- python3 tools/gemm_bench.py --only fc1 --M <256 n 4>"]
txt.append("# 8 CUs / XCD:\n" + cap(SP.pmc, os.path.join(G, "r02f/pmc_cus8"), "gemm"))
txt.append("# 32 CUs / XCD:\n" + cap(SP.pmc, os.path.join(G, "r02f/pmc_cus32"), "gemm"))
txt.append("# unprofiled durations of these two shapes (r02d sweep below): fp16x3 275.4 us at 8 CUs/XCD, 414.3 us at 32 CUs/XCD.\n"
           "# cycles/XCD = GRBM_GUI_ACTIVE / 8: 669640 (8 CUs) vs 689507 (32 CUs): the kernel takes the SAME number of cycles per CU-load,\n"
           "# clock = cycles / duration: 2.43 GHz with a quarter of the CUs busy, 1.67 GHz with all 256 busy.  The 1.5x longer full-chip launch\n"
           "# is the chip lowering its clock under the all-CU MFMA load, not contention inside the kernel.")
txt.append("\n# ---- CU-count sweep (r02d): VTQ_GEMM_CUS=<n> python3 tools/gemm_bench.py --only fc1 --M <n x 1024>; flags=8 = epilogue skipped (values then wrong by design)")
txt.append(clean("r02d/cus_sweep.txt"))
write("r02_gemm_fc1_pmc.txt", "\n".join(txt))

f = pmc_table("r02p/fetch", "gemm"); w = pmc_table("r02p/write", "gemm")
names = {"fp16x3": "gemm_pp2_kernel<f16, 3, 1>", "fp16x2": "gemm_pp2_kernel<f16, 2, 1>", "fp16": "gemm_pp2_kernel<f16, 1, 1>", "fp8": "gemm_pp2_kernel<f8, 1, 1>"}
bfk = [k for k in f if "garbled" in k]
if bfk:
    names["bf16x3"] = bfk[0]
M, N, K = 32256, 3072, 768
alg = {"fp16x3": (M * K * 4 + N * K * 4, M * N * 4), "bf16x3": (M * K * 4 + N * K * 4, M * N * 4), "fp16x2": (M * K * 4 + N * K * 2, M * N * 4),
       "fp16": (M * K * 2 + N * K * 2, M * N * 2), "fp8": (M * K + N * K, M * N)}
js = {"kernel": "gemm_pp2_kernel<T, TERMS, GELU> (fc1), M=32256 N=3072 K=768 (B=32 pairs)",
      "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gemm_bench.py --only fc1; see r02_gemm_fc1_traffic.txt",
      "note": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; L2-miss side bytes, "
              "Infinity-Cache hits included"}
tt = ["# commands: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/gemm_bench.py --only fc1 --rounds 1 --fmt fp16x3 fp16x2 fp16 bf16x3 fp8",
      "#           rocprofv3 --pmc WRITE_SIZE --kernel-trace -- (same)          separate passes; units KiB; mean per dispatch", "",
      cap(SP.pmc, os.path.join(G, "r02p/fetch"), "gemm"), cap(SP.pmc, os.path.join(G, "r02p/write"), "gemm"),
      "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B -> x2; WRITE_SIZE exact.  Both are the L2's",
      "# memory-side request counters: reads served by the 256 MiB Infinity Cache are INCLUDED (they are L2 misses, not HBM reads)."]
for mode, kn in names.items():
    if kn not in f:
        continue
    rd, wr = 2 * f[kn]["FETCH_SIZE"] * 1024, w[kn]["WRITE_SIZE"] * 1024
    a_r, a_w = alg[mode]
    js[mode] = {"FETCH_SIZE_KiB": f[kn]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[kn]["WRITE_SIZE"], "bytes_per_launch": rd + wr,
                "algorithmic_bytes_per_launch": a_r + a_w}
    tt.append(f"# {mode:7s}: read 2 x {f[kn]['FETCH_SIZE']:.0f} KiB = {rd/1e6:6.1f} MB (algorithmic A + W {a_r/1e6:6.1f} MB), write {wr/1e6:6.1f} MB "
              f"(algorithmic {a_w/1e6:6.1f} MB) -> {(rd+wr)/1e6:6.1f} MB per launch")
tt += ["#", "# Reads are 2.5-5.4x the algorithmic A + W bytes: every tile needs its whole 256-row A panel and 256-column W panel (K = 768 is one",
       "# short pass), a wave of 32 concurrent tiles per XCD touches (a + b) panels for a x b tiles, and the panels of one wave (0.8 MB each",
       "# in the 2-plane modes) already exceed the XCD's 4 MiB L2, so nothing survives to the next wave.  The re-reads are served by the",
       "# Infinity Cache (A + W + output of the whole GEMM < 256 MiB).  They do not bound the kernel: changing the tile order's column-group",
       "# width moves the read traffic by 16 % with no change in duration --",
       "# commands (r02q): VTQ_GEMM_CG=<cg> rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/gemm_bench.py --only fc1 --rounds 3 --fmt fp16x3 fp16",
       "# (durations inside these lines are under the counter pass; unprofiled sweep: r02e below)"]
for cg in (1, 3, 12):
    tt.append(f"# column group = {cg}:")
    tt.append(clean(f"r02q/sum_fetch_cg{cg}.txt", keep=lambda l: not l.startswith("# rocprofv3")).rstrip("\n"))
tt.append("# unprofiled column-group sweep (r02e): VTQ_GEMM_CG=<cg> python3 tools/gemm_bench.py --fmt fp16x3 fp16 --only qkv fc1")
tt.append(clean("r02e/cg_sweep.txt"))
write("r02_gemm_fc1_traffic.txt", "\n".join(tt))
write("r02_gemm_fc1_traffic.json", json.dumps(js, indent=1))

# ---- attention -------------------------------------------------------------------------------------------------------------
ta = pmc_table("r02p/attn_pmc", "attention")
at = ["# command: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES",
      "#          SQ_WAIT_INST_ANY --kernel-trace -- python3 tools/attn_bench.py      (64 sequences x 501 tokens x 12 heads x 64: the encoder shape at B=32)", "",
      cap(SP.pmc, os.path.join(G, "r02p/attn_pmc"), "attention"), "# derived: cycles/XCD = GRBM_GUI_ACTIVE / 8; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)"]
for k, v in ta.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    at.append(f"#   {k:52s} cycles/XCD {cyc:9.0f}  MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc*100:5.1f} %  VALU insts per MFMA inst "
              f"{v['SQ_INSTS_VALU']/v['SQ_INSTS_MFMA']:5.1f}  LDS bank conflicts {v['SQ_LDS_BANK_CONFLICT']:.0f}")
at += ["", "# unprofiled (python3 tools/attn_bench.py):", clean("r02p/attn_bench.txt")]
write("r02_attention_pmc.txt", "\n".join(at))

# ---- plain runs ------------------------------------------------------------------------------------------------------------
write("r02_gemm_bench.txt", "# python3 tools/gemm_bench.py --fmt fp16x3 fp16x2 fp16 bf16x3 bf16 fp8   (encoder GEMM shapes at B=32; median of 7 rounds x 5 launches)\n"
      + clean("r02p/gemm_bench.txt"))
write("r02_gemm_ablation.txt", "# python3 tools/gemm_bench.py under VTQ_GEMM_FLAGS (kernels.h): 0 = shipped (static per-workgroup tile lists, DMA ring chained across\n"
      "# tiles), 4 = one workgroup per tile in hardware dispatch order (the round-1 launch form, no cross-tile state), 8 = epilogue skipped (values\n"
      "# wrong by design: the epilogue's share of the launch), 12 = 8 + 4.  Same box, two passes over the flag list.\n" + clean("r02c/gemm_ab.txt"))
write("r02_class_profile.txt", "# python3 tools/class_profile.py   (HIP events on the launch stream around every kernel class; BASELINE configs[1], B=32, N=500)\n"
      + clean("r02p/class_profile.txt"))
write("r02_golden_errors.txt", "# python3 tools/golden_errors.py   (GPU box)\n" + clean("r02p/golden_errors.txt"))
write("r02_configs.txt", "# python3 tools/run_config.py ...: the other BASELINE / reference shapes end to end on one MI355X (timing + one pair against the oracle)\n"
      "# configs[3]: --variant ViT-L16 --batch 16 --patches 1024 --scales 3\n" + clean("r02p/config3_vitl.txt")
      + "# reference default topology (train_config.py:169-194): --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault\n" + clean("r02p/refdefault.txt")
      + "# long sequence (S = 2501 > the round-1 CLS-tail limit of 2048): --variant ViT-B16 --batch 4 --patches 2500 --scales 1\n" + clean("r02p/n2500.txt"))
if os.path.exists(os.path.join(G, "r02i/fp8_tests.txt")):
    write("r02_fp8_stage_parity.txt", "# python3 -m pytest tests/test_gpu_fp8.py -q -s -m gpu   (GPU box): teacher-forced stage parity and end-to-end noise level of the fp8 mode\n"
          + clean("r02i/fp8_tests.txt", keep=lambda l: l.startswith("[") or l.startswith("   ") or "passed" in l or "failed" in l))
