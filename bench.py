#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of the VTAMIQ ViT-B/16 pair forward (P=500 patches) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--patches P]
                    [--precision fp16x3|fp16x2|fp16|bf16x3|bf16|fp8]      (default fp16x3, the mode that meets the 1e-3 tolerance)

One process per GPU; every rank scores its own shard of the global batch (weak scaling: per-GPU batch fixed) and one RCCL
all-gather of the scores closes each step.  Two ways to get the N ranks:
  * the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK/LOCAL_RANK/WORLD_SIZE set);
  * `python bench.py --gpus N` on its own: the parent process starts the N ranks itself as child processes BEFORE it makes
    any GPU call (never an exec from a process that touched the GPU), forwards rank 0's JSON line and exits non-zero if
    any rank failed.
Inputs are synthetic and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.  Beside `value` (exactly K steps
after W warm-up steps, as the driver asks) the line carries `sustained`: the same workload after >= 2 s of back-to-back forwards,
timed over >= 3 s (the chip's clock under load settles over seconds), and `fidelity`: what every numerics mode does to
SROCC / KROCC / PLCC / RMSE against the fp32 oracle's scores on a distortion ladder.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6        # dense bf16 MFMA: 256 CU x 4096 flop/clk/CU x 2.4 GHz (MI355X_MICROARCH.md)
# MFMAs per product of the linear layers (the dominant kernel is a linear layer): vtamiq_amd/_lib.py MFMA_TERMS
# fp8: one e4m3 MFMA per product at twice the 16-bit rate = 0.5 bf16-MFMA-equivalents (its cap is 2x the bf16 roofline)
MFMA_PER_PRODUCT = {"bf16": 1.0, "bf16x3": 3.0, "fp16": 1.0, "fp16x2": 2.0, "fp16x3": 3.0, "fp8": 0.5}
HEADLINE = "fp16x3"          # default `value` mode; OTHER_MODES are timed beside it
OTHER_MODES = ("fp16x2", "fp16", "fp8")   # fp8: the EXPERIMENT's own library (vtamiq_amd/libvtamiq_hip_fp8.so, built by __graft_entry__.build())
                                          # through its own handle (_lib.load_fp8()); the product library does not have the mode


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N without torchrun
# ---------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int, argv, timeout_s: float) -> int:
    """Start n rank processes (this script, same argv) with the torch.distributed env contract, wait, forward rank 0's
    stdout.  The parent never imports torch or touches the GPU.  Children are stopped by PID if one fails or time runs out."""
    import threading
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    # rank 0's stdout is drained while it runs: a rank that prints more than the pipe holds must not block until the timeout
    chunks = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(procs[0].stdout.readline, "")), daemon=True)
    reader.start()
    deadline = time.time() + timeout_s
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        rc = rc or code or 1
                        print(f"[bench] rank {r} exited with code {code}", file=sys.stderr)
            if rc or time.time() > deadline:
                if not rc:
                    print(f"[bench] ranks {sorted(pending)} still running after {timeout_s:.0f} s", file=sys.stderr)
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                       # exact PIDs only
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = "".join(chunks)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"[bench] rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
def synth_inputs_on_device(torch, B, N, device, seed):
    """ref ~ U(-1,1); dist = clamp(ref + 0.1 N(0,1)); pos ~ U(0,1) <= 1-1e-6 shared (aligned)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ref = torch.rand(B, N, 3, 16, 16, device=device, generator=g) * 2 - 1
    dist = (ref + 0.1 * torch.randn(ref.shape, device=device, generator=g)).clamp_(-1, 1)
    pos = torch.rand(B, N, 2, device=device, generator=g).clamp_(max=1 - 1e-6)
    return (ref, dist), (pos, pos.clone()), (None, None)


LADDER_SIGMAS = (0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5)


def synth_ladder_on_device(torch, B, N, device, seed):
    """The same generator with the distortion strength varied per pair: sigma = LADDER_SIGMAS[i % 8] of the noise (a ladder of
    distortion levels, as an IQA test set has) -- the inputs of the `fidelity` block."""
    g = torch.Generator(device=device).manual_seed(seed)
    ref = torch.rand(B, N, 3, 16, 16, device=device, generator=g) * 2 - 1
    sig = torch.tensor([LADDER_SIGMAS[i % len(LADDER_SIGMAS)] for i in range(B)], device=device).view(B, 1, 1, 1, 1)
    dist = (ref + sig * torch.randn(ref.shape, device=device, generator=g)).clamp_(-1, 1)
    pos = torch.rand(B, N, 2, device=device, generator=g).clamp_(max=1 - 1e-6)
    return (ref, dist), (pos, pos.clone()), (None, None)


def mode_fidelity(torch, make_model, spec, sd_np, modes, device, pairs=64, N=500, chunk=32, seed=777, threads=None):
    """What the numerics modes cost in the reference's own metric (utils/misc/correlations.py:21-51): `pairs` synthetic pairs on a
    distortion ladder, the fp32 oracle's scores on the host as the target, every mode's scores as the prediction, through
    vtamiq_amd.validate.compute_correlations (the HIP rank / Kendall / Pearson kernels + the reference's logistic fit)."""
    import numpy as np
    from oracle import vtamiq_oracle as O
    from vtamiq_amd.validate import compute_correlations
    inp = synth_ladder_on_device(torch, pairs, N, device, seed)
    if threads:
        torch.set_num_threads(threads)
    sd = O.to_torch(sd_np)
    t0 = time.perf_counter()
    q_ref = []
    for i in range(0, pairs, 8):                             # the oracle in chunks of 8 pairs (memory of the attention matrices)
        sl = slice(i, min(i + 8, pairs))
        q_ref.append(O.vtamiq_forward(sd, spec, (inp[0][0][sl].cpu(), inp[0][1][sl].cpu()), (inp[1][0][sl].cpu(), inp[1][1][sl].cpu()),
                                      (None, None))[0])
    q_ref = torch.cat(q_ref)
    cpu_s = time.perf_counter() - t0
    ref_np = q_ref.double().numpy()
    rms = float(np.sqrt(np.mean(ref_np ** 2)))
    out = {"pairs": pairs, "patches": N, "ladder_sigmas": list(LADDER_SIGMAS), "target": "fp32 oracle scores (host)",
           "oracle_seconds": cpu_s, "rms_q_ref": rms, "modes": {}}
    q_dev = q_ref.to(device)
    for prec in modes:
        m = make_model(prec)
        qs = []
        with torch.no_grad():
            for i in range(0, pairs, chunk):
                sl = slice(i, min(i + chunk, pairs))
                qs.append(m((inp[0][0][sl], inp[0][1][sl]), (inp[1][0][sl], inp[1][1][sl]), (None, None))[0])
        q = torch.cat(qs)
        c = compute_correlations(q_dev, q)
        d = np.abs(q.double().cpu().numpy() - ref_np)
        big = np.abs(ref_np) >= 0.1 * rms
        out["modes"][prec] = {"SROCC": c["SROCC"], "KROCC": c["KROCC"], "PLCC": c["PLCC"], "RMSE": c["RMSE"],
                              "PLCC_NOFIT": c["PLCC_NOFIT"], "RMSE_NOFIT": c["RMSE_NOFIT"],
                              "max_rel_err_big_scores": float(np.max(d[big] / np.abs(ref_np[big]))), "max_abs_err_over_rms": float(d.max() / rms)}
        del m
        torch.cuda.empty_cache()
    return out


def e2e_validation(torch, model, B, N, device, batches=40, warmup=3, H=384, W=512, seed=99, sets=2):
    """End-to-end throughput from the dataloader boundary (train.py:583-644, data/patch_sampling.py:529-611): validation passes fed with HOST uint8
    images and host-sampled patch coordinates.  Per batch: pinned host buffers -> H2D on a copy stream -> vtamiq_amd.patches.extract_patches
    (normalise + gather on the GPU) -> the model's forward; the scores stay on the GPU and every pass ends with vtamiq_amd.validate's reductions
    (ranks / Kendall / Pearson kernels + the host's logistic fit).  Two buffer sets: the copy of batch i + 1 runs under the forward of batch i.
    `sets` passes of `batches` batches back to back -- train.py runs a validation AND a test pass per epoch -- with the DEFERRED form of the
    reductions (validate.compute_correlations_cat_flat(defer=True)): a pass's device reductions and its one D2H copy are enqueued behind its
    last forward, its logistic fit (MINPACK through scipy, 30 - 40 ms of host work for 1 280 scores) runs on a worker thread while the next
    pass's batches are enqueued; only the LAST pass's fit is waited for.  Returns pairs/s over all passes incl. every reduction."""
    import numpy as np
    import scipy.optimize                                    # noqa: F401  (validate's logistic fit: imported at start-up, as a validation process would)
    from vtamiq_amd.pipeline import ImagePairPipeline
    from vtamiq_amd.validate import compute_correlations_cat_flat
    NI = 2 * B
    rs = np.random.RandomState(seed)
    pipe = ImagePairPipeline(model, B, (H, W), N, device=device)          # pinned host buffers, copy stream, two buffer sets
    pool = [rs.randint(0, 256, size=(NI, H, W, 3), dtype=np.uint8) for _ in range(2)]        # the "decoded images" of the synthetic set
    noise = torch.from_numpy(rs.randn(batches * B).astype(np.float32)).to(device)
    # first use of the reduction kernels and of scipy's fit (code-object load, imports, the worker thread): not part of a steady-state validation pass
    compute_correlations_cat_flat([torch.linspace(0, 1, 64, device=device)], [torch.linspace(0, 1, 64, device=device) ** 2 + 0.01 * noise[:64]], defer=True).result()
    pending, t_enq = [], 0.0
    with torch.no_grad():
        t0 = None
        for s_ in range(sets):
            yps = []
            for i in range((warmup if s_ == 0 else 0) + batches):
                if s_ == 0 and i == warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                img, smp, _ = pipe.acquire()                # the loader's part: "decode" into the pinned slot, sample the coordinates (aligned pairs)
                img[:] = pool[i % 2]
                half = np.stack([rs.randint(0, H - 15, size=(B, N)), rs.randint(0, W - 15, size=(B, N))], axis=-1).astype(np.int32)
                smp[:B], smp[B:] = half, half
                q = pipe.launch()                           # range check, H2D on the copy stream, gather + forward: all enqueued
                if s_ > 0 or i >= warmup:
                    yps.append(q)
            # the set's MOS values: synthetic, correlated with the scores as a trained model's are (SROCC ~ 0.9), so that the logistic fit
            # behind PLCC / RMSE converges as it does on real data (built on the device, inside the timed region)
            te = time.perf_counter()
            qa = torch.cat(yps)
            ys = [(qa - qa.mean()) / qa.std() + 0.45 * noise]
            pending.append(compute_correlations_cat_flat(ys, [qa], defer=True))     # device reductions + the one D2H copy enqueued; the fit on a worker thread
            t_enq += time.perf_counter() - te
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0                    # every forward and every device reduction done
        corr = [p.result() for p in pending]
        dt = time.perf_counter() - t0
    bytes_pair = (2 * H * W * 3 + 2 * N * 2 * 4)
    total = sets * batches
    return {"value": total * B / dt, "unit": "image-pairs/s", "sets": sets, "batches": batches, "batch": B, "patches": N, "seconds": dt,
            "ms_per_batch": t_gpu / total * 1e3, "loop_seconds": t_gpu, "reductions_seconds": dt - t_gpu, "reductions_enqueue_seconds": t_enq,
            "value_loop_only": total * B / t_gpu, "image_hw": [H, W], "pcie_bytes_per_pair": bytes_pair,
            "pcie_gbps_at_this_rate": bytes_pair * total * B / dt / 1e9,
            "pipeline": "vtamiq_amd.pipeline.ImagePairPipeline: a host copy of the decoded uint8 images into pinned buffers + host-sampled coordinates "
                        "-> H2D on a copy stream (two buffer sets) -> extract_patches (normalise + gather) -> forward -> scores kept on the GPU -> "
                        "validate.compute_correlations_cat_flat(defer=True) at the end of each of the `sets` passes (the fit of a pass on a worker thread "
                        "under the next pass; reductions_seconds = what was still waited for after the last forward)",
            "SROCC_of_the_synthetic_targets": corr[-1]["SROCC"]}


def fc1_traffic(precision, B):
    """HBM bytes of ONE full-batch fc1 GEMM from the committed PMC passes (profiles/*_gemm_fc1_traffic.json: FETCH_SIZE x2
    gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc runs at B=32), scaled linearly with the batch.  A committed
    measurement of the same kernel, not a counter read during this run (rocprofv3 is not on bench.py's path)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_fc1_traffic.json")), reverse=True):      # newest round first
        try:
            d = json.load(open(path))
            return d[precision]["bytes_per_launch"] * (B / 32.0), "profiles/" + os.path.basename(path)
        except Exception:
            continue
    return None, None


LIVE_EXTRA = {}      # filled by live_fc1_traffic: counters of the third pass (MFMA-pipe utilisation)


def live_fc1_traffic(precision, B, S, timeout_s=75.0):
    """FETCH_SIZE and WRITE_SIZE of the fc1 GEMM measured NOW, on this box: two child processes (separate --pmc passes, as the
    MI355X guide prescribes; --kernel-trace only beside them; the program itself after `--`) of
    `rocprofv3 --pmc <counter> --kernel-trace -- python3 tools/gemm_bench.py --only fc1 --rounds 1 --fmt <precision>`,
    the same kernel on the same shape as the bench workload's fc1 launch (M = 2 B S padded, N = mlp_dim, K = hidden).
    Returns (bytes per launch with the gfx950 x2 correction on FETCH_SIZE, description) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under a ROCm profiler: no nested passes"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import summarize_prof as SP
    except Exception as e:                                   # pragma: no cover
        return None, f"tools/summarize_prof.py: {e}"
    elt = "f16" if precision.startswith("fp16") else ("f8" if precision == "fp8" else "bf16")
    terms = {"fp16x3": 3, "bf16x3": 3, "fp16x2": 2}.get(precision, 1)
    want = f"gemm_pp2_kernel<{elt}, {terms}, 1>"
    M = (2 * B * S + 255) // 256 * 256
    vals = {}
    tmp = tempfile.mkdtemp(prefix="vtq_pmc_", dir="/tmp")
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES")):
            counter = counters[0]
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "gemm_bench.py"), "--only", "fc1", "--rounds", "1", "--fmt", precision,
                   "--M", str(M)]
            env = dict(os.environ, TMPDIR="/tmp")
            # own session: on a timeout the whole group (rocprofv3 and the program under it) is stopped, by exact pgid
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                pr.wait()
                return None, f"rocprofv3 --pmc {counter} pass timed out"
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if rc != 0 or not files:
                if counter == "GRBM_GUI_ACTIVE":
                    break                                       # the utilisation pass is optional: keep the traffic
                return None, f"rocprofv3 --pmc {counter} failed (rc {rc})"
            tot, n = {c: 0.0 for c in counters}, {c: 0 for c in counters}
            for row in csv.DictReader(open(files[0])):
                name = SP.short(row["Kernel_Name"])
                if row["Counter_Name"] in tot and (name == want or (elt == "bf16" and name.startswith("gemm_pp2_kernel<bf16"))):
                    tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
            if min(n.values()) == 0:
                if counter == "GRBM_GUI_ACTIVE":
                    break
                return None, f"no {want} dispatch in the {counter} pass"
            for c in counters:
                vals[c] = tot[c] / n[c]
            if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[counter] *= 1024.0                        # KiB -> bytes, mean per dispatch
    except Exception as e:
        return None, f"live traffic measurement failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if "GRBM_GUI_ACTIVE" in vals and vals["GRBM_GUI_ACTIVE"] > 0:
        # MFMA-pipe busy fraction of the launch: busy SIMD-cycles / (1024 SIMDs x cycles per XCD); GRBM_GUI_ACTIVE is summed over 8 XCDs
        LIVE_EXTRA["mfma_busy_frac"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0)
        LIVE_EXTRA["cycles_per_xcd"] = vals["GRBM_GUI_ACTIVE"] / 8.0
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], (
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes, --kernel-trace only) on "
        f"tools/gemm_bench.py --only fc1 --fmt {precision} --M {M}; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
        f"FETCH {vals['FETCH_SIZE'] / 1e6:.1f} MB counted, WRITE {vals['WRITE_SIZE'] / 1e6:.1f} MB")


def effective_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(torch, spec, sd_np, seconds_budget=40.0):
    """the oracle (fp32 torch-CPU port of the reference path, kind "port") on this host's cores, eval/no-grad,
    at C1 (B=2, N=50) and B=8, N=500, with n = all usable cores and n = 8 threads.  Bounded: the B=8 rows are one warm-up +
    as many forwards as fit the budget (>= 1).  `value` = the B=8, N=500 rate on all cores."""
    from oracle import vtamiq_oracle as O
    from vtamiq_amd import synth
    ncores = min(effective_cores(), 64)
    sd = O.to_torch(sd_np)
    cases = []
    t_begin = time.perf_counter()
    threads = sorted({ncores, min(8, ncores)}, reverse=True)
    # SURVEY 8(d) protocol: 2 warm-ups + median of >= 5 forwards.  C1 always gets it (a forward is ~50 ms); the B = 8 x N = 500 rows
    # (2.5 - 4 s per forward) get 2 warm-ups + 5 on all cores while the budget lasts, never fewer than 1 warm-up + 1 timed forward.
    plan = [("C1", 2, 50, n, 2, 7) for n in threads] + [("B8N500", 8, 500, n, 2, 5) for n in threads]
    for name, Bc, Nc, nthr, warmups, reps in plan:
        nthr = min(nthr, ncores)
        torch.set_num_threads(nthr)
        patches, pos, _ = synth.make_inputs(spec, Bc, Nc, 4242)
        tp, tq = torch.from_numpy(patches), torch.from_numpy(pos)
        args = ((tp[:, 0], tp[:, 1]), (tq[:, 0], tq[:, 1]), (None, None))
        bounded = name != "C1"
        nwarm, warm = 0, 0.0
        while nwarm < warmups and (nwarm == 0 or not bounded or time.perf_counter() - t_begin + 2 * warm < seconds_budget):
            t0 = time.perf_counter()
            O.vtamiq_forward(sd, spec, *args)
            warm = time.perf_counter() - t0
            nwarm += 1
        times = []
        while len(times) < reps and (not times or not bounded or time.perf_counter() - t_begin + warm < seconds_budget):
            t0 = time.perf_counter()
            O.vtamiq_forward(sd, spec, *args)
            times.append(time.perf_counter() - t0)
        med = sorted(times)[len(times) // 2]
        cases.append({"case": name, "B": Bc, "N": Nc, "threads": nthr, "ms_per_forward": med * 1e3, "pairs_per_s": Bc / med,
                      "warmups": nwarm, "forwards_timed": len(times)})
    head = [c for c in cases if c["case"] == "B8N500"][0]
    return {"value": head["pairs_per_s"], "unit": "image-pairs/s", "cores": head["threads"], "kind": "port",
            "sample": f"oracle fp32 (torch CPU ops), ViT-B/16 L={spec.num_layers}: B=8 x N=500 on {head['threads']} threads "
                      f"({head['warmups']} warm-ups + median of {head['forwards_timed']}); C1 (B=2, N=50): 2 warm-ups + median of 7; all rows in `cases`",
            "ms_per_forward": head["ms_per_forward"], "cases": cases}


def practical_peak(precision, warm_s=1.0, timed_s=1.0):
    """What the matrix pipe sustains on THIS device with the operand bits of this mode (vtq_debug_mfma_stream, csrc/mfma_stream.hip): a bare
    stream of back-to-back 16x16x32 MFMAs on register operands on every CU, >= timed_s after >= warm_s of the same load.  3-term modes
    stream the hi x hi / hi x lo / lo x hi mix of gaussian planes, single-plane modes uniform random data; zeros give the issue limit."""
    import ctypes as C
    import torch
    from vtamiq_amd import _lib
    if precision == "fp8":
        return None
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    f16 = 1 if precision.startswith("fp16") else 0
    data = 0 if precision.endswith(("x3", "x2")) else 2
    out = {}
    for key, d, w, t in (("operands_of_this_mode", data, warm_s, timed_s), ("zeros_issue_limit", 1, 0.3, 0.5)):
        tf, ghz = C.c_double(0.0), C.c_double(0.0)
        _lib.check(lib.vtq_debug_mfma_stream(f16, d, w, t, C.byref(tf), C.byref(ghz), stream))
        out[key] = {"tflops": tf.value, "implied_clock_ghz": ghz.value, "warm_s": w, "timed_s": t}
    out["operands"] = ("3-term mix of gaussian hi / lo planes (hi x hi, hi x lo, lo x hi)" if data == 0 else "uniform random single planes") + \
                      (", fp16" if f16 else ", bf16")
    return out


def latency_block(torch, make_model, precision, device, N, batches=(1, 2, 4, 8, 16), steps=20):
    """The small-batch / single-query regime (the reference's own FLOP probe is batch 1 x 500 patches, modules/utils.py:68-78): ms per
    forward with the forwards queued back to back (HIP events) and one at a time (the host waits for every score), and the forward's
    fraction of the MFMA roofline, at B pairs x N patches on the bench model; GEMM launches with fewer than 64 tiles of 256x256 run the
    small-tile kernels (csrc/gemm_st.hip), bit-identical scores (profiles/r05_small_batch.txt: B = 1 2.98 -> 1.60 ms)."""
    m = make_model(precision)
    spec = m.spec
    rows = []
    with torch.no_grad():
        for Bc in batches:
            inp = synth_inputs_on_device(torch, Bc, N, device, 9000 + Bc)
            for _ in range(4):
                m(*inp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                m(*inp)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            t0 = time.perf_counter()
            for _ in range(steps):
                m(*inp)
                torch.cuda.synchronize()
            ms_sync = (time.perf_counter() - t0) / steps * 1e3
            f = spec.flops_per_pair_executed(N, cls_prune=precision != "fp8")
            rows.append({"batch": Bc, "token_rows": 2 * Bc * spec.seq_len(N), "ms_per_forward": ms, "ms_per_forward_synchronous": ms_sync,
                         "pairs_per_s": Bc / ms * 1e3, "forward_mfma_frac": Bc / ms * 1e3 * f / (PEAK_BF16_TFLOPS * 1e12)})
    del m
    torch.cuda.empty_cache()
    return rows


def long_sequence_block(torch, make_model, precision, device, patches=(2500, 5000), B=4, steps=6):
    """The long-sequence regime the reference advertises (README.md:85: "50, 500, and 5000 patches"; data/patch_sampling.py:450): B pairs x N patches on the
    bench model (ViT-B/16, L = 12), ms per forward (HIP events, forwards queued back to back), the forward's fraction of the MFMA roofline, and attention's
    share of the step from the engine's per-class events (the S^2 term is 32 % of the flops at N = 2500, 49 % at N = 5000).  Scores at N = 5000 are pinned by
    the reference (tests/golden/long_b2_n5000.npz, long5h_b1_n5000.npz: test_golden_long_sequence)."""
    from vtamiq_amd import _lib
    m = make_model(precision)
    spec = m.spec
    rows = []
    with torch.no_grad():
        for N in patches:
            inp = synth_inputs_on_device(torch, B, N, device, 7000 + N)
            for _ in range(2):
                m(*inp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                m(*inp)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            m.profile_enable(list(_lib.KERNEL_CLASSES))
            for _ in range(steps):
                m(*inp)
            prof = m.profile_collect()
            m.profile_enable([])
            tot = sum(v[0] for v in prof.values())
            S = spec.seq_len(N)
            f, f_alg = spec.flops_per_pair_executed(N, cls_prune=precision != "fp8"), spec.flops_per_pair(N)
            f_att = 2 * spec.num_layers * 4.0 * S * S * spec.hidden_size
            rows.append({"batch": B, "patches": N, "seq_len": S, "ms_per_forward": ms, "pairs_per_s": B / ms * 1e3,
                         "forward_mfma_frac": B / ms * 1e3 * f / (PEAK_BF16_TFLOPS * 1e12),
                         "attention_share_of_flops": f_att / f_alg, "attention_share_of_time": prof["attention"][0] / tot if tot > 0 else None,
                         "attention_ms_per_forward": prof["attention"][0] / steps,
                         "attention_tflops_algorithmic": (B * f_att * (spec.num_layers - 1) / spec.num_layers) / (prof["attention"][0] / steps * 1e-3) / 1e12 if prof["attention"][0] > 0 else None,
                         "workspace_mib": m.workspace_bytes(B, N) / 2 ** 20})
            del inp
            torch.cuda.empty_cache()
    del m
    torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU (BASELINE config 2: 32)")
    ap.add_argument("--patches", type=int, default=500)
    ap.add_argument("--precision", default=os.environ.get("VTAMIQ_BENCH_PRECISION", HEADLINE), choices=sorted(MFMA_PER_PRODUCT))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-mode", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC passes instead of measuring it")
    ap.add_argument("--no-sustained", action="store_true", help="skip the `sustained` block (>= 2 s warm-up + >= 3 s timed)")
    ap.add_argument("--sustained-seconds", type=float, nargs=2, default=(2.0, 3.0), metavar=("WARM", "TIMED"))
    ap.add_argument("--no-fidelity", action="store_true", help="skip the `fidelity` block (mode scores vs the fp32 oracle on a distortion ladder)")
    ap.add_argument("--fidelity-pairs", type=int, default=64)
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1: put the world-size-1 RCCL all-gather of the scores INTO the timed step as well (default: it is "
                         "executed and timed in its own block, `collective`)")
    ap.add_argument("--no-collective-check", action="store_true", help="--gpus 1: skip the world-size-1 RCCL block")
    ap.add_argument("--no-e2e", action="store_true", help="skip the `e2e` block (validation loop from host uint8 images)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the reference-default-topology row (`secondary`)")
    ap.add_argument("--no-latency", action="store_true", help="skip the `latency` block (B = 1 .. 16 pairs per forward)")
    ap.add_argument("--no-long-sequence", action="store_true", help="skip the `long_sequence` block (B = 4 pairs at 2500 and 5000 patches)")
    ap.add_argument("--no-practical-peak", action="store_true", help="skip the in-run measurement of the matrix pipe's sustained rate")
    ap.add_argument("--no-auto-overhead", action="store_true", help="skip timing the default-constructed model (precision='auto')")
    ap.add_argument("--other-modes-multi-gpu", action="store_true",
                    help="--gpus N > 1: also time the other numerics modes on every rank (triples the run; off by default there)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launch: seconds before the ranks are stopped")
    # launcher self-test on CPU (tests/test_bench_launcher.py): gloo ranks + a stub model, no HIP anywhere
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:], a.launch_timeout))

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner sits in libc's buffer until exit and
    # lands BEHIND the line): from here on file descriptor 1 is stderr, and the line goes to a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # dmabuf IPC is the only form this pool's host driver supports (RCCL across processes fails with the legacy one); already exported
    # on the boxes, kept here for a launcher that built its own environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from vtamiq_amd.dist import gather_scores

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = a.backend == "nccl"
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # One GPU: the RCCL path is still exercised -- a world-size-1 process group on the nccl backend (= RCCL) and the real
    # gather_scores -> all_gather_into_tensor on the compute stream.  A failure here is recorded, never fatal.
    collective = None
    if world == 1 and on_gpu and not a.stub and not a.no_collective_check:
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            collective = {"backend": "nccl (RCCL)", "world_size": 1}
        except Exception as e:                       # pragma: no cover
            collective = {"collective_executed": False, "error": f"init_process_group: {e!r}"[:300]}
    # Multi-GPU first-try checklist: every rank reports the device it runs on; N distinct devices or no timing.
    rank_devices = None
    if on_gpu:
        pr = torch.cuda.get_device_properties(device)
        ident = {"rank": rank, "local_rank": local_rank, "device_index": device.index, "name": torch.cuda.get_device_name(device),
                 "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0)),
                 "uuid": str(getattr(pr, "uuid", "")), "cus": pr.multi_processor_count, "hbm_gib": round(pr.total_memory / 2 ** 30, 1)}
    else:
        ident = {"rank": rank, "local_rank": local_rank, "device_index": None, "name": "cpu (launcher self-test)", "pci_bus_id": f"cpu:{rank}", "uuid": f"cpu:{rank}"}
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, ident)
        keys = {(d["pci_bus_id"], d["uuid"], d["device_index"]) for d in rank_devices}
        assert len(keys) == world, f"{world} ranks on {len(keys)} distinct devices: {rank_devices}"
        print(f"[bench] rank {rank}: device {ident['device_index']} {ident['name']} pci {ident['pci_bus_id']}", file=sys.stderr, flush=True)
    else:
        rank_devices = [ident]
    force_coll = bool(a.force_collective and collective is not None and "error" not in collective)
    if os.environ.get("VTQ_BENCH_FAIL_RANK") == str(rank):          # launcher test hook: this rank dies before the first collective
        sys.exit(3)
    rccl_ranks = dist.get_world_size() if world > 1 else 1

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    B, N = a.batch, a.patches
    global_batch = B * world
    inputs = synth_inputs_on_device(torch, B, N, device, 1234 + rank)

    rank_times = []          # per-rank wall time of the last run() (all ranks hold all of them)
    event_ms = []            # this rank's hipEvent-pair time of the last run()'s timed loop

    def run(model, steps, warmup, inp=inputs, gb=global_batch, profile_class=None):
        with torch.no_grad():
            for _ in range(warmup):
                q = gather_scores(model(*inp)[0], gb, force_collective=force_coll)
            sync()
            if profile_class:
                model.profile_enable([profile_class])
            if world > 1:
                dist.barrier()
            sync()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if on_gpu else None
            t0 = time.perf_counter()
            if ev:
                ev[0].record()                               # a hipEvent pair on the launch stream around the same K steps
            for _ in range(steps):
                q = gather_scores(model(*inp)[0], gb, force_collective=force_coll)
            if ev:
                ev[1].record()
            sync()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            if ev:
                event_ms[:] = [ev[0].elapsed_time(ev[1])]
            prof = None
            if profile_class:
                prof = model.profile_collect()[profile_class]
                model.profile_enable([])
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        if world > 1:
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_times[:] = [float(e.item()) for e in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            rank_times[:] = [dt]
        return float(t.item()), q, prof

    def run_sustained(model, warm_s, timed_s, inp=inputs, profile_class=None):
        """Back-to-back forwards (at most 3 in flight: the host never idles the GPU, the queue never grows) for warm_s seconds
        untimed, then for >= timed_s seconds timed."""
        def spin(seconds):
            evs, n = [], 0
            t_end = time.perf_counter() + seconds
            t0 = time.perf_counter()
            while time.perf_counter() < t_end:
                model(*inp)
                e = torch.cuda.Event()
                e.record()
                evs.append(e)
                n += 1
                if len(evs) > 3:
                    evs.pop(0).synchronize()
            sync()
            return n, time.perf_counter() - t0
        with torch.no_grad():
            spin(warm_s)
            if profile_class:
                model.profile_enable([profile_class])
            n, dt_s = spin(timed_s)
            prof = None
            if profile_class:
                prof = model.profile_collect()[profile_class]
                model.profile_enable([])
        return n, dt_s, prof

    if a.stub:                                # launcher / collective plumbing only; never a measurement
        from tests.bench_stub import StubModel
        dt, q, _ = run(StubModel(), a.steps, a.warmup)
        assert q.shape == (global_batch,)
        if rank == 0:
            print(file=real_stdout, flush=True, *[json.dumps({"metric": "launcher self-test (stub model, no HIP)", "value": global_batch * a.steps / dt,
                              "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f32", "data": "stub", "rccl_ranks": rccl_ranks, "backend": a.backend,
                              "rank_step_ms": {"min": min(rank_times) / a.steps * 1e3, "max": max(rank_times) / a.steps * 1e3},
                              "rank_devices": rank_devices,
                              "config": {"workload": "stub", "global_batch": global_batch, "parallelism": f"dp{world}"},
                              "q_checksum": float(q.double().sum())})])
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from vtamiq_amd import VTAMIQ, synth, _lib as _vlib
    fp8_ok = bool(on_gpu and _vlib.fp8_available())
    if a.precision == "fp8" and not fp8_ok:
        sys.exit("bench.py --precision fp8: vtamiq_amd/libvtamiq_hip_fp8.so (the fp8 experiment's library) is not built: python -m vtamiq_amd.build --fp8")

    def model_class(precision):
        if precision != "fp8":
            return VTAMIQ
        from vtamiq_amd.experimental_fp8 import VTAMIQFp8
        return VTAMIQFp8
    kw = dict(vit_config=dict(variant="ViT-B16", pretrained=False))   # L=12, T=1, r=8: BASELINE configs 2/3 and north star; seeded weights
    spec = model_class(a.precision)(**json.loads(json.dumps(kw)), precision=a.precision).spec
    sd_np = synth.make_state_dict(spec, 0)
    state = {k: torch.from_numpy(v) for k, v in sd_np.items()}

    def make_model(precision):
        m = model_class(precision)(**json.loads(json.dumps(kw)), precision=precision)
        m.load_state_dict(state)
        return m.to(device).eval()

    model = make_model(a.precision)
    DOM = "fc1"
    dt, q, prof = run(model, a.steps, a.warmup, profile_class=DOM)
    assert q.shape == (global_batch,) and bool(torch.isfinite(q).all())
    pairs_per_s = global_batch * a.steps / dt
    headline_rank_times = list(rank_times)
    headline_event_ms = event_ms[0] / a.steps if event_ms else None
    allgather_us = None
    if world > 1:                                 # the step's one collective, timed alone: a missed scaling target can be read off the record
        qloc = torch.zeros(B, device=device)
        for _ in range(10):
            gather_scores(qloc, global_batch)
        sync(); dist.barrier(); sync()
        t0 = time.perf_counter()
        for _ in range(100):
            gather_scores(qloc, global_batch)
        sync()
        tg = torch.tensor([(time.perf_counter() - t0) / 100 * 1e6], device=device, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        allgather_us = float(tg.item())
    if collective is not None and "error" not in collective:
        try:                                      # the same call sequence on ONE rank: a step's scores through the real collective, then the collective alone
            with torch.no_grad():
                q_direct = model(*inputs)[0]
                q_coll = gather_scores(model(*inputs)[0], global_batch, force_collective=True)
            sync()
            qloc = torch.zeros(B, device=device)
            for _ in range(10):
                gather_scores(qloc, global_batch, force_collective=True)
            sync()
            t0 = time.perf_counter()
            for _ in range(100):
                gather_scores(qloc, global_batch, force_collective=True)
            sync()
            allgather_us = (time.perf_counter() - t0) / 100 * 1e6
            collective.update({"collective_executed": True, "op": "all_gather_into_tensor on the compute stream (vtamiq_amd.dist.gather_scores)",
                               "scores_identical_to_direct": bool(torch.equal(q_direct, q_coll) and q_coll.data_ptr() != q_direct.data_ptr()),
                               "allgather_us": allgather_us, "in_timed_step": force_coll})
        except Exception as e:                       # pragma: no cover
            collective = {"collective_executed": False, "error": f"{e!r}"[:300]}
    auto_cost = None
    if on_gpu and world == 1 and not a.no_auto_overhead and a.precision == HEADLINE:
        # what a default-constructed model costs on top of this line's mode: precision="auto" = fp16x3 + the engine's error word read after
        # every forward (one 4-byte D2H + stream sync per call), same K steps after the same warm-up, measured in this run
        try:
            m_auto = make_model("auto")
            dt_auto, q_auto, _ = run(m_auto, a.steps, a.warmup)
            auto_cost = {"value": global_batch * a.steps / dt_auto, "ms_per_step": dt_auto / a.steps * 1e3, "overhead": dt_auto / dt - 1.0,
                         "scores_identical_to_explicit_mode": bool(torch.equal(q_auto, q)), "engine_precision": m_auto.engine_precision}
            del m_auto
        except Exception as e:                           # an optional block never costs the line
            auto_cost = {"error": f"{e!r}"[:300]}
        torch.cuda.empty_cache()
    sustained = None
    if on_gpu and world == 1 and not a.no_sustained:
        n_s, dt_s, prof_s = run_sustained(model, a.sustained_seconds[0], a.sustained_seconds[1], profile_class=DOM)
        sustained = {"value": B * n_s / dt_s, "unit": "image-pairs/s", "forwards": n_s, "seconds": dt_s, "warmup_seconds": a.sustained_seconds[0],
                     "ms_per_step": dt_s / n_s * 1e3,
                     "note": "same workload and model as `value`; back-to-back forwards (<= 3 in flight) for warmup_seconds untimed, then timed"}
        if prof_s and prof_s[1] > 0:
            sustained["fc1_avg_launch_ms"] = prof_s[0] / prof_s[1]
            sustained["fc1_launches"] = int(prof_s[1])
    f_pair = spec.flops_per_pair(N)                      # algorithmic (SURVEY 8d / BASELINE.md)
    pruned = True                                        # the product path: CLS-only last layer (vtq_config.options = 0)
    # executed flops per pair, per mode: the CLS-only last layer is used by every mode but fp8 (there the CLS row must go through
    # the same e4m3 GEMMs as every other row, engine.hip run_encoder), which executes the algorithmic count
    f_exec_of = lambda prec: spec.flops_per_pair_executed(N, cls_prune=pruned and prec != "fp8")
    f_exec = f_exec_of(a.precision)
    S = spec.seq_len(N)
    mfma_frac = lambda pps, prec=a.precision: pps / world * f_exec_of(prec) / (PEAK_BF16_TFLOPS * 1e12)
    others = [m for m in OTHER_MODES if m != a.precision and (m != "fp8" or fp8_ok)]

    out = {
        "metric": "image-pairs/sec ViT-B/16 P=500 patches, 1->8 MI355X; % bf16 MFMA roofline",
        "value": pairs_per_s, "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # the same K steps between a hipEvent pair on the launch stream (rank 0; `ms_per_step` is the barrier + synchronize bracket the contract asks for)
        "ms_per_step_hip_events": headline_event_ms,
        "dtype": "f16" if a.precision.startswith("fp16") else ("e4m3" if a.precision == "fp8" else "bf16"), "data": "synthetic",
        "rccl_ranks": rccl_ranks,
        "collective_executed": bool(world > 1 or (collective or {}).get("collective_executed", False)),
        "collective": collective,
        # what a default-constructed model runs: precision="auto" = this line's fp16x3 + the error word read after every forward
        "default_precision": "auto", "auto_overhead": (auto_cost or {}).get("overhead"), "auto_overhead_measured_in_this_run": auto_cost,
        "rank_devices": rank_devices,
        "fp8_experiment_library_loaded": fp8_ok,
        "rank_step_ms": {"min": min(headline_rank_times) / a.steps * 1e3, "max": max(headline_rank_times) / a.steps * 1e3},
        "allgather_us": allgather_us,
        "config": {"workload": f"BASELINE configs[{1 if world == 1 else 2}]: ViT-B/16 (L=12, T=1) FR pair forward, batch={B} pairs/GPU, "
                               f"{N} patches of 16x16x3, random-init seeded weights",
                   "global_batch": global_batch, "patches": N, "seq_len": S, "parallelism": f"dp{world}",
                   "numerics": a.precision,
                   "numerics_note": "16-bit MFMA operands (fp16 and bf16 MFMAs run at the same rate; peak = the bf16 dense peak), fp32 "
                                    "accumulate, fp32 LayerNorm/softmax/GELU/residual.  fp16x3 = hi/lo fp16 split of both operands, 3 "
                                    "MFMAs per product (scores at the fp32 reference's own noise floor); fp16x2 = activations split, "
                                    "weights single fp16 in the linear layers, 2 MFMAs per product (attention stays 3-term); fp16 / "
                                    "bf16 = 1 MFMA per product (throughput modes, outside the 1e-3 parity tolerance, never claimed as "
                                    "parity).  fp8 = BASELINE configs[4]: e4m3 weights (per-output-channel scales) and e4m3 activations (static "
                                    "scales) on the MX-scaled MFMA at twice the 16-bit rate, single-fp16 attention; a different MODEL "
                                    "(3 mantissa bits), checked against its own fake-quant oracle, tens of percent from the fp32 scores "
                                    "on random-init weights.  A k-MFMA mode can reach at most 1/k of the MFMA roofline "
                                    "(roofline.mode_cap)"},
        # executed flops: the last layer runs Q/attention/out-proj/MLP for the CLS row only (legal: only token 0 is consumed)
        "forward_mfma_frac": mfma_frac(pairs_per_s),
        "flops_per_pair": f_pair, "flops_per_pair_executed": f_exec,
    }

    def roofline_of(prof_entry, steps, note):
        ms_sum, launches = prof_entry
        full_layers = spec.num_layers - (1 if (pruned and a.precision != "fp8") else 0)       # fc1 GEMM launches per step (fp8 runs the full last layer)
        per_step = launches / steps
        flops_launch = 2.0 * (2 * B * S) * spec.hidden_size * spec.mlp_dim * full_layers / per_step    # algorithmic, unpadded rows
        ach = flops_launch / (ms_sum / launches * 1e-3) / 1e12
        traffic, src = fc1_traffic(a.precision, B)
        mpp = MFMA_PER_PRODUCT[a.precision]
        elt = "f16" if a.precision.startswith("fp16") else ("f8" if a.precision == "fp8" else "bf16")
        return {"bound": "mfma", "kernel": f"gemm_pp2_kernel<{elt}, {max(1, int(mpp))}, GELU> (fc1 of every full layer)",
                "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                "traffic": traffic * full_layers / per_step if traffic else None, "traffic_source": src,
                # the numerics ceiling: bf16x3 issues 3 bf16 MFMAs per algorithmic product
                "mode_cap": 1.0 / mpp, "frac_of_mode_cap": ach / PEAK_BF16_TFLOPS * mpp,
                "avg_launch_ms": ms_sum / launches, "launches": int(launches), "launches_per_step": per_step,
                "flops_per_launch": flops_launch, "note": note}

    if sustained:
        sustained["forward_mfma_frac"] = mfma_frac(sustained["value"])
        if "fc1_avg_launch_ms" in sustained:
            fl = 2.0 * (2 * B * S) * spec.hidden_size * spec.mlp_dim
            sustained["fc1_tflops"] = fl / (sustained["fc1_avg_launch_ms"] * 1e-3) / 1e12
            sustained["fc1_roofline_frac"] = sustained["fc1_tflops"] / PEAK_BF16_TFLOPS
        out["sustained"] = sustained
    if prof and prof[1] > 0:
        out["roofline"] = roofline_of(prof, a.steps, "HIP events recorded by the engine on the launch stream around every fc1 launch "
                                      "of the timed region of `value`")
        if on_gpu and world == 1 and not a.no_practical_peak:
            # the denominator the performance argument rests on, measured on THIS box in THIS run: what a bare MFMA
            # stream with this mode's operand bits sustains on all CUs; the dominant kernel's MFMA ISSUE rate (achieved x MFMAs per
            # product) against it
            try:
                pk = practical_peak(a.precision)
            except Exception as e:                       # an optional block never costs the line
                pk = None
                out["roofline"]["practical_peak_error"] = f"{e!r}"[:300]
            if pk:
                rf = out["roofline"]
                rf["practical_peak_tflops_measured_here"] = pk["operands_of_this_mode"]["tflops"]
                rf["practical_peak"] = pk
                rf["mfma_issue_tflops"] = rf["achieved"] * MFMA_PER_PRODUCT[a.precision]
                rf["frac_of_practical"] = rf["mfma_issue_tflops"] / pk["operands_of_this_mode"]["tflops"]
                rf["practical_peak_as_frac_of_nominal"] = pk["operands_of_this_mode"]["tflops"] / PEAK_BF16_TFLOPS
                out["forward_frac_of_practical"] = out["forward_mfma_frac"] * PEAK_BF16_TFLOPS * MFMA_PER_PRODUCT[a.precision] / pk["operands_of_this_mode"]["tflops"]
    q_other = {}
    if not a.no_second_mode and (world == 1 or a.other_modes_multi_gpu):      # every rank: run() is collective; N > 1: the headline mode only by default
        del model
        torch.cuda.empty_cache()
        out["other_modes"] = {}
        for other in others:
            model2 = make_model(other)
            dt2, q2, _ = run(model2, a.steps, a.warmup)
            q_other[other] = q2
            out["other_modes"][other] = {"value": global_batch * a.steps / dt2, "unit": "image-pairs/s",
                                         "forward_mfma_frac": mfma_frac(global_batch * a.steps / dt2, other),
                                         "flops_per_pair_executed": f_exec_of(other),
                                         "mfma_per_product_linear": MFMA_PER_PRODUCT[other]}
            del model2
            torch.cuda.empty_cache()
    if not a.no_north_star and world == 1:
        # BASELINE north-star operating point: B = 64 pairs on one GPU, every numerics mode (the >= 40 % target is stated on it)
        Bn = 64
        inp64 = synth_inputs_on_device(torch, Bn, N, device, 4321)
        ns = {"batch": Bn, "patches": N, "target_forward_mfma_frac": 0.40}
        for prec in [a.precision] + others:
            m = make_model(prec)
            nsteps = max(3, a.steps // 2)
            dtn, _, _ = run(m, nsteps, 2, inp=inp64, gb=Bn)
            ns[prec] = {"value": Bn * nsteps / dtn, "unit": "image-pairs/s", "ms_per_step": dtn / nsteps * 1e3,
                        "forward_mfma_frac": Bn * nsteps / dtn * f_exec_of(prec) / (PEAK_BF16_TFLOPS * 1e12),
                        "mode_cap": 1.0 / MFMA_PER_PRODUCT[prec]}
            del m
            torch.cuda.empty_cache()
        out["north_star_point"] = ns
    if not a.no_e2e and world == 1 and on_gpu:
        me = make_model(a.precision)
        out["e2e"] = e2e_validation(torch, me, B, N, device)
        out["e2e"]["frac_of_value"] = out["e2e"]["value"] / pairs_per_s
        out["e2e"]["loop_frac_of_value"] = out["e2e"]["value_loop_only"] / pairs_per_s
        out["e2e"]["numerics"] = a.precision
        del me
        torch.cuda.empty_cache()
    if not a.no_secondary and world == 1 and on_gpu:
        # SURVEY 8(d) secondary row: the reference-default topology -- what the released checkpoint runs (train_config.py:169-194,
        # 356-369: 6 kept layers, 8 register tokens, LayerScale, ca_reduction 16) at B = 16, N = 512
        kw2 = dict(vit_config=dict(variant="ViT-B16", pretrained=False, num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True), ca_reduction=16)
        m2 = model_class(a.precision)(**json.loads(json.dumps(kw2)), precision=a.precision)
        spec2 = m2.spec
        m2.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec2, 0).items()})
        m2 = m2.to(device).eval()
        B2, N2 = 16, 512
        inp2 = synth_inputs_on_device(torch, B2, N2, device, 2468)
        n2 = max(10, a.steps)
        dt2s, _, _ = run(m2, n2, 3, inp=inp2, gb=B2)
        f2 = spec2.flops_per_pair_executed(N2, cls_prune=a.precision != "fp8")
        out["secondary"] = {"workload": "reference-default topology (train_config.py:169-194: L=6, T=9, LayerScale, r=16), batch=16 pairs, 512 patches",
                            "numerics": a.precision, "value": B2 * n2 / dt2s, "unit": "image-pairs/s", "ms_per_step": dt2s / n2 * 1e3,
                            "forward_mfma_frac": B2 * n2 / dt2s * f2 / (PEAK_BF16_TFLOPS * 1e12),
                            "flops_per_pair": spec2.flops_per_pair(N2), "flops_per_pair_executed": f2, "seq_len": spec2.seq_len(N2)}
        del m2
        torch.cuda.empty_cache()
    if not a.no_latency and world == 1 and on_gpu:
        kwl = dict(vit_config=dict(variant="ViT-B16", pretrained=False, num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True), ca_reduction=16)

        def make_refdefault(precision):
            ml = model_class(precision)(**json.loads(json.dumps(kwl)), precision=precision)
            ml.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(ml.spec, 0).items()})
            return ml.to(device).eval()
        try:
            out["latency"] = {"numerics": a.precision, "patches": N, "workload": "ViT-B/16 (L=12, T=1) FR pair forward at B pairs per forward",
                              "rows": latency_block(torch, make_model, a.precision, device, N)}
            out["latency"]["reference_default_topology"] = {"workload": "L=6, T=9, LayerScale, r=16 (train_config.py:169-194), 512 patches",
                                                            "rows": latency_block(torch, make_refdefault, a.precision, device, 512, batches=(1, 16))}
        except Exception as e:                           # an optional block never costs the line
            out.setdefault("latency", {})["error"] = f"{e!r}"[:300]
    if not a.no_long_sequence and world == 1 and on_gpu:
        try:
            out["long_sequence"] = {"numerics": a.precision, "workload": "ViT-B/16 (L=12, T=1) FR pair forward, 4 pairs per forward at 2500 / 5000 patches "
                                                                         "(reference README.md:85: '50, 500, and 5000 patches')",
                                    "rows": long_sequence_block(torch, make_model, a.precision, device)}
        except Exception as e:                           # an optional block never costs the line
            out["long_sequence"] = {"error": f"{e!r}"[:300]}
    # N > 1: scaling against the committed N = 1 line of this tree's bench.py (the driver computes its own efficiency from its own runs)
    if world > 1 and rank == 0:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")), reverse=True):
            try:
                n1 = json.load(open(path))
                if n1.get("n_gpus") == 1 and n1.get("config", {}).get("numerics") == a.precision and n1.get("config", {}).get("patches") == N:
                    out["scaling_efficiency_vs_committed_n1"] = {"value": pairs_per_s / (world * n1["value"]), "n1_value": n1["value"],
                                                                 "n1_source": "profiles/" + os.path.basename(path)}
                    break
            except Exception:
                continue
    # (c) nothing rank-0-only between collectives: the process group is closed BEFORE rank 0's host-side work (oracle parity check, CPU
    # baseline, fidelity, PMC passes), so the other ranks never spin in an RCCL barrier while rank 0 computes on the host
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # parity of both modes on the first pairs of rank 0's shard, against the oracle on the host (not timed)
        from oracle import vtamiq_oracle as O
        nchk = 2
        cpu_in = ((inputs[0][0][:nchk].cpu(), inputs[0][1][:nchk].cpu()), (inputs[1][0][:nchk].cpu(), inputs[1][1][:nchk].cpu()),
                  (None, None))
        torch.set_num_threads(min(effective_cores(), 64))
        q_ref = O.vtamiq_forward(O.to_torch(sd_np), spec, *cpu_in)[0].numpy()
        rms = float(np.sqrt(np.mean(q_ref ** 2)))

        def perr(qq):
            d = np.abs(qq[:nchk].cpu().numpy() - q_ref)
            return {"max_rel": float(np.max(d / np.abs(q_ref))), "max_rel_rms": float(d.max() / rms), "max_abs": float(d.max())}
        out["parity_vs_cpu_oracle"] = {a.precision: perr(q)}
        for other, q2 in q_other.items():
            out["parity_vs_cpu_oracle"][other] = perr(q2)
        if "fp8" in out["parity_vs_cpu_oracle"]:
            out["parity_vs_cpu_oracle"]["fp8"]["note"] = ("distance of the fp8 MODEL to the fp32 reference, reported only: its parity "
                                                          "statement is tests/test_gpu_fp8.py against oracle/fp8_oracle.py")
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(torch, spec, sd_np)
        if not a.no_fidelity and world == 1 and on_gpu:
            # every mode's scores against the fp32 oracle in the reference's own metric (bounded sample; the full table on flat-init
            # and trained-like weights is profiles/r03_mode_fidelity.txt, tools/mode_fidelity.py)
            out["fidelity"] = mode_fidelity(torch, make_model, spec, sd_np, ["fp16x3", "bf16x3", "fp16x2", "fp16", "bf16"] + (["fp8"] if fp8_ok else []), device,
                                            pairs=a.fidelity_pairs, N=N, threads=min(effective_cores(), 64))
        if "roofline" in out and world == 1 and not a.no_live_traffic:
            # HBM-side bytes of the dominant kernel measured on THIS box now (the timed regions are over; the GPU is idle)
            tb, src = live_fc1_traffic(a.precision, B, S)
            committed = out["roofline"]["traffic"]
            if tb is not None:
                full_layers = spec.num_layers - (1 if (pruned and a.precision != "fp8") else 0)
                out["roofline"]["traffic"] = tb * full_layers / out["roofline"]["launches_per_step"]
                out["roofline"]["traffic_source"] = src
                out["roofline"]["traffic_committed_pass"] = committed
                # HBM-side rate of the kernel (L2-miss side bytes, Infinity-Cache hits included) against the 8 TB/s peak
                out["roofline"]["memory_side_tbps"] = out["roofline"]["traffic"] / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e12
                if "mfma_busy_frac" in LIVE_EXTRA:
                    out["roofline"]["mfma_busy_frac_measured"] = LIVE_EXTRA["mfma_busy_frac"]
                    out["roofline"]["mfma_busy_note"] = ("rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES on the same kernel in this "
                                                         "run: busy SIMD-cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)")
            else:
                out["roofline"]["traffic_live_error"] = src
        print(json.dumps(out), file=real_stdout, flush=True)


if __name__ == "__main__":
    main()
