#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of the VTAMIQ ViT-B/16 pair forward (P=500 patches) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--patches P] [--precision bf16x3|bf16]

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); every rank scores its own shard of the global batch
(weak scaling: per-GPU batch fixed) and one RCCL all-gather of the scores closes each step.  Inputs are synthetic and
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6        # dense bf16 MFMA: 256 CU x 4096 flop/clk/CU x 2.4 GHz (MI355X_MICROARCH.md)


def synth_inputs_on_device(torch, spec, B, N, device, seed):
    """ref ~ U(-1,1); dist = clamp(ref + 0.1 N(0,1)); pos ~ U(0,1) <= 1-1e-6 shared (aligned)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ref = torch.rand(B, N, 3, 16, 16, device=device, generator=g) * 2 - 1
    dist = (ref + 0.1 * torch.randn(ref.shape, device=device, generator=g)).clamp_(-1, 1)
    pos = torch.rand(B, N, 2, device=device, generator=g).clamp_(max=1 - 1e-6)
    return (ref, dist), (pos, pos.clone()), (None, None)


def fc1_traffic(precision, B):
    """HBM bytes of ONE full-batch fc1 GEMM (M = 2*B*S_pad rows) from the committed PMC passes
    (profiles/r01_gemm_fc1_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, collected at B=32 in separate
    rocprofv3 --pmc runs); scaled linearly with the batch."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_gemm_fc1_traffic.json")))
        return d[precision]["bytes_per_launch"] * (B / 32.0)
    except Exception:
        return None


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


def cpu_baseline(torch, spec, sd_np, N, seconds_budget=25.0):
    """The oracle (CPU port of the reference path, fp32, torch CPU ops) timed on this host's cores on a bounded sample:
    one pair (two 500-patch images through all layers) per forward; thread count = usable cores capped at 64 (torch's
    intra-op pool stops scaling and oversubscribes far earlier on this shape)."""
    from oracle import vtamiq_oracle as O
    from vtamiq_amd import synth
    ncores = min(effective_cores(), 64)
    torch.set_num_threads(ncores)
    sd = O.to_torch(sd_np)
    Bc = 1
    patches, pos, _ = synth.make_inputs(spec, Bc, N, 4242)
    tp, tq = torch.from_numpy(patches), torch.from_numpy(pos)
    args = ((tp[:, 0], tp[:, 1]), (tq[:, 0], tq[:, 1]), (None, None))
    t0 = time.perf_counter()
    O.vtamiq_forward(sd, spec, *args)                       # warm-up (also bounds the sample)
    warm = time.perf_counter() - t0
    times = []
    t_start = time.perf_counter()
    while len(times) < 7 and (time.perf_counter() - t_start + warm) < seconds_budget:
        t0 = time.perf_counter()
        O.vtamiq_forward(sd, spec, *args)
        times.append(time.perf_counter() - t0)
    if not times:
        times = [warm]
    med = sorted(times)[len(times) // 2]
    return {"value": Bc / med, "unit": "image-pairs/s", "cores": ncores, "kind": "port",
            "sample": f"oracle fp32 (torch CPU ops), {Bc} pair x {N} patches ViT-B/16 L={spec.num_layers}, "
                      f"1 warm-up + median of {len(times)} forwards, {ncores} threads",
            "ms_per_forward": med * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU (BASELINE config 2: 32)")
    ap.add_argument("--patches", type=int, default=500)
    ap.add_argument("--precision", default=os.environ.get("VTAMIQ_BENCH_PRECISION", "bf16x3"), choices=["bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-mode", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from vtamiq_amd import VTAMIQ, synth
    from vtamiq_amd.dist import gather_scores

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    B, N = a.batch, a.patches
    kw = dict(vit_config=dict(variant="ViT-B16"))                 # L=12, T=1, r=8: BASELINE configs 2/3 and north star
    spec_model = VTAMIQ(**json.loads(json.dumps(kw)), precision=a.precision)
    spec = spec_model.spec
    sd_np = synth.make_state_dict(spec, 0)
    state = {k: torch.from_numpy(v) for k, v in sd_np.items()}

    def make_model(precision):
        m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
        m.load_state_dict(state)
        return m.to(device).eval()

    inputs = synth_inputs_on_device(torch, spec, B, N, device, 1234 + rank)
    global_batch = B * world

    def run(model, steps, warmup, profile_class=None):
        with torch.no_grad():
            for _ in range(warmup):
                q = gather_scores(model(*inputs)[0], global_batch)
            torch.cuda.synchronize()
            if profile_class:
                model.profile_enable([profile_class])
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                q = gather_scores(model(*inputs)[0], global_batch)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            prof = None
            if profile_class:
                prof = model.profile_collect()[profile_class]
                model.profile_enable([])
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), q, prof

    model = make_model(a.precision)
    DOM = "fc1"
    dt, q, prof = run(model, a.steps, a.warmup, DOM)
    assert q.shape == (global_batch,) and bool(torch.isfinite(q).all())
    pairs_per_s = global_batch * a.steps / dt
    f_pair = spec.flops_per_pair(N)                      # algorithmic (SURVEY 8d / BASELINE.md)
    pruned = os.environ.get("VTQ_NO_CLS_PRUNE", "0") != "1"
    f_exec = spec.flops_per_pair_executed(N, cls_prune=pruned)
    S = spec.seq_len(N)

    out = {
        "metric": "image-pairs/sec ViT-B/16 P=500 patches, 1->8 MI355X; % bf16 MFMA roofline",
        "value": pairs_per_s, "unit": "image-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: ViT-B/16 (L=12, T=1) FR pair forward, batch={B} pairs/GPU, "
                               f"{N} patches of 16x16x3, random-init seeded weights",
                   "global_batch": global_batch, "patches": N, "seq_len": S, "parallelism": f"dp{world}",
                   "numerics": a.precision,
                   "numerics_note": "bf16x3 = hi/lo bf16 operand split, 3 bf16 MFMAs per product, fp32 accumulate (meets 1e-3 "
                                    "parity); bf16 = 1 MFMA per product (throughput mode, parity ~3e-2)"},
        # executed flops: the last layer runs Q/attention/out-proj/MLP for the CLS row only (legal: only token 0 is consumed)
        "forward_mfma_frac": pairs_per_s / world * f_exec / (PEAK_BF16_TFLOPS * 1e12),
        "flops_per_pair": f_pair, "flops_per_pair_executed": f_exec,
    }

    def roofline_of(prof_entry, steps, note):
        ms_sum, launches = prof_entry
        full_layers = spec.num_layers - (1 if pruned else 0)       # fc1 GEMM launches per step = full layers x part-batches
        per_step = launches / steps
        flops_launch = 2.0 * (2 * B * S) * spec.hidden_size * spec.mlp_dim * full_layers / per_step    # algorithmic, unpadded rows
        ach = flops_launch / (ms_sum / launches * 1e-3) / 1e12
        traffic = fc1_traffic(a.precision, B)
        return {"bound": "mfma", "kernel": f"gemm_pp2_kernel<{3 if a.precision == 'bf16x3' else 1}, GELU> (fc1 of every full layer)",
                "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                "traffic": traffic * full_layers / per_step if traffic else None, "avg_launch_ms": ms_sum / launches,
                "launches": int(launches), "launches_per_step": per_step, "flops_per_launch": flops_launch,
                # bf16x3 issues 3 bf16 MFMAs per algorithmic product, so its algorithmic frac is capped at 1/3
                "mfma_issue_frac": ach * (3 if a.precision == "bf16x3" else 1) / PEAK_BF16_TFLOPS, "note": note}

    if prof and prof[1] > 0:
        in_region = roofline_of(prof, a.steps, "HIP events inside the timed region of `value`: two part-batches run on two streams, so a "
                                "launch covers half the rows and shares the chip with the other stream's kernels")
        # the same kernel alone on the chip: a second timed region with ONE part-batch (full-batch launches on one stream)
        os.environ["VTQ_PARTS"] = "1"
        model_iso = make_model(a.precision)
        iso_steps = max(3, a.steps // 2)
        _, _, prof_iso = run(model_iso, iso_steps, 2, DOM)      # the engine is created lazily on the first forward
        os.environ.pop("VTQ_PARTS")
        out["roofline"] = roofline_of(prof_iso, iso_steps, "HIP events on the launch stream over a timed region with one part-batch "
                                      "(VTQ_PARTS=1): the kernel alone on the chip, full-batch launches")
        out["roofline"]["in_value_region"] = {k: in_region[k] for k in ("achieved", "frac", "avg_launch_ms", "launches_per_step", "note")}
        del model_iso
        torch.cuda.empty_cache()
    if not a.no_second_mode:                     # every rank: run() is collective
        other = "bf16" if a.precision == "bf16x3" else "bf16x3"
        del model
        torch.cuda.empty_cache()
        model2 = make_model(other)
        dt2, q2, _ = run(model2, a.steps, a.warmup)
        out["other_mode"] = {"numerics": other, "value": global_batch * a.steps / dt2, "unit": "image-pairs/s",
                             "forward_mfma_frac": global_batch * a.steps / dt2 / world * f_exec / (PEAK_BF16_TFLOPS * 1e12)}
    else:
        q2 = None
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
        if q2 is not None:
            out["parity_vs_cpu_oracle"][out["other_mode"]["numerics"]] = perr(q2)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(torch, spec, sd_np, N)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
