#!/usr/bin/env python3
"""CU-partitioned co-scheduling probe (GPU box): does a GEMM on g of the 32 CUs of every XCD, run TOGETHER with the attention kernel of another
half-batch on the remaining 32 - g, finish sooner than the same two launches one after the other on the whole chip?

Two streams from hipExtStreamCreateWithCUMask (hip_runtime_api.h:2999).  Mask bit i is taken as CU (i // 8) of XCC (i % 8) -- the KFD's
symmetric map for multi-XCC parts -- and CHECKED: vtq_debug_cu_map launches one CU-filling workgroup per owned CU on each stream and reads
back (XCC id, HW_ID); the two streams must own disjoint CU sets of g and 32 - g CUs on each of the 8 XCDs, otherwise the other bit order
(XCC-major) is tried.  The persistent grids are sized for the partition (vtq_debug_cu_partition: the GEMM's tile schedule is rebuilt for 8 g
workgroups, the attention grid for 8 (32 - g)).

Work: the encoder's shapes at HALF of BASELINE configs[1] (16 pairs = 32 sequences of 501 rows: what each of two micro-batches would run
half a layer out of phase, transformer.py:275-285), fp16x3.  Per split g: each kernel alone on its partition, the pair together, and the
serial reference (both on the unmasked stream, whole chip) -- interleaved `--rounds` times, medians.

    python3 tools/cu_partition.py [--splits 28 24 20] [--rounds 3]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes

ap = argparse.ArgumentParser()
ap.add_argument("--splits", type=int, nargs="+", default=[28, 24, 20])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=6, help="launches per timed burst")
ap.add_argument("--pairs", type=int, default=16, help="pairs of the micro-batch (2 sequences each)")
ap.add_argument("--S", type=int, default=501)
ap.add_argument("--fmt", default="fp16x3")
a = ap.parse_args()

lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]


def hchk(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"HIP error {rc} {what}")


def event():
    e = C.c_void_p()
    hchk(hip.hipEventCreate(C.byref(e)))
    return e


def elapsed(e0, e1):
    ms = C.c_float()
    hchk(hip.hipEventElapsedTime(C.byref(ms), e0, e1))
    return ms.value * 1e3           # us


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    hchk(hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words), "hipExtStreamCreateWithCUMask")
    return s


def cu_set(stream, n):
    """{(xcc, se, sh, cu)} of n CU-filling workgroups launched on `stream`"""
    out = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(lib.vtq_debug_cu_map(out.data_ptr(), n, 300, stream))
    hchk(hip.hipStreamSynchronize(stream))
    v = out.cpu().numpy().astype("uint32").reshape(n, 2)
    return {(int(x) & 15, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 15) for x, h in v}


def partition(g, order):
    """bit lists (GEMM partition: g CUs per XCD, the rest) under a bit order"""
    if order == "interleaved":       # bit i -> XCC i % 8, CU slot i // 8
        big = [i for i in range(256) if i // 8 < g]
    else:                            # bit i -> XCC i // 32, CU slot i % 32
        big = [i for i in range(256) if i % 32 < g]
    small = [i for i in range(256) if i not in set(big)]
    return big, small


def per_xcc(cs):
    n = [0] * 8
    for x, *_ in cs:
        n[x] += 1
    return n


null = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"# device CUs: {torch.cuda.get_device_properties(0).multi_processor_count}; work: {a.pairs} pairs = {2 * a.pairs} sequences x {a.S} rows, {a.fmt}")
whole = cu_set(null, 256)
print(f"# unmasked stream: {len(whole)} distinct CUs, per XCC {per_xcc(whole)}")

# ---- work items ----------------------------------------------------------------------------------------------------------------------
H, Mdim = 768, 3072
nseq = 2 * a.pairs
M = (nseq * a.S + 255) // 256 * 256
g_ = torch.Generator(device="cpu").manual_seed(0)
fmt = a.fmt


def make_gemm(N, K, epi):
    A = torch.randn(M, K, generator=g_).cuda()
    W = (torch.randn(N, K, generator=g_) * 0.03).cuda()
    bias, gamma = torch.randn(N, generator=g_).cuda(), torch.randn(N, generator=g_).cuda()
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device="cuda") if epi != 2 else None
    x = torch.randn(M, N, generator=g_).cuda() if epi == 2 else None
    keep = (Ap, Wp, bias, gamma, out, x)

    def call(stream):
        _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), epi, bias.data_ptr(),
                                  gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                  out.data_ptr() if epi != 2 else None, M * N, N, stream))
    call.keep = keep
    call.flops = 2.0 * M * N * K
    return call


def make_attention():
    rows = nseq * a.S + 128
    qkv = (torch.randn(rows, 3 * H, generator=g_) * 1.5).cuda()
    P = to_planes(qkv, fmt, "a")
    out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device="cuda")

    def call(stream):
        _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, a.S, a.S, H, num_code(fmt), stream))
    call.keep = (P, out)
    call.flops = 4.0 * nseq * (H // 64) * a.S * a.S * 64
    return call


def make_layernorm():
    x = torch.randn(M, H, generator=g_).cuda()
    w, b = torch.randn(H, generator=g_).cuda(), torch.randn(H, generator=g_).cuda()
    out = torch.zeros((2, M, H), dtype=elt_dtype(fmt), device="cuda")

    def call(stream):
        _lib.check(lib.vtq_k_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M * H, M, H, 1, 2, stream))
    call.keep = (x, w, b, out)
    call.flops = 0.0
    return call


gemms = {"qkv": make_gemm(3 * H, H, 0), "out_proj": make_gemm(H, H, 2), "fc1": make_gemm(Mdim, H, 1), "fc2": make_gemm(H, Mdim, 2)}
attn = make_attention()
ln = make_layernorm()
torch.cuda.synchronize()


def set_part(gemm_cus, attn_cus):
    _lib.check(lib.vtq_debug_cu_partition(gemm_cus, attn_cus))


def burst(items):
    """items: [(call, stream, gemm_cus_per_xcd, attention_cus)], all started behind one event; -> us per repetition until the LAST stream is done"""
    e0, ends = event(), []
    hchk(hip.hipEventRecord(e0, null))
    for _, s, _, _ in items:
        if s.value != null.value:
            hchk(hip.hipStreamWaitEvent(s, e0, 0))
    for r in range(a.reps):
        for call, s, gc, ac in items:
            set_part(gc, ac)
            call(s)
    for _, s, _, _ in items:
        e = event()
        hchk(hip.hipEventRecord(e, s))
        ends.append(e)
    for e in ends:
        hchk(hip.hipEventSynchronize(e))
    set_part(0, 0)
    return max(elapsed(e0, e) for e in ends) / a.reps


def med(f, n):
    return statistics.median(f() for _ in range(n))


# warm-up (schedules, clocks)
for c in list(gemms.values()) + [attn, ln]:
    for _ in range(3):
        c(null)
torch.cuda.synchronize()

order_ok = None
for g in a.splits:
    for order in (["interleaved", "blocked"] if order_ok is None else [order_ok]):
        big_bits, small_bits = partition(g, order)
        sb, ss = masked_stream(big_bits), masked_stream(small_bits)
        cb, cs = cu_set(sb, 8 * g), cu_set(ss, 8 * (32 - g))
        ok = per_xcc(cb) == [g] * 8 and per_xcc(cs) == [32 - g] * 8 and not (cb & cs)
        print(f"\n## split {g} / {32 - g} CUs per XCD, mask bit order '{order}': GEMM stream owns {len(cb)} CUs per XCC {per_xcc(cb)}, "
              f"small stream {len(cs)} CUs per XCC {per_xcc(cs)}, overlap {len(cb & cs)} -> {'ok' if ok else 'NOT the intended partition'}")
        if ok:
            order_ok = order
            break
        hip.hipStreamDestroy(sb); hip.hipStreamDestroy(ss)
    if not ok:
        print("   no bit order gave the partition; skipping this split")
        continue
    # warm the narrowed schedules
    for c in gemms.values():
        set_part(g, 0); c(sb)
    set_part(0, 8 * (32 - g)); attn(ss)
    hip.hipStreamSynchronize(sb); hip.hipStreamSynchronize(ss); set_part(0, 0)
    rows = {}
    for rnd in range(a.rounds):
        for name, gm in gemms.items():
            r = rows.setdefault(name, {k: [] for k in ("gemm_whole", "attn_whole", "serial", "gemm_part", "attn_part", "pair", "pair_ln")})
            r["gemm_whole"].append(burst([(gm, null, 0, 0)]))
            r["attn_whole"].append(burst([(attn, null, 0, 0)]))
            r["serial"].append(burst([(gm, null, 0, 0), (attn, null, 0, 0)]))
            r["gemm_part"].append(burst([(gm, sb, g, 0)]))
            r["attn_part"].append(burst([(attn, ss, 0, 8 * (32 - g))]))
            r["pair"].append(burst([(gm, sb, g, 0), (attn, ss, 0, 8 * (32 - g))]))
    print(f"   {'GEMM':9s} {'alone,whole':>12s} {'attn,whole':>11s} {'serial':>9s} | {'GEMM on g':>10s} {'(x whole)':>9s} {'attn on rest':>13s} {'(x whole)':>9s} | {'pair':>9s} {'pair/serial':>11s}")
    for name, r in rows.items():
        m = {k: statistics.median(v) for k, v in r.items() if v}
        print(f"   {name:9s} {m['gemm_whole']:10.1f}us {m['attn_whole']:9.1f}us {m['serial']:7.1f}us | {m['gemm_part']:8.1f}us {m['gemm_part'] / m['gemm_whole']:9.2f} "
              f"{m['attn_part']:11.1f}us {m['attn_part'] / m['attn_whole']:9.2f} | {m['pair']:7.1f}us {m['pair'] / m['serial']:11.3f}")
    # a layer's worth on the GEMM partition against ONE attention + the LayerNorms on the small one: out_proj + fc1 + fc2 + qkv || attention + 2 LN
    layer_serial, layer_pair = [], []
    for rnd in range(a.rounds):
        seq = [(gemms[n], null, 0, 0) for n in ("out_proj", "fc1", "fc2", "qkv")] + [(ln, null, 0, 0), (ln, null, 0, 0), (attn, null, 0, 0)]
        layer_serial.append(burst(seq))
        par = [(gemms[n], sb, g, 0) for n in ("out_proj", "fc1", "fc2", "qkv")] + [(ln, ss, 0, 0), (ln, ss, 0, 0), (attn, ss, 0, 8 * (32 - g))]
        layer_pair.append(burst(par))
    ls, lp = statistics.median(layer_serial), statistics.median(layer_pair)
    print(f"   one layer of one micro-batch (4 GEMMs | attention + 2 LayerNorms): serial on the whole chip {ls:.1f} us, partitioned {lp:.1f} us -> {lp / ls:.3f}")
    hip.hipStreamDestroy(sb); hip.hipStreamDestroy(ss)
