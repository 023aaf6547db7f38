#!/usr/bin/env python3
"""Where the engine's error on trained-like weights comes from: the 64-pair N = 500 ladder on stress_state(qk = 5)
weights (tests/golden/stress5_b64_n500.npz: the REFERENCE's fp32 and float64 scores).

 1. signed relative error of every fp16x3 score against the reference's float64 score: noise or bias?
 2. for the worst pairs: per-layer CLS rows of the engine (vtq_set_token_trace) against the float64 oracle on the host, beside the CPU
    emulation of the same operand scheme (tests/numerics_study.py "h3": fp16 hi/lo operands, fp32 accumulate, everything else fp32);
 3. encoder vs head: the float64 head on the engine's CLS rows, and the engine's head on the float64 CLS difference.
GPU box only (the oracle is the checker, as in the tests)."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vtamiq_oracle as O
from tests import numerics_study as NS
from tests.helpers import load_ladder_case, split_inputs
from vtamiq_amd import VTAMIQ, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="stress5_b64_n500"); ap.add_argument("--worst", type=int, default=2); ap.add_argument("--precision", default="fp16x3")
ap.add_argument("--options", type=int, default=0)
a = ap.parse_args()
torch.set_num_threads(min(os.cpu_count() or 8, 32))
g, kw, spec, sd, (patches, pos, _) = load_ladder_case(a.case)
dev = "cuda"
m = VTAMIQ(**json.loads(json.dumps(kw)), precision=a.precision, engine_options=a.options)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).eval()
p, ps, sc = split_inputs(patches, pos, None, device=dev)
qs = []
with torch.no_grad():
    for i in range(0, patches.shape[0], 32):
        qs.append(m((p[0][i:i + 32], p[1][i:i + 32]), (ps[0][i:i + 32], ps[1][i:i + 32]), (None, None))[0])
q = torch.cat(qs).double().cpu().numpy()
q64, q32 = g["q64"], g["q"].astype(np.float64)
rms = np.sqrt(np.mean(q64 ** 2))
rel = (q - q64) / q64
big = np.abs(q64) >= 0.1 * rms
print(f"# {a.case}, {a.precision}: {len(q)} scores, rms {rms:.4f}")
print(f"1. signed (q - q64) / q64 over the {int(big.sum())} scores with |q64| >= 0.1 rms: mean {rel[big].mean():+.2e}, std {rel[big].std():.2e}, min {rel[big].min():+.2e}, max {rel[big].max():+.2e}")
print(f"   absolute error (q - q64): mean {np.mean(q - q64):+.3e}, std {np.std(q - q64):.3e}   [a constant OFFSET shows here, a constant FACTOR in the line above]")
print(f"   reference fp32 vs its own fp64, same measure: mean {((q32 - q64) / q64)[big].mean():+.2e}, std {((q32 - q64) / q64)[big].std():.2e}; absolute mean {np.mean(q32 - q64):+.3e}, std {np.std(q32 - q64):.3e}")
lin = np.polyfit(q64, q - q64, 1)
print(f"   least squares (q - q64) = a q64 + b: a = {lin[0]:+.3e}, b = {lin[1]:+.3e} (rms of q64 {rms:.3f}); residual std {np.std(q - q64 - np.polyval(lin, q64)):.3e}")
order = np.argsort(-np.where(big, np.abs(rel), 0))
sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
sd32 = O.to_torch(sd)
lib = _lib.load()
L, T, H = spec.num_layers, spec.num_tokens, spec.hidden_size
for idx in order[: a.worst]:
    sl = slice(idx, idx + 1)
    pin = ((p[0][sl], p[1][sl]), (ps[0][sl], ps[1][sl]), (None, None))
    trace = torch.zeros(L + 1, 2, T, H, device=dev)
    with torch.no_grad():
        qh = m(*pin, _trace=trace)[0].double().cpu().numpy()[0]          # full last layer (the trace needs every row)
    tr = trace.double().cpu()
    cin = ((pin[0][0].cpu(), pin[0][1].cpu()), (pin[1][0].cpu(), pin[1][1].cpu()), (None, None))
    c64 = tuple(tuple(None if t is None else t.double() for t in grp) for grp in cin)
    t64 = {}
    q_or = O.vtamiq_forward(sd64, spec, *c64, trace=t64)[0].numpy()[0]
    ref_tr = torch.cat([t64["tokens_ref"], t64["tokens_dist"]], dim=1)     # (L + 1, 2, T, H)
    # the h3 emulation, with its own per-layer trace
    print(f"\n2. pair {idx}: q64 {q64[idx]:+.6f} (oracle fp64 here {q_or:+.6f}), engine {q[idx]:+.6f} (rel {rel[idx]:+.2e}); with the trace tap (full last layer) {qh:+.6f}")
    print("   layer | CLS row error / max|row| (ref, dist) | error of the DIFFERENCE cls_ref - cls_dist relative to its norm | norm of the difference / norm of a row")
    for l in range(L + 1):
        e = (tr[l] - ref_tr[l])[:, 0]
        r = ref_tr[l][:, 0]
        dd = (tr[l][0, 0] - tr[l][1, 0]) - (r[0] - r[1])
        print(f"   {l:5d} | {e[0].abs().max() / r[0].abs().max():.2e} {e[1].abs().max() / r[1].abs().max():.2e} | {dd.norm() / (r[0] - r[1]).norm():.2e} | {(r[0] - r[1]).norm() / r[0].norm():.2e}")
    # 3. encoder vs head
    def final(tok):                                             # float64 encoder_norm on token rows (2, T, H)
        return O._layer_norm(tok, sd64["transformer.encoder.encoder_norm.weight"], sd64["transformer.encoder.encoder_norm.bias"])
    f_eng, f_ref = final(tr[L]), final(ref_tr[L])
    q_head64_on_engine_rows = O.head(sd64, spec, f_eng[0:1], f_eng[1:2]).numpy()[0]
    d64 = (f_ref[0, 0] - f_ref[1, 0]) * (sd64["diff_scale.gamma"] if spec.diff_scale else 1.0)
    d_dev = d64.float().to(dev).view(1, H).contiguous()
    qo = torch.zeros(1, device=dev)
    with torch.cuda.device(0):
        _lib.check(lib.vtq_k_diffnet_head(m._engine, d_dev.data_ptr(), 1, qo.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    q_engine_head_on_ref = float(qo[0])
    print(f"3. float64 head on the ENGINE's last-layer CLS rows: {q_head64_on_engine_rows:+.6f} (rel to q64 {(q_head64_on_engine_rows - q_or) / q_or:+.2e})  = the encoder's share")
    print(f"   ENGINE head on the float64 CLS difference:        {q_engine_head_on_ref:+.6f} (rel to q64 {(q_engine_head_on_ref - q_or) / q_or:+.2e})  = the head's share")
    qe = NS.forward(sd32, spec, cin, NS.scheme("h3"))[0]
    q_fp32 = O.vtamiq_forward(sd32, spec, *cin)[0].numpy()[0]
    print(f"   CPU emulation of fp16x3 operands (fp32 everything else, fp32 head): {qe:+.6f} (rel {(qe - q_or) / q_or:+.2e}); the fp32 oracle itself: {q_fp32:+.6f} (rel {(q_fp32 - q_or) / q_or:+.2e})")
