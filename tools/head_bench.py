#!/usr/bin/env python3
"""DiffNet head + predictor (39 dependent skinny-MFMA launches for the default 4 x 4 topology): eager launches against ONE
hipGraph replay of the same chain (GPU box).  The graph is captured here with torch.cuda.CUDAGraph around
vtq_k_diffnet_head -- the same launches the engine makes, on the capturing stream.  If the chain were bound by the host's launch
rate a replay would shorten it; it is bound on the device (each stage: one exposed weight-streaming latency + the dependent-launch
boundary), so it does not."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, _lib, synth

lib = _lib.load()
dev = torch.device("cuda")
m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, pretrained=False), precision="fp16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 0).items()})
m = m.to(dev).eval()
m._ensure_engine(dev)
H = m.spec.hidden_size
for HB in (32, 64):
    d = torch.randn(HB, H, device=dev)
    q = torch.empty(HB, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: _lib.check(lib.vtq_k_diffnet_head(m._engine, d.data_ptr(), HB, q.data_ptr(), st()))
    call(); torch.cuda.synchronize()
    q_eager = q.clone()

    def timeit(fn, n=200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_eager = timeit(call)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call()
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(q, q_eager)
    t_graph = timeit(g.replay)
    print(f"DiffNet head + predictor, {HB} rows: eager {t_eager:.1f} us per call (back-to-back calls: the host stays ahead), "
          f"one hipGraph replay {t_graph:.1f} us; scores bit-identical: {same}", flush=True)
