#!/bin/bash
# fp8 mode first light: kernel tests, e2e vs fake-quant oracle, GEMM micro-bench, end-to-end timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_fp8.py -x -q -s -m gpu > $O/fp8_tests.txt 2>&1; tail -40 $O/fp8_tests.txt
true
timeout 600 python3 - > $O/e2e.txt 2>&1 <<'PY'
import json, time, torch, sys
sys.path.insert(0, '.')
from vtamiq_amd import VTAMIQ, synth
for prec in ("fp8", "fp16", "fp16x3"):
    m = VTAMIQ(vit_config=dict(variant="ViT-B16"), precision=prec, pretrained=False)
    sd = synth.make_state_dict(m.spec, 0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    for B in (32, 64):
        patches, pos, scales = synth.make_inputs(m.spec, B, 500, 7)
        tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
        args = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))
        with torch.no_grad():
            for _ in range(3): q = m(*args)[0]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): q = m(*args)[0]
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{prec} B={B}: {dt*1e3:.2f} ms/step {B/dt:.1f} pairs/s", flush=True)
        if B == 32:
            print(json.dumps(m.profile_classes(*args)) if hasattr(m, "profile_classes") else "", flush=True)
PY
cat $O/e2e.txt
