"""Stub model for the bench.py launcher self-test (tests/test_bench_launcher.py): same call signature as VTAMIQ, a closed
form score per pair, plain torch on whatever device the inputs live on.  Test infrastructure only: bench.py marks such a
line `"data": "stub"` and it is never a measurement."""
import torch


class StubModel:
    def __call__(self, patches, pos, scales):
        ref, dist = patches
        q = (ref - dist).flatten(1).abs().mean(dim=1) + pos[0].flatten(1).mean(dim=1)
        return q.float(), None
