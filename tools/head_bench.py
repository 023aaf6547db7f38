#!/usr/bin/env python3
"""Micro-benchmark of the head's small_linear kernel: warm (same weights) vs cold (rotating weight buffers + cache flush)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import stream
lib = _lib.load()
B, N, K = 32, 768, 768
x = torch.randn(B, K, device="cuda"); bias = torch.randn(N, device="cuda"); y = torch.zeros(B, N, device="cuda")
Ws = [torch.randn(N, K, device="cuda") * 0.05 for _ in range(24)]
flush = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device="cuda")
def call(W): _lib.check(lib.vtq_k_small_linear(x.data_ptr(), W.data_ptr(), bias.data_ptr(), None, None, None, y.data_ptr(), B, N, K, stream()))
def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(3): call(Ws[0])
print("warm  same W   : %.2f us/launch" % timeit(lambda i: call(Ws[0]), 200))
print("rotating 24 W  : %.2f us/launch" % timeit(lambda i: call(Ws[i % 24]), 240))
ts = []
for r in range(5):
    flush.zero_(); torch.cuda.synchronize()
    ts.append(timeit(lambda i: call(Ws[i]), 24))
print("cold (after 512MB flush), 24 different W: %.2f us/launch" % (sorted(ts)[2]))
