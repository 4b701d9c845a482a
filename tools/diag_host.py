"""Diagnostics: host cost per launch (ctypes wrappers and torch glue ops) with an otherwise idle GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(64, 1024, device=dev).bfloat16(); w = torch.randn(1024, 1024, device=dev).bfloat16(); b = torch.zeros(1024, device=dev)
def bench(name, fn, n=2000):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    tt = time.perf_counter() - t
    print("%-28s host %.1f us/call, host+drain %.1f us/call" % (name, th / n * 1e6, tt / n * 1e6))
bench("ops.linear 64x1024x1024", lambda: ops.linear(x, w, b))
a = [torch.randn(75, 1024, device=dev).bfloat16() for _ in range(25)]
bench("torch.cat 25x[75,1024]", lambda: torch.cat(a, 0))
bench("torch.empty", lambda: torch.empty((300, 1024), device=dev, dtype=torch.bfloat16))
idx = torch.arange(300, device=dev)
big = torch.randn(2000, 1024, device=dev).bfloat16()
bench("index_select", lambda: big.index_select(0, idx))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): ops.linear(x, w, b)
    with torch.cuda.stream(s2): ops.linear(x, w, b)
bench("2 streams x ops.linear", two, 1000)
print("threads", torch.get_num_threads(), "env", {k: v for k, v in os.environ.items() if k.startswith(("HIP", "AMD", "HSA", "ROC", "GPU", "OMP", "MEGA"))})
