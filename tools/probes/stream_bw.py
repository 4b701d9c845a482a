"""What do plain streaming kernels reach on the tensors of layer3's conv3 (read t2 49 MB + residual 196 MB, write 196 MB)?
torch's element-wise kernels as the yardstick for the igemm streaming class."""
import torch
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for rows, C in ((95760, 1024), (375000, 512), (95760 * 4, 1024)):
    x = torch.randn((rows, C), device=dev).to(torch.bfloat16)
    y = torch.randn((rows, C), device=dev).to(torch.bfloat16)
    o = torch.empty_like(x)
    mb = rows * C * 2 / 1e6
    t = timeit(lambda: torch.add(x, y, out=o))
    print("add   [%d x %d] bf16: %.4f ms  %.2f TB/s (2 reads + 1 write of %.0f MB)" % (rows, C, t, 3 * mb / t / 1e3, mb))
    t = timeit(lambda: o.copy_(x))
    print("copy  [%d x %d] bf16: %.4f ms  %.2f TB/s" % (rows, C, t, 2 * mb / t / 1e3))
    t = timeit(lambda: torch.relu_(o))
    print("relu_ [%d x %d] bf16: %.4f ms  %.2f TB/s (in place)" % (rows, C, t, 2 * mb / t / 1e3))
    t = timeit(lambda: o.zero_())
    print("zero_ [%d x %d] bf16: %.4f ms  %.2f TB/s (write only)" % (rows, C, t, mb / t / 1e3))
    t = timeit(lambda: x.sum())
    print("sum   [%d x %d] bf16: %.4f ms  %.2f TB/s (read only)" % (rows, C, t, mb / t / 1e3))
