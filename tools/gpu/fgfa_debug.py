import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R, NC = 300, 31
def make(seed, n):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn((R, NC), generator=g) * 0.3).to(dev)
    deltas = (torch.randn((R, NC * 4), generator=g) * 0.5).to(dev)
    ctr = torch.rand((R, 2), generator=g) * torch.tensor([192., 128.]); wh = torch.rand((R, 2), generator=g) * 60 + 8
    props = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(min=0).to(dev)
    props[n:] = 0
    return logits, deltas, props, torch.tensor([n], dtype=torch.int32, device=dev)
W = (10., 10., 5., 5.)
def run(lg, dl, pr, cn):
    return ops.postprocess(lg, dl, pr, cn, W, 192, 128, 0.001, 0.5, 300, True)
cases = [make(1, 230), make(2, 300), make(3, 150), make(4, 229), make(1, 230)]
eager = []
for c in cases:
    o = run(*c); torch.cuda.synchronize(); n = int(o[3]); eager.append((n, o[0][:n].clone(), o[1][:n].clone(), o[2][:n].clone()))
# static inputs + graph
s_in = [t.clone() for t in cases[0]]
run(*s_in); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    out = run(*s_in)
for rep in range(2):
    for i, c in enumerate(cases):
        for d_, s_ in zip(s_in, c):
            d_.copy_(s_)
        gr.replay(); torch.cuda.synchronize()
        n = int(out[3])
        ok = n == eager[i][0] and torch.equal(out[0][:n], eager[i][1]) and torch.equal(out[1][:n], eager[i][2]) and torch.equal(out[2][:n], eager[i][3])
        print("pass %d case %d (nprop %d): replay count %d eager %d -> %s" % (rep, i, int(c[3]), n, eager[i][0], "equal" if ok else "DIFFERENT"))
