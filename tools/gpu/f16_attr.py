import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_e2e_gpu import _r101_mixed_run, _fmt
dev = torch.device("cuda:0")
for fd, hd in (("float16", "float32"), ("float16", "float16"), ("float32", "bfloat16")):
    r = _r101_mixed_run(dev, fd, hd)
    for idx, m in sorted(r.items()):
        print(_fmt("%s/%s" % (fd, hd), idx, m), flush=True)
