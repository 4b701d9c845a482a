import sys, tempfile, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from PIL import Image
from mega.pytorch_amd import feed, ops, synth
dev = torch.device("cuda:0")
T, H0, W0 = 6, 180, 320
clip0 = synth.make_clip(T, H0, W0, seed=3).numpy()
d = tempfile.mkdtemp()
for t in range(T):
    Image.fromarray(clip0[t]).save(os.path.join(d, "%06d.png" % t))
src = feed.FrameSource(os.path.join(d, "%s.png"), "%06d", T, dev, workers=4)
ids = [4, 0, 5, 4]
u8 = src.fetch(ids)
torch.cuda.synchronize()
ref = np.stack([np.asarray(Image.fromarray(clip0[i]).resize((999, 562), Image.BILINEAR)) for i in ids])
g = u8.cpu().numpy()
print("fetch u8 mismatches:", (g != ref).sum(), "of", ref.size)
if (g != ref).any():
    w = np.argwhere(g != ref); print(w[:10], g[tuple(w[0])], ref[tuple(w[0])])
# direct kernel on the same frames
tb = feed.ResizeTables((H0, W0), (562, 999), dev)
g2 = ops.resize_bilinear_u8(torch.from_numpy(clip0[ids]).to(dev), (562, 999), tb).cpu().numpy()
print("direct kernel mismatches:", (g2 != ref).sum())
got = ops.preprocess_frames(torch.from_numpy(ref).to(dev), synth.PIXEL_MEAN, True).cpu()
want = synth.preprocess_cpu(torch.from_numpy(ref))
print("preprocess mismatches:", (got != want).sum().item(), (got - want).abs().max().item())
