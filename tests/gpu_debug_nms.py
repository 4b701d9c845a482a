"""Ad-hoc GPU diagnostic for the NMS kernels (not a pytest file)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mega.pytorch_amd import _lib
from oracle import native

lib = _lib.load()
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for n, thr, ties in [(300, 0.5, False), (300, 0.5, True), (200, 0.5, False), (130, 0.3, False)]:
    ctr = rng.rand(n, 2) * 500
    wh = rng.rand(n, 2) * 120 + 4
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    scores = rng.rand(n).astype(np.float32)
    if ties:
        scores[::7] = scores[3]
    order = np.lexsort((np.arange(n), -scores))
    sb = boxes[order]
    want_pos = native.nms(sb, -np.arange(n, dtype=np.float32), thr, True)
    b = torch.from_numpy(sb).to(dev)
    counts = torch.tensor([n], dtype=torch.int32, device=dev)
    keep_pos = torch.full((n,), -1, dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    nb = lib.mega_nms_workspace_bytes(1, n)
    ws = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    rc = lib.mega_nms_sorted(b.data_ptr(), counts.data_ptr(), None, None, 1, n, thr, 1, n, keep_pos.data_ptr(),
                             keep_cnt.data_ptr(), None, ws.data_ptr(), nb, None)
    torch.cuda.synchronize()
    cnt = int(keep_cnt.item())
    got = keep_pos[:cnt].cpu().numpy()
    cb = (n + 63) // 64
    mask = ws[: n * cb * 8].cpu().numpy().view(np.uint64).reshape(n, cb)
    # CPU mask
    x1, y1, x2, y2 = sb.T
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    iw = np.maximum(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]) + 1, 0)
    ih = np.maximum(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]) + 1, 0)
    inter = (iw * ih).astype(np.float32)
    iou = inter / (area[:, None] + area[None] - inter)
    sup = iou > thr
    bad = 0
    for i in range(n):
        for c in range(i // 64, cb):
            wantw = 0
            for j in range(c * 64, min(n, c * 64 + 64)):
                if j > i and sup[i, j]:
                    wantw |= 1 << (j - c * 64)
            if int(mask[i, c]) != wantw:
                bad += 1
                if bad < 4:
                    print("  mask mismatch row", i, "cb", c, hex(int(mask[i, c])), hex(wantw))
    print("n=%d thr=%g ties=%s rc=%d: gpu kept %d, oracle %d, equal=%s, mask mismatches=%d" % (
        n, thr, ties, rc, cnt, len(want_pos), np.array_equal(got, want_pos), bad))
    if not np.array_equal(got, want_pos):
        m = min(len(got), len(want_pos))
        d = np.nonzero(got[:m] != want_pos[:m])[0]
        print("  first diff at", d[:1], got[max(0, d[0] - 3): d[0] + 3] if len(d) else None,
              want_pos[max(0, d[0] - 3): d[0] + 3] if len(d) else None)
