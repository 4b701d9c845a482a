"""Per-kernel busy time of the LAST step-batch in a tools/gpu/trace.sh timeline (no-overlap run: the tail of the stream is
F(last batch), B(previous batch), B(last batch); a step-batch there = 20 key frames = 40 frames).
usage: python tools/trace_summary.py gpurun_out/<tag>/tail.csv"""
import csv
import re
import sys
from collections import defaultdict


def name(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')
    m = re.match(r'[\w:]+', n)
    base = m.group(0) if m else n[:40]
    return n[:50] if 'igemm' in base else base


def main(path):
    rows = [(int(a), int(b), n) for a, b, q, s, n in csv.reader(open(path))]
    pre = [i for i, r in enumerate(rows) if 'preprocess' in r[2] or 'stem_mfma' in r[2] or 'stem_pool' in r[2]]    # first kernel of a frame stage
    seg = rows[pre[-1]:]
    roi = [i for i, r in enumerate(seg) if 'roi_align' in r[2]][-1]
    # the frame stage ends with the first FC (one igemm launch after ROIAlign, + its split-K finalize)
    end = roi + 2
    if end < len(seg) and 'splitk_finalize' in seg[end][2]:
        end += 1
    f, b = seg[:end], seg[end:]
    for title, part, div in (("frame stage of one step-batch", f, 1), ("aggregation, per step-batch (two batches traced, halved)", b, 2)):
        agg = defaultdict(lambda: [0, 0])
        for r in part:
            k = name(r[2])
            agg[k][0] += 1
            agg[k][1] += r[1] - r[0]
        tot = sum(v[1] for v in agg.values())
        print("== %s: %d launches, busy %.2f ms, span %.2f ms" % (title, len(part) // div, tot / 1e6 / div,
                                                                    (part[-1][1] - part[0][0]) / 1e6 / div))
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            print("  %-52s n=%5.1f busy %8.1f us  avg %6.1f" % (n, c / div, t / 1e3 / div, t / 1e3 / c))


if __name__ == "__main__":
    main(sys.argv[1])
