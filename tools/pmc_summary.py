"""Summarise rocprofv3 PMC passes (one counter per pass, as the MI355X guide prescribes) per kernel.

  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                              gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv  profiles/r01_pmc

writes <prefix>_per_kernel.csv and <prefix>_summary.json.  HBM bytes per launch of a kernel =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are reported in KiB and, on gfx950 with this
rocprofv3, FETCH_SIZE counts 64 B per 128-B read request (MI355X_MICROARCH.md, HBM section) -> doubled.
Infinity-Cache hits are included in these fabric-side counters, so this is an upper bound on true HBM traffic.
"""
import collections
import csv
import json
import sys


_DEMANGLED = {}


def _demangle(name):
    """c++filt on one mangled symbol, cached.  The binutils of this image predate the `DF16_` (_Float16) builtin-type code, so it
    is handed the symbol with `Dh` (IEEE half: also a builtin type, hence no substitution index moves) and `half` is renamed back."""
    if name not in _DEMANGLED:
        import subprocess
        out = name
        try:
            r = subprocess.run(["c++filt", name.replace("DF16_", "Dh")], stdout=subprocess.PIPE, universal_newlines=True, timeout=10)
            got = r.stdout.strip()
            if got and not got.startswith("_Z"):
                import re
                out = re.sub(r"\bhalf\b", "_Float16", got)
        except Exception:  # noqa: BLE001
            pass
        _DEMANGLED[name] = out
    return _DEMANGLED[name]


def short(k):
    """rocprofv3 kernel name -> short family name.  The anonymous-namespace prefix is stripped BEFORE the argument
    list is cut off (every kernel of this library lives in `(anonymous namespace)`: cutting at the first "(" used to
    collapse all of them into one empty name).  The split-K first FC of the box head is its own template
    instantiation (igemm.hip), hence its own row."""
    if k.startswith("_Z"):      # rocprofv3 leaves names with _Float16 arguments mangled (round 6: the fp16 instantiations)
        k = _demangle(k)
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")
    k = k.replace("__hip_bfloat16", "bf16")
    name = k.split("(")[0].strip()
    return name[-96:]


def agg(path, name):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        e = d[short(r["Kernel_Name"])]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
        e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def counter_table(argv):
    """python tools/pmc_summary.py --counter NAME pmc_counter_collection.csv out.csv : mean of one counter per kernel
    (e.g. SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE collected in their own passes; MFMA-busy % = ratio of the two)."""
    name, path, out = argv
    d = agg(path, name)
    rows = sorted(((k, n, v / n, t / n / 1e3) for k, (n, v, t) in d.items()), key=lambda r: -r[1] * r[3])
    with open(out, "w") as fh:
        fh.write("kernel,launches,%s_per_launch,avg_us\n" % name)
        for k, n, v, t in rows:
            fh.write('"%s",%d,%.1f,%.2f\n' % (k, n, v, t))
    for k, n, v, t in rows[:8]:
        print("%-64s x%-5d %s %14.0f  %8.1f us" % (k[:64], n, name, v, t))


def mfma_busy(argv):
    """python tools/pmc_summary.py --mfma-busy <SQ_VALU_MFMA_BUSY_CYCLES csv> <GRBM_GUI_ACTIVE csv> out.csv :
    busy fraction per kernel = busy SIMD-cycles / (1024 SIMDs x GUI_ACTIVE cycles / 8 XCDs)  (GRBM_GUI_ACTIVE is summed
    over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 32 per v_mfma_f32_32x32x16_bf16 on its SIMD)."""
    busy, gui, out = argv
    b = agg(busy, "SQ_VALU_MFMA_BUSY_CYCLES")
    g = agg(gui, "GRBM_GUI_ACTIVE")
    rows = []
    for k, (n, v, t) in b.items():
        if k in g and g[k][1] > 0:
            gn, gv, gt = g[k]
            rows.append((k, n, v / n, gv / gn, (v / n) / (1024.0 * (gv / gn) / 8.0), t / n / 1e3))
    rows.sort(key=lambda r: -r[1] * r[5])
    with open(out, "w") as fh:
        fh.write("kernel,launches,mfma_busy_cycles_per_launch,gui_active_per_launch,mfma_busy_fraction,avg_us_under_pmc\n")
        for r in rows:
            fh.write('"%s",%d,%.0f,%.0f,%.4f,%.2f\n' % r)
    for r in rows[:12]:
        print("%-60s x%-5d MFMA busy %5.1f %%  %8.1f us" % (r[0][:60], r[1], 100 * r[4], r[5]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--counter":
        return counter_table(sys.argv[2:5])
    if len(sys.argv) > 1 and sys.argv[1] == "--mfma-busy":
        return mfma_busy(sys.argv[2:5])
    f = agg(sys.argv[1], "FETCH_SIZE")
    w = agg(sys.argv[2], "WRITE_SIZE")
    prefix = sys.argv[3]
    rows = []
    for k, (n, fs, t) in f.items():
        ws = w.get(k, [1, 0.0, 0.0])
        rows.append({"kernel": k, "launches": n, "fetch_kib_per_launch": fs / n,
                     "write_kib_per_launch": ws[1] / max(ws[0], 1), "avg_us_under_pmc": t / n / 1e3,
                     "hbm_bytes_per_launch_corrected": (2 * fs / n + ws[1] / max(ws[0], 1)) * 1024, "total_us": t / 1e3})
    rows.sort(key=lambda r: -r["total_us"])
    with open(prefix + "_per_kernel.csv", "w") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()})
    for r in rows:          # GB/s against the 8 TB/s HBM peak, from the (PMC-slowed) durations of the same pass
        r["hbm_gbps_under_pmc"] = r["hbm_bytes_per_launch_corrected"] / max(r["avg_us_under_pmc"], 1e-9) / 1e3
    json.dump({"note": __doc__.split("writes")[1].strip(), "kernels": {r["kernel"]: r for r in rows[:24]}},
              open(prefix + "_summary.json", "w"), indent=1)
    for r in rows[:10]:
        print("%-64s x%-5d fetch %9.0f KiB  write %9.0f KiB  -> %7.1f MB/launch" % (
            r["kernel"][:64], r["launches"], r["fetch_kib_per_launch"], r["write_kib_per_launch"],
            r["hbm_bytes_per_launch_corrected"] / 1e6))


if __name__ == "__main__":
    main()
