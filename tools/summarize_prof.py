"""Condense rocprofv3 CSV output (run on the GPU box) into small per-kernel summaries that fit gpurun_out/ and
are then committed under profiles/.

    python tools/summarize_prof.py stats  <dir-with-*_kernel_stats.csv>  <out.md>
    python tools/summarize_prof.py pmc    <dir-with-*_counter_collection.csv>  <out.csv>

``pmc`` groups the per-dispatch counter rows by (kernel, counter): number of dispatches, summed counter value,
summed dispatch duration -- only for kernels of libmofa_hip.so (names containing ``_kernel``).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def _find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return fs[0]


def _col(row, *names):
    low = {k.lower(): k for k in row}
    for n in names:
        if n.lower() in low:
            return low[n.lower()]
    raise KeyError(names)


def stats(d, out):
    f = _find(d, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    name, calls = _col(rows[0], "Name"), _col(rows[0], "Calls")
    tot, avg, pct = _col(rows[0], "TotalDurationNs"), _col(rows[0], "AverageNs"), _col(rows[0], "Percentage")
    mn, mx = _col(rows[0], "MinNs"), _col(rows[0], "MaxNs")
    with open(out, "w") as o:
        o.write(f"source: rocprofv3 --kernel-trace --stats ({os.path.basename(f)})\n\n")
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:40]:
            o.write(f"| `{r[name][:90]}` | {r[calls]} | {float(r[tot]) / 1e6:.2f} | {float(r[avg]) / 1e3:.1f} | "
                    f"{float(r[mn]) / 1e3:.1f} | {float(r[mx]) / 1e3:.1f} | {float(r[pct]):.2f} |\n")
    print(open(out).read()[:3000])


def pmc(d, out):
    f = _find(d, "*counter_collection.csv")
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    rd = csv.DictReader(open(f))
    first = None
    for r in rd:
        if first is None:
            first = r
            kn, cn, cv = _col(r, "Kernel_Name"), _col(r, "Counter_Name"), _col(r, "Counter_Value")
            st, en = _col(r, "Start_Timestamp"), _col(r, "End_Timestamp")
        k = r[kn]
        if "_kernel" not in k:
            continue
        a = agg[(k[:100], r[cn])]
        a[0] += 1
        a[1] += float(r[cv])
        a[2] += float(r[en]) - float(r[st])
    with open(out, "w") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "counter", "dispatches", "counter_sum", "duration_ns_sum"])
        for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            w.writerow([k, c, a[0], f"{a[1]:.6g}", f"{a[2]:.6g}"])
    print(open(out).read()[:3000])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
