#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_stats.csv into a short per-kernel table (our kernels only by default)."""
import csv
import sys


def main(path, all_kernels=False):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        name = r["Name"]
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").strip()
        short = short.split("(")[0].strip() if not short.startswith("(") else short
        if not all_kernels and not short.startswith(("k_", "ncclDevKernel", "rccl")):
            continue
        out.append((float(r["TotalDurationNs"]), short[:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print("%-60s %6s %12s %7s" % ("kernel", "calls", "avg_us", "share"))
    for t, n, c, a in out:
        print("%-60s %6d %12.1f %6.1f%%" % (n, c, a, 100 * t / tot))
    print("sum of listed kernels per call-set: %.3f ms" % (sum(o[3] for o in out) / 1e3))


def timed(trace_path, steps):
    """Per-kernel averages over the LAST `steps` launches of the kernel trace (= bench.py's timed region: the set-up and
    warm-up passes run cold -- the first score launch also pays the first touch of the 383 MB product -- and rocprofv3's
    --stats average includes them)."""
    import collections
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_path)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
        if n.startswith("k_"):
            d[n].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = []
    for n, v in d.items():
        v.sort()
        du = [x[1] for x in v]
        k = min(steps, len(du))
        out.append((sum(du[-k:]) / k, n, len(du), du[0]))
    out.sort(reverse=True)
    print()
    print("timed region (last %d launches of each kernel, p_kernel_trace.csv):" % steps)
    print("%-60s %6s %12s %12s" % ("kernel", "calls", "timed_avg_us", "first_us"))
    for a, n, c, f in out:
        print("%-60s %6d %12.1f %12.1f" % (n[:60], c, a, f))
    print("sum of timed averages: %.3f ms" % (sum(o[0] for o in out) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2].endswith(".csv"):
        main(sys.argv[1])
        timed(sys.argv[2], int(sys.argv[3]))
    else:
        main(sys.argv[1], len(sys.argv) > 2)
