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


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
