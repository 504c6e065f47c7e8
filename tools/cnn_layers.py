#!/usr/bin/env python3
"""Per-launch view of ONE batch of the CNN tile scorer from a rocprofv3 kernel trace (csv): the launches between the last
k_conv1_pool / k_conv1 and the following k_head, in order, with grid and duration.
   python tools/cnn_layers.py <kernel_trace.csv>"""
import csv
import re
import sys


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
        if n.startswith("_Z"):                     # a name the profiler's demangler gave up on (_Float16 parameters): k_name<first int argument>
            m = re.search(r"\d+(k_[a-z0-9_]+)(?:ILi(\d+))?", n)
            n = (m.group(1) + ("<%s>" % m.group(2) if m.group(2) else "")) if m else n
        if n.startswith("k_"):
            rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", "")))
    rows.sort()
    heads = [i for i, r in enumerate(rows) if r[1].startswith("k_head")]
    i1 = heads[-1]                                            # the last batch: from the launch after the previous head
    i0 = heads[-2] + 1 if len(heads) > 1 else 0
    tot = sum(r[2] for r in rows[i0:i1 + 1])
    print("%-28s %10s %8s %9s %6s" % ("kernel", "grid_x", "grid_y", "us", "share"))
    for r in rows[i0:i1 + 1]:
        print("%-28s %10s %8s %9.1f %5.1f%%" % (r[1][:28], r[3], r[4], r[2], 100 * r[2] / tot))
    print("batch total %.1f us (kernel time, %.1f us wall first start to last end)" %
          (tot, (rows[i1][0] - rows[i0][0]) / 1e3 + rows[i1][2]))


if __name__ == "__main__":
    main(sys.argv[1])
