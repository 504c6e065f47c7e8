#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.

FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section; re-calibrated in round 1 on a 4 GiB streaming read:
the counter reports 2,097,165 KB).  Counter units are KB (1024 B... the calibration fixes the scale: 1 unit = 1 KiB)."""
import csv, glob, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load(d, name):
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    # the CSV of the bench.py process itself: the one that holds k_score rows (a child process started under the profiler
    # would write its own file into the same directory; bench.py no longer starts one there)
    files = [x for x in files if "k_score" in open(x).read()] or files
    f = files[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].strip()
        if k.startswith("k_"):
            acc[k].append(float(r["Counter_Value"]))
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --in-flight 1 (tools/pmc_traffic.sh)",
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section; calibrated in round 1: a 4 GiB streaming read at 4/8/16 B per lane reports 2,097,165 KB; WRITE_SIZE exact on a 4 GiB copy)",
       "workload": "598x20000x425, p=72", "kernels": {}}
_h = hashlib.sha256()
for _f in ("cmf_score.hip", "cmf_common.h"):        # bench.py quotes this record only for the same score-kernel source
    _h.update(open(os.path.join(ROOT, "srcfinder_amd", "csrc", _f), "rb").read())
out["kernel_source_sha"] = _h.hexdigest()[:16]
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
    out["kernels"][k] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": len(fetch.get(k, [])),
                         "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
print(json.dumps(out, indent=1))
