#!/usr/bin/env python3
"""Instruction mix and register use of kernels in a hipcc -S listing:  python tools/isa_stats.py file.s <name substring>"""
import re
import sys


def main(path, pat):
    t = open(path).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s*;\s*@", t, re.M)]
    for i, (pos, name) in enumerate(starts):
        if pat not in name:
            continue
        end = starts[i + 1][0] if i + 1 < len(starts) else len(t)
        body = t[pos:end]
        meta = t[end - 1:]
        c = lambda r: len(re.findall(r, body))
        g = lambda k: (re.search(r"; %s: (\d+)" % k, body) or [None, "?"])[1]
        print("%s\n   glds %d  buffer_load %d  global_load %d  ds_read_b128 %d  ds_read_b64 %d  ds_read_b32 %d  ds_write %d  mfma %d  "
              "s_barrier %d  waitcnt %d | VGPRs %s AGPRs %s SGPRs %s scratch %s LDS %s occupancy %s"
              % (name[:90], c(r"global_load_lds"), c(r"buffer_load"), c(r"global_load_dword"), c(r"ds_read_b128"),
                 c(r"ds_read_b64"), c(r"ds_read_b32"), c(r"ds_write"), c(r"v_mfma"), c(r"s_barrier"), c(r"s_waitcnt"),
                 g("NumVgprs"), g("NumAgprs"), g("NumSgprs"), g("ScratchSize"), g("LDSByteSize"), g("Occupancy")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
