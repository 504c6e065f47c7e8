#!/usr/bin/env python3
"""Build guard (srcfinder_amd/csrc/Makefile): k_sweep4s reads LDS through asm with hand-counted s_waitcnt, so a spill or a
scratch slot inside it would shift the counts and the kernel would consume stale registers silently.  Reads the
-Rpass-analysis=kernel-resource-usage remarks of the compile, FAILS when any instantiation of the kernel spills or uses scratch
-- and also when NO instantiation is found (a renamed kernel or changed remark labels must not pass silently) --
and prints whatever else the compiler wrote (its warnings share the stderr the remarks arrive on)."""
import re
import sys


def main(path, kernel):
    t = open(path).read()
    seen = []
    for l in t.splitlines():                      # the compiler's own diagnostics, once each
        if re.search(r"\b(warning|error):", l) and l not in seen:
            seen.append(l)
    if seen:
        print("\n".join(seen))
    blocks = re.findall(r"Function Name: (\S*%s\S*)(.*?)(?=Function Name:|\Z)" % re.escape(kernel), t, re.S)
    if not blocks:
        sys.exit("%s: no instantiation of %s in the resource remarks -- the spill guard matched nothing" % (path, kernel))
    bad = []
    for name, body in blocks:
        sc = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", body)
        sp = re.search(r"VGPRs Spill: (\d+)", body)
        if sc is None or sp is None:
            sys.exit("%s: resource remarks of %s lack ScratchSize / VGPRs Spill -- the spill guard cannot read them" % (path, name))
        if int(sc.group(1)) or int(sp.group(1)):
            bad.append(name)
    if bad:
        sys.exit("%s uses scratch / spills: %s" % (kernel, bad[0]))
    print("%s: no scratch, no spills (%d instantiations)" % (kernel, len(blocks)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
