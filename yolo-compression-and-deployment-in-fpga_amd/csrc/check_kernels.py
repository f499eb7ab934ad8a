#!/usr/bin/env python3
"""Post-compile guard: the LDS-DMA kernels wait with COUNTED `s_waitcnt vmcnt(N)` whose constants assume
that the only vector-memory operations in flight are the DMAs and stores the source writes.  A register spill adds
scratch loads/stores to that in-order counter and the waits become too lenient -- silently wrong tiles under load.
This reads the -Rpass-analysis=kernel-resource-usage remarks the Makefile keeps in build/*.rpt and fails if one of
those kernels uses scratch; it also reports VGPRs / LDS / occupancy of every production kernel and warns when the
compiler is not the validated ROCm 7.2."""
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SIGFILE = os.path.join(HERE, "ring_signature.json")


def signatures(build):
    """Per counted-wait kernel of build/*.s: {LDS-DMA instructions, global stores, MFMAs, histogram of `s_waitcnt vmcnt(N)`}.
    The waits of conv3x3_ring.hip / convpx.hip / convr.hip / pxpair.hip (pxpair3r_kernel) are constants derived from how many LDS-DMA pieces and output stores a wave
    has issued since the data it waits for (vmcnt retires in issue order): if a compiler or a source edit changes any of
    these counts, the constants must be re-derived and the kernels re-validated under load before the signature is updated
    (`python3 check_kernels.py build --record`)."""
    out = {}
    for f in sorted(glob.glob(os.path.join(build, "*.s"))):
        name = None
        for line in open(f, errors="replace"):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name = m.group(1) if any(g in m.group(1) for g in ("conv3x3_i8_ring_kernel", "convpx_kernel", "convr_kernel", "pxpair3r_kernel", "convpxb_kernel")) else None
                if name:
                    out[name] = {"lds_dma": 0, "stores": 0, "mfma": 0, "vmcnt": {}}
                continue
            if not name:
                continue
            if "s_endpgm" in line:
                name = None
                continue
            t = line.strip().split()
            if not t:
                continue
            op = t[0]
            if op.startswith("global_load_lds"):
                out[name]["lds_dma"] += 1
            elif op.startswith("global_store") or op.startswith("buffer_store"):
                out[name]["stores"] += 1
            elif op.startswith("v_mfma"):
                out[name]["mfma"] += 1
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:
                    out[name]["vmcnt"][m.group(1)] = out[name]["vmcnt"].get(m.group(1), 0) + 1
    return out

GUARDED = ("conv3x3_i8_ring_kernel", "conv1_fast_kernel", "front_kernel", "convpx_kernel", "convr_kernel", "pxpair", "convpxb_kernel")


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
    sig = signatures(build)
    if "--record" in sys.argv:
        json.dump(sig, open(SIGFILE, "w"), indent=1, sort_keys=True)
        print("check_kernels: recorded the signature of %d kernels in %s" % (len(sig), os.path.basename(SIGFILE)))
    elif sig and "Y355_DIAG" not in os.environ.get("EXTRA", ""):
        try:
            want = json.load(open(SIGFILE))
        except OSError:
            want = None
        if want is None:
            print("check_kernels: WARNING: no ring_signature.json (python3 check_kernels.py build --record)")
        else:
            for k in sorted(set(sig) | set(want)):
                if sig.get(k) != want.get(k):
                    bad.append(("signature", k, -1))
                    print("check_kernels: %s: LDS-DMA / store / MFMA / vmcnt counts differ from ring_signature.json:\n   built    %s\n   recorded %s"
                          % (k, json.dumps(sig.get(k), sort_keys=True), json.dumps(want.get(k), sort_keys=True)))
    if bad:
        for f, k, n in bad:
            if n < 0:
                continue
            print("check_kernels: %s: %s uses %d bytes of scratch per lane: its counted vmcnt waits are no longer valid" % (f, k, n))
        return 1
    print("check_kernels: %d counted-wait kernel instantiations, none spills; %d signatures match" % (seen, len(sig)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "build"))
