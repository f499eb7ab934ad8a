#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV of bench.py's three-handle region: per kernel type the launch-to-launch gap on its own
queue, the duration, the CUs its grid can hold, and the CU-time integral against 256 CUs x wall time.
usage: timeline.py <kernel_trace.csv> [first_fraction last_fraction]"""
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])),
                 int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])))
rows.sort()
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.65)
# the densest part of the run: the window between fractions f0 and f1 of the dispatches of the most frequent conv kernel
ring = [r for r in rows if "ring_kernel<256, 128" in r[3]]
t_lo, t_hi = ring[int(len(ring) * f0)][0], ring[int(len(ring) * f1)][0]
win = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
queues = sorted({r[2] for r in win})
print("window %.3f ms, %d dispatches, queues %s" % ((t_hi - t_lo) / 1e6, len(win), queues))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:64]
def cus(wgs, threads, lds, vgpr):
    waves = (threads + 63) // 64
    per_cu = 64  # workgroups a CU can hold
    if lds: per_cu = min(per_cu, 163840 // lds)
    wps = max(1, (512 // max(vgpr, 1)))            # waves per SIMD by registers (512 per SIMD lane-slice)
    per_cu = min(per_cu, max(1, (min(wps, 8) * 4) // waves))
    return min(256.0, wgs / per_cu), per_cu
last_end = {}
st = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0, 0])
cu_time = 0.0
for s, e, q, n, wgs, thr, lds, vgpr in win:
    k = short(n)
    c, per = cus(wgs, thr, lds, vgpr)
    a = st[k]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if q in last_end: a[2] += max(0, s - last_end[q]) / 1e3
    a[3] += c * (e - s) / 1e3
    a[4], a[5] = wgs, per
    cu_time += c * (e - s) / 1e3
    last_end[q] = e
wall = (t_hi - t_lo) / 1e3
print("%-64s %6s %9s %9s %7s %6s %9s" % ("kernel", "n", "dur us", "gap us", "WGs", "/CU", "CU-us/launch"))
for k, a in sorted(st.items(), key=lambda kv: -kv[1][3]):
    print("%-64s %6d %9.1f %9.1f %7d %6d %9.0f" % (k, a[0], a[1] / a[0], a[2] / a[0], a[4], a[5], a[3] / a[0]))
print("sum of (CUs a grid can hold) x duration = %.0f CU-us over %.0f us wall = %.1f CUs busy on average (of 256)" % (cu_time, wall, cu_time / wall))
for q in queues:
    rq = [r for r in win if r[2] == q]
    busy = sum(r[1] - r[0] for r in rq) / 1e3
    print("queue %d: %d dispatches, busy %.0f us of %.0f (%.2f)" % (q, len(rq), busy, wall, busy / wall))
