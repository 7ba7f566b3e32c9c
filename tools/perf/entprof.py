"""Per-entity times of the mesh entity kernel's last frame (MW_ENT_PROF=<file> python bench.py --config pickup_dr ...).
usage: python tools/perf/entprof.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], np.uint64).reshape(-1, 8)
t0, t1, nm, pv, p1, p23, blk, item = [a[:, i].astype(np.int64) for i in range(8)]
tris, win, nm = (nm >> 8) & 0xFFFFFF, nm >> 32, nm & 0xFF
big, item = item >> 32, item & 0xFFFFFFFF
p2, p3 = p23 & 0xFFFFFFFF, p23 >> 32
live = nm > 0
print("entities", int(live.sum()), "in", len(np.unique(item[live] & 0xFFFFFF)), "envs,", "triangles", int(tris.sum()), "wave-sized", int(big.sum()), "winners", int(win.sum()))
if live.any():
    dur = (t1 - t0)[live] / 100.0          # s_memrealtime: 100 MHz
    start = t0[live].min()
    print("kernel span (first start -> last end) us: %.1f" % ((t1[live].max() - start) / 100.0))
    print("per entity us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f;  sum %.0f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum()))
    for lo, hi in ((0, 1024), (1024, 1 << 30)):
        m = live & (tris >= lo) & (tris < hi)
        if m.any():
            print("  entities of %d..%d triangles: %d, mean %.1f us = vertex stage %.1f + first pass %.1f + wave-sized %.1f + winners %.1f + rest %.1f" % (
                lo, hi, m.sum(), ((t1 - t0)[m]).mean() / 100.0, pv[m].mean() / 100.0, p1[m].mean() / 100.0, p2[m].mean() / 100.0, p3[m].mean() / 100.0,
                ((t1 - t0)[m] - pv[m] - p1[m] - p2[m] - p3[m]).mean() / 100.0))
    order = np.argsort(-dur)[:6]
    idx = np.nonzero(live)[0]
    for o in order:
        e = idx[o]
        print("  item %5d (env %d, entry %d): %.1f us (start +%.1f), %d triangles, %d wave-sized, %d winners, workgroup %d" % (e, item[e] & 0xFFFFFF, item[e] >> 24, dur[o], (t0[e] - start) / 100.0, tris[e], big[e], win[e], blk[e]))
    # per workgroup busy time
    busy = {}
    for e in idx:
        busy[blk[e]] = busy.get(blk[e], 0.0) + (t1[e] - t0[e]) / 100.0
    b = np.array(list(busy.values()))
    print("workgroups with work %d: busy mean %.1f max %.1f us" % (len(b), b.mean(), b.max()))
