#!/usr/bin/env python3
"""Reads the records of scratch/lanes_timeline.sh (per strip wave: start, end, ticks waiting, hw id, unit) and prints
where the launch's time goes: active waves over time, per strip level start / duration / waiting."""
import sys, struct
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
pos = 0
launches = []
while pos < len(raw):
    assert raw[pos] == 0x54494d454c494e45, hex(int(raw[pos]))
    n = int(raw[pos + 1]); pos += 4
    launches.append(raw[pos:pos + 4 * n].reshape(n, 4)); pos += 4 * n
print("launches:", len(launches), [len(l) for l in launches])
L = launches[-1]
t0 = int(L[:, 0].min())
start = (L[:, 0].astype(np.int64) - t0) / 100.0   # us (100 MHz)
end = (L[:, 1].astype(np.int64) - t0) / 100.0
wait = (L[:, 2] & 0xffffffff).astype(np.int64) / 100.0
first = start + (L[:, 2] >> 32).astype(np.int64) / 100.0   # when the wave had its first block of boundary rows
unit = (L[:, 3] & 0xffffffff).astype(np.int64)
hw = (L[:, 3] >> 32).astype(np.int64)
grp, strip = unit & 0xfffff, unit >> 20
print(f"units {len(L)}, span {end.max():.0f} us, sum of wave time {np.sum(end-start)/1e3:.1f} ms, of which waiting {wait.sum()/1e3:.1f} ms ({100*wait.sum()/np.sum(end-start):.0f} %)")
# resident and working waves over time
T = np.linspace(0, end.max(), 41)
print("  t_us resident working(approx: resident x (1 - waiting share of those waves))")
for a, b in zip(T[:-1], T[1:]):
    m = (start < b) & (end > a)
    res = m.sum()
    share = 1 - (wait[m].sum() / max(1e-9, (end[m] - start[m]).sum()))
    print(f"  {a:7.0f} {res:6d} {res*share:8.0f}")
# one group, level by level: when each strip got going
g0 = int(np.bincount(grp).argmax())
m = grp == 0
order = np.argsort(strip[m])
fs = first[m][order]; ss = strip[m][order]; en = end[m][order]; wt = wait[m][order]
print("group 0: level, first block at (us), lag behind the level before, end, waiting in all")
for k in range(len(ss)):
    if k < 12 or k % 6 == 0:
        print(f"  {ss[k]:3d} {fs[k]:8.1f} {fs[k]-fs[k-1] if k else 0:7.1f} {en[k]:8.1f} {wt[k]:8.1f}")
print("per strip level: first start, median start, median duration, median waiting (us)")
for s in sorted(set(strip.tolist())):
    m = strip == s
    if s % 4 == 0 or s == strip.max():
        print(f"  level {s:3d}: n {m.sum():4d} first {start[m].min():7.0f} median start {np.median(start[m]):7.0f} duration {np.median(end[m]-start[m]):7.0f} waiting {np.median(wait[m]):7.0f}")
xcc = (hw >> 28) & 0xf
print("waves per XCC:", np.bincount(xcc, minlength=8).tolist())
# same-group neighbours on the same XCC?
same = 0; tot = 0
key = {(int(g), int(s)): int(x) for g, s, x in zip(grp, strip, xcc)}
for (g, s), x in key.items():
    if s > 0 and (g, s - 1) in key:
        tot += 1; same += key[(g, s - 1)] == x
print(f"left neighbour on the same XCC: {same} of {tot}")
