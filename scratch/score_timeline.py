#!/usr/bin/env python3
"""Reads the per-wave accumulators of the SW_SCORE_TIMELINE build (scratch/score_timeline.sh) after one database
search and prints where the score kernel's wave time goes.  usage: score_timeline.py <qlen> <n_targets> <target_len>"""
import sys, os, ctypes, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dlsa_amd
from dlsa_amd import bridge
args = [a for a in sys.argv[1:] if not a.startswith("--")]
qlen, n, tlen = int(args[0]), int(args[1]), int(args[2])
rng = np.random.default_rng(7)
lut = np.frombuffer(b"ATCG", dtype=np.uint8)
q = lut[rng.integers(0, 4, size=qlen, dtype=np.uint8)].tobytes()
lens = np.full(n, tlen)
offs = np.zeros(n + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
flat = lut[rng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)]
db = bridge.Database(flat=flat, offsets=offs)
sc = dlsa_amd.AlignmentScore(2, -1, -1)
lib = ctypes.CDLL(os.environ["SW_HIP_LIB"])
lib.sw_hip_debug_score_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for rep in range(2):
    db.search_topk(q, sc, 5)
st = bridge.last_stats()
buf = np.zeros(256 * 16 * 8, dtype=np.uint32)
rc = lib.sw_hip_debug_score_timeline(buf.ctypes.data, buf.size)
assert rc == 0, rc
A = buf.reshape(256, 16, 8).astype(np.float64) / 100.0  # us
tot, queue, setup, ready, done, glb, strips, tail = [A[:, :, i] for i in range(8)]
strips = strips * 100.0
m = tot > 0
span = tot[m].max()
print(json.dumps({"qlen": qlen, "n": n, "tlen": tlen, "score_kernel_ms": round(st["score_kernel_ms"], 3), "score_launches": st["score_launches"]}))
print(f"waves {m.sum()}, kernel span {span:.0f} us; per wave (mean, us): total {tot[m].mean():.0f}, ticket/queue wait {queue[m].mean():.1f}, "
      f"set-up {setup[m].mean():.1f} ({setup[m].sum()/strips[m].sum():.2f} per strip, {strips[m].mean():.1f} strips), left neighbour {ready[m].mean():.1f}, "
      f"ring room {done[m].mean():.1f}, other workgroup {glb[m].mean():.1f}, idle after the last unit {(span - tail[m]).mean():.1f}")
share = lambda x: 100 * x[m].sum() / (span * m.sum())
print(f"share of waves x span: queue {share(queue):.1f} %, set-up {share(setup):.1f} %, left {share(ready):.1f} %, room {share(done):.1f} %, "
      f"other wg {share(glb):.1f} %, rows (rest) {100 - share(queue) - share(setup) - share(ready) - share(done) - share(glb) - 100 * (span - tot[m]).sum() / (span * m.sum()):.1f} %, "
      f"not resident {100 * (span - tot[m]).sum() / (span * m.sum()):.1f} %")
print("per wave index (mean us): wave queue setup left room other")
for w in range(16):
    mm = m[:, w]
    if mm.any():
        print(f"  {w:2d} {queue[:, w][mm].mean():8.1f} {setup[:, w][mm].mean():8.1f} {ready[:, w][mm].mean():8.1f} {done[:, w][mm].mean():8.1f} {glb[:, w][mm].mean():8.1f}")
# distribution over workgroups: when each finished, and the ticket waits of a middle wave
if "--wg" in sys.argv:
    ends = tot.max(axis=1)
    print("workgroup end (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (ends[ends > 0].min(), np.percentile(ends[ends > 0], 10), np.median(ends[ends > 0]), np.percentile(ends[ends > 0], 90), ends.max()))
    q8 = queue[:, 8]
    print("ticket wait of wave 8 per workgroup (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (q8.min(), np.percentile(q8, 10), np.median(q8), np.percentile(q8, 90), q8.max()))
    st8 = strips[:, 8]; st0 = strips[:, 0]; st15 = strips[:, 15]
    print("strips of wave 0 / 8 / 15 per workgroup: %s / %s / %s" % (np.bincount(st0.astype(int)).tolist(), np.bincount(st8.astype(int)).tolist(), np.bincount(st15.astype(int)).tolist()))
    lastu = tail.max(axis=1)
    print("start of a workgroup's last unit (us): min %.0f median %.0f max %.0f" % (lastu.min(), np.median(lastu), lastu.max()))
