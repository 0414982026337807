#!/usr/bin/env python3
"""The literal drop-in, measured (GPU box): the plain-C stand-in for the Go worker (tests/c_harness/worker_harness:
N threads, static chunks, ONE pair per find_alignment_low_memory call, C.GoString copies, one free per result --
internal/worker/worker_impl.go:170-238, worker_rust_bridge.go:105-124) on
  * the reference's default synthetic distribution (utils/generate_synthetic_data.py:7-12: queries 200-1000 nt,
    targets 10-200 kb) and
  * pairs of ~1 kb,
with 1 / 15 / 64 caller threads, and beside every figure the oracle's CPU restatement of the same Rust path on the
same pairs (a bounded prefix) with the same number of threads, driven the same way.
Usage: python3 scratch/dropin_bench.py > gpurun_out/r03_dropin.txt"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

HARNESS = os.path.join(ROOT, "tests", "c_harness", "worker_harness")
rng = random.Random(2024)


def r(n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def shapes():
    synth_q = [r(rng.randint(200, 1000)) for _ in range(10)]
    synth_t = [r(rng.randint(10000, 200000)) for _ in range(40)]
    synth = [(q, t) for q in synth_q for t in synth_t]                       # 400 pairs, as a work package lists them
    q1k = [r(1000) for _ in range(8)]
    one_kb = [(q1k[i % 8], r(rng.randint(900, 1100))) for i in range(3000)]
    return {"synthetic default (q 200-1000 x t 10k-200k)": (synth, 60), "1 kb pairs": (one_kb, 600)}


def harness(pairs, threads):
    path = "/tmp/dropin_pairs.txt"
    with open(path, "w") as f:
        for q, t in pairs:
            f.write(f"-1 2 -1 {q} {t}\n")
    p = subprocess.run([HARNESS, "pairs", path, str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    line = [x for x in p.stderr.decode().splitlines() if "pairs/s" in x][-1]
    secs = float(line.split("threads:")[1].split("s,")[0])
    return secs


def main():
    print("drop-in ABI, one pair per call (find_alignment_low_memory through the C harness) vs the CPU restatement, same pairs")
    print(f"{'shape':46s} {'threads':>7s} {'GPU pairs/s':>12s} {'GPU us/pair':>12s} {'GPU GCUPS':>10s} | {'CPU GCUPS':>10s} {'CPU pairs':>9s} | GPU / CPU")
    for name, (pairs, n_cpu) in shapes().items():
        cells = sum(len(q) * len(t) for q, t in pairs)
        for threads in (1, 15, 64):
            harness(pairs[: max(8, len(pairs) // 8)], threads)  # warm-up (buffers, code objects)
            secs = harness(pairs, threads)
            cpu_threads = min(threads, 15)
            sub = pairs[: max(cpu_threads, n_cpu if threads > 1 else max(4, n_cpu // 6))]
            csecs, ccells, _ = oracle_lib.bench_pairs([q.encode() for q, _ in sub], [t.encode() for _, t in sub], -1, 2, -1, cpu_threads)
            gpu_g, cpu_g = cells / secs / 1e9, ccells / csecs / 1e9
            print(f"{name:46s} {threads:7d} {len(pairs) / secs:12.0f} {secs / len(pairs) * 1e6:12.1f} {gpu_g:10.2f} | "
                  f"{cpu_g:10.2f} {len(sub):9d} | {gpu_g / cpu_g:6.2f}x   (CPU threads: {cpu_threads})", flush=True)


if __name__ == "__main__":
    main()
