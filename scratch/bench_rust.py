import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlsa_amd import bridge
L = bridge.lib()
for _ in range(2):
    t0 = time.perf_counter(); v = L.sw_benchmark_rust(0.010, 1, 2); dt = time.perf_counter() - t0
    print("sw_benchmark_rust(10 ms, 1, 2) = %.3f GCUPS  (%.2f s)" % (v / 1e9, dt), bridge.package_stats())
print("sw_hip_benchmark_cups = %.1f GCUPS" % (L.sw_hip_benchmark_cups() / 1e9))
