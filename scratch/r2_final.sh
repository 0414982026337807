#!/bin/bash
# round 2, final tree: bench line + trace + PMC (profile_all.sh), shape sweep, C4 counters of the exact kernel, capacity figure
set -e
export TMPDIR=/tmp
bash scratch/profile_all.sh r02f > gpurun_out/r02f_profile_all.txt 2>&1
timeout -k 10 400 python3 scratch/sweep.py c1 c1big c2x4 c3 c5s c5 c4 c4m b1 b2 b5 b3 w2 w5 > gpurun_out/r02f_shape_sweep.jsonl 2> gpurun_out/r02f_shape_sweep.err
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02f_c4_trace -o t -- python3 scratch/sweep.py c4 c4m > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/r02f_c4_pmc -o p -- python3 scratch/sweep.py c4 > /dev/null 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/r02f_c4_pmc/p_counter_collection.csv")))
agg=collections.defaultdict(float); calls=collections.Counter()
for r in rows:
    k=r["Kernel_Name"]
    name="exact" if "sw_exact" in k else ("walk" if "sw_walk" in k else None)
    if name: agg[(name,r["Counter_Name"])]+=float(r["Counter_Value"])
with open("gpurun_out/r02f_c4_pmc_summary.txt","w") as f:
    f.write("C4 (100 kb x 100 kb, two calls): counters summed over both calls\n")
    for k,v in sorted(agg.items()): f.write("%s %s %.5g\n"%(k[0],k[1],v))
PY
cat gpurun_out/r02f_c4_trace/t_kernel_stats.csv | cut -c1-170 > gpurun_out/r02f_c4_kernel_stats.csv
timeout -k 10 120 python3 -m dlsa_amd benchmark --protocol > gpurun_out/r02f_capacity.txt 2>&1
timeout -k 10 120 python3 -m dlsa_amd benchmark >> gpurun_out/r02f_capacity.txt 2>&1
timeout -k 10 60 python3 scratch/pair_latency.py > gpurun_out/r02f_pair_latency.txt 2>&1
