#!/bin/bash
# run on the GPU box: full-size bench, kernel trace, and PMC passes (each in its own run)
set -e
export TMPDIR=/tmp
R=$1
timeout -k 10 300 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_bench_traced.json 2>/dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${R}_pmc_sq -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,collections,glob
for d in ("sq","fetch","write"):
    rows=list(csv.DictReader(open("gpurun_out/${R}_pmc_%s/p_counter_collection.csv"%d)))
    agg=collections.defaultdict(float)
    for r in rows:
        k=r["Kernel_Name"]
        name="score" if "sw_score" in k else ("exact" if "sw_exact" in k else ("walk" if "sw_walk" in k else None))
        if name: agg[(name,r["Counter_Name"])]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(d,k,"%.5g"%v)
PY
cat gpurun_out/${R}_trace/t_kernel_stats.csv
cut -c1-1800 gpurun_out/${R}_bench.json
