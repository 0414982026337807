#!/bin/bash
# usage: scratch/pmc.sh <outdir-name>   (run on the GPU box)
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$1 -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/$1/p_counter_collection.csv")))
agg=collections.defaultdict(float)
for r in rows:
    if "sw_score" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items()): print(k, "%.4g"%v)
PY
