#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/c5_pmc -o p -- python3 scratch/sweep.py c5 > gpurun_out/c5_pmc.txt 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/c5_pmc/p_counter_collection.csv")))
agg=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    if "sw_score" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k,v in sorted(agg.items()): print(k, "%.5g"%(v/n[k]), "per launch (", n[k], "rows )")
PY
