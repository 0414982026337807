#!/bin/bash
# usage: scratch/pmc_lanes.sh <outdir-name> Q T N [nq]  (GPU box): SQ counters of the lane kernels for one batch shape
export TMPDIR=/tmp SW_HIP_LANES=1
out=$1; shift
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/$out -o p -- python3 scratch/lanes_ab.py "$@" > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${out}b -o p -- python3 scratch/lanes_ab.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,collections
for d in ("gpurun_out/$out", "gpurun_out/${out}b"):
    try: rows=list(csv.DictReader(open(d+"/p_counter_collection.csv")))
    except Exception as e: print("no file", d, e); continue
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in rows:
        if "sw_lanes" in r["Kernel_Name"]:
            k=("walk " if "walk" in r["Kernel_Name"] else "fill ")+r["Counter_Name"]
            agg[k]+=float(r["Counter_Value"]); n[k]+=1
    for k,v in sorted(agg.items()): print(k, "%.4g per launch"%(v/n[k]), "launches", n[k])
PY
