#!/bin/bash
# walker staging cost experiment: C4 shapes with each variant library
set -e
mkdir -p gpurun_out
for v in wkb16 st2 st3; do
  echo "== $v" >> gpurun_out/wkb_sweep.txt
  SW_HIP_LIB=$PWD/scratch/variants/libsw_$v.so timeout -k 10 120 python3 scratch/sweep.py c4 c4m >> gpurun_out/wkb_sweep.txt 2>&1
done
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o s -- python3 scratch/sweep.py c4 > /dev/null 2>&1
cat gpurun_out/prof_c4/*kernel_stats.csv | cut -c1-160 >> gpurun_out/wkb_sweep.txt
