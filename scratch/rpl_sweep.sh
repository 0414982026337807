#!/bin/bash
# exact kernel: forced rows-per-lane against the host's choice (exact_choose_rpl), long pairs and per-call pairs
set -e
for r in 0 1 2 4; do
  echo "== SW_HIP_EXACT_RPL=$r"
  SW_HIP_EXACT_RPL=$r timeout -k 10 120 python3 scratch/sweep.py c4 p20k p50k strips | cut -c1-140
  SW_HIP_EXACT_RPL=$r timeout -k 10 60 python3 scratch/pair_latency.py
done
