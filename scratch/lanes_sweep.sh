#!/bin/bash
# 15 (and 30) single-target sets in one batch, on 1 .. 8 context lanes; and with more hardware queues
for L in 1 2 3 4 6 8; do
  echo "== SW_HIP_PACKAGE_LANES=$L"
  SW_HIP_PACKAGE_LANES=$L timeout -k 10 100 python3 -u scratch/lanes15.py 15 | tail -1 || exit 1
done
for Q in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$Q SW_HIP_PACKAGE_LANES=8"
  GPU_MAX_HW_QUEUES=$Q SW_HIP_PACKAGE_LANES=8 timeout -k 10 100 python3 -u scratch/lanes15.py 15 | tail -1 || exit 1
done
echo "== 30 pairs, 8 lanes"
SW_HIP_PACKAGE_LANES=8 timeout -k 10 100 python3 -u scratch/lanes15.py 30 | tail -1
