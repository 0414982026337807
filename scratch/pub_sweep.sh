#!/bin/bash
for pb in 4 8 16 32; do echo -n "PUB $pb: "; SW_HIP_SCORE_PUB=$pb timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | grep -o "kernel_ms\": [0-9.]*"; done
