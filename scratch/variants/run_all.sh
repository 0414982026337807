#!/bin/bash
# times every libsw_*.so in this directory; stops at the first failure
cd "$(dirname "$0")/../.."
for so in scratch/variants/libsw_*.so; do
  SW_HIP_LIB=$PWD/$so timeout -k 10 120 python3 scratch/variants/time_score.py || exit 1
done
