#!/bin/bash
set -o pipefail
O=gpurun_out/r6g; mkdir -p $O
LCS_HOST_TIMING=1 timeout -k 10 120 python tools/pcie_rate.py > $O/pcie_timing.txt 2>&1; grep "lc_lcs_host" $O/pcie_timing.txt
