#!/bin/bash
# Cost order (64 levels by mixed views, short XCD runs) against the four work levels, by grid size.  Usage: tools/gpu_cost_order_ab.sh <tag>
set -u
TAG=${1:-cost_order}; mkdir -p gpurun_out
for wl in cfg2 128x64@640x480 384x96@640x480 cfg3; do
  rounds=9; [ $wl = cfg3 ] && rounds=5
  timeout -k 10 500 python tools/gpu_exp.py tools/exp_list_head.txt --workload $wl --rounds $rounds --variants 2097152,1048576 --scenes dense,speckle --tag ${TAG}_${wl%%@*} > gpurun_out/${TAG}_${wl%%@*}.log 2>&1 || { tail -5 gpurun_out/${TAG}_${wl%%@*}.log; exit 1; }
  grep -E " fuse " gpurun_out/${TAG}_${wl%%@*}.log
done
