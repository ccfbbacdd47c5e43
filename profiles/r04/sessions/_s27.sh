#!/bin/bash
# Tooling: round-4 session 27: wave account of the polygon kernel (-DIRBPP_AB_POLY_ACCOUNT build) at 512 / 4096 bins
O=gpurun_out/r04_s27; mkdir -p $O
export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_polyacct.so
for spec in blockout:512 blockout:4096 general:4096; do
  wl=${spec%%:*}; bins=${spec##*:}
  timeout 300 python tools/polygon_profile.py --workload $wl --bins $bins > $O/polygon_$wl$bins.json 2> $O/err_$wl$bins.txt; tail -c 1500 $O/polygon_$wl$bins.json; tail -3 $O/err_$wl$bins.txt
done
