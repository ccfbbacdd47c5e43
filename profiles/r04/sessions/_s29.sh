#!/bin/bash
# Tooling: round-4 session 29: when do the polygon kernel's waves start and end (wall-clock stamps of the account build)
O=gpurun_out/r04_s29; mkdir -p $O
for spec in blockout:4096 blockout:1024; do
  wl=${spec%%:*}; bins=${spec##*:}
  IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_polyacct.so timeout 300 python tools/polygon_profile.py --workload $wl --bins $bins > $O/polygon_$wl$bins.json 2> $O/err_$wl$bins.txt; tail -2 $O/polygon_$wl$bins.json | cut -c1-1200
done
