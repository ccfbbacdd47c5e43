#!/bin/bash
# Tooling: round-4 session 34: account of the emit kernel's workgroups (-DIRBPP_AB_EMIT_ACCOUNT build)
O=gpurun_out/r04_s34; mkdir -p $O
for spec in blockout:4096 blockout:1024 general:4096; do
  wl=${spec%%:*}; bins=${spec##*:}
  IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_emitacct.so timeout 300 python tools/emit_profile.py --workload $wl --bins $bins > $O/emit_$wl$bins.json 2> $O/err_$wl$bins.txt; tail -1 $O/emit_$wl$bins.json; tail -2 $O/err_$wl$bins.txt | grep -v amdgpu.ids
done
