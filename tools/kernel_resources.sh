#!/bin/bash
# Tooling: register / spill / occupancy figures of every kernel as the compiler reports them.
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -shared -fPIC -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage irbpp_amd/csrc/irbpp_capi.hip -o /tmp/_res.so 2>&1 | \
  grep -E "Function Name|SGPRs:|VGPRs:|Spill|Occupancy|ScratchSize" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - -
