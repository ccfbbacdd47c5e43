#!/bin/bash
# Tooling: an A/B build of the library under another name.  tools/build_variant.sh NAME -DFLAG=... ;
# select it with IRBPP_LIBRARY=irbpp_amd/libirbpp_var_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -shared -fPIC -Wno-unused-value "$@" \
  $R/irbpp_amd/csrc/irbpp_capi.hip -o $R/irbpp_amd/libirbpp_var_$N.so
