#!/bin/bash
# Tooling: an A/B build of the library under another name.  tools/build_variant.sh NAME -DFLAG=... ;
# select it with IRBPP_LIBRARY=irbpp_amd/libirbpp_var_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
H=$(cd $R && python -c "from irbpp_amd import build; print(build.source_hash())")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -shared -fPIC -Wno-unused-value "-DIRBPP_SOURCE_HASH=\"$H+$N\"" "$@" \
  $R/irbpp_amd/csrc/irbpp_capi.hip -o $R/irbpp_amd/libirbpp_var_$N.so
