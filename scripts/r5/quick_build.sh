#!/bin/bash
# rebuild only the translation units named (default: lbfgsb_x) and relink liblbfgsx.so -- for kernel iterations that touch
# lbfgsb_x.cuh only (lbfgspp_amd/_build.py rebuilds every unit when any header changes: 5+ minutes for lbfgsb.hip)
set -e
cd "$(dirname "$0")/../.."
units=${@:-lbfgsb_x}
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -c lbfgspp_amd/csrc/$u.hip -o lbfgspp_amd/build/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC lbfgspp_amd/build/*.o -o lbfgspp_amd/liblbfgsx.so -Wl,--version-script=lbfgspp_amd/csrc/export.map
touch lbfgspp_amd/build/*.o lbfgspp_amd/liblbfgsx.so lbfgspp_amd/liblbfgsx_solver.so
echo built: $units
