#!/bin/bash
# Builds tools/bin/libdot_probe{0..7}.so (tools/dot_probe.hip with GKOMI_DOT_PROBE = k).  Runs here: hipcc cross-compiles.
set -e
cd "$(dirname "$0")"
mkdir -p bin
for k in 0 1 2 3 4 5 6 7; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -DGKOMI_DOT_PROBE=$k \
        dot_probe.hip -o bin/libdot_probe$k.so -L../repo-8852-ginkgo_amd/lib -lgkomi '-Wl,-rpath,$ORIGIN/../../repo-8852-ginkgo_amd/lib' &
    if (( k % 4 == 3 )); then wait; fi
done
wait
ls -la bin/libdot_probe*.so
