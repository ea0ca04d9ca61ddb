#!/bin/bash
# build/variants/libgpe_fpoff.so: the whole library with -ffp-contract=off (hipcc's default is fast-honor-pragmas) -- a rounding-level
# variant of every kernel, for the tests that must not depend on ulp-level arithmetic
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/variants
cd $root/gross-pitaevskii-eigenvalue-problem_amd/csrc
for u in gpe_engine gpe_wide; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -c $u.hip -o $root/build/variants/fpoff_$u.o 2>&1 | grep -E "error" || true &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/variants/libgpe_fpoff.so $root/build/variants/fpoff_gpe_engine.o $root/build/variants/fpoff_gpe_wide.o
ls -la $root/build/variants/libgpe_fpoff.so
