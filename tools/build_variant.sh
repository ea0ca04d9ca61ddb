#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]  ->  build/variants/libgpe_<name>.so
# Kernel-tuning builds: -DGPE_FAST_BUILD compiles only the H = 64, n_out = 1, C in {1, 5} kernels (the NS workload), ~20 s.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $root/build/variants
cd $root/gross-pitaevskii-eigenvalue-problem_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DGPE_FAST_BUILD "$@" -I ../../include \
    -o $root/build/variants/libgpe_$name.so gpe_engine.hip 2>&1 | grep -E "error|ScratchSize" || true
ls -la $root/build/variants/libgpe_$name.so
