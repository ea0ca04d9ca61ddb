#!/bin/bash
# usage: tools/bench_kernels.sh <workload> <outcsv> [lib]   -- rocprofv3 kernel stats of a short bench run (optionally with an alternative library build)
wl=$1; out=$PWD/$2; lib=$3
R=$PWD
export TMPDIR=/tmp
[ -n "$lib" ] && export GPE_HIP_LIB=$R/$lib
d=$(mktemp -d /tmp/ks.XXXX)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $out.json 2> $out.err
cp $d/*/*kernel_stats.csv $out; rm -rf $d
python3 $R/tools/kstats.py $out | head -6
