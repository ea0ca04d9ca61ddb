#!/bin/bash
# usage (GPU box): tools/head_sweep.sh <outdir> [workload]  -- k_head_pde / k_seed_pde time (rocprofv3 kernel stats of a short bench run with the
# head in its own kernel, GPE_FUSE_HEAD=0) against threads per workgroup x workgroups per CU (GPE_HEAD_THREADS, GPE_HEAD_WG_PER_CU)
out=$1; wl=${2:-ns_2d_4x64}; mkdir -p $out; R=$PWD; export TMPDIR=/tmp
for spec in 256:2 512:2 1024:1 1024:2; do
  d=$(mktemp -d /tmp/hs.XXXX)
  export GPE_HEAD_THREADS=${spec%%:*} GPE_HEAD_WG_PER_CU=${spec##*:} GPE_FUSE_HEAD=0
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --blocks 2 --no-parity-check --no-cpu-baseline --no-alt-mode > /dev/null 2> $R/$out/err_$spec.txt)
  echo "$wl threads:wg_per_cu=$spec" >> $R/$out/head_sweep.txt
  python3 - $d >> $R/$out/head_sweep.txt <<PY
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_head_pde", "k_seed_pde")):
            print("  %-28s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $d
done
cat $R/$out/head_sweep.txt
