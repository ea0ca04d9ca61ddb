#!/bin/bash
# usage (GPU box): tools/head_sweep.sh <outdir>  -- k_head_pde / k_seed_pde time of the NS workload against workgroups per CU (GPE_HEAD_WG_PER_CU)
out=$1; mkdir -p $out; R=$PWD; export TMPDIR=/tmp
for g in 2 4 8 16; do
  d=$(mktemp -d /tmp/hs.XXXX)
  export GPE_HEAD_WG_PER_CU=$g
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 3 --blocks 2 --no-parity-check --no-cpu-baseline --no-alt-mode > /dev/null 2> $R/$out/err_$g.txt)
  echo "wg_per_cu=$g" >> $R/$out/head_sweep.txt
  python3 - $d >> $R/$out/head_sweep.txt <<PY
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_head_pde", "k_seed_pde", "k_grad_reduce", "k_update", "f_backward", "f_forward")):
            print("  %-28s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $d
done
cat $R/$out/head_sweep.txt
