#!/bin/bash
out=gpurun_out/r4x; mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_ns.json 2> $out/bench_ns.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4x/bench_ns.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity_check',{}).get('ok'), d.get('parity_check',{}).get('points'))
P
