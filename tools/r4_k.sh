#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
line() { python - "$@" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
try:
    a=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-52s %.4f ms/step  fwd %.4f (%.3f)  bwd %.4f (%.3f)  %s"%(tag,a["ms_per_step"],a["roofline_forward"]["avg_launch_ms"],a["roofline_forward"]["frac"],a["roofline"]["avg_launch_ms"],a["roofline"]["frac"],a["roofline_forward"]["kernel"]))
except Exception as ex: print(tag,"ERR",ex)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-parity-check --blocks 15"
export GPE_HIP_LIB=$PWD/build/variants/libgpe_fastbase.so
for rep in 1 2; do
python bench.py $B > $out/ns_default.json 2>/dev/null; line "NS default (2 WG/CU, head in forward)" $out/ns_default.json
GPE_FUSE_HEAD=0 python bench.py $B > $out/ns_nohead.json 2>/dev/null; line "NS FUSE_HEAD=0 (2 WG/CU)" $out/ns_nohead.json
GPE_FUSE_HEAD=0 GPE_FWD_WG_PER_CU=3 python bench.py $B > $out/ns_nohead_3wg.json 2>/dev/null; line "NS FUSE_HEAD=0, 3 WG/CU forward" $out/ns_nohead_3wg.json
done
unset GPE_HIP_LIB
bash tools/profile_round4.sh gpurun_out/r4k/prof 2
