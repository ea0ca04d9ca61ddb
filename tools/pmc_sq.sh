#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_sq.sh <workload> <outdir> [points_per_gpu]
# SQ / LDS / wait counters of the bench command, in their own rocprofv3 passes (8 SQ slots per pass; never mixed with trace
# domains other than --kernel-trace), plus the FETCH_SIZE / WRITE_SIZE passes.  Summaries: <outdir>/mfma_util_<workload>.json,
# <outdir>/traffic_<workload>.json.  Copy what is to be judged into profiles/<round>/.
wl=$1
out=$PWD/$2
pts=${3:-1048576}
R=$PWD
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --blocks 1 --no-parity-check --no-cpu-baseline --no-alt-mode"
pass() {   # name, counters...
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$name -- $CMD > /dev/null 2> $out/pmc_$name.err || { echo "pass $name failed"; tail -5 $out/pmc_$name.err; return 1; }
    echo "pass $name done"
}
pass a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE &&
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES &&
pass c SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE
cd $R
python3 tools/pmc_sq_summary.py $out/mfma_util_$wl.json "$CMD" $out/pmc_a/*/*counter_collection.csv $out/pmc_b/*/*counter_collection.csv $out/pmc_c/*/*counter_collection.csv
python3 tools/pmc_traffic.py $out/pmc_fetch/*/*counter_collection.csv $out/pmc_write/*/*counter_collection.csv $pts $out/traffic_$wl.json \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-alt-mode"
rm -rf $out/pmc_a $out/pmc_b $out/pmc_c $out/pmc_fetch $out/pmc_write
