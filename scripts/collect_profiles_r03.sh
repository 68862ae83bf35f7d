#!/bin/bash
# Round-3 evidence committed under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/final3/).
set -o pipefail
R=$(pwd); O=$R/gpurun_out/final3; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_fetch.out 2> $O/pmc_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_write.out 2> $O/pmc_write.err || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_sq.out 2> $O/pmc_sq.err || exit 5
cd $R
python3 scripts/pmc_summary_r02.py $(find $O/pmc_fetch -name "*counter_collection.csv") $(find $O/pmc_write -name "*counter_collection.csv") 16384 4 $O/pmc.json > $O/pmc_summary.out 2>&1 || exit 6
python3 scripts/pmc_sq_summary.py $(find $O/pmc_sq -name "*counter_collection.csv") 14 > $O/pmc_sq.txt 2>&1 || exit 7
{ echo "== scripts/bench_configs.py (in-library event timing on: hipGraph replay off)"; timeout -k 10 500 python3 scripts/bench_configs.py; echo "== scripts/bench_c3.py (no profiler)"; timeout -k 10 120 python3 scripts/bench_c3.py 4096 30; timeout -k 10 120 python3 scripts/bench_c3.py 4096 60; echo "== scripts/bench_deepresnet.py"; BATCHES=1024 timeout -k 10 300 python3 scripts/bench_deepresnet.py; echo "== scripts/bench_fused.py"; timeout -k 10 200 python3 scripts/bench_fused.py; echo "== scripts/phase_timing.py"; timeout -k 10 300 python3 scripts/phase_timing.py 2>&1 | grep phases; } > $O/other_configs.txt 2>&1 || exit 8
echo collected
