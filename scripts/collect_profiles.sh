#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/final/).
set -o pipefail
R=$(pwd); O=$R/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_fetch.out 2> $O/pmc_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $O/pmc_write.out 2> $O/pmc_write.err || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/drn_stats -- python3 $R/scripts/prof_deepresnet.py 1024 bf16 > $O/drn_stats.out 2> $O/drn_stats.err || exit 5
cd $R
python3 scripts/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv") $(find $O/pmc_write -name "*counter_collection.csv") 16384 $O/pmc_embed.json > $O/pmc_summary.out 2>&1 || exit 6
{ echo "== scripts/bench_deepresnet.py"; BATCHES=16,256,1024 timeout -k 10 500 python3 scripts/bench_deepresnet.py; echo "== scripts/bench_configs.py"; timeout -k 10 400 python3 scripts/bench_configs.py; echo "== scripts/bench_midbatch.py"; timeout -k 10 300 python3 scripts/bench_midbatch.py; } > $O/other_configs.txt 2>&1 || exit 7
echo collected
