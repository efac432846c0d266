#!/bin/bash
# Re-measures everything the profiles/ directory holds, on the GPU box, into gpurun_out/refresh/.  Every GPU step runs
# under its own time limit and the chain stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O && cd $R || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.txt 2>&1 &&
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err &&
timeout -k 10 200 python bench.py --workload stream --steps 300 --warmup 30 > $O/stream.json 2> $O/stream.err &&
timeout -k 10 300 python bench.py --workload stream --sweep 256,1024,4096,8192 --steps 300 --warmup 30 > $O/stream_sweep.json 2> $O/stream_sweep.err &&
timeout -k 10 200 python bench.py --workload longform --steps 3 --warmup 1 > $O/longform.json 2> $O/longform.err &&
timeout -k 10 300 python bench.py --workload mixed --utterances 2000 --steps 1 --warmup 1 > $O/mixed.json 2> $O/mixed.err &&
timeout -k 10 300 python tools/layer_profile.py > $O/layers.txt 2>&1 &&
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > $O/bench_under_rocprof.json 2> $O/stats.err &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > /dev/null 2> $O/pmc_fetch.err &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > /dev/null 2> $O/pmc_write.err &&
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 tools/one_forward.py --batch 64 > /dev/null 2> $O/pmc_mfma.err &&
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json &&
python tools/pmc_mfma_util.py $O/pmc_mfma > $O/pmc_mfma_util.txt
rc=$?
# keep the merged-back output small: per-dispatch counter tables are summarised above
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
echo "refresh rc=$rc"; tail -2 $O/pytest_gpu.txt; cat $O/bench.json | cut -c1-300
exit $rc
