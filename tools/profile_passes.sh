set -o pipefail
O=gpurun_out/rp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > $O/bench_under_rocprof.json 2> $O/stats.err &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > /dev/null 2> $O/pmc_fetch.err &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-context-check --no-stream-record > /dev/null 2> $O/pmc_write.err &&
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
rc=$?
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cut -c1-200 $O/bench_under_rocprof.json; exit $rc
