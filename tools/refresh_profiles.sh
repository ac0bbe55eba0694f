set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency > $O/bench_under_rocprof.json 2> $O/kt.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 tools/prof_run.py > $O/pmc_f.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 tools/prof_run.py > $O/pmc_w.log 2>&1
echo "pmc write done"
for w in sv12_trotter2 sv20_l40 sv20_trotter2 mat10_l40 mat5_cyc180; do
  python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w done"
done
find $O -name "*.csv" | head -20
