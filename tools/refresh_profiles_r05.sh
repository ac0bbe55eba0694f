# Round-5 evidence run (GPU box): GPU suite, default bench (1024 lanes), 64 lanes, kernel trace of the bench command, PMC traffic and
# SQ counters of the headline's launches (sparse route and dense route), the config-4 driver with its kernel trace, lockstep MPS lanes.
# Usage: bash tools/refresh_profiles_r05.sh [part ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ev
mkdir -p $O
parts="${@:-tests bench trace pmc sq cfg4}"
for part in $parts; do case $part in
tests)
  python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || true
  tail -3 $O/gpu_tests.log ;;
bench)
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err     # the driver's command: headline + every config
  python bench.py --batch 64 --no-configs --no-cpu-baseline --no-objective-object > $O/bench_b64.json 2> $O/bench_b64.err
  AQC_SPARSE_SWEEP=0 python bench.py --no-configs --no-cpu-baseline --no-objective-object --no-latency > $O/bench_dense_route.json 2> $O/bench_dense_route.err
  AQC_PROJECTED=0 python bench.py --no-configs --no-cpu-baseline --no-objective-object --no-latency > $O/bench_full_size_stages.json 2> $O/bench_full_size_stages.err
  AQC_PROJECTED_VDAG=0 python bench.py --no-configs --no-cpu-baseline --no-objective-object --no-latency > $O/bench_vdag_by_stages.json 2> $O/bench_vdag_by_stages.err
  echo "bench done" ;;
trace)
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-configs --no-cpu-baseline --no-latency --no-objective-object --sustain-seconds 0 > $O/bench_under_rocprof.json 2> $O/kt.err
  echo "kernel trace done" ;;
pmc)
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 tools/prof_run5.py > $O/pmc_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 tools/prof_run5.py > $O/pmc_w.log 2>&1
  echo "pmc traffic done" ;;
sq)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/sq1 -o s --output-format csv -- python3 tools/prof_run5.py > $O/sq1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/sq2 -o s --output-format csv -- python3 tools/prof_run5.py > $O/sq2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/sq3 -o s --output-format csv -- python3 tools/prof_run5.py > $O/sq3.log 2>&1 || echo "sq3 counters not available"
  echo "sq done" ;;
cfg4)
  timeout -k 10 500 python bench.py --workload cfg4_driver > $O/bench_cfg4_driver.json 2> $O/bench_cfg4_driver.err || echo "cfg4_driver failed"
  timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $O/kt_cfg4 -o kt --output-format csv -- python3 bench.py --workload cfg4_driver > $O/bench_cfg4_under_rocprof.json 2> $O/kt_cfg4.err || echo "cfg4 trace failed"
  echo "cfg4 done" ;;
esac; done
