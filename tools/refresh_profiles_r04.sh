# Round-4 evidence run (GPU box): GPU suite, default bench (1024 lanes), 256 and 64 lanes, kernel trace, PMC traffic, SQ counters,
# the full records of the round's new workloads (the other configs ride in the default line's `configs`), the config-4 job mix with its own kernel trace.  Usage: bash tools/refresh_profiles_r04.sh [part ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ev
mkdir -p $O
parts="${@:-tests bench trace pmc sq workloads cfg4 lockstep}"
for part in $parts; do case $part in
tests)
  python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || true
  tail -3 $O/gpu_tests.log ;;
bench)
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err     # the driver's command: headline + every config
  python bench.py --batch 64 --no-configs --no-cpu-baseline --no-objective-object > $O/bench_b64.json 2> $O/bench_b64.err
  echo "bench done" ;;
trace)
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-configs --no-cpu-baseline --no-latency --no-objective-object --sustain-seconds 0 > $O/bench_under_rocprof.json 2> $O/kt.err
  echo "kernel trace done" ;;
pmc)
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 tools/prof_run3.py > $O/pmc_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 tools/prof_run3.py > $O/pmc_w.log 2>&1
  echo "pmc traffic done" ;;
sq)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/sq1 -o s --output-format csv -- python3 tools/prof_run3.py > $O/sq1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/sq2 -o s --output-format csv -- python3 tools/prof_run3.py > $O/sq2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $O/sq3 -o s --output-format csv -- python3 tools/prof_run3.py > $O/sq3.log 2>&1 || echo "sq3 counters not available"
  echo "sq done" ;;
workloads)
  for w in cd5_cyc180 mps32_trotter2_engine mps32_trotter2_opt mat10_l40_k16; do
    python bench.py --workload $w --steps 40 --warmup 10 --sustain-seconds 0 > $O/bench_$w.json 2> $O/bench_$w.err
    echo "bench $w done"
  done ;;
cfg4)
  timeout -k 10 400 python bench.py --workload cfg4_jobs --steps 2 --warmup 1 > $O/bench_cfg4_jobs.json 2> $O/bench_cfg4_jobs.err || echo "cfg4_jobs failed"
  rocprofv3 --kernel-trace --stats -d $O/kt_cfg4 -o kt --output-format csv -- python3 bench.py --workload cfg4_jobs --steps 1 --warmup 1 > $O/bench_cfg4_under_rocprof.json 2> $O/kt_cfg4.err || echo "cfg4 trace failed"
  echo "cfg4 done" ;;
lockstep)
  rm -f $O/lockstep_lanes.log
  for lanes in 1 16 256 1024 4096; do
    timeout -k 10 300 python tools/mps_lockstep_profile.py $lanes 2 >> $O/lockstep_lanes.log 2>&1
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_lockstep -o kt --output-format csv -- python3 tools/mps_lockstep_profile.py 256 3 > $O/lockstep_under_rocprof.log 2>&1
  if [ -f aqc_research_amd/libaqc_hip_tuning.so ]; then
    AQC_HIP_LIB=$GRAFT_REPO_ROOT/aqc_research_amd/libaqc_hip_tuning.so timeout -k 10 300 python tools/mps_lockstep_profile.py 256 3 > $O/lockstep_stamps.log 2>&1 || echo "stamps failed"
  fi
  echo "lockstep done" ;;
esac; done
