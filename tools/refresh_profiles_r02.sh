# Round-2 evidence run (GPU box): full GPU suite, default bench, kernel trace, PMC traffic, SQ counters, all workloads.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || true
tail -3 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --batch 64 --no-cpu-baseline > $O/bench_b64.json 2> $O/bench_b64.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-latency > $O/bench_under_rocprof.json 2> $O/kt.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 tools/prof_run3.py > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 tools/prof_run3.py > $O/pmc_w.log 2>&1
echo "pmc traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/sq1 -o s --output-format csv -- python3 tools/prof_run3.py > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES -d $O/sq2 -o s --output-format csv -- python3 tools/prof_run3.py > $O/sq2.log 2>&1
echo "sq done"
for w in sv12_trotter2 sv20_l40 sv20_trotter2 mat10_l40 mat5_cyc180 mps16_l40_chi16 mps16_l40_chi64 mps16_l40_chi256; do
  python bench.py --workload $w --steps 40 --warmup 10 > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w done"
done
timeout -k 10 400 python bench.py --workload cfg4_jobs --steps 2 --warmup 1 > $O/bench_cfg4_jobs.json 2> $O/bench_cfg4_jobs.err || echo "cfg4_jobs failed"
echo "bench cfg4_jobs done"
