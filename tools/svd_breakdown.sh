cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/svd
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_mps_engine.py -x -q 2>&1 | tail -5 || exit 1
rocprofv3 --kernel-trace --stats -d $O/prod -o p --output-format csv -- python3 tools/svd_probe.py 512 > $O/prod.log 2>&1 || true
export AQC_HIP_LIB=$PWD/aqc_research_amd/libaqc_hip_tuning.so
for d in 16 17 18 20 24; do
  AQC_SVD_DEBUG=$d rocprofv3 --kernel-trace --stats -d $O/d$d -o p --output-format csv -- python3 tools/svd_probe.py 512 > $O/d$d.log 2>&1 || true
done
for d in prod d16 d17 d18 d20 d24; do echo $d; grep -h "jacobi_block\|jacobi_round" $O/$d/p_kernel_stats.csv | cut -d, -f1-4; done
cat $O/prod.log | grep blocked
