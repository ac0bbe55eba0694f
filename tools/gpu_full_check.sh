#!/bin/bash
# GPU box: the whole GPU suite, then the bench workloads named in $WORKLOADS (default: the 32-qubit engine workload)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1
for w in ${WORKLOADS:-mps32_trotter2_engine}; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
