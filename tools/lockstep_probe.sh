#!/bin/bash
# GPU box: the MPS test files, the lockstep lanes of the 32-qubit engine workload at 256 / 1024 / 4096 lanes, a kernel trace at 256 lanes
# and (when the tuning library was built: make -C aqc_research_amd/csrc tuning) the in-kernel stamps of lanes_gate2_kernel.
#   gpurun -- 'bash tools/lockstep_probe.sh'    -> gpurun_out/lockstep_*.log, gpurun_out/prof_lockstep/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_hip_round4.py tests/test_hip_mps_engine.py tests/test_hip_parity_mps.py -x -q > gpurun_out/lockstep_tests.log 2>&1
rm -f gpurun_out/lockstep_lanes.log
for lanes in 256 1024 4096; do
  timeout -k 10 300 python tools/mps_lockstep_profile.py $lanes 2 >> gpurun_out/lockstep_lanes.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lockstep -- python3 tools/mps_lockstep_profile.py 256 3 > gpurun_out/lockstep_prof.log 2>&1
if [ -f aqc_research_amd/libaqc_hip_tuning.so ]; then
  AQC_HIP_LIB=$GRAFT_REPO_ROOT/aqc_research_amd/libaqc_hip_tuning.so timeout -k 10 300 python tools/mps_lockstep_profile.py 256 3 > gpurun_out/lockstep_stamps.log 2>&1
fi
