#!/bin/bash
# GPU box: SQ counters of the lockstep-lane kernels (32-qubit workload, 1024 lanes), two rocprofv3 --pmc passes -> gpurun_out/lockstep_sq{1,2}
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/lockstep_sq1 -o s --output-format csv -- python3 tools/mps_lockstep_profile.py 1024 1 > gpurun_out/lockstep_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/lockstep_sq2 -o s --output-format csv -- python3 tools/mps_lockstep_profile.py 1024 1 > gpurun_out/lockstep_sq2.log 2>&1
