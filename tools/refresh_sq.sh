set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01c
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq1 -o s --output-format csv -- python3 tools/prof_run.py > $O/sq1.log 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES -d $O/sq2 -o s --output-format csv -- python3 tools/prof_run.py > $O/sq2.log 2>&1
echo "sq2 done"
python tools/tune.py skip > $O/tune_skip.txt 2>&1
python tools/tune.py nodots > $O/tune_nodots.txt 2>&1
echo "skip done"
