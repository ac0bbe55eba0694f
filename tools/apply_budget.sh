# usage (GPU box): bash tools/apply_budget.sh [lanes] -- times the headline V^H launch pair with parts of the sub-stage loop compiled out
# (variant libraries libaqc_hip_skip<bits>.so: make variant NAME=skip<bits> EXTRA=-DAQC_EXP_APPLY_SKIP=<bits>)
lanes=${1:-1024}
for m in base 8 16 32 1 2 3 24 48 56 27 59 base; do
  if [ "$m" = base ]; then lib=$PWD/aqc_research_amd/libaqc_hip.so; else lib=$PWD/aqc_research_amd/libaqc_hip_skip$m.so; fi
  AQC_HIP_LIB=$lib timeout -k 10 120 python tools/apply_budget.py $lanes $m || exit 1
done
