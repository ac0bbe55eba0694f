# usage (GPU box): bash tools/variants.sh out_file name1 name2 ...   (names of aqc_research_amd/libaqc_hip_<name>.so; "base" = shipped)
out=$1; shift
mkdir -p $(dirname $out); : > $out
for v in "$@"; do
  if [ "$v" = base ]; then lib=$PWD/aqc_research_amd/libaqc_hip.so; else lib=$PWD/aqc_research_amd/libaqc_hip_$v.so; fi
  AQC_HIP_LIB=$lib python tools/variant_time.py 256 64 >> $out 2>&1 || exit 1
done
