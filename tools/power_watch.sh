# usage (GPU box): bash tools/power_watch.sh out_file -- command ...   samples rocm-smi (power, clocks) once a second while the command runs
out=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';' ; echo; sleep 1; done ) > $out 2>&1 &
watch=$!
"$@"
rc=$?
kill $watch
exit $rc
