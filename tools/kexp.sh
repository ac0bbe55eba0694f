for lb in 3 4 5 2; do
  AQC_LOW_BITS=$lb python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-latency --no-objective-object --sustain-seconds 0 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('low bits $lb:', o['config']['tile_bits'], o['config']['launches_per_eval_step'], o['roofline']['substages'], 'value', round(o['value']), 'kernel_ms', {k:round(v,3) for k,v in o['kernel_ms_per_step'].items()})"
done
