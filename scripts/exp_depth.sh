for d in 1 2; do python bench.py --steps 20 --warmup 3 --depth $d --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('depth $d value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'others', {k: (v['ms_per_step'], v['kernel_ms']) for k, v in d['other_modes'].items()})
"; done
