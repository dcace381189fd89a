for v in 0 1 2 3 4 5; do VOXCARVE_LUT_VARIANT=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $v', 'kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'surv', d['config']['survivors'])
"; done
