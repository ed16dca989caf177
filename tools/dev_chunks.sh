for c in 2 4 8 16 32 64; do python bench.py --no-cpu --steps 10 --warmup 2 --chunk $c | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('chunk', d['config']['chunk'], 'fps %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in r['kernel_ms_per_step'].items()})
"; done
