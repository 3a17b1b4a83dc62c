for b in 1 2 4 8 16; do
  python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch %2d: %8.1f pairs/s  %7.3f ms/step  mfma_util %.3f' % ($b, d['value'], d['ms_per_step'], d.get('mfma_util_whole_forward', 0)))"
done
