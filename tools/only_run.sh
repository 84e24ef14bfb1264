for k in 0 12 345 67; do
IE_ZDCE_ONLY=$k python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('only=$k', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
