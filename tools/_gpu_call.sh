export TMPDIR=/tmp
python bench.py --steps 10 --no-configs --no-cpu-baseline --no-feed-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(d[k],2) for k in ['ms_per_step','beam5_ms_per_batch','greedy_ms_per_batch','beam5_batch1_ms_per_sentence']})"
