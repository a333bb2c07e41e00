#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
for pair in 1 0; do
 for mode in cold warm; do
  rm -rf /tmp/ap_$pair$mode
  NM_ATTN_PAIR=$pair NM_B=128 NM_S=50 NM_A=1024 NM_C=1024 rocprofv3 --kernel-trace --output-format csv -d /tmp/ap_$pair$mode -- python $ROOT/tools/attn_only.py 1 23 $mode > /dev/null 2>&1
  python $ROOT/tools/pmc_summary.py --out /tmp/ap_$pair$mode.json --match attn_ --trace /tmp/ap_$pair$mode --algorithmic-bytes 53528576 --note x > /dev/null
  python -c "
import json; d=json.load(open('/tmp/ap_$pair$mode.json')); print('pair=$pair $mode', {k: round(v.get('avg_us',0),2) for k,v in d['kernels'].items()}, 'frac', round(d['frac_of_8TBps'],3))"
 done
done
cd $ROOT && timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2
