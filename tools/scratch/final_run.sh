#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/suite.txt 2>&1; grep -E "passed|failed" $OUT/suite.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
bash tools/round_evidence.sh r05 > $OUT/evidence.log 2>&1; tail -3 $OUT/evidence.log
