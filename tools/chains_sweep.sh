#!/bin/bash
# independent half / quarter-batch chains on parallel graph branches (DSC_CHAINS) on the small configurations, where
# every launch is latency-bound and fills <= 192 CUs with one wave per SIMD.  Sampling only (the captured training step has no chain form).
#   bash tools/chains_sweep.sh <tag>     -> gpurun_out/<tag>_chains_sweep.txt
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_chains_sweep.txt
: > $O
for c in text bedroom21 complete arrange living80; do
  for n in 1 2 4; do
    DSC_CHAINS=$n timeout 300 python $R/bench.py --config $c --mode sample --steps 40 --warmup 10 --no-cpu-baseline --no-full-loop 2>/dev/null | tail -1 | \
      python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-10s chains=%s  sampling %.3f ms per step  (%.1f steps/s)' % ('$c', '$n', d['ms_per_step'], d['value']))" >> $O 2>&1 || echo "$c chains=$n failed" >> $O
  done
done
cat $O
