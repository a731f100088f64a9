#!/bin/bash
# A/B of bench.py with the coarse stage on a second stream (--overlap 1) and on the same stream (--overlap 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for o in 0 1 0 1; do
  python $ROOT/bench.py --overlap $o --no-cpu-baseline --no-parity --no-other-modes --no-e2e 2>/dev/null | tail -1 > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('overlap', $o, round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step, regress launch', round(d['roofline']['avg_launch_ms'],2), 'ms')"
done
