#!/bin/bash
# Sweep consensus-layer-2 tilings (P2P_NC2_TILE = tb,tc,tdr,ta,nthreads) for a batch size: bash tools/nc2_sweep.sh [BATCH]
export BATCH=${1:-1} REPS=30
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "default:"; python $ROOT/tools/coarse_bench.py 2>/dev/null | tail -1
for t in 4,6,5 5,10,5 3,5,5 2,8,5 4,8,5 6,6,5 3,10,5 2,4,5 4,4,5 5,5,5; do
  for ta in 2 3 5 6 10 15 30; do
    for nt in 128 256; do
      IFS=, read tb tc tdr <<< "$t"
      [ $((tb*tc*tdr)) -gt $nt ] && continue
      r=$(P2P_NC2_TILE=$t,$ta,$nt python $ROOT/tools/coarse_bench.py 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) us.*/\1/')
      echo "$t,$ta,$nt $r"
    done
  done
done | sort -k2 -n | head -12
