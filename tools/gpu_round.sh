#!/bin/bash
# One GPU-box visit: tests, bench, accuracy of the arithmetic modes, profiles.  bash tools/gpu_round.sh TAG [what...]
# what: tests bench check prof (default: all)
set -u
TAG=${1:-r02}
shift || true
WHAT=${@:-tests bench check prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/${TAG}_pytest.log 2>&1; tail -5 $OUT/${TAG}_pytest.log;;
    bench) timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 3000 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err;;
    benchE) timeout 900 python bench.py --config E --steps 6 --warmup 2 > $OUT/${TAG}_benchE.json 2> $OUT/${TAG}_benchE.err; tail -c 2500 $OUT/${TAG}_benchE.json; tail -3 $OUT/${TAG}_benchE.err;;
    check) timeout 600 python tools/split_check.py > $OUT/${TAG}_split_check.txt 2>&1; cat $OUT/${TAG}_split_check.txt;;
    prof) bash tools/collect_profiles.sh $TAG fp16x2w f32;;
    prof3) bash tools/collect_profiles.sh $TAG fp16x2;;
  esac
done
