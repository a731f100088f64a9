#!/bin/bash
# The records of a round in two GPU-box visits:  bash tools/final_round.sh TAG part1|part2
#   all: everything below in one visit, profiles first so that the bench line finds the traffic record of its own sources
#   part1: the -m gpu suite, the default bench line (A + other_configs.E + e2e + cpu_baseline)
#   part2: rocprofv3 kernel stats + PMC passes at A and E (regress_traffic.json for the final sources), the pair stream,
#          the pyramid producer (bench, tile sweep, kernel stats)
set -u
TAG=${1:-r03}
PART=${2:-part1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "$PART" = all ]; then
  bash tools/collect_profiles.sh $TAG fp16x2w 2>&1 | tail -6
  P2P_CONFIG=E bash tools/collect_profiles.sh $TAG fp16x2w 2>&1 | tail -6
  cp $OUT/regress_traffic.json $ROOT/profiles/regress_traffic.json      # (on the box: the bench line below then carries the traffic)
  cp $OUT/kernel_times.json $ROOT/profiles/kernel_times.json
  cd $ROOT; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
  for B in 1 16; do BATCH=$B timeout 120 python tools/coarse_bench.py 2>&1 | tail -1; done > $OUT/${TAG}_coarse_bench.txt; H=960 W=1280 BATCH=2 REPS=10 timeout 120 python tools/coarse_bench.py 2>&1 | tail -1 >> $OUT/${TAG}_coarse_bench.txt; cat $OUT/${TAG}_coarse_bench.txt
  cd $ROOT
  timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 1200 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
  timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/${TAG}_pytest.log 2>&1; tail -4 $OUT/${TAG}_pytest.log
  timeout 300 python bench.py --pairs 2048 --no-e2e > $OUT/${TAG}_stream.json 2> $OUT/${TAG}_stream.err; tail -c 400 $OUT/${TAG}_stream.json
  timeout 300 python tools/backbone_bench.py > $OUT/${TAG}_backbone_bench.txt 2>&1; grep -v amdgpu.ids $OUT/${TAG}_backbone_bench.txt | tail -16
  timeout 300 python tools/conv_sweep.py > $OUT/${TAG}_conv_sweep.txt 2>&1; grep "x16" $OUT/${TAG}_conv_sweep.txt | cut -c1-200
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_bb
  NB=16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bb -o bb -- python $ROOT/tools/backbone_once.py > /tmp/bb.log 2>&1
  python $ROOT/tools/prof_summary.py $(find /tmp/prof_bb -name "*.db" | head -1) > $OUT/${TAG}_backbone_kernel_stats.txt; head -16 $OUT/${TAG}_backbone_kernel_stats.txt | cut -c1-160
elif [ "$PART" = part1 ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/${TAG}_pytest.log 2>&1; tail -4 $OUT/${TAG}_pytest.log
  timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 1500 $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
else
  bash tools/collect_profiles.sh $TAG fp16x2w 2>&1 | tail -12
  P2P_CONFIG=E bash tools/collect_profiles.sh $TAG fp16x2w 2>&1 | tail -12
  timeout 300 python bench.py --pairs 2048 --no-e2e > $OUT/${TAG}_stream.json 2> $OUT/${TAG}_stream.err; tail -c 600 $OUT/${TAG}_stream.json
  timeout 300 python tools/backbone_bench.py > $OUT/${TAG}_backbone_bench.txt 2>&1; grep -v amdgpu.ids $OUT/${TAG}_backbone_bench.txt | tail -16
  timeout 300 python tools/conv_sweep.py > $OUT/${TAG}_conv_sweep.txt 2>&1; grep "x16" $OUT/${TAG}_conv_sweep.txt | cut -c1-200
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_bb
  NB=16 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bb -o bb -- python $ROOT/tools/backbone_once.py > /tmp/bb.log 2>&1
  python $ROOT/tools/prof_summary.py $(find /tmp/prof_bb -name "*.db" | head -1) > $OUT/${TAG}_backbone_kernel_stats.txt; head -16 $OUT/${TAG}_backbone_kernel_stats.txt | cut -c1-160
fi
