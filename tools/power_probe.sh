#!/bin/bash
# Sample board power and shader clock (rocm-smi) while the regress launch runs back to back.
#   gpurun -- 'bash tools/power_probe.sh fp16x2 [lib.so]'   -> gpurun_out/power_<mode>.txt
MODE=${1:-fp16x2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
[ -n "${2:-}" ] && export P2P_LIB_PATH=$2
NPROP=6400 NITER=500 timeout 120 python $ROOT/tools/regress_bench.py $MODE > $OUT/power_${MODE}_bench.txt 2>&1 &
PID=$!
sleep 12
: > $OUT/power_${MODE}.txt
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';' >> $OUT/power_${MODE}.txt
  echo >> $OUT/power_${MODE}.txt
  sleep 0.25
done
tail -4 $OUT/power_${MODE}.txt; cat $OUT/power_${MODE}_bench.txt
