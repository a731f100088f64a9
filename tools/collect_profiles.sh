#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh TAG [MODE ...]   -> gpurun_out/TAG_<mode>_{kernel_stats.txt,pmc.txt}, gpurun_out/regress_traffic.json
# rocprofv3 is run from /tmp (its scratch files), counters in their own passes with --kernel-trace only (FETCH_SIZE
# and WRITE_SIZE do not fit one pass: 3 + 2 of 4 TCC slots), every pass under its own timeout (a counter pass once ate
# a whole round's GPU budget).  Copy what you want judged into profiles/.
set -u
TAG=${1:-r02}
shift || true
MODES=${@:-fp16x2w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f $ROOT/profiles/regress_traffic.json ] && [ ! -f $OUT/regress_traffic.json ] && cp $ROOT/profiles/regress_traffic.json $OUT/regress_traffic.json      # (a second call of one visit keeps the first one's records)
[ -f $ROOT/profiles/kernel_times.json ] && [ ! -f $OUT/kernel_times.json ] && cp $ROOT/profiles/kernel_times.json $OUT/kernel_times.json
# P2P_CONFIG=E profiles BASELINE configs[4] (960x1280, 2 pairs x 6400 proposals per step); files are tagged TAG_E_<mode>_*
CFG=${P2P_CONFIG:-A}
B="--no-cpu-baseline --no-parity --no-other-modes --no-e2e --no-other-configs --config $CFG"
[ "$CFG" != "A" ] && TAG=${TAG}_${CFG}
export P2P_CONFIG=$CFG
for MODE in $MODES; do
  rm -rf /tmp/prof_k /tmp/prof_p1 /tmp/prof_p2 /tmp/prof_p3
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o bench -- python $ROOT/bench.py --mode $MODE --steps 8 --warmup 2 $B > $OUT/${TAG}_${MODE}_profiled_bench.json 2>/dev/null
  python $ROOT/tools/prof_summary.py $(find /tmp/prof_k -name "*.db" | head -1) > $OUT/${TAG}_${MODE}_kernel_stats.txt
  head -8 $OUT/${TAG}_${MODE}_kernel_stats.txt
  timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_p1 -o bench -- python $ROOT/bench.py --mode $MODE --steps 3 --warmup 1 $B > /dev/null 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_p3 -o bench -- python $ROOT/bench.py --mode $MODE --steps 3 --warmup 1 $B > /dev/null 2>&1
  timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace -d /tmp/prof_p2 -o bench -- python $ROOT/bench.py --mode $MODE --steps 3 --warmup 1 $B > /dev/null 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/${TAG}_${MODE}_pmc.txt $OUT/regress_traffic.json $MODE $(find /tmp/prof_p1 /tmp/prof_p2 /tmp/prof_p3 -name "*.db")
  # per-kernel times + MFMA busy x clock of every p2p kernel of the step (bench.py: coarse_roofline), default mode only
  [ "$MODE" = "fp16x2w" ] && python $ROOT/tools/kernel_times.py $OUT/kernel_times.json $CFG "profiles/${TAG}_${MODE}_kernel_stats.txt + profiles/${TAG}_${MODE}_pmc.txt" $(find /tmp/prof_k -name "*.db" | head -1) $(find /tmp/prof_p1 /tmp/prof_p2 /tmp/prof_p3 -name "*.db")
done
cat $OUT/regress_traffic.json
