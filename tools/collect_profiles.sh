#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh TAG      -> gpurun_out/TAG_{bench.json,kernel_stats.txt,pmc.txt,traffic.json}
# rocprofv3 is run from /tmp (its scratch files), counters in their own passes with --kernel-trace only, every pass
# under its own timeout (a counter pass once ate a whole round's GPU budget).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o bench -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/prof_k -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.txt
head -12 $OUT/${TAG}_kernel_stats.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/prof_p1 -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace -d /tmp/prof_p2 -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc.txt $OUT/${TAG}_traffic.json $(find /tmp/prof_p1 /tmp/prof_p2 -name "*.db")
cat $OUT/${TAG}_traffic.json
