#!/bin/bash
# Kernel times + counters of ANY command on the GPU box:  bash tools/kprof.sh TAG [--traffic] -- <command ...>
#   -> gpurun_out/TAG_kernel_stats.txt, gpurun_out/TAG_pmc.txt.  Counter passes with --kernel-trace only, each under its own
#   timeout, run from /tmp (see NOTES.md).  --traffic adds the FETCH_SIZE / WRITE_SIZE passes.
set -u
TAG=$1; shift
TRAFFIC=0
if [ "$1" == "--traffic" ]; then TRAFFIC=1; shift; fi
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_k /tmp/kp_1 /tmp/kp_2 /tmp/kp_3
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp_k -o x -- "$@" > $OUT/${TAG}_cmd.log 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/kp_k -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats.txt
head -14 $OUT/${TAG}_kernel_stats.txt
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace -d /tmp/kp_1 -o x -- "$@" > /dev/null 2>&1
if [ $TRAFFIC == 1 ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/kp_2 -o x -- "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/kp_3 -o x -- "$@" > /dev/null 2>&1
fi
python $ROOT/tools/pmc_table.py $(find /tmp/kp_1 /tmp/kp_2 /tmp/kp_3 -name "*.db" 2>/dev/null) > $OUT/${TAG}_pmc.txt
cat $OUT/${TAG}_pmc.txt
