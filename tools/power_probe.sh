#!/bin/bash
# Is the regress launch bound by the package power limit?  (run on the GPU box; needs tools/exp/lib_cap128.so, lib_cap64.so
# from `bash tools/ab_variants.sh cap128=-DXF_GRID_CAP=128 cap64=-DXF_GRID_CAP=64`)
#   1. the same launch on 256 / 128 / 64 compute units: with a fixed clock the time per proposal and CU would not change
#   2. all-zero convolution weights: the same instruction stream without operand toggling in the matrix cores
#   3. rocm-smi power / clock samples while the launch repeats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export NPROP=6400 NPAIRS=16 NITER=9
MODE=${MODE:-fp16x2w}          # fp16x2w: the whole fine stage of a step (conv1 launches + Winograd GEMMs + FC); fp16x2: round 4's one launch
echo "== 256 CUs"; timeout 200 python tools/regress_bench.py $MODE 2>&1 | grep median
[ -f $ROOT/tools/exp/lib_cap128.so ] && { echo "== 128 CUs"; P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH=$ROOT/tools/exp/lib_cap128.so timeout 200 python tools/regress_bench.py $MODE 2>&1 | grep median; }
[ -f $ROOT/tools/exp/lib_cap64.so ] && { echo "== 64 CUs";  P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH=$ROOT/tools/exp/lib_cap64.so timeout 200 python tools/regress_bench.py $MODE 2>&1 | grep median; }
echo "== 256 CUs, zero weights"; ZERO_W=1 timeout 200 python tools/regress_bench.py $MODE 2>&1 | grep median
echo "== 256 CUs, exact f32 kernel (for the clock comparison)"; NITER=3 timeout 200 python tools/regress_bench.py f32 2>&1 | grep median
echo "== rocm-smi while the $MODE fine stage repeats"
(NITER=3000 timeout 100 python tools/regress_bench.py $MODE > /dev/null 2>&1 &)
sleep 22
for i in 1 2 3 4 5 6; do timeout 10 rocm-smi --showpower --showclocks 2>/dev/null | grep -i -E "power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 2; done
sleep 22
echo "== rocm-smi idle"; timeout 10 rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i -E "power|sclk" | tr -s ' '
