#!/bin/bash
# A/B of kernel variants selected by -D flags (experiment builds: -DP2P_EXPERIMENT is added, the library then identifies
# itself as one and is only loaded under P2P_ALLOW_EXPERIMENT=1).  Builds tools/exp/lib_<name>.so HERE (no GPU needed):
#   bash tools/ab_variants.sh name=-DFLAG1,-DFLAG2 ...      (name "cur" with no flags = the working tree as it is)
# then on the GPU box:
#   gpurun -- 'bash tools/ab_variants.sh --run <mode> name1 name2 ...'   times them with tools/wino_ab.py (NPAIRS x 400 proposals).
# Only the files in FILES (default: the regressor sources) are recompiled with the flags; the others come from the main build.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp"
FILES=${FILES:-"api regress regress_h2 regress_wino"}
if [ "${1:-}" = "--run" ]; then
    mode=$2; shift 2
    for name in "$@"; do
        echo "== $name"
        P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH="$ROOT/tools/exp/lib_$name.so" ORACLE=0 REPS=${REPS:-8} timeout 300 python "$ROOT/tools/wino_ab.py" ${NPAIRS:-16} 400 480 640 $mode 2>&1 | grep "round 1" || true
    done
    exit 0
fi
python -m patch2pix_amd.build > /dev/null
for spec in "$@"; do
    name=${spec%%=*}; flags=""
    [ "$spec" != "$name" ] && flags="-DP2P_EXPERIMENT $(echo "${spec#*=}" | tr ',' ' ')"
    tmp=$(mktemp -d)
    objs=""
    for f in api backbone coarse consensus filter regress regress_h2 regress_wino; do
        if [ -n "$flags" ] && echo " $FILES " | grep -q " $f "; then
            /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$ROOT/patch2pix_amd/csrc/$f.hip" -o "$tmp/$f.o" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|error" | grep -B3 -A0 "ScratchSize \[bytes/lane\]: [1-9]\|error" || true
            objs="$objs $tmp/$f.o"
        else
            objs="$objs $ROOT/patch2pix_amd/csrc/$f.o"
        fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/lib_$name.so" $objs
    rm -rf "$tmp"
    echo "built lib_$name.so ($flags)"
done
