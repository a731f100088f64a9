#!/bin/bash
# A/B of regress_h2 kernel variants selected by -D flags: builds tools/exp/lib_<name>.so HERE (no GPU needed), then
#   gpurun -- 'bash tools/ab_variants.sh --run <mode> name1 name2 ...'   times them with tools/regress_bench.py.
#   bash tools/ab_variants.sh name=-DFLAG1,-DFLAG2 ...      (name "cur" with no flags = the working tree as it is)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp"
if [ "${1:-}" = "--run" ]; then
    mode=$2; shift 2
    for name in "$@"; do
        echo "== $name"
        P2P_LIB_PATH="$ROOT/tools/exp/lib_$name.so" NPROP=${NPROP:-6400} timeout 200 python "$ROOT/tools/regress_bench.py" $mode 2>&1 | grep median || true
    done
    exit 0
fi
for spec in "$@"; do
    name=${spec%%=*}; flags=""
    [ "$spec" != "$name" ] && flags=$(echo "${spec#*=}" | tr ',' ' ')
    tmp=$(mktemp -d)
    objs=""
    for f in api backbone coarse consensus filter regress regress_h2; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$ROOT/patch2pix_amd/csrc/$f.hip" -o "$tmp/$f.o" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A7 "regress_h2_kernel" | grep -E "VGPRs:|ScratchSize" || true
        objs="$objs $tmp/$f.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/lib_$name.so" $objs
    rm -rf "$tmp"
    echo "built lib_$name.so ($flags)"
done
