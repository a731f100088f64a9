#!/bin/bash
# A/B of kernel variants that live on branches (see NOTES.md): builds libp2p_hip.so of every given ref into
# tools/exp/lib_<ref>.so HERE (no GPU needed), then `gpurun -- bash tools/ab_branches.sh --run <refs>` times them
# with tools/regress_bench.py / tools/coarse_bench.py on the GPU box through P2P_LIB_PATH.
#   bash tools/ab_branches.sh main exp/l3-dedupe exp/l23-dedupe           # build step (CPU container)
#   gpurun --timeout 600 -- 'bash tools/ab_branches.sh --run main exp/l3-dedupe exp/l23-dedupe'
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp"
if [ "${1:-}" = "--run" ]; then
    shift
    for ref in "$@"; do
        lib="$ROOT/tools/exp/lib_$(echo "$ref" | tr '/' '_').so"
        echo "== $ref"
        P2P_LIB_PATH="$lib" NPROP=${NPROP:-2000} timeout 120 python "$ROOT/tools/regress_bench.py" bf16x2 2>&1 | grep median || true
        P2P_LIB_PATH="$lib" timeout 120 python "$ROOT/tools/split_check.py" 2>&1 | tail -2 || true
    done
    exit 0
fi
for ref in "$@"; do
    tmp=$(mktemp -d)
    git -C "$ROOT" archive "$ref" patch2pix_amd/csrc include | tar -x -C "$tmp"
    objs=""
    for f in api coarse regress regress_split; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$tmp/patch2pix_amd/csrc/$f.hip" -o "$tmp/$f.o"
        objs="$objs $tmp/$f.o"
    done
    out="$ROOT/tools/exp/lib_$(echo "$ref" | tr '/' '_').so"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
    rm -rf "$tmp"
    echo "built $out"
done
