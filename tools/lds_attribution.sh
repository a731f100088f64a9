#!/bin/bash
# Which phase of regress_h2_kernel produces its LDS bank conflicts, and what does each phase cost in (power-limited) time?
# Needs tools/exp/lib_{skipconv2,skipc,skipp,skipfold,pinw}.so (bash tools/ab_variants.sh skipconv2=-DXF_SKIP_CONV2 ...).
# Output: per variant the launch time (no profiler) and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the launch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export NPROP=6400 NPAIRS=16
for v in default skipconv2 skipc skipp skipfold pinw; do
  lib=$ROOT/tools/exp/lib_$v.so; [ $v = default ] && lib=$ROOT/patch2pix_amd/csrc/libp2p_hip.so
  echo "== $v"
  P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH=$lib NITER=7 timeout 120 python $ROOT/tools/regress_bench.py fp16x2 2>&1 | grep median
  rm -rf /tmp/pl
  P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH=$lib NITER=2 timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace -d /tmp/pl -o r -- python $ROOT/tools/regress_bench.py fp16x2 > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/pl/**/*.db", recursive=True)
if db:
    c = sqlite3.connect(db[0])
    q = "select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%regress_h2%' group by counter_name"
    r = {n: v for n, k, v in c.execute(q)}
    print("   ", {k: f"{v:.4g}" for k, v in r.items()}, "conflict/active = %.3f" % (r.get("SQ_LDS_BANK_CONFLICT", 0) / max(r.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done
