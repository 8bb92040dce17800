#!/bin/bash
# LDS / issue counters of the grouped weight-gradient launch under the three block bodies (DSC_TN_FORM), one rocprofv3 --pmc pass each.
#   bash tools/tn_pmc.sh <tag>  -> gpurun_out/<tag>_tn_pmc.txt
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_tn_pmc.txt
for f in 1 2; do
  DSC_TN_FORM=$f timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    -d $O/${TAG}_tnpmc_$f -o p -- python $R/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_tnpmc_$f.log 2>&1
  python - <<PY >> $O/${TAG}_tn_pmc.txt
import glob, sqlite3
db = glob.glob("$O/${TAG}_tnpmc_$f/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
dur = dict(cur.execute("select name, avg(end - start) / 1e3 from kernels group by name").fetchall())
rows = {}
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "gemm_tn_split" in k:
        rows.setdefault(k, {})[c] = v
for k, c in rows.items():
    print("DSC_TN_FORM=$f %s: %.1f us" % (k[:60], dur.get(k, 0)))
    for name in sorted(c):
        print("    %-28s %.4g" % (name, c[name]))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print("    lds bank conflict / active   %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
    if c.get("GRBM_GUI_ACTIVE"):
        print("    mfma busy per SIMD           %.3f   clock %.3f GHz" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), c["GRBM_GUI_ACTIVE"] / 8 / dur.get(k, 1) / 1e3))
PY
done
find $O -name "*.db" -size +2M -delete 2>/dev/null
cat $O/${TAG}_tn_pmc.txt
