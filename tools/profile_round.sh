#!/bin/bash
# Round-end evidence on the GPU box: tests, smoke, default bench, rocprofv3 kernel traces and PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/${TAG}_tests.log 2>&1; tail -2 $O/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -1 $O/${TAG}_bench_default.json | cut -c1-300
python bench.py --mode sample --steps 30 --warmup 3 > $O/${TAG}_bench_sample.json 2>/dev/null
python bench.py --mode train --steps 10 --warmup 3 > $O/${TAG}_bench_train.json 2>/dev/null
python bench.py --mode sample --steps 30 --warmup 3 --objects 21 > $O/${TAG}_bench_sample_n21.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_sample -o s -- python $R/bench.py --mode sample --steps 20 --warmup 3 > $O/${TAG}_trace_sample.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_train -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 > $O/${TAG}_trace_train.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/${TAG}_pmc_sample -o p -- python $R/bench.py --mode sample --steps 3 --warmup 1 > $O/${TAG}_pmc_sample.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o f -- python $R/bench.py --mode sample --steps 3 --warmup 1 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o w -- python $R/bench.py --mode sample --steps 3 --warmup 1 > $O/${TAG}_pmc_write.log 2>&1
ls $O | grep $TAG
