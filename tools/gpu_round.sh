#!/bin/bash
# One gpurun call: GPU tests, smoke, benches, rocprofv3 kernel traces (each step under its own timeout).
# Usage (repo root on the GPU box):  bash tools/gpu_round.sh <tag> [steps...]   steps default: tests smoke bench trace
set -u
TAG=${1:-r02}; shift || true
STEPS=${*:-tests smoke bench trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for s in $STEPS; do
case $s in
tests) rm -f $O/${TAG}_parity.txt; DSC_PARITY_LOG=$O/${TAG}_parity.txt timeout 1800 python -m pytest tests -m gpu -q --durations=15 > $O/${TAG}_tests.log 2>&1; tail -40 $O/${TAG}_tests.log | grep -v "^$" | tail -30 ;;
twolayer) timeout 300 python tools/two_layer_probe.py > $O/${TAG}_two_layer.txt 2>&1; cat $O/${TAG}_two_layer.txt | tail -14 ;;
smoke) timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log ;;
bench) timeout 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -1 $O/${TAG}_bench_default.json | cut -c1-600 ;;
benchall) for c in bedroom21 text complete arrange; do timeout 400 python bench.py --config $c > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err; tail -1 $O/${TAG}_bench_$c.json | cut -c1-300; done ;;
gemmbench) timeout 300 python tools/gemm_split_bench.py > $O/${TAG}_gemm_split_bench.txt 2>&1; tail -9 $O/${TAG}_gemm_split_bench.txt ;;
profile) timeout 300 python tools/plan_profile.py living80 > $O/${TAG}_plan_profile_sample_living80.txt 2>&1; timeout 300 python tools/plan_profile.py living80 --train > $O/${TAG}_plan_profile_train_living80.txt 2>&1; grep -A 12 aggregated $O/${TAG}_plan_profile_train_living80.txt | cut -c1-150 ;;
trace) cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_sample -o s -- python $R/bench.py --mode sample --steps 20 --warmup 3 --no-cpu-baseline --no-full-loop > $O/${TAG}_trace_sample.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_train -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_trace_train.log 2>&1
  cd $R; python tools/rocpd_summary.py $(find $O/${TAG}_trace_sample -name "*.db" | head -1) > $O/${TAG}_sample_kernel_trace.txt 2>&1; python tools/rocpd_summary.py $(find $O/${TAG}_trace_train -name "*.db" | head -1) > $O/${TAG}_train_kernel_trace.txt 2>&1; head -12 $O/${TAG}_sample_kernel_trace.txt; head -25 $O/${TAG}_train_kernel_trace.txt ;;
pmc) cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_pmc_sample -o p -- python $R/bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-full-loop > $O/${TAG}_pmc_sample.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o f -- python $R/bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-full-loop > $O/${TAG}_pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o w -- python $R/bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline --no-full-loop > $O/${TAG}_pmc_write.log 2>&1
  cd $R; python tools/rocpd_summary.py $(find $O/${TAG}_pmc_sample -name "*.db" | head -1) > $O/${TAG}_sample_pmc.txt 2>&1
  python tools/pmc_summary.py $TAG > $O/${TAG}_pmc_summary.json 2>&1; tail -25 $O/${TAG}_pmc_summary.json ;;
pmctrain) cd /tmp && export TMPDIR=/tmp
  # training-side counters (separate passes, kernel trace only): MFMA busy, then HBM fetch, then HBM write
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/${TAG}_pmct_sq -o p -- python $R/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmct_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmct_fetch -o f -- python $R/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmct_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmct_write -o w -- python $R/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmct_write.log 2>&1
  cd $R; for k in sq fetch write; do python tools/rocpd_summary.py $(find $O/${TAG}_pmct_$k -name "*.db" | head -1) > $O/${TAG}_train_pmc_$k.txt 2>&1; done
  python tools/pmc_train_summary.py $TAG > $O/${TAG}_train_pmc.txt 2>&1; head -30 $O/${TAG}_train_pmc.txt ;;
skinny) # the K-parallel small-launch GEMM: phase stamps (staged form, then the direct-fragment form it replaced), the B = 1 plan with the kernel off / on
  ( for k in 512 1024; do ./tools/skinny_probe 1 $k 21 1 400 1; ./tools/skinny_probe 1 $k 21 1 400 1 0 1; done; ./tools/skinny_probe 0 512 21 1 400 1; ./tools/skinny_probe 0 3072 21 1 400 1; ./tools/skinny_probe 1 512 12 4 400 1 ) > $O/${TAG}_skinny_probe.txt 2>&1
  for s in 0 1; do echo "== DSC_SKINNY=$s"; DSC_SKINNY=$s timeout 300 python tools/plan_profile.py bedroom21 --batch 1 2>&1 | grep -v amdgpu.ids; done > $O/${TAG}_plan_b1_skinny.txt 2>&1
  grep "launches, sum\|us per launch" $O/${TAG}_plan_b1_skinny.txt $O/${TAG}_skinny_probe.txt | head -30 ;;
guard) rm -f $O/${TAG}_guard.log; timeout 300 python tools/stale_plan_repro.py arrange > $O/${TAG}_stale_plan_repro.txt 2> $O/${TAG}_stale_plan_repro.err; cat $O/${TAG}_stale_plan_repro.txt | cut -c1-700
  DSC_GUARD_LOG=$O/${TAG}_guard.log timeout 1500 python -m pytest tests/test_gpu_guard.py -q -x --durations=8 > $O/${TAG}_guard_tests.log 2>&1; tail -25 $O/${TAG}_guard_tests.log ;;
guard20) timeout 1500 python tools/guard_run.py --mode normal --loops 20 --T 3 --train-steps 3 --out $O/${TAG}_guard_loops20.json > /dev/null 2> $O/${TAG}_guard_loops20.log; tail -4 $O/${TAG}_guard_loops20.log | cut -c1-400 ;;
ddprehearsal) DSC_BENCH_DDP_REHEARSAL=1 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-full-loop > $O/${TAG}_bench_ddp_rehearsal.json 2> $O/${TAG}_bench_ddp_rehearsal.err; grep -c "ok$" $O/${TAG}_bench_ddp_rehearsal.err ;;
ddp) timeout 600 python bench.py --ddp-selftest > $O/${TAG}_bench_ddp_selftest.json 2> $O/${TAG}_bench_ddp_selftest.err; tail -1 $O/${TAG}_bench_ddp_selftest.json | cut -c1-1500 ;;
benchf32) DSC_GEMM=f32 timeout 600 python bench.py > $O/${TAG}_bench_default_f32.json 2> $O/${TAG}_bench_default_f32.err; tail -1 $O/${TAG}_bench_default_f32.json | cut -c1-600 ;;
*) echo "unknown step $s" ;;
esac
done
# the rocprofv3 databases are summarised above; gpurun only copies back <= 64 MiB
find $O -name "*.db" -size +2M -delete 2>/dev/null
ls $O | grep $TAG | head -40
