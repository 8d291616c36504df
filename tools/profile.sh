#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# usage: tools/profile.sh <tag> <workload>        outputs under gpurun_out/prof/<tag>/
set -u
TAG=${1:-r02}; WL=${2:-cfg3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_$C.json 2> $OUT/$C.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_SQ -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_SQ.json 2> $OUT/SQ.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/pmc_SQ2 -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_SQ2.json 2> $OUT/SQ2.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_SQ3 -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_SQ3.json 2> $OUT/SQ3.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_TCC -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_TCC.json 2> $OUT/TCC.err
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_TA -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_TA.json 2> $OUT/TA.err
python tools/summarize_prof.py $OUT $WL > $OUT/summary.json
cat $OUT/summary.json
