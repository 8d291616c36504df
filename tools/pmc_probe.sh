#!/bin/bash
# rocprofv3 PMC passes over tools/probe_case.py: tools/pmc_probe.sh <tag> <data> <mode>
set -u
TAG=$1; DATA=$2; MODE=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python tools/probe_case.py $DATA $MODE > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pg_nsq_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
