#!/bin/bash
# Resource usage + ISA of the all-pairs engine instances for one group count:  tools/isa.sh [G=2] [extra hipcc flags]
# writes /tmp/g$G.s and, per kNN list size, /tmp/knn20.s / /tmp/knn64.s (5-bit instances)
G=${1:-2}; shift
cd /root/repo/prograph_amd/csrc || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed -Wno-unused-variable -mllvm -amdgpu-mfma-vgpr-form=1 -DPG_G=$G $*"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c pg_nsq_inst.hip -o /tmp/g$G.o 2>&1 \
  | grep -E "error|Function Name: _Z12pg_mm|VGPRs:|ScratchSize|Spill|LDS Size" | grep -A5 "pg_mm\|error" | sed 's/.*remark: //' \
  | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: _Z12pg_mm_kernelI13HammingMetricI//; s/EEv9NsqParams//' | paste - - - - - - | tr -s ' \t' ' '
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only pg_nsq_inst.hip -o /tmp/g$G.s 2>&1 | grep -E "error" | head -3
for k in 20 64; do
  awk -v pat="^_Z12pg_mm_kernelI13HammingMetricILi${G}ELi5EELi1ELi${k}EEv9NsqParams:" '$0 ~ pat{p=1} p{print} /s_endpgm/{if(p) exit}' /tmp/g$G.s > /tmp/knn$k.s
done
wc -l /tmp/knn20.s /tmp/knn64.s | head -2
