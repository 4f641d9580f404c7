#!/bin/bash
# PMC passes over a micro-benchmark (separate passes, --kernel-trace only). Usage: tools/pmc_attn.sh <substr> <out.txt> -- <python tool + args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sub=$1; out=$2; shift 3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_p$i -- "$@" > gpurun_out/pmc_p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_p$i.log; }
done
python3 tools/pmc_kernel.py "$sub" gpurun_out/pmc_p1 gpurun_out/pmc_p2 gpurun_out/pmc_p3 gpurun_out/pmc_p4 gpurun_out/pmc_p5 > "$out"
rm -rf gpurun_out/pmc_p1 gpurun_out/pmc_p2 gpurun_out/pmc_p3 gpurun_out/pmc_p4 gpurun_out/pmc_p5
cat "$out"
