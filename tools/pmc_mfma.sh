#!/bin/bash
# MFMA utilisation of the bench's kernels from SQ / GRBM counters, per kernel family: counters only +
# kernel trace, one pass per counter set.  usage: tools/pmc_mfma.sh <tag> [bench args]; output gpurun_out/<tag>_pmc_mfma.json
tag=${1:-r}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_mfma/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-other-configs --streams 1 "$@" > gpurun_out/${tag}_pmc_mfma_p$i.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_mfma_p$i.log; exit 1; }
done
python3 tools/pmc_mfma_summary.py gpurun_out/${tag}_pmc_mfma > gpurun_out/${tag}_pmc_mfma.json && head -c 2500 gpurun_out/${tag}_pmc_mfma.json
