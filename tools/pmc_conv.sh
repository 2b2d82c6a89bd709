#!/bin/bash
# PMC passes for one conv_micro case (rocprofv3, counters only + kernel trace); results under gpurun_out/pmc_<tag>/
tag=$1; case_idx=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MICRO_ITERS=2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python tools/conv_micro.py bf16 $case_idx > gpurun_out/pmc_${tag}_p$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/pmc_${tag}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,d in agg.items():
    print(k[:90])
    for c,v in sorted(d.items()): print(f"   {c:32s} {v/cnt[(k,c)]:16.0f}  (per dispatch, {cnt[(k,c)]} dispatches)")
PY
