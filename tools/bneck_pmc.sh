#!/bin/bash
# SQ / LDS counters of the 128-channel bottleneck kernel (tools/bneck_micro.py), one pass per counter set; prints per-kernel means.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bneck_pmc
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/bneck_pmc/p$i -- python tools/bneck_micro.py > gpurun_out/bneck_pmc_p$i.log 2>&1 || { tail -5 gpurun_out/bneck_pmc_p$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for p in (1, 2):
    f = glob.glob(f"gpurun_out/bneck_pmc/p{p}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            k = "bneck128" if "bneck128" in k else "halo128" if "conv_halo_kernel" in k else "stream" if "conv_stream" in k else None
            if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
