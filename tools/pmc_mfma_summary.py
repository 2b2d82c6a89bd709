#!/usr/bin/env python3
"""Summarise rocprofv3 SQ / GRBM counter passes per kernel family of the SkyEye engine.

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the share of SIMD-cycles in which the
matrix pipe was busy while the kernel ran (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, GRBM_GUI_ACTIVE
is summed over the 8 XCDs).  Counters are per dispatch, averaged over the dispatches of a family."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]


def family(k):
    if "stem_down_kernel" in k:
        return "stem_down"
    if "csp_stage_kernel" in k:
        return "csp_stage"
    if "bneck64w" in k:
        return "halo_cv1"            # BottleneckBlock(64, 64) fused: k_bneck_w64.hip (default) or the halo-tile kernel's CV1 form
    if "bneck128" in k:
        return "bneck128"
    if "conv3x3_deep_kernel" in k:
        return "deep3x3"
    if "gemm1x1_kernel" in k:
        return "gemm1x1"
    if "cv3_head_kernel" in k:
        return "cv3_head"
    if "head_stream_kernel" in k:
        return "tile"                # detection levels: same family as the tile kernel's head launches
    if "conv_halo_kernel" in k and (re.search(r"Lb1EEEvNS_8ConvArgsE$", k.split("(")[0]) or re.search(r", true>$", k.split("(")[0])):
        return "halo_cv1"            # last template argument CV1 = true: the fused bottleneck
    if "conv_halo_small" in k:
        return "halo_narrow"
    if "conv_halo_kernel" in k:
        return "halo3x3"
    if "conv_stream_kernel" in k:
        m = re.search(r"conv_stream_kernelI\w+?Li\dELi\dELi\dELb(\d)E", k) or re.search(r"conv_stream_kernel<[^,]+, \d, \d, \d, (true|false)", k)
        ring = m and m.group(1) in ("1", "true")
        return "stream_ring" if ring else "stream_resident"
    if "conv_igemm" in k:
        return "tile"
    if "attention" in k or "layernorm" in k or "cla_" in k:
        return "attention"
    if "nms_" in k:
        return "nms"
    if "sky" in k:
        return "other_sky"
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{root}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[fam][r["Counter_Name"]] += 1
out = {}
for fam, d in agg.items():
    e = {c: v / cnt[fam][c] for c, v in d.items()}
    e["dispatches"] = max(cnt[fam].values())
    gui = e.get("GRBM_GUI_ACTIVE")
    if gui:
        simd_cycles = gui / 8.0 * 256 * 4
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
        if "GRBM_TA_BUSY" in e:
            e["ta_busy_frac"] = round(e["GRBM_TA_BUSY"] / gui, 4)
    if e.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        e["mfma_busy_over_sq_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"], 4)
    out[fam] = {k: (round(v, 1) if isinstance(v, float) and k.isupper() else v) for k, v in sorted(e.items())}
print(json.dumps(dict(note="per-dispatch means per kernel family; bench.py --steps 2 --warmup 1 --no-graph", families=out), indent=1))
