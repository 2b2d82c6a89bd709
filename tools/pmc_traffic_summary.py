#!/usr/bin/env python3
"""Summarise rocprofv3 FETCH_SIZE / WRITE_SIZE passes per kernel: launches, KB fetched/written, corrected HBM bytes per launch."""
import collections, csv, glob, json, re, sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: dict(launches=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"]
            agg[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                agg[k]["launches"] += 1
out = {}
for k, v in agg.items():
    if not v["launches"]:
        continue
    m = re.search(r"conv_stream_kernelI(DF16b|f|NS_5fp8_tE)Li(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d)E", k)
    name = k.split("(")[0][:100]
    variant = None
    m2 = re.search(r"conv_stream_kernel<[^,]+, \d, (\d), (\d), (true|false), (true|false)", k)      # <T, KS, MF, NF, RING, UTAP, ...>
    if m:
        variant = (3000 if m.group(5) == "1" else 2000) + int(m.group(4)) * 16
    elif m2:
        variant = (3000 if m2.group(3) == "true" else 2000) + int(m2.group(2)) * 16
    elif "conv_halo_small_kernel" in k:
        mh = re.search(r"conv_halo_small_kernelI(?:DF16b|f|NS_5fp8_tE)Li(\d+)ELi(\d)E", k) or re.search(r"conv_halo_small_kernel<.*?(\d+), (\d),", k)
        variant = 5000 + int(mh.group(2)) * 16 if mh else 5000
    elif "stem_down_kernel" in k:
        variant = 8064
    elif "csp_stage_kernel" in k:
        variant = 8564
    elif "bneck128w8" in k:
        variant = 7257
    elif "bneck64w8" in k:
        variant = 7066
    elif "bneck64w" in k:
        variant = 7065
    elif "bneck128w" in k:
        variant = 7256
    elif "bneck128" in k:
        variant = 7128
    elif "conv3x3_deep_kernel" in k:
        variant = 4728
    elif "cv3_head_kernel" in k:
        variant = 1628
    elif "gemm1x1_kernel" in k:
        variant = 3256
    elif "head_stream_kernel" in k:
        variant = 1548
    elif "conv_halo_kernel" in k:
        # <T, NF, SQ, S2, FC, TO, CV1> (mangled or demangled); CV1 = fused bottleneck (7064)
        mh = re.search(r"conv_halo_kernelI(?:DF16b|f|NS_5fp8_tE)Li(\d)ELb(\d)ELb(\d)ELi(\d+)E(?:DF16b|f|NS_5fp8_tE|S1_)Lb(\d)E", k)
        mh2 = re.search(r"conv_halo_kernel<[^,]+, (\d), (true|false), (true|false), (\d+), [^,]+, (true|false)>", k)
        if mh:
            variant = 7064 if mh.group(5) == "1" else (6000 if mh.group(3) == "1" else 4000) + int(mh.group(1)) * 16
        elif mh2:
            variant = 7064 if mh2.group(5) == "true" else (6000 if mh2.group(3) == "true" else 4000) + int(mh2.group(1)) * 16
    elif "conv_igemm" in k:
        variant = 1000
    # bytes: counters are in KB; FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)
    hbm = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    out[name] = dict(variant=variant, launches=v["launches"], fetch_kb=v["FETCH_SIZE"], write_kb=v["WRITE_SIZE"],
                     hbm_bytes_per_launch=hbm / v["launches"])
by_variant = collections.defaultdict(lambda: dict(launches=0, hbm_bytes=0.0))
for v in out.values():
    if v["variant"]:
        by_variant[str(v["variant"])]["launches"] += v["launches"]
        by_variant[str(v["variant"])]["hbm_bytes"] += v["hbm_bytes_per_launch"] * v["launches"]
for v in by_variant.values():
    v["hbm_bytes_per_launch"] = v["hbm_bytes"] / v["launches"]
total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
def _library_sources():
    """Hash of the kernel sources the measured library was built from (bench.py compares it with the running library's)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "skyeye-aerial-object-detection-using-yolo_amd"))
    try:
        from skyeye import _native
        return _native.build_info()
    except Exception:  # noqa: BLE001
        return None


print(json.dumps(dict(note="FETCH_SIZE*2 + WRITE_SIZE, KB -> bytes; bench.py --steps 2 --warmup 1 --no-graph (+ calibration forwards of 2 frames, + the timing-bucket passes of bench.py)",
                      library_sources=_library_sources(),
                      total_hbm_bytes_all_kernels=total,
                      by_variant=by_variant, kernels=out), indent=1))
