#!/usr/bin/env python3
"""Static guard for the store-data hazard of DESIGN.md section 3: compile the kernel sources to gfx950 assembly and list every vector
store of more than 64 bits whose `soffset` is an SGPR (LLVM then inserts no wait state before the data registers are overwritten)
together with the distance, in instructions, to the next write of one of its data registers.  Exit status 1 if any such store is
followed by a VALU write within 2 instructions.

    python tools/isa_store_hazard_scan.py [file.hip ...]        # default: every csrc/*.hip
"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def makefile_flags():
    """FLAGS and the per-file FLAGS_<stem> of csrc/Makefile: the scanned ISA must be the SHIPPED ISA (k_conv3x3_deep.hip is built with
    -mllvm -pragma-unroll-threshold=200000, which unrolls its 36-step body completely and changes the schedule around the stores)."""
    common, per = [], {}
    for ln in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"^FLAGS\s*:=\s*(.*)$", ln)
        if m:
            common = [t for t in m.group(1).split() if t not in ("-fPIC",) and "$(" not in t]
        m = re.match(r"^FLAGS_(\w+)\s*:=\s*(.*)$", ln)
        if m:
            per[m.group(1)] = m.group(2).split()
    return common, per


def compile_asm(src, out):
    common, per = makefile_flags()
    stem = os.path.splitext(os.path.basename(src))[0]
    flags = common + ["--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", CSRC] + per.get(stem, [])
    subprocess.run([HIPCC] + flags + ["-o", out, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def scan(asm_path):
    findings = []
    name, body = None, []
    kernels = {}
    for ln in open(asm_path):
        m = re.match(r"^(_Z\S+):", ln)
        if m:
            name = m.group(1)
            kernels[name] = []
            continue
        if name is not None:
            kernels[name].append(ln)
        if ".end_amdhsa_kernel" in ln:
            name = None
    for k, body in kernels.items():
        ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        for i, l in enumerate(ins):
            if not re.match(r"buffer_store_dwordx[234]\b", l):
                continue
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            soff = ops[3].split()[0] if len(ops) > 3 else ""
            if not re.match(r"s\d+$", soff):
                continue
            data = regs(ops[0])
            dist, what = None, None
            for j in range(i + 1, min(i + 64, len(ins))):
                t = ins[j].split()
                if len(t) < 2 or t[0].startswith(("s_", "buffer_store", "global_store", "ds_write", "scratch_store")):
                    continue
                if regs(t[1].rstrip(",")) & data:
                    dist, what = j - i, ins[j]
                    break
            findings.append((k, l, dist, what))
    return findings


def main(argv):
    srcs = argv or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        jobs = {ex.submit(compile_asm, s, os.path.join(tmp, os.path.basename(s) + ".s")): s for s in srcs}
        for fut in concurrent.futures.as_completed(jobs):
            for k, store, dist, what in scan(fut.result()):
                valu = what is not None and what.startswith("v_")
                flag = valu and dist is not None and dist <= 2
                bad += flag
                print(f"{'HAZARD ' if flag else 'note   '}{os.path.basename(jobs[fut])} {k[:70]}: `{store}` -> +{dist}: `{what}`")
    print(f"{bad} wide store(s) with a register soffset overwritten by the VALU within 2 instructions")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
