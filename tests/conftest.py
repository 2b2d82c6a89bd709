import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Parity margins of every comparison of the run (tests/parity.py): worst |d| / limit and 1 - min IoU per case."""
    try:
        import parity
    except Exception:  # noqa: BLE001
        return
    if not parity.MARGINS:
        return
    parity.write_margins(os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    agree = [m for m in parity.MARGINS if m.get("kind") == "agreement"]
    if agree:
        terminalreporter.section("reduced-precision agreement rates")
        for m in agree:
            rates = "  ".join(f"{k}={v}" for k, v in m.items() if k not in ("where", "kind", "case", "worst_ratio"))
            terminalreporter.write_line(f"{m['case']:<34s} {rates}")
    worst = {}
    for m in parity.MARGINS:
        if m.get("kind") == "agreement":
            continue
        w = worst.setdefault(m["where"], dict(ratio=0.0, iou=0.0, n=0))
        w["ratio"] = max(w["ratio"], m["worst_ratio"])
        w["iou"] = max(w["iou"], m.get("one_minus_min_iou", 0.0))
        w["n"] += 1
    tr = terminalreporter
    tr.section("parity margins (worst error / limit; 1 - min IoU)")
    for k, w in sorted(worst.items(), key=lambda kv: -kv[1]["ratio"])[:40]:
        tr.write_line(f"{w['ratio']:8.3f}  {w['iou']:.2e}  x{w['n']:<3d} {k}")
