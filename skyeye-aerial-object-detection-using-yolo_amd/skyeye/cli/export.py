"""export -- counterpart of the reference's (empty) skyeye/cli/export.py: dump the engine's packed weight file.

    python -m skyeye.cli.export --cfg skyeye_s.yaml --weights w.pt --img 1280 --half --out skyeye_s_1280.npz
"""
import argparse

import torch

from ..core.models.detector import SkyEyeDetector


def run(cfg="skyeye_s.yaml", weights=None, img=640, batch=1, half=False, out="skyeye_engine.npz", device=0):
    model = SkyEyeDetector(cfg).eval()
    if weights:
        model.load_from_pretrained(weights)
    if half:
        model.half()
    x = torch.zeros(batch, 3, img, img, dtype=torch.uint8, device=torch.device("cuda", device))
    index = model.export_engine(out, x)
    print(f"{out}: {len(index)} packed convolutions ({'bf16' if half else 'fp32'}), BatchNorm folded, K = (ky, kx, cin)")
    return index


def parse_opt():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="skyeye_s.yaml")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--out", default="skyeye_engine.npz")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args()


if __name__ == "__main__":
    run(**vars(parse_opt()))
