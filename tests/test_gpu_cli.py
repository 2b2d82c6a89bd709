"""The validate / detect counterparts run the hot path end to end (BASELINE configs[0]: skyeye_s, 640x640)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_validate_counterpart_runs_config0():
    from skyeye.cli.validate import validate
    r = validate(cfg="skyeye_s.yaml", batch_size=2, img_size=640, num_batches=1, half=True, verbose=True)
    assert r["images"] == 2 and len(r["results"]) == 2 and all(x >= 0 for x in r["speed_ms"])


def test_detect_counterpart_scales_boxes():
    from skyeye.cli.detect import run
    frames = np.random.default_rng(1).integers(0, 256, size=(2, 3, 320, 320), dtype=np.uint8)
    out = run(source=frames, imgsz=320, orig_shapes=[(640, 640), (300, 320)], conf_thres=0.5)
    assert len(out) == 2
    for det, shp in zip(out, [(640, 640), (300, 320)]):
        if det.shape[0]:
            assert det[:, [0, 2]].max() <= shp[1] and det[:, [1, 3]].max() <= shp[0] and det[:, :4].min() >= 0
