"""GPU parity of the evaluation accounting: sky_box_iou through the C ABI against the fixture the
reference's box_iou produced and against the oracle; process_batch against the oracle; mAP of the bf16 engine
measured against the fp32 engine's detections as labels."""
import os

import numpy as np
import pytest
import torch

from skyeye.utils import metrics as M

pytestmark = pytest.mark.gpu
E = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def test_box_iou_matches_reference_fixture():
    a = torch.from_numpy(E["iou.a"]).cuda()
    b = torch.from_numpy(E["iou.b"]).cuda()
    lit = M.box_iou(a.T.contiguous(), b).cpu().numpy()                    # the file's own indexing: box1 is [4, N]
    rows = M.box_iou(a, b, layout="rows").cpu().numpy()
    assert lit.shape == (37, 53)
    # fp32, same operation order, no FMA contraction: 1 ulp of the division at most (tolerance 1e-6 relative)
    np.testing.assert_allclose(lit, E["iou.out"], rtol=1e-6, atol=0)
    assert np.array_equal(lit, rows)
    assert int((lit != E["iou.out"]).sum()) == 0, "expected bit-identical IoU on this fixture"


def test_box_iou_shapes_and_errors():
    O = _oracle()
    r = np.random.default_rng(5)
    for n, m in ((1, 1), (0, 7), (5, 0), (300, 1000)):
        a = r.uniform(0, 100, (n, 2)).astype(np.float32)
        a = np.concatenate([a, a + r.uniform(1, 50, (n, 2)).astype(np.float32)], 1)
        b = r.uniform(0, 100, (m, 2)).astype(np.float32)
        b = np.concatenate([b, b + r.uniform(1, 50, (m, 2)).astype(np.float32)], 1)
        got = M.box_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), layout="rows").cpu().numpy()
        np.testing.assert_allclose(got, O.box_iou(a, b, literal=False), rtol=1e-6)
    with pytest.raises(Exception):
        M.box_iou(torch.zeros(3, 4), torch.zeros(3, 4))                    # CPU tensors: no CPU path
    with pytest.raises(ValueError):
        M.box_iou(torch.zeros(5, 4).cuda(), torch.zeros(3, 4).cuda())      # literal layout wants [4, N]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_process_batch_matches_oracle(seed):
    O = _oracle()
    r = np.random.default_rng(seed)
    m, n = 30, 90
    lab_xy = r.uniform(0, 500, (m, 2)).astype(np.float32)
    lab = np.concatenate([r.integers(0, 4, (m, 1)).astype(np.float32), lab_xy, lab_xy + r.uniform(10, 80, (m, 2)).astype(np.float32)], 1)
    src = r.integers(0, m, n)
    det_box = lab[src, 1:5] + r.normal(0, 3.0, (n, 4)).astype(np.float32)        # jittered copies: several candidates per label
    cls = np.where(r.uniform(0, 1, n) < 0.8, lab[src, 0], r.integers(0, 4, n)).astype(np.float32)
    det = np.concatenate([det_box, r.uniform(0, 1, (n, 1)).astype(np.float32), cls[:, None]], 1).astype(np.float32)
    iouv = torch.linspace(0.5, 0.95, 10).cuda()
    got = M.process_batch(torch.from_numpy(det).cuda(), torch.from_numpy(lab).cuda(), iouv).cpu().numpy()
    ref = O.process_batch(det, lab, iouv.cpu().numpy())
    assert got.dtype == bool and got.shape == (n, 10)
    assert np.array_equal(got, ref)
    assert got[:, 0].sum() >= got[:, -1].sum() and got[:, 0].sum() > 10
    assert M.process_batch(torch.zeros(0, 6).cuda(), torch.from_numpy(lab).cuda(), iouv).shape == (0, 10)


def test_bf16_engine_map_against_fp32_engine():
    """The accuracy statement that survives reduced precision: detections of the bf16 engine scored against the fp32
    engine's detections as ground truth (same seeded weights, structured scenes, YOLOv5-semantics NMS)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import build_model, calibrate_objectness
    from seeded import seeded_scene
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(seeded_scene(4, 320, 320, seed=11)).to(dev)
    outs = {}
    shift = None
    for prec in ("fp32", "bf16"):
        model, _ = build_model("skyeye_s", prec, dev)
        if shift is None:
            shift = calibrate_objectness(model, x, 0.01, 0.25)
        else:
            no = model.detection_head.detection_layers[0].bias.numel() // 3
            with torch.no_grad():
                for layer in model.detection_head.detection_layers:
                    layer.bias.view(-1, no)[:, 4] += shift
            model.refresh_weights()
        det, _ = model(x)
        outs[prec] = M.non_max_suppression(det, 0.25, 0.45, mode="corrected")
    labels = [torch.cat([o[:, 5:6], o[:, :4]], 1) for o in outs["fp32"]]
    assert sum(l.shape[0] for l in labels) > 20, "calibration should leave a realistic number of boxes"
    res = M.mean_average_precision(outs["bf16"], labels)
    same = M.mean_average_precision(outs["fp32"], labels)
    assert same["map"] > 0.99                                    # the fp32 engine against itself
    assert res["map50"] > 0.85 and res["map"] > 0.7, res
