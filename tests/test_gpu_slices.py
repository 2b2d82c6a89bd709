"""``parallel_slices``: a batch run as equal slices on parallel HIP streams (one plan per slice, the slices write into ONE output
tensor) is bit-identical to the whole-batch call -- frames are independent units --, also under capture_graph and with the raw
levels skipped."""
import numpy as np
import pytest
import torch

from helpers import build_detector, detector_params, variant_cfg
from seeded import seeded_scene
from skyeye.utils.metrics import nms_raw
from skyeye.utils.torch_utils import capture_graph

pytestmark = pytest.mark.gpu


def _model(prec="bf16"):
    m = build_detector(variant_cfg("skyeye_s"))
    m.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in detector_params("skyeye_s").items()}, strict=True)
    return m.eval().set_precision(prec)


@pytest.mark.parametrize("nsl,B,hw", [(2, 8, (320, 320)), (4, 8, (256, 384)), (2, 32, (640, 640))])
def test_sliced_batch_equals_whole_batch(nsl, B, hw):
    x = torch.from_numpy(seeded_scene(B, hw[0], hw[1], 31)).cuda()
    whole = _model()
    det0, raw0 = whole(x)
    sl = _model().parallel_slices(nsl)
    det1, raw1 = sl(x)
    assert torch.equal(det0, det1)
    assert all(torch.equal(a, b) for a, b in zip(raw0, raw1))
    det2, raw2 = sl(x, return_raw=False)
    assert raw2 == [] and torch.equal(det0, det2)
    det3, _ = sl(x[:3])                       # a batch the slices do not divide: the whole-batch plan
    assert torch.equal(det3, det0[:3])


def test_sliced_batch_in_a_captured_graph():
    x = torch.from_numpy(seeded_scene(8, 320, 320, 32)).cuda()
    ref = _model()
    d0, _ = ref(x, return_raw=False)
    r0, c0 = nms_raw(d0, 0.25, 0.45)
    m = _model().parallel_slices(2).reuse_output_buffers(True)

    def step():
        d, _ = m(x, return_raw=False)
        return nms_raw(d, 0.25, 0.45)

    graph, (rows, counts) = capture_graph(step, warmup=2)
    for _ in range(3):
        rows.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, c0) and torch.equal(rows, r0)


def test_detect_nms_equals_forward_then_nms():
    x = torch.from_numpy(seeded_scene(8, 320, 320, 33)).cuda()
    ref = _model()
    d0, _ = ref(x, return_raw=False)
    r0, c0 = nms_raw(d0, 0.2, 0.45)
    r1, c1 = ref.detect_nms(x, 0.2, 0.45)
    assert torch.equal(c0, c1) and torch.equal(r0, r1)
    m = _model().parallel_slices(2).reuse_output_buffers(True)
    graph, (rows, counts) = capture_graph(lambda: m.detect_nms(x, 0.2, 0.45), warmup=2)
    rows.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, c0) and torch.equal(rows, r0)
    assert int(c0.max()) > 0


def test_detect_nms_pipelined_equals_detect_nms_per_batch():
    """NMS of batch k beside the forward pass of batch k + 1 (two buffer sets): every batch's rows / counts equal the unpipelined pair's, eagerly
    (parity toggling) and as two captured hipGraphs replayed alternately (the bench's form)."""
    from skyeye.utils.torch_utils import capture_graph
    P = detector_params("skyeye_s")
    m = build_detector(variant_cfg("skyeye_s"))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}, strict=True)
    m = m.eval().set_precision("bf16")
    m.reuse_output_buffers(True)
    m.parallel_slices(2)
    xs = [torch.from_numpy(seeded_scene(4, 320, 320, 90 + i)).cuda() for i in range(4)]
    want = []
    for x in xs:
        r, c = m.detect_nms(x, 0.05, 0.45, max_detections=100)
        want.append((r.clone(), c.clone()))
    got = []
    for x in xs:
        res = m.detect_nms_pipelined(x, 0.05, 0.45, max_detections=100)
        if res is not None:
            got.append((res[0].clone(), res[1].clone()))
    r, c = m.detect_nms_flush()
    got.append((r.clone(), c.clone()))
    assert len(got) == len(want)
    for (r0, c0), (r1, c1) in zip(want, got):
        assert int(c0.sum()) > 0 and torch.equal(c0, c1) and torch.equal(r0, r1)
    # two graphs, replayed alternately on a static input buffer
    xin = xs[0].clone()
    graphs = [capture_graph(lambda p=p: m.detect_nms_pipelined(xin, 0.05, 0.45, max_detections=100, parity=p), warmup=2) for p in (0, 1)]
    got = []
    for i, x in enumerate(xs):
        xin.copy_(x)
        g, held = graphs[i & 1]
        g.replay()
        torch.cuda.synchronize()
        if i:
            got.append((held[0].clone(), held[1].clone()))
    r, c = m.detect_nms_flush(parity=(len(xs) - 1) & 1)
    got.append((r.clone(), c.clone()))
    assert len(got) == len(want)
    for (r0, c0), (r1, c1) in zip(want, got):
        assert torch.equal(c0, c1) and torch.equal(r0, r1)


def test_unequal_slices_equal_whole_batch():
    """parallel_slices((5, 3)): explicit slice sizes (batches of exactly their sum are sliced that way) -- bit-identical again."""
    x = torch.from_numpy(seeded_scene(8, 256, 320, 33)).cuda()
    det0, raw0 = _model()(x)
    sl = _model().parallel_slices((5, 3))
    det1, raw1 = sl(x)
    assert torch.equal(det0, det1) and all(torch.equal(a, b) for a, b in zip(raw0, raw1))
    r0, c0 = nms_raw(det0, 0.05, 0.45, max_detections=100)
    r1, c1 = sl.detect_nms(x, 0.05, 0.45, max_detections=100)
    assert torch.equal(c0, c1) and torch.equal(r0, r1)
    y = torch.from_numpy(seeded_scene(6, 256, 320, 34)).cuda()          # another batch size: not sliced, still right
    assert torch.equal(_model()(y)[0], sl(y)[0])


def test_fp8_slices_need_calibration_frames_of_the_slices_geometry():
    """fp8 plans take their activation scales from the frames given to ``calibrate()`` -- when those have the plan's geometry.  Calibrated at
    ANOTHER resolution, every slice plan would fall back to calibrating itself on its own half of the batch (other scales than the whole
    batch: results that depend on the batch's composition).  ``_sliceable`` applies ``_engine_entry``'s predicate, so such a call takes the
    whole-batch plan: bit-identical to the unsliced module; with matching calibration frames the slices run and are bit-identical too."""
    x = torch.from_numpy(seeded_scene(8, 320, 320, 41)).cuda()
    other = torch.from_numpy(seeded_scene(4, 256, 256, 42)).cuda()
    ref = _model("fp8")
    ref.calibrate(other)                                     # (not used by a 320 x 320 plan: it calibrates itself on the whole batch)
    d0, _ = ref(x, return_raw=False)
    sl = _model("fp8").parallel_slices(2)
    sl.calibrate(other)
    assert not sl._sliceable([sl._prepare_input(x)], 2)
    d1, _ = sl(x, return_raw=False)
    assert torch.equal(d0, d1)
    # matching geometry: common scales for both slices, sliced == whole batch
    ref2 = _model("fp8")
    ref2.calibrate(x[:4])
    e0, _ = ref2(x, return_raw=False)
    sl2 = _model("fp8").parallel_slices(2)
    sl2.calibrate(x[:4])
    assert sl2._sliceable([sl2._prepare_input(x)], 2)
    e1, _ = sl2(x, return_raw=False)
    assert torch.equal(e0, e1)
