"""Host logic of the batch-slicing / pipelining extensions (skyeye/core/models/_base.py, detector.py), no compute: which batches a
``parallel_slices`` setting applies to, argument checks, and that the pipelined detect call refuses to run where it cannot keep two batches apart
or without the HIP device (no CPU path)."""
import pytest
import torch

from helpers import build_detector, variant_cfg


def _model():
    return build_detector(variant_cfg("skyeye_s")).eval()


def test_equal_slices_apply_to_divisible_batches_only():
    m = _model().parallel_slices(2)
    assert m._sliceable([torch.zeros(8, 3, 64, 64)], 2)
    assert not m._sliceable([torch.zeros(7, 3, 64, 64)], 2)
    assert not m._sliceable([torch.zeros(2, 3, 64, 64)], 2)          # fewer than two frames per slice: not worth a fork
    assert m.parallel_slices(1).__dict__["_slices"] == 1


def test_explicit_slice_sizes():
    m = _model().parallel_slices((5, 3))
    assert m.__dict__["_slices"] == 2 and m.__dict__["_slice_sizes"] == (5, 3)
    assert m._sliceable([torch.zeros(8, 3, 64, 64)], 2)
    assert not m._sliceable([torch.zeros(6, 3, 64, 64)], 2)          # another batch size runs unsliced
    with pytest.raises(ValueError):
        m.parallel_slices((4, 0))
    m.parallel_slices((8,))                                           # one slice = no slicing
    assert m.__dict__["_slice_sizes"] is None and m.__dict__["_slices"] == 1
    m.parallel_slices(4)
    assert m.__dict__["_slice_sizes"] is None and m.__dict__["_slices"] == 4


def test_pipelined_detect_needs_reusable_buffers_and_eval_mode():
    m = _model()
    x = torch.zeros(2, 3, 64, 64, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="reuse_output_buffers"):
        m.detect_nms_pipelined(x)
    with pytest.raises(RuntimeError, match="nothing in flight"):
        m.detect_nms_flush()
    m.train()
    with pytest.raises(RuntimeError, match="eval-mode"):
        m.detect_nms_pipelined(x)


def test_no_cpu_path_behind_the_new_calls():
    """The product path fails loudly without the HIP device: there is no CPU fallback behind detect_nms / detect_nms_pipelined."""
    if torch.cuda.is_available():
        pytest.skip("runs where there is no GPU")
    m = _model()
    m.reuse_output_buffers(True)
    x = torch.zeros(2, 3, 64, 64, dtype=torch.uint8)
    for call in (lambda: m.detect_nms(x), lambda: m.detect_nms_pipelined(x)):
        with pytest.raises(Exception) as ei:
            call()
        assert "HIP" in str(ei.value) or "cuda" in str(ei.value).lower() or "device" in str(ei.value).lower()
