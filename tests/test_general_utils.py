"""Host-side helpers around the path (reference general.py:234-268 and the undefined box utilities its CLIs import)."""
import torch

from skyeye.utils.general import check_img_size, make_divisible, scale_boxes, xywh2xyxy, xyxy2xywh


def test_make_divisible_and_check_img_size():
    assert make_divisible(641, 32) == 672 and make_divisible(640, 32) == 640          # general.py:234-245
    assert check_img_size(640, s=32) == 640 and check_img_size(650, stride=32) == 672 # both spellings (D10)
    assert check_img_size([100, 640], s=torch.tensor([8, 16, 32])) == [128, 640]


def test_box_conversions_roundtrip():
    b = torch.tensor([[10.0, 20.0, 4.0, 6.0], [100.0, 50.0, 30.0, 10.0]])
    c = xywh2xyxy(b)
    assert torch.equal(c, torch.tensor([[8.0, 17.0, 12.0, 23.0], [85.0, 45.0, 115.0, 55.0]]))
    assert torch.allclose(xyxy2xywh(c), b)


def test_scale_boxes_inverts_letterbox():
    # 300x400 image letterboxed to 640x640: gain 1.6, pad (0, 80)
    boxes = torch.tensor([[160.0, 240.0, 320.0, 400.0]])
    out = scale_boxes((640, 640), boxes, (300, 400))
    assert torch.allclose(out, torch.tensor([[100.0, 100.0, 200.0, 200.0]]))
    assert scale_boxes((640, 640), torch.tensor([[-50.0, 0.0, 9999.0, 9999.0]]), (300, 400)).tolist() == [[0.0, 0.0, 400.0, 300.0]]
