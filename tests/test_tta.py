"""Test-time augmentation + tiling front end.

CPU: the oracle's scale_img against tests/golden/tta.npz (the reference's own scale_img, torch_utils.py:262-288, run in the
build container), geometry arithmetic, tile cover, de-scaling properties.  GPU: sky_scale_img / sky_map_detections /
sky_tile_gather through the C ABI bit for bit against the oracle, ``model(x, augment=True)`` and ``detect_tiled`` against the
same steps composed by hand around the engine's own forward."""
import os

import numpy as np
import pytest
import torch

from cases import TTA_CASES
from seeded import seeded_input, seeded_scene

from skyeye.utils import tta as T
from skyeye.utils.torch_utils import scale_img, scale_img_geometry

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tta.npz"))
# fp32 bilinear on [0, 1) inputs: ATen's CPU kernel adds the four taps in another order -> a few ulp of 1.0 (observed 1.8e-7)
SCALE_IMG_ATOL = 5e-7


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


def _case_input(name):
    B, H, W = TTA_CASES[name][:3]
    return seeded_input("tta." + name, (B, 3, H, W), 7)


def _expected(name):
    if name.endswith("_big"):
        return tuple(G[name + ".shape"]), G[name + ".rows"]
    return G[name].shape, G[name]


# ----------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", list(TTA_CASES))
def test_oracle_scale_img_matches_reference_fixture(name):
    O = _oracle()
    B, H, W, ratio, flip, same, gs = TTA_CASES[name]
    y = O.scale_img(_case_input(name), ratio, same_shape=same, gs=gs, flip=flip)
    shape, ref = _expected(name)
    assert y.shape == tuple(shape)
    got = y[:, :, ::37] if name.endswith("_big") else y
    np.testing.assert_allclose(got, ref, rtol=0, atol=SCALE_IMG_ATOL)
    if ratio == 1.0:
        assert np.array_equal(got, ref)
    else:                                                   # the border is the literal 0.447
        s, p = O.scale_img_geometry(H, W, ratio, same, gs)
        assert (y[:, :, s[0]:] == np.float32(0.447)).all() and (y[:, :, :, s[1]:] == np.float32(0.447)).all()


def test_scale_img_geometry_is_the_reference_arithmetic():
    O = _oracle()
    assert scale_img_geometry(1280, 1280, 0.83) == ((1062, 1062), (1088, 1088))
    assert scale_img_geometry(1280, 1280, 0.67) == ((857, 857), (864, 864))
    assert scale_img_geometry(640, 480, 0.83, same_shape=True) == ((531, 398), (640, 480))
    assert scale_img_geometry(160, 224, 0.67, gs=64) == ((107, 150), (128, 192))
    for h, w, r in ((1280, 1280, 0.83), (97, 61, 0.67), (640, 480, 0.5)):
        assert scale_img_geometry(h, w, r) == O.scale_img_geometry(h, w, r)


def test_tile_origins_cover_the_frame():
    O = _oracle()
    for (h0, w0, th, tw, ov) in ((3000, 4000, 1280, 1280, 0.2), (1280, 1280, 1280, 1280, 0.2), (500, 2000, 640, 640, 0.25),
                                 (1281, 1280, 1280, 1280, 0.0), (300, 400, 128, 96, 0.5)):
        org = T.tile_origins(h0, w0, th, tw, ov)
        assert np.array_equal(org, O.tile_origins(h0, w0, th, tw, ov)) and org.dtype == np.int32
        cover = np.zeros((h0, w0), bool)
        for y, x in org:
            assert y >= 0 and x >= 0 and (y + th <= h0 or h0 <= th) and (x + tw <= w0 or w0 <= tw)
            cover[y:y + th, x:x + tw] = True
        assert cover.all()
        assert len({tuple(o) for o in org}) == len(org)
    assert T.tile_origins(3000, 4000, 1280, 1280).shape == (12, 2)


def test_clip_rows_follow_the_level_ratio():
    # three levels at strides 8 / 16 / 32: rows per level 16 : 4 : 1
    n = [3 * (160 * 160 + 80 * 80 + 40 * 40), 3 * (136 * 136 + 68 * 68 + 34 * 34), 3 * (108 * 108 + 54 * 54 + 27 * 27)]
    keep = T.clip_rows(n)
    assert keep[0] == (0, n[0] - 3 * 40 * 40) and keep[1] == (0, n[1]) and keep[2] == (3 * 108 * 108, n[2] - 3 * 108 * 108)


def test_oracle_map_detections_round_trip():
    O = _oracle()
    r = np.random.default_rng(3)
    d = r.uniform(0, 640, (2, 50, 15)).astype(np.float32)
    assert np.array_equal(O.map_detections(d), d)
    m = O.map_detections(d, 0.5, 3, (480, 640))
    assert np.array_equal(m[..., 0], np.float32(640) - d[..., 0] / np.float32(0.5)) and np.array_equal(m[..., 4:], d[..., 4:])
    org = np.array([[10, 20], [300, 400]], np.int32)
    m = O.map_detections(d, origins=org)
    assert np.array_equal(m[1, :, 0], d[1, :, 0] + np.float32(400)) and np.array_equal(m[0, :, 1], d[0, :, 1] + np.float32(10))


def test_no_cpu_path():
    with pytest.raises(Exception):
        scale_img(torch.zeros(1, 3, 64, 64), 0.5)
    with pytest.raises(Exception):
        T.map_detections(torch.zeros(1, 4, 15))
    with pytest.raises(Exception):
        T.tile_gather(torch.zeros(64, 64, 3, dtype=torch.uint8), torch.zeros(1, 2, dtype=torch.int32), 32, 32)


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TTA_CASES))
def test_scale_img_kernel_matches_oracle_and_fixture(name):
    O = _oracle()
    B, H, W, ratio, flip, same, gs = TTA_CASES[name]
    x = _case_input(name)
    y = scale_img(torch.from_numpy(x).cuda(), ratio, same_shape=same, gs=gs, flip=flip).cpu().numpy()
    assert np.array_equal(y, O.scale_img(x, ratio, same_shape=same, gs=gs, flip=flip)), "kernel and oracle share one expression"
    shape, ref = _expected(name)
    assert y.shape == tuple(shape)
    np.testing.assert_allclose(y[:, :, ::37] if name.endswith("_big") else y, ref, rtol=0, atol=SCALE_IMG_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ratio,flip,gs", [((2, 3, 96, 130), 0.83, 3, 32), ((1, 3, 61, 47), 0.67, 2, 32), ((1, 1, 40, 40), 1.0, 3, 32),
                                                 ((1, 3, 64, 64), 1.5, 0, 32)])
def test_scale_img_uint8_and_ragged(shape, ratio, flip, gs):
    O = _oracle()
    u = np.random.default_rng(shape[2]).integers(0, 256, shape, dtype=np.uint8)
    y = scale_img(torch.from_numpy(u).cuda(), ratio, gs=gs, flip=flip)
    ref = O.scale_img(u.astype(np.float32) / np.float32(255.0), ratio, gs=gs, flip=flip)
    assert y.dtype == torch.float32 and np.array_equal(y.cpu().numpy(), ref)
    yf = scale_img(torch.from_numpy(u.astype(np.float32) / np.float32(255.0)).cuda(), ratio, gs=gs, flip=flip)
    assert torch.equal(y, yf)                                # uint8 == float input, bit for bit
    x = torch.from_numpy(u).cuda()
    assert scale_img(x, 1.0) is x


@pytest.mark.gpu
def test_map_detections_kernel_bit_exact():
    O = _oracle()
    r = np.random.default_rng(9)
    d = r.uniform(0, 900, (4, 333, 15)).astype(np.float32)
    dg = torch.from_numpy(d).cuda()
    for scale, flip in ((1.0, None), (0.83, 3), (0.67, 2), (0.5, None)):
        got = T.map_detections(dg, scale, flip, (608, 800)).cpu().numpy()
        assert np.array_equal(got, O.map_detections(d, scale, flip or 0, (608, 800)))
    org = np.array([[0, 0], [0, 512], [384, 0], [384, 512]], np.int32)
    got = T.map_detections(dg, origins=torch.from_numpy(org).cuda(), tiles_per_image=4).cpu().numpy()
    assert got.shape == (1, 4 * 333, 15) and np.array_equal(got[0], O.map_detections(d, origins=org).reshape(-1, 15))
    # a row window of the source into a row window of a larger destination; everything else untouched
    out = torch.full((4, 500, 15), -1.0, device="cuda")
    T.map_detections(dg, 0.83, 3, (608, 800), rows=(33, 100), out=out, out_row0=50)
    o = out.cpu().numpy()
    assert np.array_equal(o[:, 50:150], O.map_detections(d[:, 33:133], 0.83, 3, (608, 800))) and (o[:, :50] == -1).all() and (o[:, 150:] == -1).all()
    with pytest.raises(Exception):
        T.map_detections(dg, rows=(300, 100))                # past the last row


@pytest.mark.gpu
@pytest.mark.parametrize("h0,w0,th,tw,chw,rev", [(300, 400, 128, 128, False, False), (100, 90, 128, 96, False, True), (257, 131, 64, 50, True, False)])
def test_tile_gather_kernel_bit_exact(h0, w0, th, tw, chw, rev):
    O = _oracle()
    f = np.random.default_rng(h0).integers(0, 256, (h0, w0, 3), dtype=np.uint8)
    org = T.tile_origins(h0, w0, th, tw, 0.25)
    src = torch.from_numpy(f.transpose(2, 0, 1).copy() if chw else f).cuda()
    got = T.tile_gather(src, torch.from_numpy(org).cuda(), th, tw, chw=chw, reverse_channels=rev).cpu().numpy()
    assert np.array_equal(got, O.tile_gather(f, org, th, tw, reverse_channels=rev))


def _model(prec="fp32"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from bench import build_model
    return build_model("skyeye_s", prec, torch.device("cuda", 0))[0]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_forward_augment_equals_hand_composition(prec):
    """model(x, augment=True) == the 1 / 0.83-flipped / 0.67 passes run one by one through the same engine on the oracle's
    scale_img, mapped back by the oracle's arithmetic, concatenated (bit for bit: every stage is deterministic)."""
    O = _oracle()
    model = _model(prec)
    u = seeded_scene(2, 256, 320, seed=5)
    x = torch.from_numpy(u).cuda()
    y, none = model(x, augment=True)
    assert none is None

    def fwd(xi):
        return model(torch.from_numpy(xi).cuda())[0].cpu().numpy()
    ref = O.tta_forward(fwd, u.astype(np.float32) / np.float32(255.0), gs=32)
    n1 = 3 * (32 * 40 + 16 * 20 + 8 * 10)
    s2, p2 = scale_img_geometry(256, 320, 0.83)
    s3, p3 = scale_img_geometry(256, 320, 0.67)
    n2, n3 = (3 * sum((p[0] // s) * (p[1] // s) for s in (8, 16, 32)) for p in (p2, p3))
    assert y.shape == (2, n1 + n2 + n3, 15) and ref.shape == tuple(y.shape)
    assert np.array_equal(y.cpu().numpy(), ref)
    # pass 1 is the plain forward
    assert torch.equal(y[:, :n1], model(x)[0])
    # clipped form keeps the documented row windows
    yc = T.forward_augment(lambda xi: model(xi)[0], x, clip=True)
    keep = T.clip_rows([n1, n2, n3])
    assert yc.shape[1] == sum(k[1] for k in keep)
    assert torch.equal(yc[:, :keep[0][1]], y[:, :keep[0][1]]) and torch.equal(yc[:, -keep[2][1]:], y[:, -keep[2][1]:])


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [None, 5])
def test_detect_tiled_equals_hand_composition(batch):
    O = _oracle()
    from skyeye.utils.metrics import non_max_suppression
    model = _model("fp32")
    frame = np.ascontiguousarray(seeded_scene(1, 300, 420, seed=13)[0].transpose(1, 2, 0))      # HWC uint8
    org = T.tile_origins(300, 420, 128, 160, 0.2)
    assert len(org) == 12                                   # batch 5 -> three chunks, the last one padded with repeated windows
    merged, org2 = T.detect_tiled(model, torch.from_numpy(frame).cuda(), tile=(128, 160), overlap=0.2, batch=batch, return_raw=True)
    assert np.array_equal(org, org2)
    tiles = O.tile_gather(frame, org, 128, 160)
    det = np.concatenate([model(torch.from_numpy(tiles[i:i + 1]).cuda())[0].cpu().numpy() for i in range(len(org))], 0)
    # per-tile batch-1 forwards vs one batched forward: the engine's kernels do not mix batch entries
    ref = O.map_detections(det, origins=org).reshape(1, -1, det.shape[2])
    assert merged.shape == ref.shape
    np.testing.assert_allclose(merged.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    rows = T.detect_tiled(model, torch.from_numpy(frame).cuda(), tile=(128, 160), overlap=0.2, batch=batch, conf_thres=0.25)
    want = non_max_suppression(merged, 0.25, 0.45, max_det=1000, mode="corrected")[0]
    assert torch.equal(rows, want) and rows.shape[1] == 6
