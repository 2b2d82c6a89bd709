"""Test-time augmentation and large-frame tiling around the batch path.

The reference plumbs ``augment=`` from both CLIs into ``model(img, augment=...)`` (validate.py:245, detect.py:140) and ships
``scale_img`` (utils/torch_utils.py:262-288), but ``SkyEyeDetector.forward`` (detector.py:300-324) has no body for the flag.
The schedule here is the one that signature comes from (YOLOv5 ``_forward_augment``): scales 1 / 0.83 / 0.67, the middle pass
flipped left-right, each pass de-scaled / un-flipped and concatenated along the detection axis.  Tiling is the aerial use
case of the same batch path: overlapping windows of a large frame -> one batch -> rows shifted back -> one NMS.

Every per-element step runs in libskyeye_hip.so (``sky_scale_img``, ``sky_map_detections``, ``sky_tile_gather``); this module is
geometry and plumbing.  No CPU path."""
import ctypes

import numpy as np
import torch

from .. import _native as N
from .metrics import _handle, non_max_suppression
from .torch_utils import scale_img

TTA_SCALES = (1.0, 0.83, 0.67)
TTA_FLIPS = (None, 3, None)


def map_detections(det, scale=1.0, flip=None, img_hw=(0, 0), origins=None, tiles_per_image=1, rows=None, out=None, out_row0=0):
    """Rows ``rows = (first, count)`` (default all) of det [B, N, no] mapped back to the original frame: ``[..., :4] /= scale``;
    flip 3: ``cx = img_w - cx``; flip 2: ``cy = img_h - cy``; ``origins`` (int32 [B, 2] = (y, x) on the device): ``cx += x``,
    ``cy += y``.  Written to ``out[b // tiles_per_image, out_row0 + (b % tiles_per_image) * count + r]`` (allocated when None)."""
    if not (torch.is_tensor(det) and det.is_cuda and det.dim() == 3 and det.dtype == torch.float32):
        raise N.SkyEyeNativeError("map_detections: det must be a float32 [B, N, no] tensor on the HIP device (no CPU path)")
    det = det.contiguous()
    B, Nrows, no = det.shape
    row0, count = (0, Nrows) if rows is None else rows
    if out is None:
        out = torch.empty((B // tiles_per_image, out_row0 + tiles_per_image * count, no), dtype=torch.float32, device=det.device)
    if not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.dim() == 3 and out.shape[2] == no
            and out.shape[0] * tiles_per_image == B):
        raise ValueError(f"map_detections: out {tuple(out.shape)} does not fit det {tuple(det.shape)} / {tiles_per_image} tiles per image")
    if origins is not None and not (origins.is_cuda and origins.dtype == torch.int32 and origins.is_contiguous()
                                    and tuple(origins.shape) == (B, 2)):
        raise ValueError("map_detections: origins must be a contiguous int32 [B, 2] tensor on the device")
    h = _handle(det.device.index or 0)
    stream = torch.cuda.current_stream(det.device).cuda_stream
    N.check(h.L.sky_map_detections(h.h, det.data_ptr(), B, Nrows, no, row0, count, float(scale), int(flip or 0), float(img_hw[0]), float(img_hw[1]),
                                   origins.data_ptr() if origins is not None else None, tiles_per_image, out.data_ptr(), out.shape[1], out_row0,
                                   ctypes.c_void_p(stream)), h.h)
    return out


def clip_rows(n_rows, n_levels=3):
    """YOLOv5 ``_clip_augmented``: (first, count) kept per pass -- the full-scale pass drops its coarsest level (its last
    n / (4^0 + .. + 4^(nl-1)) rows), the smallest pass its finest level (its first n / g * 4^(nl-1) rows)."""
    g = sum(4 ** k for k in range(n_levels))
    keep = [(0, n) for n in n_rows]
    keep[0] = (0, n_rows[0] - n_rows[0] // g)
    first = (n_rows[-1] // g) * 4 ** (n_levels - 1)
    keep[-1] = (first, n_rows[-1] - first)
    return keep


def forward_augment(forward, x, gs=32, scales=TTA_SCALES, flips=TTA_FLIPS, clip=False):
    """``forward(xi) -> detections [B, N_i, no]`` on every scaled / flipped copy of ``x`` [B, 3, H, W] (uint8 or float);
    returns the concatenation [B, sum N_i, no] in the frame of ``x``."""
    H, W = x.shape[2:]
    dets = [forward(scale_img(x, s, gs=gs, flip=f)) for s, f in zip(scales, flips)]
    keep = clip_rows([d.shape[1] for d in dets]) if clip else [(0, d.shape[1]) for d in dets]
    out = torch.empty((x.shape[0], sum(k[1] for k in keep), dets[0].shape[2]), dtype=torch.float32, device=x.device)
    at = 0
    for d, s, f, k in zip(dets, scales, flips, keep):
        map_detections(d, s, f, (H, W), rows=k, out=out, out_row0=at)
        at += k[1]
    return out


# ----------------------------------------------------------------------------- tiling
def tile_origins(h0, w0, tile_h, tile_w, overlap=0.2):
    """(y, x) corners of the overlapping windows covering an h0 x w0 frame, row-major: step = int(tile * (1 - overlap)); the
    last window of each axis is pulled back flush with the border; an axis shorter than the tile has the single origin 0."""
    def axis(n, t):
        if n <= t:
            return [0]
        step = max(int(t * (1.0 - overlap)), 1)
        return list(range(0, n - t, step)) + [n - t]
    return np.array([(y, x) for y in axis(h0, tile_h) for x in axis(w0, tile_w)], np.int32)


def tile_gather(frame, origins, tile_h, tile_w, chw=False, pad=114, reverse_channels=False):
    """frame uint8 [H0, W0, 3] (or [3, H0, W0] with ``chw``) on the device -> uint8 [n, 3, tile_h, tile_w] (the engine's input);
    ``origins`` int32 [n, 2] on the device; outside the frame = ``pad``."""
    if not (torch.is_tensor(frame) and frame.is_cuda and frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[0 if chw else 2] == 3):
        raise N.SkyEyeNativeError("tile_gather: frame must be a uint8 [H, W, 3] (or [3, H, W]) tensor on the HIP device (no CPU path)")
    frame = frame.contiguous()
    H0, W0 = (frame.shape[1], frame.shape[2]) if chw else (frame.shape[0], frame.shape[1])
    n = origins.shape[0]
    out = torch.empty((n, 3, tile_h, tile_w), dtype=torch.uint8, device=frame.device)
    h = _handle(frame.device.index or 0)
    stream = torch.cuda.current_stream(frame.device).cuda_stream
    N.check(h.L.sky_tile_gather(h.h, frame.data_ptr(), H0, W0, int(chw), origins.data_ptr(), n, out.data_ptr(), tile_h, tile_w, int(pad),
                                int(reverse_channels), ctypes.c_void_p(stream)), h.h)
    return out


@torch.no_grad()
def detect_tiled(model, frame, tile=1280, overlap=0.2, batch=None, chw=False, reverse_channels=False, conf_thres=0.25, iou_thres=0.45,
                 classes=None, agnostic=False, max_det=1000, nms_mode="corrected", return_raw=False):
    """Detect on a frame larger than the network input: overlapping ``tile`` x ``tile`` windows -> batches of ``batch`` windows
    through ``model`` -> rows shifted by the window origin -> ONE non_max_suppression over the whole frame.

    Returns the NMS rows of the frame (``[k, 6]``: in "corrected" mode x1, y1, x2, y2, conf, cls in frame pixels), or with
    ``return_raw`` the merged ``[1, n_tiles * N, no]`` tensor and the origins."""
    th, tw = (tile, tile) if isinstance(tile, int) else tile
    H0, W0 = (frame.shape[1], frame.shape[2]) if chw else (frame.shape[0], frame.shape[1])
    org = tile_origins(H0, W0, th, tw, overlap)
    n = len(org)
    batch = n if batch is None else max(1, min(int(batch), n))
    n_pad = -(-n // batch) * batch
    org_pad = np.concatenate([org, np.repeat(org[-1:], n_pad - n, 0)], 0)          # one planned batch size for every chunk
    origins = torch.from_numpy(org_pad).to(frame.device)
    tiles = tile_gather(frame, origins, th, tw, chw=chw, reverse_channels=reverse_channels)
    merged = None
    for i in range(0, n_pad, batch):
        det = model(tiles[i:i + batch])
        det = det[0] if isinstance(det, (tuple, list)) else det
        valid = min(batch, n - i)
        if merged is None:
            merged = torch.empty((1, n * det.shape[1], det.shape[2]), dtype=torch.float32, device=frame.device)
        map_detections(det[:valid], origins=origins[i:i + valid], tiles_per_image=valid, out=merged, out_row0=i * det.shape[1])
    if return_raw:
        return merged, org
    return non_max_suppression(merged, conf_thres, iou_thres, classes, agnostic, max_det=max_det, mode=nms_mode)[0]


# ----------------------------------------------------------------------------- tiling across ranks ("Tiled (C5)")
# The tiles of ONE frame may sit on different GPUs.  The path is two-stage whatever the number of ranks, so that its result
# does not depend on it ("1-GPU tiled == 8-GPU tiled, exactly"):
#   1. every rank: its contiguous shard of the tiles (skyeye.distributed.shard_bounds) -> forward -> per-tile NMS (<= tile_max_det
#      survivors per tile, corner rows x1, y1, x2, y2, conf, cls) -> sky_offset_boxes moves them into frame coordinates;
#   2. ONE all-gather of the fixed-capacity survivor blocks (RCCL over xGMI; every rank contributes ceil(n_tiles / world) blocks,
#      the missing ones empty), which keeps the tiles in row-major order;
#   3. the frame's owner rank: cross-tile NMS over the gathered rows (sky_nms mode 2, "rows").
def tile_shard(n_tiles, rank, world):
    """-> (lo, hi, per): this rank's contiguous tiles [lo, hi) and the common block count per rank of the gather."""
    from ..distributed import shard_bounds
    lo, hi = shard_bounds(n_tiles, rank, world)
    return lo, hi, -(-n_tiles // world)


@torch.no_grad()
def tile_survivors(model, frame, origins_np, tile_hw, per, batch=None, chw=False, reverse_channels=False, conf_thres=0.25, iou_thres=0.45,
                   classes=None, agnostic=False, tile_max_det=300):
    """Stage 1 for the tiles at ``origins_np`` (int32 [t, 2], t <= per): -> (rows [per, tile_max_det, 7], counts [per]) on the
    device, boxes in FRAME pixels; blocks t .. per-1 are empty (zero rows, count 0)."""
    from .metrics import nms_raw
    th, tw = tile_hw
    dev = frame.device
    rows = torch.zeros((per, tile_max_det, 7), dtype=torch.float32, device=dev)
    counts = torch.zeros((per,), dtype=torch.int32, device=dev)
    t = len(origins_np)
    if t == 0:
        return rows, counts
    batch = t if batch is None else max(1, min(int(batch), t))
    t_pad = -(-t // batch) * batch
    org_pad = np.concatenate([origins_np, np.repeat(origins_np[-1:], t_pad - t, 0)], 0).astype(np.int32)
    origins = torch.from_numpy(org_pad).to(dev)
    tiles = tile_gather(frame, origins, th, tw, chw=chw, reverse_channels=reverse_channels)
    h = _handle(dev.index or 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for i in range(0, t_pad, batch):
        det = model(tiles[i:i + batch])
        det = det[0] if isinstance(det, (tuple, list)) else det
        r, c = nms_raw(det, conf_thres, iou_thres, classes, agnostic, max_detections=tile_max_det, mode="corrected")
        valid = min(batch, t - i)
        r, c = r[:valid].contiguous(), c[:valid].contiguous()
        N.check(h.L.sky_offset_boxes(h.h, r.data_ptr(), c.data_ptr(), valid, tile_max_det, 7, origins[i:i + valid].data_ptr(),
                                     ctypes.c_void_p(stream)), h.h)
        rows[i:i + valid] = r
        counts[i:i + valid] = c
    return rows, counts


def merge_tile_survivors(rows, counts, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=1000):
    """Stage 3: cross-tile NMS over gathered survivor blocks rows [T, R, 7] -> [k, 6] (x1, y1, x2, y2, conf, cls).  Empty
    blocks and the zero rows past each count have conf 0 and never pass the threshold."""
    from .metrics import non_max_suppression
    flat = rows.reshape(1, -1, rows.shape[-1]).contiguous()
    return non_max_suppression(flat, conf_thres, iou_thres, classes, agnostic, max_det=max_det, mode="rows")[0]


@torch.no_grad()
def detect_tiled_sharded(model, frame, tile=1280, overlap=0.2, batch=None, chw=False, reverse_channels=False, conf_thres=0.25,
                         iou_thres=0.45, classes=None, agnostic=False, max_det=1000, tile_max_det=300, group=None, owner=None):
    """Tiled detection of one frame with its tiles sharded over the ranks of ``group`` (every rank holds ``frame``; None or an
    uninitialised process group = one rank).  ``owner``: the rank that runs the cross-tile NMS and returns the [k, 6] rows (the
    others return None); None = every rank computes the (identical) result.  The result is the same for any world size."""
    import torch.distributed as dist
    from ..distributed import all_gather_detections
    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    th, tw = (tile, tile) if isinstance(tile, int) else tile
    H0, W0 = (frame.shape[1], frame.shape[2]) if chw else (frame.shape[0], frame.shape[1])
    org = tile_origins(H0, W0, th, tw, overlap)
    lo, hi, per = tile_shard(len(org), rank, world)
    rows, counts = tile_survivors(model, frame, org[lo:hi], (th, tw), per, batch, chw, reverse_channels, conf_thres, iou_thres, classes,
                                  agnostic, tile_max_det)
    if world > 1:
        rows, counts = all_gather_detections(rows, counts, group)
    if owner is not None and rank != owner:
        return None
    return merge_tile_survivors(rows, counts, conf_thres, iou_thres, classes, agnostic, max_det)
