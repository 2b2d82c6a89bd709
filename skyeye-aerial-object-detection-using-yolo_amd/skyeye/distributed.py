"""Multi-GPU data path: batch sharding + one fused all-gather of the post-NMS boxes (RCCL over xGMI).

The detection path has no cross-image term (the reference loops over images even inside NMS, metrics.py:400), so a
batch shards over the 8 GPUs of a node as independent units: contiguous slices, weights replicated, every rank runs
forward + decode + per-image NMS locally.  The only exchange step is the one the north star names -- gathering the
fixed-capacity box buffers so every rank (or the frame's owner in tiled mode) holds all results.  The payload is tiny
(B_local x max_det x 7 floats + B_local counts, ~270 KB per rank at B_local = 32), i.e. latency-bound on xGMI, so it
is sent as ONE fused buffer.  One process per GPU; torch.distributed backend "nccl" is RCCL on ROCm ("gloo" in the
CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` independent images for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_detections(rows, counts):
    """[B, max_det, 7] float32 + [B] int32 -> one [B, max_det*7 + 1] float32 buffer; the count travels as float(count)
    (exact up to 2^24; a bit-cast int32 would be a denormal that a flush-to-zero copy on the way may erase)."""
    B = rows.shape[0]
    packed = torch.empty((B, rows.shape[1] * rows.shape[2] + 1), dtype=torch.float32, device=rows.device)
    packed[:, :-1] = rows.reshape(B, -1)
    packed[:, -1] = counts.to(torch.float32)
    return packed


def unpack_detections(packed, max_det, cols=7):
    rows = packed[:, :-1].reshape(packed.shape[0], max_det, cols)
    counts = packed[:, -1].to(torch.int32)
    return rows, counts


def all_gather_detections(rows, counts, group=None):
    """Every rank contributes its [B_local, max_det, 7] rows / [B_local] counts and receives the rank-ordered
    concatenation ([world*B_local, max_det, 7], [world*B_local]).  Equal B_local on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return rows, counts
    world = dist.get_world_size(group)
    if world == 1:
        return rows, counts
    packed = pack_detections(rows, counts)
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    if dist.get_backend(group) == "gloo" and not packed.is_cuda:
        # the CPU test backend: list form (gloo has no all_gather_into_tensor for CPU tensors on every build)
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
        out = torch.stack(parts, 0)
    else:
        # RCCL: ONE fused collective; an error here is an RCCL / xGMI failure and must surface, not be retried on another path
        dist.all_gather_into_tensor(out, packed, group=group)
    return unpack_detections(out.reshape(world * packed.shape[0], -1), rows.shape[1], rows.shape[2])


def detections_to_list(rows, counts, cols):
    """Device buffers -> the reference's list-of-tensors format (metrics.py:457)."""
    host = counts.cpu().tolist()
    return [rows[i, :n, :cols] for i, n in enumerate(host)]
