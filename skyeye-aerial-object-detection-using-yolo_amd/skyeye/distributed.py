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


class BoxExchange:
    """The exchange step with every buffer allocated ONCE (per geometry): nothing is allocated, packed or unpacked per step, so the
    step is a fixed sequence of launches on fixed addresses (capturable in a hipGraph together with the forward pass and the NMS).

    Layout of a rank's block (int32 storage; a collective only moves bytes, no arithmetic ever touches the words):
        image b:  [ max_det * cols float32 rows | int32 count ]           = max_det * cols + 1 words, images back to back
    ``rows`` / ``counts`` are strided VIEWS of the local block: hand them to ``nms_raw(..., out=ex.rows, counts=ex.counts)`` (or to
    ``detect_nms``) and the NMS kernel writes the block in place.  ``gather()`` sends it with one ``all_gather_into_tensor`` (RCCL)
    into the [world * B_local] block buffer; ``all_rows`` / ``all_counts`` are views of that buffer in rank order: the result of the
    1-process run, bit for bit.  world == 1 (or no process group): the views of the local block, no copy."""

    def __init__(self, b_local, max_det=300, cols=7, device="cpu", group=None, always_collective=False):
        """``always_collective``: run the collective at world size 1 too (readiness tests on one GPU: the RCCL call the 8-GPU run makes)."""
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.collective = self.world > 1 or (inited and always_collective)
        self.b_local, self.max_det, self.cols = int(b_local), int(max_det), int(cols)
        self.words = self.max_det * self.cols + 1
        self.local = torch.zeros((self.b_local, self.words), dtype=torch.int32, device=device)
        self.rows, self.counts = self._views(self.local)
        if self.collective:
            self.gathered = torch.zeros((self.world * self.b_local, self.words), dtype=torch.int32, device=device)
            self.all_rows, self.all_counts = self._views(self.gathered)
        else:
            self.gathered, self.all_rows, self.all_counts = self.local, self.rows, self.counts

    def _views(self, blocks):
        n = blocks.shape[0]
        rows = blocks.view(torch.float32).as_strided((n, self.max_det, self.cols), (self.words, self.cols, 1))
        counts = blocks.as_strided((n,), (self.words,), self.words - 1)
        return rows, counts

    def gather(self):
        """-> (all_rows [world * B_local, max_det, cols], all_counts [world * B_local]); asynchronous on the current stream."""
        if self.collective:
            if dist.get_backend(self.group) == "gloo" and not self.local.is_cuda:
                parts = list(self.gathered.view(self.world, self.b_local, self.words).unbind(0))      # views: gloo writes in place
                dist.all_gather(parts, self.local, group=self.group)
            else:
                # RCCL: ONE fused collective; an error here is an RCCL / xGMI failure and must surface
                dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        return self.all_rows, self.all_counts


def detections_to_list(rows, counts, cols):
    """Device buffers -> the reference's list-of-tensors format (metrics.py:457)."""
    host = counts.cpu().tolist()
    return [rows[i, :n, :cols] for i, n in enumerate(host)]
