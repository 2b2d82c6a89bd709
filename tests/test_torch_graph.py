"""tests/torch_graph.py (the torch-CPU comparator of bench.py's cpu_baseline) computes the same detections as the oracle."""
import numpy as np
import torch

from helpers import detector_params
from oracle import skyeye_oracle as O
from seeded import seeded_scene
from torch_graph import detector_forward


def test_torch_graph_matches_oracle():
    P = detector_params("skyeye_s")
    frames = seeded_scene(2, 96, 128, 5).astype(np.float32) / np.float32(255.0)
    ref, _ = O.detector_forward(P, frames, 10)
    Pt = {k: torch.from_numpy(np.asarray(v)) for k, v in P.items() if np.asarray(v).dtype != np.int64}
    got = detector_forward(Pt, torch.from_numpy(frames), 10).numpy()
    assert got.shape == ref.shape
    scale = np.maximum(np.abs(ref).max(axis=(0, 1), keepdims=True), 1.0)
    assert float((np.abs(got - ref) / scale).max()) < 2e-4
    assert (got[..., 5:].argmax(-1) == ref[..., 5:].argmax(-1)).mean() > 0.999
