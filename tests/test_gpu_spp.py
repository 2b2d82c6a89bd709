"""SPPBlock (reference blocks.py:126-149) on the two pooling paths of the engine: the single-launch pyramid kernel (maps of up
to 2048 pixels: the three cascaded 5 x 5 max pools on an LDS-resident plane) and the three separable launches (larger
maps), both against the CPU oracle on seeded inputs.  max is exact, so the tolerance is the 1 x 1 convolutions' alone."""
import os

import numpy as np
import pytest
import torch

from helpers import load_seeded, seeded_state_for
from parity import close
from seeded import seeded_input

import skyeye.core.models as M

pytestmark = pytest.mark.gpu


def _oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import skyeye_oracle as O
    return O


@pytest.mark.parametrize("shape,cin,cout", [((2, 64, 40, 40), 64, 64), ((1, 64, 45, 45), 64, 96), ((3, 32, 7, 5), 32, 32),
                                            ((1, 64, 48, 48), 64, 64), ((1, 128, 33, 70), 128, 64)],
                         ids=["pyramid_40", "pyramid_45_odd", "pyramid_tiny", "separable_48", "separable_33x70"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_spp_paths_match_oracle(shape, cin, cout, prec):
    O = _oracle()
    m = load_seeded(M.SPPBlock(cin, cout), 31).set_precision(prec)
    P = seeded_state_for(M.SPPBlock(cin, cout), 31)
    x = seeded_input("spp." + "x".join(map(str, shape)), shape, 31, -1.0, 1.0)
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = O.spp(P, "", x)
    assert got.shape == ref.shape
    if prec == "fp32":
        close(got, ref, 1e-4)
    else:
        assert np.abs(got - ref).max() <= 0.05 * max(1.0, np.abs(ref).max())


def test_pyramid_equals_three_pools_bit_for_bit(tmp_path):
    """The same seeded block and input through the three separable launches (SKY_NO_SPP_PYRAMID=1, read once per process, hence a
    child process) and through the pyramid kernel: identical bits in both precisions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path[:0] = [{os.path.join(root, 'tests')!r}, {os.path.join(root, 'tests', 'golden')!r}, "
        f"{os.path.join(root, 'skyeye-aerial-object-detection-using-yolo_amd')!r}]\n"
        "from helpers import load_seeded\nfrom seeded import seeded_input\nimport skyeye.core.models as M\n"
        "x = torch.from_numpy(seeded_input('spp.eq', (2, 64, 40, 40), 5, -1.0, 1.0)).cuda()\n"
        "for prec in ('fp32', 'bf16'):\n"
        "    m = load_seeded(M.SPPBlock(64, 64), 5).set_precision(prec)\n"
        "    np.save(sys.argv[1] + prec + '.npy', m(x).cpu().numpy())\n")
    env = dict(os.environ, SKY_NO_SPP_PYRAMID="1")
    subprocess.run([sys.executable, "-c", code, str(tmp_path) + os.sep], check=True, env=env, timeout=300)
    x = torch.from_numpy(seeded_input("spp.eq", (2, 64, 40, 40), 5, -1.0, 1.0)).cuda()
    for prec in ("fp32", "bf16"):
        m = load_seeded(M.SPPBlock(64, 64), 5).set_precision(prec)
        assert np.array_equal(m(x).cpu().numpy(), np.load(str(tmp_path) + os.sep + prec + ".npy"))
