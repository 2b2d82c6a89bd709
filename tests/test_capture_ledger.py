"""CaptureLedger (skyeye/utils/torch_utils.py): the fork / join rules of a stream capture, checked without a device.  The topologies
are the ones the package enqueues under capture_graph (parallel_slices, detect_nms_pipelined) and the round-3 experiment whose capture
faulted in hipStreamEndCapture (experiments/detect_nms_chain.py)."""
from skyeye.utils.torch_utils import CaptureLedger


def _wait(L, dst, src):
    L.wait(dst, L.record(src))                      # torch: dst.wait_stream(src) = src.record_event() + dst.wait_event()


def test_fork_join_is_legal():
    L = CaptureLedger("cur")
    for s in ("s0", "s1"):
        _wait(L, s, "cur")                          # _run_sliced: fork
    for s in ("s0", "s1"):
        _wait(L, "cur", s)                          # join
    assert L.unjoined() == [] and not L.problems and L.captured == {"cur", "s0", "s1"}


def test_pipelined_nms_inside_slices_is_legal():
    """detect_nms_pipelined on top of parallel_slices: NMS side stream forked first, slices forked and joined, NMS joined last."""
    L = CaptureLedger("cur")
    _wait(L, "nms", "cur")
    for s in ("s0", "s1"):
        _wait(L, s, "cur")
    for s in ("s0", "s1"):
        _wait(L, "cur", s)
    _wait(L, "cur", "nms")
    assert L.unjoined() == [] and not L.problems


def _chain(L, n, join):
    for s in ("s0", "s1"):
        _wait(L, s, "cur")
    for k in range(n):
        for s in ("s0", "s1"):
            if k >= 2:
                _wait(L, s, "nms")
        for s in ("s0", "s1"):
            _wait(L, "nms", s)
    if join:
        _wait(L, "cur", "nms")


def test_round3_chain_is_an_unjoined_capture():
    """The form that faulted: three streams enter the capture, none is ordered before the caller's stream at the end -- for n = 2 already."""
    for n in (2, 4):
        L = CaptureLedger("cur")
        _chain(L, n, join=False)
        assert sorted(L.unjoined()) == ["nms", "s0", "s1"]
        assert bool(L.problems) == (n > 2)              # from the third step on the two-buffer chain also has the mutual waits (below)


def test_chain_with_the_final_join_is_legal():
    """One edge (caller waits for the NMS stream) orders every tail before the caller: the slices' last work is behind the last NMS."""
    L = CaptureLedger("cur")
    _chain(L, 2, join=True)
    assert L.unjoined() == [] and not L.problems
    L = CaptureLedger("cur")
    _chain(L, 4, join=True)                             # joined, and still refused: slice streams wait for the NMS stream and vice versa
    assert L.unjoined() == [] and L.problems and "mutual waits" in L.problems[0]


def test_work_after_the_join_needs_another_join():
    L = CaptureLedger("cur")
    _wait(L, "s0", "cur")
    _wait(L, "cur", "s0")
    _wait(L, "s0", "cur")                           # second fork, never joined
    assert L.unjoined() == ["s0"]


def test_wait_for_an_event_from_outside_the_capture():
    L = CaptureLedger("cur")
    tok = L.record("other")                         # a stream that is not capturing
    L.wait("cur", tok)
    assert L.problems and "outside the capture" in L.problems[0]
    L2 = CaptureLedger("cur")
    L2.wait("elsewhere", L2.record("other"))        # two streams outside the capture: none of the ledger's business
    assert not L2.problems and L2.captured == {"cur"}


def test_mutual_waits_between_forked_streams_are_flagged_before_the_edge_is_made():
    """Round 4 (experiments/stagger_probe.py): an 8-step chain with two detection buffers -- slice streams wait for the NMS of batch k - 2,
    the NMS stream waits for the slices -- is joined and legal, and this runtime's hipStreamEndCapture faults on it."""
    L = CaptureLedger("cur")
    for s in ("s0", "s1"):
        _wait(L, s, "cur")
    for k in range(2):
        for s in ("s0", "s1"):
            assert L.wait("nms", L.record(s)) is None        # forward edges only (slices -> NMS): fine
    msg = L.wait("s0", L.record("nms"))                      # the first back edge (k = 2): the slice stream waits for the NMS stream
    assert msg and "mutual waits" in msg and L.problems == [msg]
    # the shipped topologies have no edge between two forked streams at all
    L2 = CaptureLedger("cur")
    _wait(L2, "nms", "cur")
    for s in ("s0", "s1"):
        _wait(L2, s, "cur")
        _wait(L2, "cur", s)
    _wait(L2, "cur", "nms")
    assert not L2.problems
