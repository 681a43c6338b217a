"""LaunchTape bookkeeping without a GPU (mis_hip/lib.py): what is recorded, what is not, replay order and failure."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))


class _FakeLib:
    """Three kinds of entry points: a launch (status 0, last argument a stream), a refused launch (MIS_ERR_UNSUPPORTED: the
    caller falls through to another entry point) and a query without a stream."""

    def __init__(self):
        self.log = []
        self.fail_next = False

    def mis_launch(self, x, stream):
        self.log.append(("launch", x))
        if self.fail_next:
            return 3
        return 0

    def mis_refused(self, x, stream):
        self.log.append(("refused", x))
        return -2

    def mis_query(self, x):
        self.log.append(("query", x))
        return 0


def test_launch_tape_records_only_successful_launches_and_replays_them_in_order():
    from mis_hip import lib
    fake = _FakeLib()
    rec = lib._RecordingLib(fake)
    st = lib.StreamPtr(0)
    tape = lib.LaunchTape()
    assert rec.mis_launch(1, st) == 0 and len(tape) == 0            # not recording: calls go through, nothing is kept
    with tape.recording():
        assert lib.TAPE is tape
        assert rec.mis_launch(2, st) == 0
        assert rec.mis_refused(3, st) == -2                           # launched nothing: must not be replayed
        assert rec.mis_query(4) == 0                                  # no stream argument: not a launch
        assert rec.mis_launch(5, st) == 0
        with pytest.raises(RuntimeError, match="already being recorded"):
            with lib.LaunchTape().recording():
                pass
    assert lib.TAPE is None
    assert [a[0] for _, a in tape.items] == [2, 5] and len(tape) == 2
    fake.log.clear()
    tape.replay()
    tape.replay()
    assert fake.log == [("launch", 2), ("launch", 5)] * 2
    # a launch that fails during a replay surfaces with its entry point's name, and nothing after it runs
    fake.log.clear()
    fake.fail_next = True
    with pytest.raises(RuntimeError, match="mis_launch failed with status 3"):
        tape.replay()
    assert fake.log == [("launch", 2)]


def test_a_failed_recording_leaves_no_tape_behind():
    from mis_hip import lib
    tape = lib.LaunchTape()
    with pytest.raises(ValueError):
        with tape.recording():
            raise ValueError("the eager step failed")
    assert lib.TAPE is None
    with tape.recording():                                            # a new recording can start
        pass
