"""The one-forward-in-flight guard of the patches (tome/patch/_common.py), CPU side: the bookkeeping happens at the ENTRY
of a patched forward, under a per-device lock -- a second host thread that starts a forward while the first is still
enqueueing waits, and then orders its stream behind the first forward's end event.  Streams and events are fakes (no
GPU call); the GPU behaviour itself is covered by test_models_gpu.py
(test_second_forward_on_another_stream_is_ordered_not_concurrent)."""
import threading
import warnings

import pytest
import torch

from tome.patch import _common


class FakeEvent:
    def __init__(self, stream):
        self.stream, self.done = stream, False

    def query(self):
        return self.done


class FakeStream:
    def __init__(self, sid):
        self.sid, self.waited = sid, []

    def wait_event(self, ev):
        self.waited.append(ev)


class FakeDevice:
    type, index = "cuda", 0


@pytest.fixture
def fakes(monkeypatch):
    tls = threading.local()
    streams = {}

    def current(device):
        sid = getattr(tls, "sid", 1)
        return sid, streams.setdefault(sid, FakeStream(sid))
    monkeypatch.setattr(_common, "_guarded", lambda device: True)
    monkeypatch.setattr(_common, "_current_stream", current)
    monkeypatch.setattr(_common, "_record_event", lambda stream: FakeEvent(stream))
    monkeypatch.setattr(_common, "_warned_two_streams", False)
    _common._in_flight.clear()
    _common._issue_locks.clear()
    yield tls, streams
    _common._in_flight.clear()
    _common._issue_locks.clear()


def test_second_thread_waits_at_the_entry_and_is_ordered_behind_the_first(fakes):
    tls, streams = fakes
    dev = FakeDevice()
    a_inside, a_may_leave, b_entered = threading.Event(), threading.Event(), threading.Event()
    order = []

    def thread_a():
        tls.sid = 1
        with _common.one_forward_at_a_time(dev):
            order.append("a in")
            a_inside.set()
            assert a_may_leave.wait(10)
            order.append("a out")

    def thread_b():
        tls.sid = 2
        assert a_inside.wait(10)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            with _common.one_forward_at_a_time(dev):
                order.append("b in")
                b_entered.set()

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start()
    tb.start()
    assert a_inside.wait(10)
    # B is at the entry while A is still enqueueing: it must NOT get in (the old guard let it through here, because A
    # had recorded nothing yet)
    assert not b_entered.wait(0.3)
    a_may_leave.set()
    ta.join(10)
    tb.join(10)
    assert order == ["a in", "a out", "b in"]
    # and once in, B's stream was ordered behind A's end event (recorded on A's stream at A's exit)
    assert len(streams[2].waited) == 1 and streams[2].waited[0].stream is streams[1]
    assert _common._in_flight[0][0] == 2  # B's own end event is the newest entry


def test_same_stream_and_finished_forwards_pass_without_a_wait(fakes):
    tls, streams = fakes
    dev = FakeDevice()
    with _common.one_forward_at_a_time(dev):
        with _common.one_forward_at_a_time(dev):  # re-entrant: a patched model inside a patched forward
            pass
    assert streams[1].waited == []
    _common._in_flight[0][1].done = True  # the forward has finished on the device
    tls.sid = 3
    with _common.one_forward_at_a_time(dev):
        pass
    assert streams[3].waited == []


def test_a_forward_that_raises_still_leaves_its_end_event_and_frees_the_lock(fakes):
    tls, streams = fakes
    dev = FakeDevice()
    with pytest.raises(ZeroDivisionError):
        with _common.one_forward_at_a_time(dev):
            1 / 0
    assert _common._in_flight[0][0] == 1 and not _common._in_flight[0][1].done
    tls.sid = 2
    with pytest.warns(RuntimeWarning, match="ordered behind"):
        with _common.one_forward_at_a_time(dev):
            pass
    assert len(streams[2].waited) == 1


def test_raise_mode_refuses_and_frees_the_lock(fakes, monkeypatch):
    tls, streams = fakes
    dev = FakeDevice()
    with _common.one_forward_at_a_time(dev):
        pass
    tls.sid = 2
    monkeypatch.setenv("TOME_ONE_FORWARD", "raise")
    with pytest.raises(RuntimeError, match="one forward at a time"):
        with _common.one_forward_at_a_time(dev):
            pass
    monkeypatch.setenv("TOME_ONE_FORWARD", "order")
    acquired = _common._issue_lock(dev).acquire(timeout=1)  # the refused entry did not keep the lock
    assert acquired
    _common._issue_lock(dev).release()


def test_cpu_models_are_not_guarded():
    ctx = _common.one_forward_at_a_time(torch.device("cpu"))
    assert ctx.on is False
    with ctx:
        pass
    assert _common.one_forward_at_a_time(None).on is False
