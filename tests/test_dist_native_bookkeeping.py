"""The N-rank path's BOOKKEEPING without hardware (CPU): tactics2d_amd.dist.NativeGather and the bench's fragment alignment
(bench.Runner.steps_chain with align = True) driven against a fake pool that enforces what t2d_gather enforces -- a gather of K
steps only where the pool's step count is a multiple of K, K dividing the record ring with at least two fragments in it -- so
that the first real 8-GPU run can fail for hardware reasons only.  (No reference counterpart: SURVEY 8e is a new design.)"""
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")


class FakePool:
    """the calls NativeGather / Runner make, with t2d_gather's own argument checks (t2d_api.hip t2d_gather)"""
    RING = 64

    def __init__(self, n_env, start_steps=0):
        self.n_env, self.steps = n_env, start_steps
        self.calls = []          # ("step_n", first step, n) / ("gather", step count, n_steps, out ptr)

    def step_count(self):
        return self.steps

    def bind_actions(self, a0, a1, stride=1, extent=None):
        pass

    def step_n(self, n, interval_ms, act_step_stride=0, stream=None):
        assert 1 <= n <= self.RING
        self.calls.append(("step_n", self.steps, n))
        self.steps += n

    def step(self, interval_ms, stream=None):
        self.step_n(1, interval_ms, 0, stream)

    def gather(self, n_steps, out_ptr, stream=None, comm=None):
        assert n_steps >= 1 and self.RING % n_steps == 0 and self.RING // n_steps >= 2, n_steps
        assert self.steps > 0 and self.steps % n_steps == 0, (self.steps, n_steps)   # "n_steps divides the steps taken so far"
        self.calls.append(("gather", self.steps, n_steps, out_ptr))

    def gather_wait(self, stream=None, block_host=False):
        self.calls.append(("wait", self.steps))


def _runner(pool, n_participants=640):
    import bench
    a = types.SimpleNamespace(data_ptr=lambda: 4096, numel=lambda: 32 * n_participants)
    r = types.SimpleNamespace(pool=pool, a0=a, a1=a, k=0, align=True, ring_bound=False, stream=types.SimpleNamespace(cuda_stream=None),
                              scene=types.SimpleNamespace(interval_ms=100, n=n_participants))
    r.steps_chain = types.MethodType(bench.Runner.steps_chain, r)
    r.steps_single = types.MethodType(bench.Runner.steps_single, r)
    return r


@pytest.mark.parametrize("every,start,phases", [(8, 0, (20, 25)), (32, 0, (100, 25)), (16, 5, (20, 25, 64)), (4, 3, (1, 2, 3, 30)),
                                                (32, 3000 % 64, (100, 1000))])
def test_fragments_end_where_gathers_are_due_and_buffers_alternate(every, start, phases):
    """warm-up + timed regions of arbitrary lengths on a pool that has stepped before (the clock ramp): every chained fragment
    ends on a multiple of `every`, a gather is issued exactly there, into buffers 0, 1, 0, ..."""
    from tactics2d_amd import dist as D
    pool = FakePool(10, start)
    g = D.NativeGather(pool, world=8, every=every)
    assert g.out[0].shape == (8, every, 10, 2) and g.out[0].dtype == torch.int32
    run = _runner(pool)
    issued = []

    def hook():
        k = g.launch(None, None)
        if k is not None:
            issued.append((pool.steps, k))

    for n in phases:
        run.steps_chain(n, every, hook)
    assert pool.steps == start + sum(phases) and run.k == sum(phases)
    frags = [c for c in pool.calls if c[0] == "step_n"]
    assert all(1 <= c[2] <= every for c in frags)
    for c in frags:   # a fragment never crosses a multiple of `every`
        assert c[1] // every == (c[1] + c[2] - 1) // every, c
    due = [s for s in range(start + 1, pool.steps + 1) if s % every == 0]
    assert [s for s, _ in issued] == due
    assert [k for _, k in issued] == [i & 1 for i in range(len(due))]
    gathers = [c for c in pool.calls if c[0] == "gather"]
    assert [c[3] for c in gathers] == [g.out[i & 1].data_ptr() for i in range(len(due))]
    # result(k) of the last fragment waits on the host first, then unpacks the LAST step's records of every rank
    if issued:
        g.out[issued[-1][1]].zero_()
        rw, st = g.result(issued[-1][1])
        assert pool.calls[-1][0] == "wait" and rw.shape == (80,) and st.shape == (80, 4)


def test_one_launch_per_step_mode_gathers_every_k_steps():
    from tactics2d_amd import dist as D
    pool = FakePool(4, 7)
    g = D.NativeGather(pool, world=2, every=8)
    run = _runner(pool)
    got = []
    run.steps_single(30, lambda: got.append((pool.steps, g.launch(None, None))))
    assert [s for s, k in got if k is not None] == [8, 16, 24, 32]
    assert [k for s, k in got if k is not None] == [0, 1, 0, 1]


@pytest.mark.parametrize("every", [0, 3, 48, 64])
def test_fragment_lengths_the_ring_cannot_hold_are_rejected(every):
    from tactics2d_amd import dist as D
    with pytest.raises(ValueError):
        D.NativeGather(FakePool(4), world=2, every=every)
    with pytest.raises(ValueError):
        D.ResultGather(torch.zeros((64, 4, 2), dtype=torch.int32), 2, every=every)


def test_shards_are_contiguous_equal_and_cover_the_job():
    from tactics2d_amd import dist as D
    for total, world in ((8192, 8), (4096, 4), (12, 3)):
        spans = [D.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert len({hi - lo for lo, hi in spans}) == 1
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 4)
    assert [D.shard_range(10, r, 4, allow_uneven=True) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
