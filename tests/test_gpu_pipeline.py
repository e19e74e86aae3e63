"""Env groups on separate streams (tactics2d_amd/pipeline.py): same results as one pool, step for step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("groups", [2, 4])
def test_env_groups_equal_the_single_pool(groups):
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pipeline import EnvGroups
    from tactics2d_amd.pool import ParticipantPool
    sc = S.mixed(48, 32, seed=9)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    acts = [sc.sample_actions(rng) for _ in range(6)]
    fields = (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_STATUS, L.F_REWARD, L.F_CNT_STEP, L.F_ENV_FLAGS)

    one = ParticipantPool(sc.n_env, sc.A)
    sc.load(one); one.set_auto_reset(True)
    want = []
    for a0, a1 in acts:
        one.set_actions(a0, a1); one.step(100)
        want.append([one.download(f) for f in fields])
    one.close()

    eg = EnvGroups(sc, groups)
    eg.configure(lambda p: p.set_auto_reset(True))
    try:
        for k, (a0, a1) in enumerate(acts):
            t0, t1 = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
            eg.bind_actions(t0, t1)
            eg.fork()                     # the uploads above ran on the current stream
            eg.step(100)
            eg.join()
            torch.cuda.current_stream().synchronize()
            for f, w in zip(fields, want[k]):
                assert np.array_equal(eg.download(f), w), (k, f)
        rec = eg.download(L.F_RECORD)
        assert rec.shape == (L.RECORD_RING, sc.n_env, 2)
    finally:
        eg.close()
    with pytest.raises(ValueError):
        EnvGroups(sc, 5)
