"""BASELINE.json config 4 (the batch sharded over N ranks with a device-side all-gather of the observation blocks)
exercised with N > 1 on ONE GPU through the in-process transport (gs_comm_init_loopback): every rank's shard is a handle
of this process, built exactly as rank r of an N-GPU job builds it (first_instance = r * B, global-index seeds), and the
exchange runs the RCCL transport's device code -- compaction of the changing columns, slot offsets, expansion into
[world * B][obs_dim], constant columns written once, the two rotating observation buffers and their events -- with
device-to-device copies where ncclAllGather would cross xGMI.  The comparator is a single handle that owns all world * B
instances: every member's gathered block must equal its observation block bit for bit."""
import ctypes

import numpy as np
import pytest

import grid_fed_rl_gym_amd as P
from grid_fed_rl_gym_amd._lib import Handle
from grid_fed_rl_gym_amd.components import PowerFlowError

pytestmark = pytest.mark.gpu

KW = dict(solver="fbs", stochastic_loads=True, weather_variation=True)


def _whole(spec, total, seeds, acts, **kw):
    env = P.BatchedGridEnvironment(spec, num_envs=total, **dict(KW, **kw))
    env.reset(seed=seeds)
    env.handle.upload_actions(acts)
    return env


def _members(spec, world, B, seeds, acts, **kw):
    grp = P.LoopbackShards(spec, world * B, world, **dict(KW, **kw))
    for s in grp.shards:
        s.env.reset(seed=seeds[s.start:s.stop])
        s.env.handle.upload_actions(acts[:, s.start:s.stop])
    return grp


@pytest.mark.parametrize("maker,world,B,kw", [
    (lambda: P.ieee123_like(), 2, 96, {}),                     # ragged shards (96 = 1.5 slab groups): the padding of a shard must not travel
    (lambda: P.ieee123_like(), 8, 96, {}),
    (lambda: P.ieee123_like(), 8, 64, {"solver": "nr"}),
    (lambda: P.ieee13_like("epsilon"), 4, 40, {"solver": "nr"}),
])
def test_gathered_block_of_every_member_equals_one_handle_that_owns_the_whole_batch(maker, world, B, kw):
    spec = maker(); total = world * B; K = 5
    rng = np.random.default_rng(world * 1000 + B)
    acts = rng.uniform(-1, 1, (K, total, spec.action_dim))
    seeds = np.arange(total, dtype=np.uint64) * 31 + 7
    whole = _whole(spec, total, seeds, acts, **kw)
    grp = _members(spec, world, B, seeds, acts, **kw)
    hs = grp.handles
    order = rng.permutation(world)                              # members reach the collective in any order
    want_prev = None
    for t in range(K):
        whole.handle.step_device(t)
        for h in hs:
            h.step_device(t)                                    # step t is queued while gather t - 1 may still be running ...
        if want_prev is not None:                               # ... and gather t - 1 must still deliver step t - 1's observations
            for r in (0, world - 1):
                assert np.array_equal(hs[r].allgather_obs_download(), want_prev), (t, r)
        want = whole.handle.download_step()["obs"]
        if t % 2 == 0:
            for r in order:
                hs[r].allgather_obs(to_host=False)              # per-member calls, completes with the last one
        else:
            Handle.allgather_obs_shards(hs, to_host=False)      # SURVEY 8(b)'s form: one call for all members
        want_prev = want
    for r in range(world):
        got = hs[r].allgather_obs_download()
        assert got.shape == (total, spec.obs_dim)
        assert np.array_equal(got, want), r
        # and the member's own step output is its slice
        assert np.array_equal(hs[r].download_step()["obs"], want[r * B:(r + 1) * B])
    # the host-output form: every member hands a buffer, all are filled when the last member has called
    whole.handle.step_device(0)
    want = whole.handle.download_step()["obs"]
    bufs = []
    for h in hs:
        h.step_device(0)
    for r in range(world - 1):
        bufs.append(np.full((total, spec.obs_dim), np.nan))
        hs[r]._check(hs[r]._lib.gs_allgather_obs(hs[r]._h, bufs[-1].ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    last = hs[world - 1].allgather_obs(to_host=True)
    for b in bufs + [last]:
        assert np.array_equal(b, want)
    whole.close(); grp.close()


def test_gather_survives_masked_resets_checkpoints_and_rollouts_between_steps():
    """Everything else an N-rank job does between two exchanges: a masked reset of finished instances, a checkpoint round
    trip, a device rollout -- each rewrites the observation buffers the gather reads."""
    spec = P.ieee123_like(); world, B = 4, 64; total = world * B
    rng = np.random.default_rng(99)
    acts = rng.uniform(-1, 1, (4, total, spec.action_dim))
    seeds = np.arange(total, dtype=np.uint64) + 500
    whole = _whole(spec, total, seeds, acts)
    grp = _members(spec, world, B, seeds, acts)
    hs = grp.handles
    mask = (rng.random(total) < 0.3).astype(np.uint8)

    def both(fn_whole, fn_member):
        fn_whole(whole)
        for s in grp.shards:
            fn_member(s)

    def check():
        want = whole.handle.download_step()["obs"]
        full = Handle.allgather_obs_shards(hs, to_host=True)
        assert np.array_equal(full, want)
        for h in hs[1:]:
            assert np.array_equal(h.allgather_obs_download(), want)

    both(lambda e: e.handle.step_device(0), lambda s: s.env.handle.step_device(0)); check()
    both(lambda e: e.handle.reset(seeds + np.uint64(9), mask, want_obs=False),
         lambda s: s.env.handle.reset(seeds[s.start:s.stop] + np.uint64(9), mask[s.start:s.stop], want_obs=False))
    both(lambda e: e.handle.step_device(1), lambda s: s.env.handle.step_device(1)); check()
    both(lambda e: e.set_state(e.get_state()), lambda s: s.env.set_state(s.env.get_state()))
    both(lambda e: e.handle.step_device(2), lambda s: s.env.handle.step_device(2)); check()
    both(lambda e: e.handle.rollout(6, "random", seed=3), lambda s: s.env.handle.rollout(6, "random", seed=3))
    both(lambda e: e.handle.step_device(3), lambda s: s.env.handle.step_device(3)); check()
    whole.close(); grp.close()


def test_loopback_call_sequence_errors():
    spec = P.ieee13_like("epsilon")
    envs = [P.BatchedGridEnvironment(spec, num_envs=8, first_instance=8 * r, solver="nr") for r in range(2)]
    odd = P.BatchedGridEnvironment(spec, num_envs=16, solver="nr")
    with pytest.raises(PowerFlowError, match="equal shards|differs"):
        Handle.comm_init_loopback([envs[0].handle, odd.handle])
    with pytest.raises(PowerFlowError, match="before gs_comm_init"):
        envs[0].handle.allgather_obs()
    Handle.comm_init_loopback([e.handle for e in envs])
    with pytest.raises(PowerFlowError, match="already belongs"):
        Handle.comm_init_loopback([e.handle for e in envs])
    for e in envs:
        e.reset(seed=1)
    envs[0].handle.allgather_obs()
    with pytest.raises(PowerFlowError, match="twice"):
        envs[0].handle.allgather_obs()
    with pytest.raises(PowerFlowError, match="not complete"):
        envs[0].handle.allgather_obs_download()
    full = envs[1].handle.allgather_obs(to_host=True)           # completes the round; reset() observations are gatherable too
    assert full.shape == (16, spec.obs_dim) and np.isfinite(full).all()
    envs[1].handle.comm_destroy()
    with pytest.raises(PowerFlowError, match="has left"):
        envs[0].handle.allgather_obs()
    for e in envs + [odd]:
        e.close()


def test_full_size_two_ranks_with_the_split_step_and_a_pending_gather():
    """Two members of BASELINE's per-GPU batch (8192 each: the step goes out as two half-grid launches on two streams, both of
    which must wait for a pending gather before they rewrite its buffer) against one handle of 16384, 6 steps with the gather
    left pending across the following step."""
    spec = P.ieee123_like(); world, B = 2, 8192; total = world * B
    rng = np.random.default_rng(4)
    acts = rng.uniform(-1, 1, (3, total, spec.action_dim))
    seeds = np.arange(total, dtype=np.uint64) + 11
    whole = _whole(spec, total, seeds, acts)
    grp = _members(spec, world, B, seeds, acts)
    hs = grp.handles
    assert hs[0].describe()["step_launches"] == 2
    for t in range(6):
        whole.handle.step_device(t % 3)
        for h in hs:
            h.step_device(t % 3)
        Handle.allgather_obs_shards(hs, to_host=False)
    want = whole.handle.download_step()["obs"]
    for h in hs:
        assert np.array_equal(h.allgather_obs_download(), want)
    whole.close(); grp.close()


def test_gather_self_check_and_comm_info_through_the_loopback_transport():
    """The self-check an N-rank bench run prints (``gather_verified``: SHA-256 of every rank's own block against the slot it
    landed in on every rank, sharding.verify_gathered_block) rehearsed on one GPU, and gs_comm_info."""
    from grid_fed_rl_gym_amd.sharding import block_checksum
    spec = P.ieee123_like(); world, B = 4, 80
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, (2, world * B, spec.action_dim))
    seeds = np.arange(world * B, dtype=np.uint64) + 5
    grp = _members(spec, world, B, seeds, acts)
    hs = grp.handles
    for t in range(2):
        for h in hs:
            h.step_device(t)
        Handle.allgather_obs_shards(hs)
    sums = [block_checksum(h.download_step()["obs"]) for h in hs]
    assert len(set(sums)) == world                                  # the shards differ (global-index seeds), so a swap would show
    for h in hs:
        mine = h.allgather_obs_download()
        assert [block_checksum(mine[q * B:(q + 1) * B]) for q in range(world)] == sums
    for r, h in enumerate(hs):
        info = h.comm_info()
        assert info["transport"] == "loopback" and info["nranks"] == world and info["rank"] == r
        assert len(info["device_uuid"]) == 32
    grp.close()
