"""The stochastic mode of the step (stochastic loads + weather), pinned without reference to its own author:

* the generator IS Philox4x32-10: the three known answers of the Random123 distribution (Salmon et al., SC'11,
  `kat_vectors`) through the NumPy oracle and the C oracle;
* the transforms give the distributions the reference draws from -- load noise N(0, 0.1) (dynamics.py:66-70),
  irradiance factor U(0.8, 1.2), wind walk N(0, 0.5), temperature noise N(0, 2), cloud walk N(0, 0.1)
  (grid_env.py:669-681) -- checked by moments, Kolmogorov-Smirnov distance and independence across the counter words,
  on the CPU restatements and (``-m gpu``) on what the HIP kernel realises for 65 536 instances in one step;
* reset without a seed runs the stream on (a new seed from the old one), reset(seed=k) is reproducible.
"""
import ctypes as C
import math

import numpy as np
import pytest
from scipy import stats

from oracle import oracle_c as OC
from oracle import oracle_np as O

KAT = [  # counter, key, expected output (Random123 kat_vectors, philox4x32 10 rounds)
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def _c_philox(ctr, key):
    lib = OC.lib()
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib.orc_philox4x32(c, k, o)
    return tuple(int(x) for x in o)


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox4x32_10_known_answers(ctr, key, want):
    assert O.philox4x32(ctr, key) == want
    assert _c_philox(ctr, key) == want


def _c_normals(seed, insts, steps, draw):
    lib = OC.lib()
    lib.orc_rng_normal.restype = C.c_double
    lib.orc_rng_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    return np.array([[[lib.orc_rng_normal(seed, i, s, draw, k) for k in range(4)] for s in steps] for i in insts])


def test_normal_quad_same_in_both_oracles_and_standard_normal():
    z = _c_normals(12345, range(2500), range(1, 5), O.DRAW_LOAD0)          # [2500, 4, 4] = 40 000 normals
    for i, s in [(0, 1), (7, 2), (2499, 4)]:
        ref = O.rng_normal_quad(12345, i, s, O.DRAW_LOAD0)
        assert np.allclose(z[i, s - 1], ref, rtol=0, atol=5e-15)
    flat = z.ravel()
    n = flat.size
    assert abs(flat.mean()) < 4.0 / math.sqrt(n)
    assert abs(flat.std() - 1.0) < 4.0 / math.sqrt(2 * n)
    assert abs(stats.skew(flat)) < 0.06 and abs(stats.kurtosis(flat)) < 0.12
    assert stats.kstest(flat, "norm").pvalue > 1e-3
    # the four components of a call, neighbouring instances, neighbouring steps: uncorrelated
    lim = 4.5 / math.sqrt(2500 * 4)
    comp = z.reshape(-1, 4)
    cc = np.corrcoef(comp.T)
    assert np.max(np.abs(cc - np.eye(4))) < lim
    assert abs(np.corrcoef(z[:-1, :, 0].ravel(), z[1:, :, 0].ravel())[0, 1]) < lim
    assert abs(np.corrcoef(z[:, :-1, 0].ravel(), z[:, 1:, 0].ravel())[0, 1]) < 4.5 / math.sqrt(2500 * 3)
    # Box-Muller pairs: cos and sin branch of one radius are uncorrelated but not independent in r; r^2 ~ chi2(2)
    r2 = comp[:, 0] ** 2 + comp[:, 1] ** 2
    assert stats.kstest(r2, "chi2", args=(2,)).pvalue > 1e-3


def test_uniform_pair_is_uniform_and_matches_between_oracles():
    lib = OC.lib()
    lib.orc_rng_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    u = np.empty((20000, 2))
    a, b = C.c_double(), C.c_double()
    for i in range(20000):
        lib.orc_rng_pair(99, i, 3, O.DRAW_IRRADIANCE, C.byref(a), C.byref(b))
        u[i] = a.value, b.value
    assert O.rng_uniform_pair(99, 17, 3, O.DRAW_IRRADIANCE) == (u[17, 0], u[17, 1])
    assert 0.0 < u.min() and u.max() < 1.0
    assert stats.kstest(u[:, 0], "uniform").pvalue > 1e-3 and stats.kstest(u[:, 1], "uniform").pvalue > 1e-3
    assert abs(np.corrcoef(u[:, 0], u[:, 1])[0, 1]) < 4.5 / math.sqrt(20000)


def test_different_draw_indices_and_seeds_are_independent_streams():
    a = _c_normals(1, range(4000), [5], O.DRAW_LOAD0)[:, 0, 0]
    b = _c_normals(1, range(4000), [5], O.DRAW_LOAD0 + 1)[:, 0, 0]
    c = _c_normals(2, range(4000), [5], O.DRAW_LOAD0)[:, 0, 0]
    w = _c_normals(1, range(4000), [5], O.DRAW_WEATHER)[:, 0, 0]
    lim = 4.5 / math.sqrt(4000)
    for x, y in [(a, b), (a, c), (a, w), (b, c)]:
        assert abs(np.corrcoef(x, y)[0, 1]) < lim


def test_next_episode_seed_chain():
    s0 = 42
    chain = [s0]
    for _ in range(2000):
        chain.append(O.next_episode_seed(chain[-1], 7))
    assert len(set(chain)) == len(chain)                               # no short cycle
    assert O.next_episode_seed(s0, 7) == chain[1]                      # deterministic
    assert O.next_episode_seed(s0, 8) != chain[1]                      # per instance
    assert all(0 <= s < 2 ** 64 for s in chain)
    # the new seed's stream is unrelated to the old one's
    lim = 4.5 / math.sqrt(3000)
    a = _c_normals(chain[0], range(3000), [1], O.DRAW_LOAD0)[:, 0, 0]
    b = _c_normals(chain[1], range(3000), [1], O.DRAW_LOAD0)[:, 0, 0]
    assert abs(np.corrcoef(a, b)[0, 1]) < lim


def test_rollout_random_actions_are_uniform_on_the_open_interval():
    a = np.array([O.rollout_random_actions(5, i, t, 8) for i in range(500) for t in range(8)])
    assert a.shape == (4000, 8) and -1.0 < a.min() and a.max() < 1.0
    assert abs(a.mean()) < 4.0 / math.sqrt(3 * a.size)
    assert abs(a.var() - 1.0 / 3.0) < 0.01
    assert stats.kstest(a.ravel(), "uniform", args=(-1.0, 2.0)).pvalue > 1e-3
    assert np.max(np.abs(np.corrcoef(a.T) - np.eye(8))) < 4.5 / math.sqrt(4000)
    assert O.rollout_random_actions(5, 3, 2, 5).tolist() == O.rollout_random_actions(5, 3, 2, 8)[:5].tolist()


# ---------------------------------------------------------------------------------------------------------------
# what the HIP kernel realises
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_one_step_of_65536_instances_has_the_reference_distributions():
    import grid_fed_rl_gym_amd as P
    fs = P.ieee123_like()
    B = 65536
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, tolerance=1e-6)
    env.reset(seed=np.arange(B, dtype=np.uint64) * 7 + 3)
    st = env.get_state()
    lay = env.state_layout()
    t0 = 11.5 * 3600.0
    st[:, lay["time"]] = t0
    st[:, lay["cloud"]] = 0.5                      # three sigma of the cloud walk away from both clamps
    env.set_state(st)
    before = env.get_state()
    env.step(np.zeros((B, env.action_dim)))
    after = env.get_state()
    t1 = t0 + env.timestep
    hour = (t1 / 3600.0) % 24.0
    # load model, dynamics.py:54-75: base * profile(hour) * (1 + N(0, 0.1)), clipped at 0 (never active at 10 sigma)
    prof = O.load_profile_power(t1, 1.0)[0]
    loadp = env.handle.debug_read_rows("LOADP")
    mult = loadp / (fs.load_base[None, :] * prof)
    n = mult.size
    assert abs(mult.mean() - 1.0) < 4.0 * 0.1 / math.sqrt(n)
    assert abs(mult.std() - 0.1) < 4.0 * 0.1 / math.sqrt(2 * n)
    sub = (mult[::8, ::7].ravel() - 1.0) / 0.1
    assert stats.kstest(sub, "norm").pvalue > 1e-4
    cc = np.corrcoef(mult[:, :8].T)                # loads 0..7 = two Philox calls, all four components of each
    assert np.max(np.abs(cc - np.eye(8))) < 4.5 / math.sqrt(B)
    assert abs(np.corrcoef(mult[:-1, 0], mult[1:, 0])[0, 1]) < 4.5 / math.sqrt(B)     # neighbouring instances
    # weather, grid_env.py:653-681
    base = 1000.0 * math.sin(math.pi * (hour - 6.0) / 12.0)
    fac = after[:, lay["irradiance"]] / base
    assert 0.8 <= fac.min() and fac.max() <= 1.2
    assert stats.kstest(fac, "uniform", args=(0.8, 0.4)).pvalue > 1e-4
    dw = after[:, lay["wind"]] - before[:, lay["wind"]]
    assert abs(dw.mean()) < 4.0 * 0.5 / math.sqrt(B) and abs(dw.std() - 0.5) < 4.0 * 0.5 / math.sqrt(2 * B)
    assert stats.kstest(dw / 0.5, "norm").pvalue > 1e-4
    dt_ = after[:, lay["temperature"]] - (25.0 + 10.0 * math.sin(2.0 * math.pi * (hour - 12.0) / 24.0))
    assert abs(dt_.mean()) < 4.0 * 2.0 / math.sqrt(B) and abs(dt_.std() - 2.0) < 4.0 * 2.0 / math.sqrt(2 * B)
    assert stats.kstest(dt_ / 2.0, "norm").pvalue > 1e-4
    dc = after[:, lay["cloud"]] - before[:, lay["cloud"]]
    assert abs(dc.mean()) < 4.0 * 0.1 / math.sqrt(B) and abs(dc.std() - 0.1) < 4.0 * 0.1 / math.sqrt(2 * B)
    assert stats.kstest(dc / 0.1, "norm").pvalue > 1e-4
    for x, y in [(dw, dt_), (dw, dc), (dt_, dc), (dw, fac), (dw, mult[:, 0])]:
        assert abs(np.corrcoef(x, y)[0, 1]) < 4.5 / math.sqrt(B)
    # and draw for draw what the oracle defines (instance 12345)
    b = 12345
    seed = int(b * 7 + 3)
    z = [O.rng_normal_quad(seed, b, 1, O.DRAW_LOAD0 + l // 4)[l & 3] for l in range(fs.n_loads)]
    want = np.array([O.load_profile_power(t1, fs.load_base[l], noise=0.1 * z[l])[0] for l in range(fs.n_loads)])
    assert np.allclose(loadp[b], want, rtol=1e-13, atol=0)
    env.close()


@pytest.mark.gpu
def test_gpu_reset_without_seed_runs_the_stream_on_and_seeded_reset_is_reproducible():
    import grid_fed_rl_gym_amd as P
    fs = P.ieee13_like("epsilon")
    B = 70
    env = P.BatchedGridEnvironment(fs, num_envs=B, stochastic_loads=True, weather_variation=True, first_instance=500)
    lay = env.state_layout()
    z = np.zeros((B, env.action_dim))

    def episode(seed):
        env.reset(seed=seed)
        st = env.get_state()
        return st[:, [lay["seed_lo"], lay["seed_hi"]]].copy(), [env.step(z)[0].copy() for _ in range(2)]

    s_a, ep_a = episode(np.arange(B, dtype=np.uint64) + 9)
    s_b, ep_b = episode(None)                       # the stream runs on: new seeds, new noise
    s_c, ep_c = episode(None)
    s_d, ep_d = episode(np.arange(B, dtype=np.uint64) + 9)
    assert np.array_equal(s_a, s_d) and all(np.array_equal(x, y) for x, y in zip(ep_a, ep_d))
    assert not np.array_equal(ep_a[0], ep_b[0]) and not np.array_equal(ep_b[0], ep_c[0])
    for b in (0, 33, 69):                           # the chain the oracle defines, keyed by the GLOBAL instance index
        want = O.next_episode_seed(int(b + 9), 500 + b)
        got = int(s_b[b, 0]) | (int(s_b[b, 1]) << 32)
        assert got == want
        got2 = int(s_c[b, 0]) | (int(s_c[b, 1]) << 32)
        assert got2 == O.next_episode_seed(want, 500 + b)
    # masked reset without seeds: only the masked instances move on
    env.reset(seed=np.arange(B, dtype=np.uint64) + 9)
    mask = np.zeros(B, dtype=np.uint8); mask[[3, 40]] = 1
    env.reset(seed=None, mask=mask)
    st = env.get_state()
    assert int(st[4, lay["seed_lo"]]) == 4 + 9 and int(st[3, lay["seed_lo"]]) != 3 + 9
    env.close()
