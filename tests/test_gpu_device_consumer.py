"""A consumer on the same GPU: `BatchedGridEnvironment.step_device` (gs_step_device_ptr / gs_step_device_view) with PyTorch-ROCm
as the policy side.  PyTorch is the CONSUMER here, as a learner would be -- the product path neither imports nor needs it;
the test skips where torch or its GPU build is missing."""
import numpy as np
import pytest

import grid_fed_rl_gym_amd as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("B,use_stream", [(96, True), (96, "side"), (96, False), (8192, True), (8192, "side")])
def test_a_policy_on_the_gpu_steps_the_environment_without_host_copies(B, use_stream):
    """actions = tanh(obs @ W) computed by torch on the device from the observation block the step left there, handed back
    as a device pointer: every observation, reward and flag equals the host-array path driven with the same actions.
    With `use_stream` the two sides are ordered by events on the device only (no host synchronisation in the loop): True = torch's
    default stream -- the LEGACY default stream, handle 0, which crosses the C ABI as hipStreamLegacy (round 3: it used to be
    taken for "no stream", and once in a dozen runs the step read its actions before tanh had written them) --, "side" = a
    stream torch created."""
    if not torch.cuda.is_available():
        pytest.skip("torch without a GPU")
    fs = P.ieee123_like()
    kw = dict(num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True)
    dev_env, host_env = P.BatchedGridEnvironment(fs, **kw), P.BatchedGridEnvironment(fs, **kw)
    seeds = np.arange(B, dtype=np.uint64) + 11
    obs0, _ = dev_env.reset(seed=seeds)
    host_env.reset(seed=seeds)
    gen = torch.Generator(device="cpu").manual_seed(3)
    W = (torch.randn(fs.obs_dim, fs.action_dim, generator=gen, dtype=torch.float64) * 0.05).to("cuda")
    side = torch.cuda.Stream() if use_stream == "side" else None
    if side is not None:
        torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream().cuda_stream if use_stream else None
    assert (stream == 0) == (use_stream is True)
    obs_t = torch.as_tensor(obs0, device="cuda")                      # the first observation comes from reset()
    for t in range(5):
        actions = torch.tanh(obs_t @ W).contiguous()                 # [B, A] float64 on the device
        if not use_stream:
            torch.cuda.synchronize()
        obs_d, rew_d, term_d, trunc_d = dev_env.step_device(actions, stream=stream)
        obs_t = torch.as_tensor(obs_d, device="cuda")
        rew_t, term_t, trunc_t = (torch.as_tensor(x, device="cuda") for x in (rew_d, term_d, trunc_d))
        assert obs_t.data_ptr() == obs_d.ptr                          # zero copy
        o, r, te, tr, _ = host_env.step(actions.cpu().numpy())
        assert np.array_equal(obs_t.cpu().numpy(), o), t
        assert np.array_equal(rew_t.cpu().numpy(), r) and np.array_equal(term_t.cpu().numpy().astype(bool), te)
        assert np.array_equal(trunc_t.cpu().numpy().astype(bool), tr)
    if side is not None:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
    dev_env.close(); host_env.close()


def test_rollout_buffers_are_readable_on_the_gpu_without_a_copy():
    """`rollout_device_arrays`: the [T + 1, B, obs_dim] observation sequence, actions, rewards, flags and the terminal-observation
    side list as torch tensors over the library's own device memory; equal to what `rollout_download` copies to the host
    (next_observations = obs_seq[1:] with the terminal observations scattered in)."""
    if not torch.cuda.is_available():
        pytest.skip("torch without a GPU")
    fs = P.ieee123_like(); B, T = 96, 30
    env = P.BatchedGridEnvironment(fs, num_envs=B, solver="fbs", stochastic_loads=True, weather_variation=True, episode_length=7)   # episodes end inside the rollout
    env.reset(seed=np.arange(B, dtype=np.uint64) + 2)
    h = env.handle
    h.rollout(T, "random", seed=9)
    dev = {k: torch.as_tensor(v, device="cuda") for k, v in h.rollout_device_arrays().items() if v.shape[0] > 0}
    host = h.rollout_download()
    assert host["n_terminal"] > 0 and dev["terminal_index"].shape[0] == host["n_terminal"]
    assert np.array_equal(dev["obs_seq"][:-1].cpu().numpy(), host["observations"])
    assert np.array_equal(dev["actions"].cpu().numpy(), host["actions"]) and np.array_equal(dev["rewards"].cpu().numpy(), host["rewards"])
    assert np.array_equal(dev["terminals"].cpu().numpy(), host["terminals"])
    nxt = dev["obs_seq"][1:].clone()
    idx = dev["terminal_index"].long()
    nxt[idx[:, 0], idx[:, 1]] = dev["terminal_obs"]                  # what gs_rollout_download does on the host
    assert np.array_equal(nxt.cpu().numpy(), host["next_observations"])
    env.close()
