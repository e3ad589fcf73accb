"""SURVEY §8 N1 — the rollout policy's inference step as one launch (np_policy_act, neuralplane_amd/policy.py FusedPolicy) against the CPU
restatement (oracle/f16_actor.inc f16o_policy_act: bit for bit) and the reference's PPOPolicy.get_actions recording
(tests/golden/policy_kat.npz, tools/gen_golden.py gen_policy: tolerances of tests/policy_kat.py)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _t(x, dev='cuda:0'):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _setup(golden_dir, act_dim, numerics='fp32', obs_dim=22):
    from neuralplane_amd.policy import FusedPolicy, pack_policy_actor, pack_policy_critic
    from oracle.f16_oracle import PolicyOracle
    from tests.policy_kat import load
    g, sa, sc = load(golden_dir, act_dim, obs_dim)
    fp = FusedPolicy((sa, sc), device='cuda:0', numerics=numerics)
    wa, A, _ = pack_policy_actor(sa)
    assert fp.obs_dim == obs_dim
    o = PolicyOracle(wa, pack_policy_critic(sc), np.float32(fp.std), np.float32(fp.log_std), numerics, obs_dim)
    assert A == act_dim == fp.act_dim and same(np.float32(fp.std), g['std']) and same(np.float32(fp.log_std), g['log_std'])
    return g, sa, sc, fp, o


@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
@pytest.mark.parametrize('act_dim,obs_dim', [(4, 22), (3, 22), (4, 15)])
def test_get_actions_equals_the_restatement_bit_for_bit_and_the_reference_recording(golden_dir, act_dim, obs_dim, numerics):
    """Five chained get_actions calls on the recorded inputs and normal draws (recurrent states fed back on the device): every output equals
    the restatement's bit for bit (both numerics: the fp32 chains, f16_actor.inc, and the block fixed point, f16_actor_i8.inc) and the
    REFERENCE's recording within the bounds of tests/policy_kat.py; act(deterministic=True) and get_values are the same launch with one network."""
    from tests.policy_kat import check_step
    g, _, _, fp, o = _setup(golden_dir, act_dim, numerics, obs_dim)
    n = g['obs'].shape[1]
    ha = hc = torch.zeros((n, 1, 128), device='cuda:0')
    ha_o = hc_o = np.zeros((n, 128), np.float32)
    for t in range(g['obs'].shape[0]):
        obs, m = _t(g['obs'][t]), _t(g['masks'][t])
        mean, ha_det = fp.act(obs, ha, m, deterministic=True)
        v_only = fp.get_values(obs, hc, m)
        a_s, ha_s = fp.act(obs, ha, m, noise=_t(g['eps'][t]))
        values, actions, logp, ha, hc = fp.get_actions(obs, ha, hc, m, noise=_t(g['eps'][t]))
        assert values.shape == (n, 1) and actions.shape == (n, act_dim) and logp.shape == (n, 1) and ha.shape == hc.shape == (n, 1, 128)
        v_o, a_o, lp_o, ha_o2, hc_o2 = o.run(g['obs'][t], ha_o, hc_o, g['masks'][t], g['eps'][t])
        mean_o = o.run(g['obs'][t], ha_o, hc_o, g['masks'][t], flags=o.ACTOR | o.DETERMINISTIC)[1]
        ha_o, hc_o = ha_o2, hc_o2
        assert same(values.cpu().numpy(), v_o) and same(actions.cpu().numpy(), a_o) and same(logp.cpu().numpy(), lp_o), t
        assert same(ha.cpu().numpy()[:, 0], ha_o) and same(hc.cpu().numpy()[:, 0], hc_o), t
        assert same(mean.cpu().numpy(), mean_o) and same(ha_det.cpu().numpy(), ha.cpu().numpy()), t
        assert same(v_only.cpu().numpy(), v_o) and same(a_s.cpu().numpy(), a_o) and same(ha_s.cpu().numpy(), ha.cpu().numpy()), t
        assert np.max(np.abs(mean.cpu().numpy() - g['means'][t])) < 2e-5
        check_step(g, t, values.cpu().numpy(), actions.cpu().numpy(), logp.cpu().numpy(), ha.cpu().numpy(), hc.cpu().numpy())


@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
@pytest.mark.parametrize('n', [1, 31, 33, 1000, 20001])
def test_ragged_batches_and_wild_inputs_equal_the_restatement(golden_dir, n, numerics):
    """Batch sizes around the 32-row tile, large observations, |h| > 1, masked rows."""
    g, _, _, fp, o = _setup(golden_dir, 4, numerics)
    rng = np.random.RandomState(n)
    obs = (rng.normal(0, 1, (n, 22)) * rng.uniform(0.1, 30, (1, 22))).astype(np.float32)
    ha, hc = rng.normal(0, 0.7, (n, 128)).astype(np.float32), rng.normal(0, 0.7, (n, 128)).astype(np.float32)
    mk = (rng.uniform(0, 1, (n, 1)) > 0.2).astype(np.float32)
    eps = rng.normal(0, 1, (n, 4)).astype(np.float32)
    out = fp.get_actions(_t(obs), _t(ha), _t(hc), _t(mk), noise=_t(eps))
    ref = o.run(obs, ha, hc, mk, eps)
    for k, (x, y) in enumerate(zip(out, ref)):
        assert same(x.cpu().numpy().reshape(y.shape), y), k


def test_sampling_draws_from_torchs_generator_like_the_reference(golden_dir):
    """Without `noise` the draws are torch.randn's on this device: re-seeding reproduces the step, and the actions are
    fl(fl(randn * std) + mean) of exactly those draws (what the reference's FixedNormal.sample() computes: tools/gen_golden.py asserts it)."""
    g, _, _, fp, _ = _setup(golden_dir, 3)
    n = g['obs'].shape[1]
    obs, m = _t(g['obs'][0]), _t(g['masks'][0])
    h0 = torch.zeros((n, 1, 128), device='cuda:0')
    torch.manual_seed(11)
    eps = torch.randn((n, 3), device='cuda:0')
    torch.manual_seed(11)
    v1, a1, lp1, _, _ = fp.get_actions(obs, h0, h0.clone(), m)
    torch.manual_seed(11)
    v2, a2, lp2, _, _ = fp.get_actions(obs, h0, h0.clone(), m)
    mean, _ = fp.act(obs, h0, m, deterministic=True)
    assert torch.equal(a1, a2) and torch.equal(lp1, lp2) and torch.equal(v1, v2)
    assert torch.equal(a1, eps * torch.tensor(fp.std, device='cuda:0') + mean)
    # log-probabilities: torch.distributions.Normal on the same device, a tolerance (its reduction order is its own)
    ref = torch.distributions.Normal(mean, torch.tensor(fp.std, device='cuda:0')).log_prob(a1).sum(-1, keepdim=True)
    assert float((ref - lp1).abs().max()) < 5e-6


def test_refresh_follows_the_source_policy(golden_dir):
    """After an in-place update of the source networks refresh() (or auto_refresh at the next call) re-packs them; a stale FusedPolicy keeps the old weights."""
    from neuralplane_amd.policy import FusedPolicy
    from tests.policy_kat import load
    g, sa, sc = load(golden_dir, 4)

    class Net:
        def __init__(self, sd):
            self.p = {k: torch.nn.Parameter(torch.from_numpy(v.copy()).to('cuda:0')) for k, v in sd.items()}

        def state_dict(self):
            return {k: v.detach() for k, v in self.p.items()}

        def parameters(self):
            return self.p.values()

    class Pol:
        pass
    pol = Pol()
    pol.actor, pol.critic = Net(sa), Net(sc)
    stale, auto = FusedPolicy(pol, 'cuda:0'), FusedPolicy(pol, 'cuda:0', auto_refresh=True)
    fixed = FusedPolicy((sa, sc), 'cuda:0')
    obs, m = _t(g['obs'][0]), _t(g['masks'][0])
    h = torch.zeros((96, 1, 128), device='cuda:0')
    eps = _t(g['eps'][0])
    base = stale.get_actions(obs, h, h.clone(), m, noise=eps)
    assert all(torch.equal(x, y) for x, y in zip(auto.get_actions(obs, h, h.clone(), m, noise=eps), base))
    # the state_dict source differs only in std = exp(log_std), evaluated where the parameter lives (this GPU / the host): an ulp
    for x, y in zip(fixed.get_actions(obs, h, h.clone(), m, noise=eps), base):
        assert float((x - y).abs().max()) < 1e-5
    assert torch.equal(fixed.get_values(obs, h, m), base[0])
    with torch.no_grad():
        pol.actor.p['act.mlp.fc.0.bias'].add_(0.05)
        pol.critic.p['value_out.bias'].add_(1.0)
        pol.actor.p['act.action_out.log_std'].add_(0.1)
    out_auto = auto.get_actions(obs, h, h.clone(), m, noise=eps)
    assert auto.refreshes == 2 and not torch.equal(out_auto[1], base[1])
    assert torch.allclose(out_auto[0], base[0] + 1.0, atol=1e-5)
    assert all(torch.equal(x, y) for x, y in zip(stale.get_actions(obs, h, h.clone(), m, noise=eps), base)) and stale.refreshes == 1
    stale.refresh()
    assert all(torch.equal(x, y) for x, y in zip(stale.get_actions(obs, h, h.clone(), m, noise=eps), out_auto))
    assert abs(auto.log_std[0] - (float(g['log_std'][0]) + 0.1)) < 1e-6


def test_policy_act_argument_errors(golden_dir):
    from neuralplane_amd import _lib
    from neuralplane_amd.policy import ACTOR, CRITIC, NpPolicyStep
    _, _, _, fp, _ = _setup(golden_dir, 4)
    lib = _lib.load()
    n = 40
    bufs = {k: torch.zeros(s, device='cuda:0') for k, s in (('obs', (n, 22)), ('m', (n,)), ('eps', (n, 4)), ('ha', (n, 128)), ('hc', (n, 128)), ('ha2', (n, 128)),
                                                           ('hc2', (n, 128)), ('v', (n,)), ('a', (n, 4)), ('lp', (n,)))}

    def step(**kw):
        q = NpPolicyStep()
        q.n, q.act_dim, q.flags = n, 4, ACTOR | CRITIC
        q.actor_weights, q.critic_weights = fp.weights[0].data_ptr(), fp.weights[1].data_ptr()
        for j in range(4):
            q.std[j], q.log_std[j] = 0.5, -0.7
        q.obs, q.masks, q.noise = bufs['obs'].data_ptr(), bufs['m'].data_ptr(), bufs['eps'].data_ptr()
        q.rnn_states_actor_in, q.rnn_states_critic_in = bufs['ha'].data_ptr(), bufs['hc'].data_ptr()
        q.rnn_states_actor_out, q.rnn_states_critic_out = bufs['ha2'].data_ptr(), bufs['hc2'].data_ptr()
        q.values, q.actions, q.action_log_probs = bufs['v'].data_ptr(), bufs['a'].data_ptr(), bufs['lp'].data_ptr()
        for k, v in kw.items():
            setattr(q, k, v)
        return lib.np_policy_act(C.byref(q), 0, _lib.stream_ptr(torch.device('cuda:0')))

    assert step() == 0
    for bad in (dict(flags=0), dict(flags=8 | ACTOR), dict(act_dim=5), dict(act_dim=0), dict(noise=None), dict(values=None), dict(n=-1),
                dict(rnn_states_actor_out=bufs['ha'].data_ptr()), dict(rnn_states_critic_in=bufs['hc'].data_ptr() + 4)):
        assert step(**bad) != 0, bad
        assert lib.np_last_error()
    assert step(flags=ACTOR, values=None, critic_weights=None, rnn_states_critic_in=None, rnn_states_critic_out=None) == 0   # one network: the other's buffers may be NULL
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        from neuralplane_amd.policy import FusedPolicy
        FusedPolicy(fp._source, device='cpu')


@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
def test_fused_policy_against_a_torch_module_of_the_same_weights_on_the_gpu(numerics):
    """tools/collect_loop.py's torch policy (the PPO actor-critic shape, eager torch on this GPU) and FusedPolicy of its parameters agree on
    the same draws to rounding: actions 2e-5, values 1e-4, log-probabilities 5e-5 (rocBLAS GEMMs / ATen kernels vs the ordered chains)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from neuralplane_amd.policy import FusedPolicy
    from tools.collect_loop import TorchPolicy
    torch.manual_seed(5)
    tp = TorchPolicy().to('cuda:0').eval()
    with torch.no_grad():
        tp.logstd.add_(-0.5)
        tp.actor.head.weight.mul_(3.0)
    fp = FusedPolicy(tp.state_dicts(), 'cuda:0', numerics=numerics)
    n = 3000
    obs = torch.randn((n, 22), device='cuda:0') * 2
    ha, hc = torch.randn((n, 128), device='cuda:0') * 0.5, torch.randn((n, 128), device='cuda:0') * 0.5
    m = (torch.rand((n, 1), device='cuda:0') > 0.1).float()
    torch.manual_seed(9)
    v_t, a_t, lp_t, ha_t, hc_t = tp.get_actions(obs, ha, hc, m)
    torch.manual_seed(9)
    v_f, a_f, lp_f, ha_f, hc_f = fp.get_actions(obs, ha, hc, m)
    assert float((a_t - a_f).abs().max()) < 2e-5 and float((v_t - v_f).abs().max()) < 1e-4 and float((lp_t - lp_f).abs().max()) < 5e-5
    assert float((ha_t - ha_f.reshape(n, 128)).abs().max()) < 5e-5 and float((hc_t - hc_f.reshape(n, 128)).abs().max()) < 5e-5


def test_numpy_inputs_as_the_reference_runner_passes_them(golden_dir):
    """F16SimRunner.collect hands PPOPolicy.get_actions numpy arrays (np.concatenate of the buffer's slices, runner/F16sim_runner.py:125-128) and
    converts the returned tensors with _t2n: FusedPolicy takes the same arrays ([n, 1, 128] recurrent states, [n, 1] masks) and returns tensors."""
    g, _, _, fp, _ = _setup(golden_dir, 4, 'i8')
    n = g['obs'].shape[1]
    ha, hc = np.zeros((n, 1, 128), np.float32), np.zeros((n, 1, 128), np.float32)
    eps = _t(g['eps'][0])
    out_np = fp.get_actions(g['obs'][0], ha, hc, g['masks'][0], noise=eps)
    out_t = fp.get_actions(_t(g['obs'][0]), _t(ha), _t(hc), _t(g['masks'][0]), noise=eps)
    assert all(isinstance(x, torch.Tensor) and x.is_cuda for x in out_np)
    assert all(torch.equal(x, y) for x, y in zip(out_np, out_t))
    v = fp.get_values(g['obs'][0].astype(np.float64), hc, g['masks'][0])      # other dtypes are converted
    assert torch.equal(v, out_t[0])
    a, h = fp.act(g['obs'][0], ha, g['masks'][0], deterministic=True)
    assert a.shape == (n, 4) and h.shape == (n, 1, 128)


@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
@pytest.mark.parametrize('act_dim,scale,obs_dim', [(1, 1.0, 22), (2, 0.05, 15), (3, 8.0, 22), (4, 1.0, 15)])
def test_one_to_four_actions_and_weight_scales_equal_the_restatement(act_dim, scale, obs_dim, numerics):
    """Random networks (every parameter random; weight scales 0.05 … 8, so the i8 path sees small and large exponents), 1 … 4 actions, three chained
    steps at a ragged size: every output equals the restatement bit for bit."""
    from neuralplane_amd.policy import FusedPolicy, pack_policy_actor, pack_policy_critic
    from oracle.f16_oracle import PolicyOracle
    from tests.policy_kat import random_state_dicts
    sa, sc = random_state_dicts(act_dim, 100 * act_dim + (numerics == 'i8'), scale, obs_dim)
    fp = FusedPolicy((sa, sc), 'cuda:0', numerics=numerics)
    wa, A, _ = pack_policy_actor(sa)
    assert A == act_dim
    o = PolicyOracle(wa, pack_policy_critic(sc), np.float32(fp.std), np.float32(fp.log_std), numerics, obs_dim)
    n = 777
    rng = np.random.RandomState(act_dim)
    ha = hc = np.zeros((n, 128), np.float32)
    dha = dhc = torch.zeros((n, 1, 128), device='cuda:0')
    for t in range(3):
        obs = (rng.normal(0, 1, (n, obs_dim)) * rng.uniform(0.1, 5, (1, obs_dim))).astype(np.float32)
        mk = (rng.uniform(0, 1, (n, 1)) > 0.1).astype(np.float32)
        eps = rng.normal(0, 1, (n, act_dim)).astype(np.float32)
        v, a, lp, dha, dhc = fp.get_actions(_t(obs), dha, dhc, _t(mk), noise=_t(eps))
        v_o, a_o, lp_o, ha, hc = o.run(obs, ha, hc, mk, eps)
        assert same(v.cpu().numpy(), v_o) and same(a.cpu().numpy(), a_o) and same(lp.cpu().numpy(), lp_o), t
        assert same(dha.cpu().numpy()[:, 0], ha) and same(dhc.cpu().numpy()[:, 0], hc), t
        assert np.all(np.isfinite(lp_o)) and np.all(np.isfinite(v_o))


@pytest.mark.parametrize('task,in_place', [('heading', True), ('heading', False), ('tracking', True)])
def test_device_collector_equals_the_three_calls(task, in_place):
    """neuralplane_amd.collect.DeviceCollector (addresses computed, the policy writing into the buffer's slot in place; in_place: no insert launch,
    the next policy launch applies the insert rule from the env's flags) against the same collect steps spelled as FusedPolicy.get_actions -> DeviceVecEnv.step -> DeviceReplayBuffer.insert_step: every storage array bit-identical after a
    rollout that wraps the buffer, same torch seed, same env seed; then compute_returns."""
    from neuralplane_amd.actor import NUM_FLOATS, FusedActor
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.collect import DeviceCollector
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    from neuralplane_amd.envs.planning_env import PlanningEnv
    from neuralplane_amd.policy import FusedPolicy
    from tests.policy_kat import random_state_dicts
    n, T, act_dim = (700, 6, 4) if task == 'heading' else (300, 3, 3)

    class Args:
        buffer_size, n_rollout_threads = T, n
        gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
        recurrent_hidden_size, recurrent_hidden_layers = 128, 1
    sds = random_state_dicts(act_dim, 3)

    def make():
        if task == 'heading':
            envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=5, device='cuda:0')])
        else:
            ctrl = FusedActor(np.random.RandomState(0).normal(0, 0.08, NUM_FLOATS).astype(np.float32), 'cuda:0', numerics='i8')
            envs = DeviceVecEnv([lambda: PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=5, device='cuda:0', controller=ctrl)])
        buf = DeviceReplayBuffer(Args, 1, envs.observation_space, envs.action_space, device='cuda:0')
        buf.obs[0].copy_(envs.reset())
        return FusedPolicy(sds, 'cuda:0'), envs, buf

    pol, envs, buf = make()
    torch.manual_seed(21)
    for _ in range(T + 2):                                   # wraps: steps T, T + 1 overwrite slots 0, 1 as the reference's buffer does
        s = buf.step
        v, a, lp, ha, hc = pol.get_actions(buf.obs[s].reshape(n, -1), buf.rnn_states_actor[s].reshape(n, 128), buf.rnn_states_critic[s].reshape(n, 128),
                                           buf.masks[s].reshape(n, 1))
        obs, rew, d, bd, tm, _ = envs.step(a)
        buf.insert_step(obs, a, rew, d, bd, tm, lp, v, ha, hc)
    pol2, envs2, buf2 = make()
    col = DeviceCollector(pol2, envs2, buf2, in_place=in_place)
    assert col.in_place == (in_place and task == 'heading')      # PlanningEnv steps through its own path: the insert launch stays
    torch.manual_seed(21)
    for k in range(T + 2):
        col.step()
        if k == 2:
            col.finish()                                     # settling a slot early changes nothing
    col.finish()
    assert buf2.step == buf.step == 2
    for k in buf._STORAGE:
        assert torch.equal(getattr(buf, k), getattr(buf2, k)), k
    assert torch.equal(envs.env.model.s, envs2.env.model.s)
    col.compute_returns()
    buf.compute_returns(pol.get_values(buf.obs[-1].reshape(n, -1), buf.rnn_states_critic[-1].reshape(n, 128), buf.masks[-1].reshape(n, 1)).reshape(n, 1, 1))
    assert torch.equal(buf.returns, buf2.returns) and float(buf.actions.abs().sum()) > 0
    with pytest.raises(ValueError):
        DeviceCollector(FusedPolicy(random_state_dicts(2, 1), 'cuda:0'), envs2, buf2)       # a 2-action policy on a 4-action buffer


@pytest.mark.parametrize('in_place', [True, False])
@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
def test_a_rollout_equals_the_oracle_chain_bit_for_bit(numerics, in_place):
    """End to end: 40 collect steps of DeviceCollector — policy step, fused env.step, insert, the policy's outputs feeding the env and
    the env's observations / end-of-episode masks feeding the policy back through the rollout storage — against the same chain on the CPU: the
    policy restatement (f16o_policy_act / _i8) on the same normal draws, the env oracle with the same seed, and the reference's insert rule in
    numpy (runner/F16sim_runner.py:131-154: recurrent states of ended envs zeroed, masks / bad_masks).  Every storage array and the flight state
    equal bit for bit; episodes end and re-start inside the window."""
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.collect import DeviceCollector
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    from neuralplane_amd.policy import FusedPolicy, pack_policy_actor, pack_policy_critic
    from oracle.f16_oracle import Oracle, PolicyOracle
    from tests.policy_kat import random_state_dicts
    n, T, seed = 333, 40, 9

    class Args:
        buffer_size, n_rollout_threads = T, n
        gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
        recurrent_hidden_size, recurrent_hidden_layers = 128, 1
    sa, sc = random_state_dicts(4, 17)
    sa['act.action_out.mu_net.fc.0.weight'] *= np.float32(4.0)          # lively commands: some aircraft leave the envelope within the window
    pol = FusedPolicy((sa, sc), 'cuda:0', numerics=numerics)
    envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=seed, device='cuda:0')])
    buf = DeviceReplayBuffer(Args, 1, envs.observation_space, envs.action_space, device='cuda:0')
    buf.obs[0].copy_(envs.reset())
    col = DeviceCollector(pol, envs, buf, in_place=in_place)
    torch.manual_seed(77)
    eps = [torch.randn((n, 4), device='cuda:0').cpu().numpy() for _ in range(T)]     # the draws the collector is about to make
    torch.manual_seed(77)
    for _ in range(T):
        col.step()
    # ---- the same rollout on the CPU
    o, st = Oracle('heading'), Oracle.new_state(n)
    po = PolicyOracle(pack_policy_actor(sa)[0], pack_policy_critic(sc), np.float32(pol.std), np.float32(pol.log_std), numerics)
    r = {'obs': np.zeros((T + 1, n, 22), np.float32), 'actions': np.zeros((T, n, 4), np.float32), 'rewards': np.zeros((T, n), np.float32),
         'masks': np.ones((T + 1, n), np.float32), 'bad_masks': np.ones((T + 1, n), np.float32), 'action_log_probs': np.zeros((T, n), np.float32),
         'value_preds': np.zeros((T + 1, n), np.float32), 'rnn_states_actor': np.zeros((T + 1, n, 128), np.float32),
         'rnn_states_critic': np.zeros((T + 1, n, 128), np.float32)}
    r['obs'][0] = o.reset(st, seed=seed, call_idx=0)
    ended = 0
    for t in range(T):
        v, a, lp, ha, hc = po.run(r['obs'][t], r['rnn_states_actor'][t], r['rnn_states_critic'][t], r['masks'][t], eps[t])
        obs, rew, done, bad, tmo = o.step(st, a, seed=seed, call_idx=t + 1)
        done, bad, tmo = done.astype(bool), bad.astype(bool), tmo.astype(bool)
        reset_env = done | bad | tmo
        ended += int(reset_env.sum())
        ha[reset_env], hc[reset_env] = 0.0, 0.0
        r['obs'][t + 1], r['actions'][t], r['rewards'][t], r['action_log_probs'][t], r['value_preds'][t] = obs, a, rew.reshape(n), lp.reshape(n), v.reshape(n)
        r['masks'][t + 1], r['bad_masks'][t + 1] = (~done).astype(np.float32), (~bad).astype(np.float32)
        r['rnn_states_actor'][t + 1], r['rnn_states_critic'][t + 1] = ha, hc
    assert ended > 0, 'no episode ended inside the window: the masks / zeroed recurrent states were not exercised'
    for k, ref in r.items():
        got = getattr(buf, k).cpu().numpy().reshape(ref.shape)
        assert same(got, ref), (k, int(np.argmax((got != ref).reshape(ref.shape[0], -1).any(1))))
    assert same(envs.env.model.s.cpu().numpy(), st['s'])


def test_collector_noise_blocks_use_the_draws_of_one_randn():
    """noise_block = K: step k of a block uses rows [k] of ONE torch.randn((K, n, A)) — the same rollout as K-step blocks handed to get_actions."""
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.collect import DeviceCollector
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    from neuralplane_amd.policy import FusedPolicy
    from tests.policy_kat import random_state_dicts
    n, T, K = 500, 7, 3

    class Args:
        buffer_size, n_rollout_threads = T, n
        gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
        recurrent_hidden_size, recurrent_hidden_layers = 128, 1
    sds = random_state_dicts(4, 8)

    def make():
        envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=2, device='cuda:0')])
        buf = DeviceReplayBuffer(Args, 1, envs.observation_space, envs.action_space, device='cuda:0')
        buf.obs[0].copy_(envs.reset())
        return FusedPolicy(sds, 'cuda:0'), envs, buf
    pol, envs, buf = make()
    torch.manual_seed(4)
    for t in range(T):
        if t % K == 0:
            block = torch.randn((K, n, 4), device='cuda:0')
        s = buf.step
        v, a, lp, ha, hc = pol.get_actions(buf.obs[s].reshape(n, -1), buf.rnn_states_actor[s].reshape(n, 128), buf.rnn_states_critic[s].reshape(n, 128),
                                           buf.masks[s].reshape(n, 1), noise=block[t % K])
        obs, rew, d, bd, tm, _ = envs.step(a)
        buf.insert_step(obs, a, rew, d, bd, tm, lp, v, ha, hc)
    pol2, envs2, buf2 = make()
    col = DeviceCollector(pol2, envs2, buf2, noise_block=K)
    torch.manual_seed(4)
    for _ in range(T):
        col.step()
    for k in buf._STORAGE:
        assert torch.equal(getattr(buf, k), getattr(buf2, k)), k


@pytest.mark.parametrize('numerics', ['fp32', 'i8'])
def test_prev_flags_equal_insert_then_policy(numerics):
    """np_policy_step.prev_flags on synthetic flags (done, bad_done and exceed_time_limit each set on a random fifth of the rows, in every
    combination): one launch that applies the runner's insert rule itself == np_rollout_insert followed by the plain launch — every policy output,
    the masks / bad_masks it writes and the recurrent states it zeroes in place, bit for bit."""
    from neuralplane_amd import _lib
    from neuralplane_amd.policy import ACTOR, CRITIC, FusedPolicy, NpPolicyStep
    from tests.policy_kat import random_state_dicts
    fp = FusedPolicy(random_state_dicts(3, 44), 'cuda:0', numerics=numerics)
    n, dev = 1234, 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(6)
    obs = torch.randn((n, 22), generator=g).to(dev)
    ha0, hc0 = (torch.randn((n, 128), generator=g) * 0.5).to(dev), (torch.randn((n, 128), generator=g) * 0.5).to(dev)
    eps = torch.randn((n, 3), generator=g).to(dev)
    flags = (torch.rand((3, n), generator=g) < 0.2).to(torch.uint8).to(dev)
    assert int(flags[0].sum()) > 50 and int((flags[0] & flags[1]).sum()) > 5
    # reference: the insert rule in torch, then the plain launch
    ended = (flags != 0).any(0)                    # (uint8.any() stays uint8: compare first)
    ha_ref, hc_ref = ha0 * (~ended).float().unsqueeze(1), hc0 * (~ended).float().unsqueeze(1)
    masks_ref, bad_ref = (flags[0] == 0).float(), (flags[1] == 0).float()
    ref = fp.get_actions(obs, ha_ref, hc_ref, masks_ref.reshape(n, 1), noise=eps)
    # one launch with prev_flags
    ha, hc = ha0.clone(), hc0.clone()
    out = {k: torch.empty(s, device=dev) for k, s in (('v', (n, 1)), ('a', (n, 3)), ('lp', (n, 1)), ('ha', (n, 1, 128)), ('hc', (n, 1, 128)), ('m', (n,)), ('bm', (n,)))}
    q = NpPolicyStep()
    C.memmove(C.byref(q), C.byref(fp._q), C.sizeof(q))
    q.n, q.flags = n, ACTOR | CRITIC
    q.obs, q.masks, q.noise = obs.data_ptr(), None, eps.data_ptr()
    q.prev_flags, q.masks_out, q.bad_masks_out = flags.data_ptr(), out['m'].data_ptr(), out['bm'].data_ptr()
    q.rnn_states_actor_in, q.rnn_states_critic_in = ha.data_ptr(), hc.data_ptr()
    q.rnn_states_actor_out, q.rnn_states_critic_out = out['ha'].data_ptr(), out['hc'].data_ptr()
    q.values, q.actions, q.action_log_probs = out['v'].data_ptr(), out['a'].data_ptr(), out['lp'].data_ptr()
    _lib.check(_lib.load().np_policy_act(C.byref(q), 0, _lib.stream_ptr(torch.device(dev))))
    for k, (x, y) in enumerate(zip((out['v'], out['a'], out['lp'], out['ha'], out['hc']), ref)):
        assert torch.equal(x, y), (k, float((x - y).abs().max()), int((x != y).sum()))
    assert torch.equal(out['m'], masks_ref) and torch.equal(out['bm'], bad_ref)
    assert torch.equal(ha, ha_ref) and torch.equal(hc, hc_ref)                 # zeroed in place where an env ended, untouched elsewhere
    q.masks_out = None
    assert _lib.load().np_policy_act(C.byref(q), 0, _lib.stream_ptr(torch.device(dev))) != 0      # prev_flags without somewhere to put the masks


def test_collector_with_a_policy_of_another_shape_runs_its_torch_modules_on_the_device_loop():
    """a policy built with --hidden-size "64 64" --recurrent-hidden-size 64 (the reference's command line allows it,
    /root/reference/config.py:48-285) has no fused kernel: FusedPolicy raises a ValueError naming the supported shape, fuse_or_torch warns and
    hands the policy back, and DeviceCollector keeps collecting with the policy's own torch get_actions — storage and flight state
    bit-identical to the same steps spelled as policy.get_actions -> DeviceVecEnv.step -> DeviceReplayBuffer.insert_step (the FusedPolicy-free path)."""
    import torch.nn as nn
    from neuralplane_amd.buffer import DeviceReplayBuffer
    from neuralplane_amd.collect import DeviceCollector, fuse_or_torch
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    from neuralplane_amd.policy import FusedPolicy
    n, T, H = 500, 5, 64

    class Tower(nn.Module):          # PPOActor / PPOCritic with the reference's state_dict key names, hidden 64
        def __init__(self, out, critic):
            super().__init__()
            mlp = 'mlp' if critic else 'act_mlp'
            self.mlp_name = mlp
            self.ln0, self.l1, self.n1, self.l2, self.n2 = nn.LayerNorm(22), nn.Linear(22, H), nn.LayerNorm(H), nn.Linear(H, H), nn.LayerNorm(H)
            self.gru, self.n3 = nn.GRU(H, H, num_layers=1), nn.LayerNorm(H)
            self.a1, self.n4, self.a2, self.n5, self.head = nn.Linear(H, H), nn.LayerNorm(H), nn.Linear(H, H), nn.LayerNorm(H), nn.Linear(H, out)

        def forward(self, obs, h, masks):            # h [n, 1, H] as the reference's runner passes it
            x = self.n2(torch.relu(self.l2(self.n1(torch.relu(self.l1(self.ln0(obs)))))))
            y, hn = self.gru(x.unsqueeze(0), (h * masks.unsqueeze(-1)).transpose(0, 1).contiguous())
            x = self.n3(y.squeeze(0))
            x = self.n5(torch.relu(self.a2(self.n4(torch.relu(self.a1(x))))))
            return self.head(x), hn.transpose(0, 1)

        def state_dict(self, *a, **k):
            names = {'ln0': 'base.feature_norm', 'l1': 'base.mlp.fc.0', 'n1': 'base.mlp.fc.2', 'l2': 'base.mlp.fc.3', 'n2': 'base.mlp.fc.5', 'n3': 'rnn.norm',
                     'a1': ('mlp' if self.mlp_name == 'mlp' else 'act.mlp') + '.fc.0', 'n4': ('mlp' if self.mlp_name == 'mlp' else 'act.mlp') + '.fc.2',
                     'a2': ('mlp' if self.mlp_name == 'mlp' else 'act.mlp') + '.fc.3', 'n5': ('mlp' if self.mlp_name == 'mlp' else 'act.mlp') + '.fc.5',
                     'head': 'value_out' if self.mlp_name == 'mlp' else 'act.action_out.mu_net.fc.0', 'gru': 'rnn.gru'}
            return {names[k.split('.')[0]] + '.' + k.split('.', 1)[1]: v for k, v in super().state_dict().items()}

    class Policy:                    # PPOPolicy's inference surface (algorithms/ppo/ppo_policy.py:26-57)
        def __init__(self):
            torch.manual_seed(3)
            self.actor, self.critic = Tower(4, False).cuda(), Tower(1, True).cuda()
            self.log_std = torch.zeros(4, device='cuda:0')
            sd = self.actor.state_dict
            self.actor.state_dict = lambda: {**sd(), 'act.action_out.log_std': self.log_std}

        def get_actions(self, obs, ha, hc, masks):
            mu, ha = self.actor(obs, ha, masks)
            mu = torch.tanh(mu)
            a = torch.randn_like(mu) * self.log_std.exp() + mu
            lp = (-0.5 * (a - mu) ** 2 - 0.9189385).sum(-1, keepdim=True)
            v, hc = self.critic(obs, hc, masks)
            return v, a, lp, ha, hc

        def get_values(self, obs, hc, masks):
            return self.critic(obs, hc, masks)[0]

    class Args:
        buffer_size, n_rollout_threads = T, n
        gamma, use_proper_time_limits, use_gae, gae_lambda = 0.99, True, True, 0.95
        recurrent_hidden_size, recurrent_hidden_layers = H, 1

    pol = Policy()
    with pytest.raises(ValueError, match='128 128'):
        FusedPolicy(pol, 'cuda:0')
    with pytest.warns(RuntimeWarning, match='128 128'):
        assert fuse_or_torch(pol, 'cuda:0') is pol

    def make():
        envs = DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='heading', model='F16', random_seed=5, device='cuda:0')])
        buf = DeviceReplayBuffer(Args, 1, envs.observation_space, envs.action_space, device='cuda:0')
        buf.obs[0].copy_(envs.reset())
        return envs, buf

    envs, buf = make()
    torch.manual_seed(21)
    with torch.no_grad():
        for _ in range(T + 2):
            s = buf.step
            v, a, lp, ha, hc = pol.get_actions(buf.obs[s].reshape(n, -1), buf.rnn_states_actor[s].reshape(n, 1, H), buf.rnn_states_critic[s].reshape(n, 1, H),
                                               buf.masks[s].reshape(n, 1))
            obs, rew, d, bd, tm, _ = envs.step(a)
            buf.insert_step(obs, a, rew, d, bd, tm, lp, v, ha, hc)
        buf.compute_returns(pol.get_values(buf.obs[-1].reshape(n, -1), buf.rnn_states_critic[-1].reshape(n, 1, H), buf.masks[-1].reshape(n, 1)).reshape(n, 1, 1))
    envs2, buf2 = make()
    col = DeviceCollector(fuse_or_torch(pol, 'cuda:0') if False else pol, envs2, buf2)
    assert col.fused is False and col.in_place is False
    torch.manual_seed(21)
    for _ in range(T + 2):
        col.step()
    col.finish()
    col.compute_returns()
    for k in buf._STORAGE + ('returns',):
        assert torch.equal(getattr(buf, k), getattr(buf2, k)), k
    assert torch.equal(envs.env.model.s, envs2.env.model.s) and float(buf2.actions.abs().sum()) > 0 and float(buf2.rnn_states_actor.abs().sum()) > 0


@pytest.mark.parametrize('numerics', ['i8', 'fp32'])
def test_fused_policy_chained_over_200_steps_vs_the_reference_and_the_oracle(golden_dir, numerics):
    """FusedPolicy.get_actions chained for 200 calls on its own recurrent states with episode ends through the masks,
    against the REFERENCE's PPOPolicy.get_actions chained the same way (tests/golden/policy_long_kat.npz; ppo_policy.py:26-32): at every
    10th step actions <= 2e-5, values <= 1e-4, log-probabilities <= 5e-5, recurrent states <= 5e-5 — for the default block-fixed-point
    numerics and the fp32 one — and bit-identical to the CPU restatement chained the same way.  Worst errors -> gpurun_out/parity_policy_long.json."""
    import json
    import os
    from neuralplane_amd.policy import FusedPolicy, pack_policy_actor, pack_policy_critic
    from oracle.f16_oracle import PolicyOracle
    from tests.policy_kat import check_long_chain, load_long
    g, sa, sc = load_long(golden_dir)
    pol = FusedPolicy((sa, sc), 'cuda:0', numerics=numerics)
    wa, A, log_std = pack_policy_actor(sa)
    o = PolicyOracle(wa, pack_policy_critic(sc), pol.std, pol.log_std, numerics, 22)
    st = {'ha': np.zeros((48, 128), np.float32), 'hc': np.zeros((48, 128), np.float32)}

    def run(obs, ha, hc, m, eps):
        v, a, lp, ha2, hc2 = [x.cpu().numpy() for x in pol.get_actions(_t(obs), _t(ha), _t(hc), _t(m), noise=_t(eps))]
        ov, oa, olp, oha, ohc = o.run(obs, st['ha'], st['hc'], m, noise=eps)
        st['ha'], st['hc'] = oha, ohc
        assert same(v, ov) and same(a, oa) and same(lp, olp) and same(ha2.reshape(-1, 128), oha) and same(hc2.reshape(-1, 128), ohc), 'HIP != oracle'
        return v, a, lp, ha2, hc2
    worst = check_long_chain(g, run, f'hip {numerics}')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        path = os.path.join(root, 'gpurun_out', 'parity_policy_long.json')
        rep = json.load(open(path)) if os.path.exists(path) else {}
        rep[numerics] = {'get_actions_calls_chained': 200, 'rows': 48, 'episode_ends': 172, 'worst_vs_reference_at_every_10th_step': worst, 'bit_identical_to_oracle': True}
        json.dump(rep, open(path, 'w'), indent=1)
    except OSError:
        pass
