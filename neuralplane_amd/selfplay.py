"""The one real exchange of sharded SingleCombat self-play: opponent observations out, opponent actions back.

Reference: runner/selfplay_F16sim_runner.py:49-55 (the opponent policies and the env slice each one serves,
`opponent_env_split = np.array_split(arange(n_rollout_threads), num_opponents)`), :62-67 (obs split into the ego half
`obs[:, :A//2]` and the opponent half `obs[:, A//2:]`), :90-100 (every opponent policy acts on the opponent observations of
its env slice; `actions = concatenate((ego, opponent), axis=1)`).

Sharded over W ranks the engagements are partitioned BY ENV (physics, rewards and terminations never cross a rank,
sharding.shard_rows), and opponent policy p — the one that serves the envs of rank p, as in `opponent_env_split` — is hosted
on rank (p + shift) mod W: the league's frozen opponents are spread over the node next to somebody else's environments.  Per
env.step this needs exactly two collectives, each ONE `all_gather_into_tensor` over RCCL/xGMI (BASELINE.json's north star
prescribes the all-gather; payload at 1e5 engagements: 15 x 4 B x 1e5 = 6 MB of observations out, 1.6 MB of actions back):

    obs[E_loc, 2, 15] --opponent half--> all-gather --> [E_total, 15] --slice served here--> opponent policy
    opponent actions of that slice --> all-gather --> [E_total, 4] --own envs--> merge with the ego actions --> env.step

Both collectives run on a SIDE stream, so the ego policy's forward on the main stream overlaps them (`lag = 0`, the
reference's semantics: both sides act on the current observation).  With `lag = 1` the opponent acts on the PREVIOUS step's
observation (an asynchronous league opponent): the exchange for step t+1 is then in flight while the env kernel of step t
runs and never sits on the critical path.  On CPU tensors (the gloo tests) the same code runs without streams.
"""
import torch

from . import sharding


class OpponentExchange:
    def __init__(self, num_envs_local, env0, num_envs_total, dist=None, device=None, opponent_policy=None, lag=0, shift=1):
        """opponent_policy(opp_obs[E, 15], env_ids[E]) -> actions[E, 4]: the frozen opponent served on THIS rank (it receives
        the global env indices of the slice it serves so that per-env policy state can be kept by the caller)."""
        self.e_loc, self.env0, self.e_total = int(num_envs_local), int(env0), int(num_envs_total)
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.policy = opponent_policy
        if lag not in (0, 1):
            raise ValueError('lag must be 0 (opponent acts on the current observation) or 1 (on the previous one)')
        self.lag = lag
        shift = shift % self.world if self.world > 1 else 0
        # the env slice whose opponent policy is hosted here: the shard of rank (rank - shift) mod W
        served = (self.rank - shift) % self.world
        self.served0, self.served_n = sharding.shard_rows(self.e_total, self.world, served)
        self.sizes = [sharding.shard_rows(self.e_total, self.world, r)[1] for r in range(self.world)]
        # after the action all-gather, rank q's block holds the actions of shard (q - shift) mod W: my envs sit in the block
        # contributed by rank (rank + shift) mod W
        self.src_rank = (self.rank + shift) % self.world
        self.act_sizes = [self.sizes[(q - shift) % self.world] for q in range(self.world)]
        self.act_off = sum(self.act_sizes[:self.src_rank])
        self.use_streams = self.device.type == 'cuda'
        self.side = torch.cuda.Stream(device=self.device) if self.use_streams else None
        self._pending = None     # (opponent actions of my envs, event) produced by the exchange started last
        self._served_ids = torch.arange(self.served0, self.served0 + self.served_n, device=self.device)

    # -- the exchange itself ---------------------------------------------------------------------------------------------
    def _exchange(self, opp_obs):
        """opponent observations of my envs [E_loc, 15] -> opponent actions of my envs [E_loc, 4] (two all-gathers)."""
        allobs = sharding.all_gather_opponent(opp_obs, self.dist, self.sizes if self.world > 1 else None)
        mine = allobs[self.served0:self.served0 + self.served_n]
        act = self.policy(mine, self._served_ids)
        allact = sharding.all_gather_opponent(act.contiguous(), self.dist, self.act_sizes if self.world > 1 else None)
        return allact[self.act_off:self.act_off + self.e_loc]

    def start(self, obs):
        """Begin the exchange for the interleaved observation obs[2 * E_loc, 15] on the side stream (returns at once)."""
        _, opp_obs = sharding.split_ego_opponent(obs, self.e_loc)
        self.start_opp(opp_obs, keep=obs)

    def start_opp(self, opp_obs, keep=None):
        """Begin the exchange for the opponent half opp_obs[E_loc, 15] (a contiguous buffer — SingleCombatEnv.step_split's — is
        read by the all-gather as it is: no copy)."""
        if not self.use_streams:
            self._pending = (self._exchange(opp_obs), None)
            return
        self.side.wait_stream(torch.cuda.current_stream(self.device))      # the observation was produced on the main stream
        with torch.cuda.stream(self.side):
            out = self._exchange(opp_obs)
            (keep if keep is not None else opp_obs).record_stream(self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self._pending = (out, ev)

    def finish(self):
        """Opponent actions [E_loc, 4] of the exchange started last; the main stream waits for it (no host sync)."""
        out, ev = self._pending
        self._pending = None
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            out.record_stream(torch.cuda.current_stream(self.device))
        return out

    # -- one self-play step ------------------------------------------------------------------------------------------------
    def actions(self, obs, ego_policy):
        """obs[2 * E_loc, 15] of the CURRENT state -> the [2 * E_loc, 4] action rows env.step consumes.

        lag 0: start the exchange, run the ego policy meanwhile, wait, merge.
        lag 1: merge the ego actions with the opponent actions computed from the PREVIOUS observation (zeros on the very
               first step), then start the exchange for this observation — it overlaps the env.step the caller launches next."""
        ego_obs, _ = sharding.split_ego_opponent(obs, self.e_loc)
        if self.lag == 0:
            self.start(obs)
            ego_act = ego_policy(ego_obs)
            opp_act = self.finish()
        else:
            opp_act = self.finish() if self._pending is not None else torch.zeros((self.e_loc, 4), dtype=obs.dtype, device=obs.device)
            ego_act = ego_policy(ego_obs)
            self.start(obs)
        return sharding.merge_actions(ego_act, opp_act)

    def actions_split(self, obs_ego, obs_opp, ego_policy):
        """The same step on the split layout (SingleCombatEnv.reset_split / step_split): obs_ego[E_loc, 15], obs_opp[E_loc, 15] ->
        (ego actions[E_loc, 4], opponent actions[E_loc, 4]) for env.step_split.  Nothing is copied on this rank: the all-gather
        reads the kernel's opponent-observation buffer, the env kernel reads its opponent actions out of the gathered action
        buffer (a contiguous slice of it)."""
        if self.lag == 0:
            self.start_opp(obs_opp)
            ego_act = ego_policy(obs_ego)
            opp_act = self.finish()
        else:
            opp_act = self.finish() if self._pending is not None else torch.zeros((self.e_loc, 4), dtype=obs_ego.dtype, device=obs_ego.device)
            ego_act = ego_policy(obs_ego)
            self.start_opp(obs_opp)
        return ego_act, opp_act
