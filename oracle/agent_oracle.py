"""CPU restatement of one full TCE ``agent.step()`` (rollout on the synthetic
env + GAE / segment advantage + critic epochs + trust-region policy epochs),
same operation order as the reference (torch-CPU: Python reverse-loop GAE,
one-hot einsum segment advantage, torch.distributions Gaussians, nn.Linear MLPs
+ torch.optim.Adam).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Used by
``bench.py``'s ``cpu_baseline`` leg (kind "port") and by tests; follows
``mprl/rl/agent/temporal_correlated_agent.py:38-100,323-639`` and
``mprl/rl/sampler/temporal_correlated_sampler.py:91-344``.
"""
import numpy as np
import torch

from . import env_oracle as E
from . import kl_oracle as KO
from . import tce_oracle as O
from .prodmp_oracle import ProDMPOracle, pair_log_prob


class OracleTCE:
    def __init__(self, cfg, num_env, seed=0, total_iterations=7600):
        """cfg: the ``params`` dict of tce_rl_amd.config.tce_config (same
        structure as the reference YAML).  Every agent / policy switch of the
        reference is honoured: ``clip_critic`` / ``clip_advantages`` /
        ``clip_grad_norm``, the three ``segment_advantage`` modes,
        ``use_gae``, ``num_minibatchs`` (numpy's global generator, as
        util_data_structure.py:378-391), ``entropy_penalty_coef``,
        ``set_variance``, weight decay, the LinearLR schedules, and the
        ``std_only`` / ``contextual`` covariance heads
        (abstract_policy.py:65-187)."""
        torch.manual_seed(seed)
        self.total_iterations = total_iterations
        self.cfg = cfg
        a = cfg["agent"]["args"]
        mpa = dict(cfg["mp"]["args"])
        self.dtype = torch.float32 if "32" in str(mpa.pop("dtype")) \
            else torch.float64
        mpa.pop("device", None)
        self.mp = ProDMPOracle(dtype=self.dtype, **mpa)
        self.dof, self.K = self.mp.num_dof, self.mp.num_dof * self.mp.num_basis_g
        self.N, self.dt = num_env, self.mp.dt
        self.T = {0.0125: 500, 0.02: 100, 0.008: 350}[self.dt]
        self.task, self.d_task = E.FAMILY[self.T]
        self.D = self.d_task + 1 + 2 * self.dof
        self.a = a
        pa, ca = cfg["policy"]["args"], cfg["critic"]["args"]
        d_in = self.D - 2 * self.dof
        mk = lambda args, d_out, gain, act: torch.nn.ParameterList(
            [torch.nn.Parameter(t) for Wb in O.mlp_init(
                d_in, d_out, O.mlp_arch_3_params(**args), gain, self.dtype)
             for t in Wb])
        self.p_act, self.c_act = pa["act_func_hidden"], ca["act_func_hidden"]
        self.pnet = mk(pa["mean_net_args"], self.K, pa["out_layer_gain"],
                       self.p_act)
        self.cnet = mk(ca["hidden"], 1, ca["out_layer_gain"], self.c_act)
        self.min_std = float(pa["min_std"])
        vna = dict(pa["variance_net_args"])
        self.std_only = bool(vna.pop("std_only", False))
        self.contextual = bool(vna.pop("contextual", False))
        K = self.K
        n_var = K if self.std_only else K + K * (K - 1) // 2
        if self.contextual:        # a second MLP (abstract_policy.py:98-109)
            self.vnet = mk(vna, n_var, pa["out_layer_gain"], self.p_act)
            self.var = None
            var_params = list(self.vnet)
        else:
            self.vnet = None
            self.var = torch.nn.Parameter(
                O.initial_variance_vector(K, self.std_only, self.dtype))
            var_params = [self.var]
        self.var_params = var_params
        self.p_opt = torch.optim.Adam(
            list(self.pnet) + var_params, lr=a["lr_policy"],
            weight_decay=float(a.get("wd_policy", 0.0)))
        self.c_opt = torch.optim.Adam(
            list(self.cnet), lr=a["lr_critic"],
            weight_decay=float(a.get("wd_critic", 0.0)))
        self.rms = O.RunningMeanStd((self.D,), self.dtype)
        self.proj = cfg["projection"]["args"]
        self.initial_entropy = None
        self.it = 0
        self.gen = torch.Generator().manual_seed(seed)
        self.forced_reset = None      # (goal, init_pos) injected by tests
        self.forced_eps = None        # parameter noise injected by tests
        self.last = {}                # rollout tensors of the last step

    def _mlp(self, net, x, act):
        ps = list(net)
        return O.mlp_forward([(ps[i], ps[i + 1]) for i in range(0, len(ps), 2)],
                             x, act)

    def _policy(self, obs):
        mean = self._mlp(self.pnet, obs, self.p_act)
        if self.contextual:
            vec = self._mlp(self.vnet, obs, self.p_act)
            L = O.vector_to_cholesky(vec, self.K, self.min_std, self.std_only)
        else:
            L = O.vector_to_cholesky(self.var[None], self.K, self.min_std,
                                     self.std_only).expand(obs.shape[0], -1, -1)
        return mean, L

    def _reset(self):
        r = lambda *s: torch.rand(*s, generator=self.gen, dtype=self.dtype)
        if self.forced_reset is not None:           # tests: same env state
            self.goal, pos = self.forced_reset
        else:
            self.goal = r(self.N, self.dof) * 2 - 1
            pos = 0.1 * (r(self.N, self.dof) * 2 - 1)
        z = torch.zeros(self.N, self.dof, dtype=self.dtype)
        return E.reset_obs(self.task, self.d_task, self.goal, pos, z)

    def rollout(self, training=True, deterministic=False):
        """TemporalCorrelatedSampler.run (temporal_correlated_sampler.py:91-344)
        on the synthetic env: with ``training`` the observation statistics are
        updated and applied before the critic (:244-249); an evaluation run
        (``deterministic``: the mean parameters instead of a sample, :305-315)
        feeds the critic raw states and leaves the statistics alone.  The pair
        offsets are drawn in both cases (:136)."""
        N, T, D2 = self.N, self.T, 2 * self.dof
        with torch.no_grad():
            s0 = self._reset()
            pairs = O.get_time_pairs(T, dict(num_select=25,
                                             fixed_interval=True))
            t0 = s0[:, -D2 - 1]
            y0, v0 = s0[:, -D2:-self.dof], s0[:, -self.dof:]
            mean_old, L_old = self._policy(s0[:, :-D2])
            times = O.get_times(t0, self.dt, T)
            if deterministic:
                eps = torch.zeros(N, self.K, dtype=self.dtype)
            elif self.forced_eps is not None:
                eps = self.forced_eps
            else:
                eps = torch.randn(N, self.K, generator=self.gen,
                                  dtype=self.dtype)
            pos, vel = self.mp.sample_trajectories(times, mean_old, L_old, t0,
                                                   y0, v0, eps)
            actions = torch.cat([pos, vel], -1)
            lp_old = pair_log_prob(self.mp, actions, mean_old, L_old, times,
                                   t0, y0, v0, pairs)
            states, rewards, flags, metrics = E.rollout(
                self.task, actions, s0, self.dof, self.d_task, self.dt)
            episode_reward = rewards.sum(-1)
            if self.task == "table_tennis":     # make_mdp_reward on hit_ball
                rewards = O.make_mdp_reward(rewards, flags)
            if training:
                self.rms.update(states.view(-1, self.D))
                nstates = self.rms.normalise(states)
            else:
                nstates = states
            values = self._mlp(self.cnet, nstates[..., :-D2],
                               self.c_act).squeeze(-1)
        return dict(step_actions=actions, segment_log_prob_estimate=lp_old,
                    step_values=values, step_rewards=rewards,
                    episode_reward=episode_reward,
                    segment_reward=rewards.sum(-1), pred_pairs=pairs,
                    segment_params_mean=mean_old, segment_params_L=L_old,
                    step_states=nstates, segment_state=s0, times=times,
                    success=metrics[:, 0])

    def evaluate(self):
        """AbstractAgent.evaluate(evaluate_deterministic=True)
        (abstract_agent.py:219-255): one deterministic test rollout."""
        return self.rollout(training=False, deterministic=True)

    def step(self):
        self.it += 1
        N, T, D2 = self.N, self.T, 2 * self.dof
        a = self.a
        ro = self.rollout(training=True)
        actions, lp_old, values = ro["step_actions"], \
            ro["segment_log_prob_estimate"], ro["step_values"]
        rewards, pairs, nstates = ro["step_rewards"], ro["pred_pairs"], \
            ro["step_states"]
        mean_old, L_old, s0, times = ro["segment_params_mean"], \
            ro["segment_params_L"], ro["segment_state"], ro["times"]
        t0 = s0[:, -D2 - 1]
        y0, v0 = s0[:, -D2:-self.dof], s0[:, -self.dof:]
        with torch.no_grad():
            dones = torch.zeros(N, T, dtype=torch.bool)
            dones[:, -1] = True
            tl = torch.zeros_like(dones)
            adv, ret = O.gae(rewards, values, dones, tl, a["discount_factor"],
                             a["gae_scaling"], a["use_gae"])
            seg_adv = O.segment_advantage(
                a["segment_advantage"], rewards, values, adv, pairs,
                a["discount_factor"], a["norm_advantages"],
                a["clip_advantages"])
        self.last = dict(step_actions=actions, segment_log_prob_estimate=lp_old,
                         step_values=values, step_rewards=rewards,
                         step_advantages=adv, step_returns=ret,
                         segment_advantage=seg_adv, pred_pairs=pairs,
                         segment_params_mean=mean_old, step_states=nstates)
        # ---- critic epochs (temporal_correlated_agent.py:323-379)
        import numpy as np
        cs = nstates[:, :-1, :-D2].reshape(N * T, -1)
        cr, cv = ret.reshape(-1), values[:, :-1].reshape(-1)
        clip_gn = float(a.get("clip_grad_norm", 0.0))
        nmb = int(a.get("num_minibatchs", 1))
        for _ in range(a["epochs_critic"]):
            if nmb == 1:        # the permutation does not change a full-batch mean
                splits = [None]
            else:               # generate_minibatches: numpy's global generator
                idx = np.arange(N * T)
                np.random.shuffle(idx)
                splits = [torch.as_tensor(x) for x in np.array_split(idx, nmb)]
            for sel in splits:
                xs, rs, vs = (cs, cr, cv) if sel is None else \
                    (cs[sel], cr[sel], cv[sel])
                v = self._mlp(self.cnet, xs, self.c_act).squeeze(-1)
                loss = O.value_loss(v, rs, vs, a["clip_critic"])
                self.c_opt.zero_grad(set_to_none=True)
                loss.backward()
                O.grad_norm_clip(clip_gn, [q.grad for q in self.cnet])
                self.c_opt.step()
        # ---- policy epochs
        if self.initial_entropy is None:
            self.initial_entropy = KO.entropy(L_old).mean()
        p = self.proj
        beta = KO.entropy_schedule(p["entropy_schedule"], self.initial_entropy,
                                   p["target_entropy"], p["temperature"],
                                   self.it, self.total_iterations, self.K)
        ent_coef = float(a.get("entropy_penalty_coef", 0.0))
        set_var = bool(a.get("set_variance", False))
        # get_trust_region_loss(..., set_variance): the covariance part is
        # dropped when the variance is SET from the projection afterwards
        include_cov = self.contextual or not set_var
        pparams = list(self.pnet) + self.var_params

        def project():
            mean_new, L_new = self._policy(s0[:, :-D2])
            pm, pL = KO.project(mean_new, L_new, mean_old, L_old,
                                p["mean_bound"], p["cov_bound"], beta,
                                contextual_std=self.contextual,
                                entropy_eq=bool(p.get("entropy_eq", False)),
                                entropy_first=bool(p.get("entropy_first",
                                                         False)))
            return mean_new, L_new, pm, pL
        # balance check (temporal_correlated_agent.py:447-522): in iterations
        # with num_iterations % balance_check == 1 every epoch first runs the
        # surrogate loss alone and the trust region loss alone, each with its
        # own forward / backward pass, for the norm of its parameter gradient
        bal = a.get("balance_check", 10)
        check = isinstance(bal, int) and self.it % bal == 1
        sur_gn, tr_gn = [], []

        def grad_norm():
            return float(sum(float(q.grad.norm(2)) ** 2 for q in pparams
                             if q.grad is not None) ** 0.5)
        for _ in range(a["epochs_policy"]):
            if check:
                mean_new, L_new, pm, pL = project()
                lp = pair_log_prob(self.mp, actions, pm, pL, times, t0, y0, v0,
                                   pairs)
                s_loss, _ = O.surrogate_loss(seg_adv, lp, lp_old)
                self.p_opt.zero_grad(set_to_none=True)
                s_loss.backward()
                sur_gn.append(grad_norm())
                mean_new, L_new, pm, pL = project()
                tr = KO.trust_region_loss(mean_new, L_new, pm, pL,
                                          p["trust_region_coeff"], include_cov)
                self.p_opt.zero_grad(set_to_none=True)
                tr.backward()
                tr_gn.append(grad_norm())
            mean_new, L_new, pm, pL = project()
            lp = pair_log_prob(self.mp, actions, pm, pL, times, t0, y0, v0,
                               pairs)
            s_loss, _ = O.surrogate_loss(seg_adv, lp, lp_old)
            e_loss = -ent_coef * KO.entropy(pL).mean()     # :741-745
            tr = KO.trust_region_loss(mean_new, L_new, pm, pL,
                                      p["trust_region_coeff"], include_cov)
            total = s_loss + e_loss + tr
            self.p_opt.zero_grad(set_to_none=True)
            total.backward()
            for q in pparams:              # a parameter the loss does not reach
                if q.grad is None:
                    q.grad = torch.zeros_like(q)
            O.grad_norm_clip(clip_gn, [q.grad for q in pparams])
            self.p_opt.step()
        self.last["surrogate_grad_norm"] = sur_gn
        self.last["trust_region_grad_norm"] = tr_gn
        if set_var and not self.contextual:                 # :626-637
            with torch.no_grad():
                _, _, _, pL = project()
                self.var.copy_(O.cholesky_to_vector(pL[:1], self.min_std,
                                                    self.std_only)[0])
        # LinearLR(1 -> 0.01, total_iterations), stepped once per agent.step()
        # (abstract_agent.py:91-96, temporal_correlated_agent.py:63-70)
        for opt, base, on in ((self.c_opt, a["lr_critic"],
                               a.get("schedule_lr_critic", False)),
                              (self.p_opt, a["lr_policy"],
                               a.get("schedule_lr_policy", False))):
            if on:
                for g in opt.param_groups:
                    g["lr"] = O.linear_lr(base, self.it, self.total_iterations)
        return N * T


class OracleBBRL:
    """CPU restatement of one ``BlackBoxAgent.step()``: episode-level policy
    (param-space Gaussian, diagonal or full), the env turns the sampled MP
    parameters into a trajectory and an episode return, advantage R - V(s0),
    critic regresses the return, trust-region projected policy update.
    Follows ``mprl/rl/agent/black_box_agent.py:34-389`` and
    ``mprl/rl/sampler/black_box_sampler.py:158-249``.  Test infrastructure."""

    def __init__(self, mp_args, num_env, dim_obs, policy_hidden, critic_hidden,
                 act, std_only, min_std, out_layer_gain, lr, epochs, mean_bound,
                 cov_bound, tr_coeff, set_variance, norm_advantages=True,
                 clip_advantages=0.0, clip_critic=0.0, dtype=torch.float32,
                 balance=False, weight_decay=0.0, num_minibatchs=1):
        mpa = dict(mp_args)
        mpa.pop("dtype", None), mpa.pop("device", None)
        self.mp = ProDMPOracle(dtype=dtype, **mpa)
        # balance: this step is one with num_iterations % balance_check == 1
        # (black_box_agent.py:218-284) -- two extra passes per epoch whose
        # gradient norms are kept in self.balance_norms
        self.balance, self.balance_norms = balance, []
        # per policy epoch the 12 means of kl_old_new_proj: {mean, cov, shape,
        # volume} x {(new, old), (new, proj), (proj, old)}
        self.kl_rows = []
        self.dof, self.K = self.mp.num_dof, self.mp.num_dof * self.mp.num_basis_g
        self.N, self.dt, self.dtype = num_env, self.mp.dt, dtype
        self.T = {0.0125: 500, 0.02: 100, 0.008: 350}[self.dt]
        self.D = dim_obs
        self.task = "push"              # metaworld push-v2 stand-in (BASELINE C4)
        mk = lambda hid, d_out, gain: torch.nn.ParameterList(
            [torch.nn.Parameter(t) for Wb in O.mlp_init(dim_obs, d_out, hid,
                                                       gain, dtype) for t in Wb])
        self.act = act
        self.pnet = mk(policy_hidden, self.K, out_layer_gain)
        self.cnet = mk(critic_hidden, 1, 1.0)
        self.std_only, self.min_std = std_only, float(min_std)
        self.var = torch.nn.Parameter(
            O.initial_variance_vector(self.K, std_only, dtype))
        self.p_opt = torch.optim.Adam(list(self.pnet) + [self.var], lr=lr,
                                      weight_decay=weight_decay)
        self.c_opt = torch.optim.Adam(list(self.cnet), lr=lr,
                                      weight_decay=weight_decay)
        self.epochs = epochs
        self.mean_bound, self.cov_bound = mean_bound, cov_bound
        self.tr_coeff, self.set_variance = tr_coeff, set_variance
        self.norm_advantages, self.clip_advantages = norm_advantages, \
            clip_advantages
        self.clip_critic = clip_critic
        # critic minibatches (black_box_agent.py:124-131; class default 10)
        self.num_minibatchs = int(num_minibatchs)
        self.forced_reset = self.forced_eps = None
        self.last = {}

    def _mlp(self, net, x):
        ps = list(net)
        return O.mlp_forward([(ps[i], ps[i + 1]) for i in range(0, len(ps), 2)],
                             x, self.act)

    def _policy(self, obs):
        L = O.vector_to_cholesky(self.var[None], self.K, self.min_std,
                                 self.std_only).expand(obs.shape[0], -1, -1)
        return self._mlp(self.pnet, obs), L

    def step(self):
        N, T = self.N, self.T
        goal, pos0 = self.forced_reset
        with torch.no_grad():
            v0 = torch.zeros(N, self.dof, dtype=self.dtype)
            full0 = E.reset_obs(self.task, self.D, goal, pos0, v0)
            obs = full0[:, :self.D]
            mean_old, L_old = self._policy(obs)
            action = O.mvn_rsample(mean_old, L_old, self.forced_eps)
            lp_old = O.mvn_log_prob(action, mean_old, L_old)
            values = self._mlp(self.cnet, obs).squeeze(-1)
            t0 = torch.zeros(N, dtype=self.dtype)
            pos, vel = self.mp.traj(O.get_times(t0, self.dt, T), action, t0,
                                    pos0, v0)
            _, rew, _, _ = E.rollout(self.task, torch.cat([pos, vel], -1),
                                     full0, self.dof, self.D, self.dt)
            reward = rew.sum(-1)
            adv = O.bbrl_advantage(reward, values, self.norm_advantages,
                                   self.clip_advantages)
        self.last = dict(segment_action=action, segment_log_prob=lp_old,
                         segment_value=values, segment_reward=reward,
                         segment_advantage=adv)
        no_beta = torch.tensor(-float("inf"), dtype=self.dtype)  # no entropy control
        for _ in range(self.epochs):
            if self.num_minibatchs == 1:   # the permutation leaves a full-batch mean
                splits = [slice(None)]
            else:   # generate_minibatches (util_data_structure.py:378-391)
                idx = np.arange(N)
                np.random.shuffle(idx)
                splits = [torch.as_tensor(x) for x in
                          np.array_split(idx, self.num_minibatchs)]
            for sel in splits:
                v = self._mlp(self.cnet, obs[sel]).squeeze(-1)
                loss = O.value_loss(v, reward[sel], values[sel],
                                    self.clip_critic)
                self.c_opt.zero_grad(set_to_none=True)
                loss.backward()
                self.c_opt.step()
        params = list(self.pnet) + [self.var]

        def grad_norm():                 # util.grad_norm_clip(0.0, params)[0]
            return torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params
                                  if p.grad is not None)).item()
        for _ in range(self.epochs):
            if self.balance:             # black_box_agent.py:226-284
                mean_new, L_new = self._policy(obs)
                pm, pL = KO.project(mean_new, L_new, mean_old, L_old,
                                    self.mean_bound, self.cov_bound, no_beta,
                                    contextual_std=False)
                s_loss, _ = O.surrogate_loss(
                    adv, O.mvn_log_prob(action, pm, pL), lp_old)
                self.p_opt.zero_grad(set_to_none=True)
                s_loss.backward()
                sg = grad_norm()
                mean_new, L_new = self._policy(obs)
                pm, pL = KO.project(mean_new, L_new, mean_old, L_old,
                                    self.mean_bound, self.cov_bound, no_beta,
                                    contextual_std=False)
                tr = KO.trust_region_loss(mean_new, L_new, pm, pL,
                                          self.tr_coeff, not self.set_variance)
                self.p_opt.zero_grad(set_to_none=True)
                tr.backward()
                self.balance_norms.append((sg, grad_norm()))
            mean_new, L_new = self._policy(obs)
            pm, pL = KO.project(mean_new, L_new, mean_old, L_old,
                                self.mean_bound, self.cov_bound, no_beta,
                                contextual_std=False)
            lp = O.mvn_log_prob(action, pm, pL)
            s_loss, _ = O.surrogate_loss(adv, lp, lp_old)
            with torch.no_grad():        # kl_old_new_proj, black_box_agent.py:391-436
                self.kl_rows.append([
                    d.mean().item()
                    for p, q in (((mean_new, L_new), (mean_old, L_old)),
                                 ((mean_new, L_new), (pm, pL)),
                                 ((pm, pL), (mean_old, L_old)))
                    for d in KO.gaussian_kl_details(p[0], p[1], q[0], q[1])])
            tr = KO.trust_region_loss(mean_new, L_new, pm, pL, self.tr_coeff,
                                      not self.set_variance)
            total = s_loss + tr
            self.p_opt.zero_grad(set_to_none=True)
            total.backward()
            self.p_opt.step()
        if self.set_variance:            # black_box_agent.py:377-386
            with torch.no_grad():
                m, L = self._policy(obs)
                _, pL = KO.project(m, L, mean_old, L_old, self.mean_bound,
                                   self.cov_bound, no_beta,
                                   contextual_std=False)
                self.var.copy_(O.cholesky_to_vector(pL[:1], self.min_std,
                                                    self.std_only)[0])
        return N * T
