"""Experiment configurations as Python dicts (same structure as the
reference's YAML ``params`` documents, mprl/config/*/tcp|bbrl/entire/shared.yaml)
for the BASELINE.json benchmark points."""


def tce_config(env="metaworld", num_env=4096, num_basis=8, dtype="float32",
               device="cuda", epochs=50, iterations=7600, seed=0,
               evaluation_interval=0, num_env_test=None):
    fam = {
        # (env_id, dof, tau, delay, alpha, bbf, w_scale, g_scale, rel_goal, dt,
        #  act, policy hidden, critic hidden, mean_bound, cov_bound, min_std,
        #  lr_policy, lr_critic, weight decay, entropy_schedule, target_entropy)
        # -- the values of mprl/config/<task>/tcp/entire/shared.yaml, checked
        # against the resolved documents in tests/golden/resolved/
        # (tests/test_config_cpu.py)
        "metaworld": ("metaworld_ProDMP_TCE/reach-v2", 4, 5.0, 0.0, 10, 5, 0.1,
                      0.1, True, 0.0125, "relu", (128, 2), (128, 2), 0.005,
                      0.0005, 1e-5, 3e-4, 3e-4, 0.0, "linear", 0),
        "box_push": ("fancy_ProDMP_TCE/BoxPushingDense-v0", 7, 2.0, 0.0, 10, 3,
                     0.3, 0.3, False, 0.02, "leaky_relu", (128, 2), (256, 2),
                     0.05, 0.0005, 1e-4, 1e-4, 1e-3, 5e-5, "linear", 0.0),
        "table_tennis": ("fancy_ProDMP_TCE/TableTennisRndInit-v0", 7, 0.75, 0.3,
                         25, 3, 0.7, 0.1, True, 0.008, "tanh", (256, 1),
                         (256, 2), 0.005, 0.00025, 1e-5, 3e-4, 3e-4, 1e-5,
                         False, -1),
    }[env]
    (env_id, dof, tau, delay, alpha, bbf, ws, gs, rel, dt, act, ph, ch, mb, cb,
     min_std, lr_p, lr_c, wd, ent_sched, ent_target) = fam
    mp = {"type": "prodmp", "args": dict(
        num_dof=dof, tau=tau, delay=delay, alpha_phase=3, num_basis=num_basis,
        basis_bandwidth_factor=bbf, num_basis_outside=0, alpha=alpha,
        disable_goal=False, relative_goal=rel, auto_scale_basis=True,
        weights_scale=ws, goal_scale=gs, dt=dt, dtype=dtype, device=device)}
    critic_act = "leaky_relu" if env == "table_tennis" else act
    params = {
        "agent": {"type": "TemporalCorrelatedAgent", "args": dict(
            lr_policy=lr_p, lr_critic=lr_c, wd_policy=wd, wd_critic=wd,
            schedule_lr_policy=True, schedule_lr_critic=True, clip_critic=0.0,
            clip_grad_norm=0.0, entropy_penalty_coef=0.0, discount_factor=1,
            gae_scaling=0.95, epochs_policy=epochs, epochs_critic=epochs,
            num_minibatchs=1, norm_advantages=True, clip_advantages=0.0,
            use_gae=True, segment_advantage="value_subtraction",
            set_variance=False, balance_check=25,
            evaluation_interval=evaluation_interval, dtype=dtype,
            device=device)},
        "mp": mp,
        "policy": {"type": "TemporalCorrelatedPolicy", "args": dict(
            mean_net_args=dict(avg_neuron=ph[0], num_hidden=ph[1], shape=0.0),
            variance_net_args=dict(std_only=False, contextual=False),
            init_method="orthogonal", out_layer_gain=0.01, min_std=min_std,
            act_func_hidden=act, act_func_last=None, dtype=dtype,
            device=device, mp=mp)},
        "critic": {"type": "ValueFunction", "args": dict(
            hidden=dict(avg_neuron=ch[0], num_hidden=ch[1], shape=0.0),
            init_method="orthogonal", out_layer_gain=1,
            act_func_hidden=critic_act, act_func_last=None, dtype=dtype,
            device=device)},
        "projection": {"type": "KLProjectionLayer", "args": dict(
            proj_type="kl", mean_bound=mb, cov_bound=cb,
            trust_region_coeff=1.0, scale_prec=True, entropy_schedule=ent_sched,
            target_entropy=ent_target, temperature=0.7, entropy_eq=False,
            entropy_first=False, do_regression=False, dtype=dtype,
            device=device)},
        "sampler": {"type": "TemporalCorrelatedSampler", "args": dict(
            env_id=env_id, num_env_train=num_env,
            num_env_test=num_env_test or min(num_env, 64),
            episodes_per_train_env=1, episodes_per_test_env=1,
            norm_step_obs=True,
            time_pairs_config=dict(num_select=25, fixed_interval=True),
            dtype=dtype, device=device, seed=seed,
            task_specified_metrics=["success"])},
    }
    return {"name": "tce_" + env, "seed": seed, "iterations": iterations,
            "verbose_level": 2, "params": params}


def bbrl_config(num_env=4096, dtype="float32", device="cuda", epochs=100,
                iterations=10000, seed=0, evaluation_interval=0,
                num_env_test=None, env_id="metaworld_ProDMP/push-v2"):
    """The black-box baseline on Metaworld (BASELINE.json configs[3]): the
    ``params`` document of mprl/config/metaworld/bbrl/entire/shared.yaml:26-121
    (32 x 2 relu nets :61-91, diagonal covariance :69, K = 4 dof x (4 + 1)
    :52-53, 100 + 100 epochs :38-39, trust_region_coeff 10 :99, set_variance
    :43, no entropy schedule :101) with the env count as the free parameter.
    ``env_id``: BASELINE names push-v2, the YAML's default is button-press-v2;
    both map to the same synthetic stand-in family."""
    mp = {"type": "prodmp", "args": dict(
        num_dof=4, num_basis=4, weights_scale=0.1, goal_scale=0.1,
        relative_goal=True, disable_goal=False, tau=5.0,
        basis_bandwidth_factor=5, alpha_phase=3, alpha=10, dt=0.0125,
        dtype=dtype, device=device)}
    net = dict(avg_neuron=32, num_hidden=2, shape=0.0)
    params = {
        "agent": {"type": "BlackBoxAgent", "args": dict(
            lr_policy=3e-4, lr_critic=3e-4, wd_policy=0.0, wd_critic=0.0,
            clip_critic=0.0, clip_grad_norm=0.0, entropy_penalty_coef=0.0,
            discount_factor=1, epochs_policy=epochs, epochs_critic=epochs,
            num_minibatchs=1, norm_advantages=True, clip_advantages=0.0,
            set_variance=True, balance_check=25,
            evaluation_interval=evaluation_interval, dtype=dtype,
            device=device)},
        "mp": mp,
        "policy": {"type": "BlackBoxPolicy", "args": dict(
            mean_net_args=dict(net),
            variance_net_args=dict(std_only=True, contextual=False),
            init_method="orthogonal", out_layer_gain=0.01, min_std=1e-5,
            act_func_hidden="relu", act_func_last=None, dtype=dtype,
            device=device)},
        "critic": {"type": "ValueFunction", "args": dict(
            hidden=dict(net), init_method="orthogonal", out_layer_gain=1,
            act_func_hidden="relu", act_func_last=None, dtype=dtype,
            device=device)},
        "projection": {"type": "KLProjectionLayer", "args": dict(
            proj_type="kl", mean_bound=0.005, cov_bound=0.0005,
            trust_region_coeff=10.0, scale_prec=True, entropy_schedule=False,
            target_entropy=0.0, temperature=0.5, entropy_eq=False,
            entropy_first=False, do_regression=False, dtype=dtype,
            device=device)},
        "sampler": {"type": "BlackBoxSampler", "args": dict(
            env_id=env_id, num_env_train=num_env,
            num_env_test=num_env_test or min(num_env, 64),
            dtype=dtype, device=device, seed=seed,
            task_specified_metrics=["success"], mp=mp)},
    }
    return {"name": "bbrl_metaworld", "seed": seed, "iterations": iterations,
            "verbose_level": 2, "params": params}
