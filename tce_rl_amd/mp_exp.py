"""Config-driven entry: mirror of mprl/mp_exp.py (MPExperiment.initialize /
iterate / save_state, get_dim_in, dim_policy_out) without the cw2 / Slurm /
W&B plumbing (out of scope: orchestration, no arithmetic).

    python -m tce_rl_amd.mp_exp <config.yaml> [--iterations N]

The YAML is the reference's experiment document (the one holding ``params``);
``seed: auto`` becomes 0 (cw2 sets it to the repetition index).
"""
import copy
import os
import sys

import torch
import yaml

from . import util
from .rl import (agent_factory, critic_factory, policy_factory,
                 projection_factory, sampler_factory)


def get_dim_in(cfg, sampler):
    if "TemporalCorrelated" in cfg["sampler"]["type"]:
        return sampler.observation_shape[-1] - cfg["mp"]["args"]["num_dof"] * 2
    return sampler.observation_shape[-1]


def dim_policy_out(cfg):
    a = cfg["mp"]["args"]
    if cfg["mp"]["type"] == "prodmp":
        dim_out = a["num_dof"] * (a["num_basis"] + 1)
        if a.get("disable_goal", False):
            dim_out -= a["num_dof"]
    elif cfg["mp"]["type"] == "promp":
        dim_out = a["num_dof"] * a["num_basis"]
    else:
        raise NotImplementedError
    return dim_out + int(a.get("learn_tau", False)) + \
        int(a.get("learn_delay", False))


def _resolve_auto(d, seed):
    for k, v in d.items():
        if isinstance(v, dict):
            _resolve_auto(v, seed)
        elif k == "seed" and v == "auto":
            d[k] = seed


class MPExperiment:
    def initialize(self, cw_config, rep=0, logger=None):
        cw_config = copy.deepcopy(cw_config)
        seed = cw_config.get("seed", 0)
        seed = rep if seed == "auto" else seed
        _resolve_auto(cw_config, seed)
        cfg = cw_config["params"]
        util.set_global_random_seed(seed)
        self.verbose_level = cw_config.get("verbose_level", 1)
        load_model_dir = cw_config.get("load_model_dir", None)
        self.training = load_model_dir is None
        if self.training and cw_config.get("save_model_dir") is not None:
            self.save_model_dir = os.path.abspath(cw_config["save_model_dir"])
            self.save_model_interval = max(
                cw_config["iterations"] // cw_config["num_checkpoints"], 1)
        else:
            self.save_model_dir = self.save_model_interval = None
        s_args = dict(cfg["sampler"]["args"])
        s_args.setdefault("mp", cfg.get("mp"))
        self.sampler = sampler_factory(cfg["sampler"]["type"],
                                       cpu_cores=cw_config.get("cpu_cores"),
                                       **s_args)
        self.policy = policy_factory(cfg["policy"]["type"],
                                     dim_in=get_dim_in(cfg, self.sampler),
                                     dim_out=dim_policy_out(cfg),
                                     **cfg["policy"]["args"])
        self.critic = critic_factory(cfg["critic"]["type"],
                                     dim_in=get_dim_in(cfg, self.sampler),
                                     dim_out=1, **cfg["critic"]["args"])
        p_args = dict(cfg["projection"]["args"])
        p_args.setdefault("total_train_steps", cw_config.get("iterations"))
        self.projection = projection_factory(cfg["projection"]["type"],
                                             action_dim=dim_policy_out(cfg),
                                             **p_args)
        a_args = dict(cfg["agent"]["args"])
        a_args.setdefault("total_iterations", cw_config.get("iterations"))
        self.agent = agent_factory(cfg["agent"]["type"], policy=self.policy,
                                   critic=self.critic, sampler=self.sampler,
                                   projection=self.projection, **a_args)
        if not self.training:
            self.agent.load_agent(load_model_dir,
                                  cw_config.get("load_model_epoch"))

    def iterate(self, cw_config=None, rep=0, n=0):
        if self.training:
            result = self.agent.step()
            if self.verbose_level == 0:
                return {}
            if self.verbose_level == 1:
                return {k: v for k, v in result.items()
                        if "exploration" not in k}
            return result
        return self.agent.evaluate(render=False)[0]

    def save_state(self, cw_config, rep, n):
        if self.save_model_dir and (
                (n + 1) % self.save_model_interval == 0
                or (n + 1) == cw_config["iterations"]):
            os.makedirs(self.save_model_dir, exist_ok=True)
            self.agent.save_agent(log_dir=self.save_model_dir, epoch=n + 1)


def load_config(path):
    """The experiment document of a (possibly multi-document) cw2 YAML."""
    with open(path) as f:
        docs = [d for d in yaml.safe_load_all(f) if d]
    for d in docs:
        if "params" in d:
            return d
    raise ValueError("no document with a `params` block in %s" % path)


def main(argv):
    cfg = load_config(argv[1])
    if "--iterations" in argv:
        cfg["iterations"] = int(argv[argv.index("--iterations") + 1])
    cfg.setdefault("iterations", 10)
    exp = MPExperiment()
    exp.initialize(cfg, 0, None)
    for n in range(cfg["iterations"]):
        res = exp.iterate(cfg, 0, n)
        keep = {k: v for k, v in res.items() if k.endswith("_time")
                or k in ("num_global_steps",) or "episode_reward_mean" in k}
        print(n, keep, flush=True)
        exp.save_state(cfg, 0, n)


if __name__ == "__main__":
    main(sys.argv)
