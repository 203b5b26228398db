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
        # env sharding (one process per GPU): rank g of `world`
        # holds num_env_train / world envs and its own env / noise random
        # streams (seed + g); weights and segment pairs are rank 0's (broadcast
        # by the agent / sampler), so rank 0 consumes its generators exactly
        # like a single process
        from .dist import DistContext
        dc = DistContext()
        rank, world = (dc.rank, dc.world) if dc.active else (0, 1)
        util.set_global_random_seed(seed + rank)
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
        if world > 1:
            n_train = int(s_args.get("num_env_train", 1))
            if n_train % world:
                raise ValueError("num_env_train %d is not divisible by the "
                                 "%d ranks" % (n_train, world))
            s_args["num_env_train"] = n_train // world
        if dc.active and isinstance(s_args.get("seed", 1), int):
            s_args["seed"] = s_args.get("seed", 1) + rank
        self.sampler = sampler_factory(cfg["sampler"]["type"],
                                       cpu_cores=cw_config.get("cpu_cores"),
                                       **s_args)
        if dc.active:
            for rms in (getattr(self.sampler, "obs_rms", None),
                        getattr(self.sampler, "rwd_rms", None)):
                if rms is not None:
                    rms.equal_shards = True       # num_env_train / world each
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
            # the reference's iterate() returns a plain dict that cw2's loggers
            # json-encode / type-check: hand out the resolved metrics (one wait
            # for the device) unless the caller opts into the lazy mapping
            # (cw_config["lazy_result"]: util.LazyMetrics, read on first access)
            if hasattr(result, "resolve") and not (
                    cw_config or {}).get("lazy_result", False):
                return result.resolve()
            return result
        return self.agent.evaluate(render=False)[0]

    def finalize(self, surrender=None, crash=False):
        """mp_exp.py:105-108 (cw2's end-of-repetition hook: nothing to do)."""

    # mp_exp.py:110-162 keeps these two as static methods of the experiment
    get_dim_in = staticmethod(get_dim_in)
    dim_policy_out = staticmethod(dim_policy_out)

    def save_state(self, cw_config, rep, n):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() \
                and dist.get_rank() != 0:
            return                        # replicas are identical: rank 0 writes
        if self.save_model_dir and (
                (n + 1) % self.save_model_interval == 0
                or (n + 1) == cw_config["iterations"]):
            os.makedirs(self.save_model_dir, exist_ok=True)
            self.agent.save_agent(log_dir=self.save_model_dir, epoch=n + 1)


def _deep_merge(base, over):
    """`over` wins; dicts merge recursively (cw2 semantics for `params`)."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _deep_merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _set_path(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


def _expand(doc):
    """cw2 `grid` (cartesian product) and `list` (zipped) sections -> one
    experiment document per parameter combination; keys are dotted paths
    below `params`."""
    import itertools
    doc = copy.deepcopy(doc)
    grid, lst = doc.pop("grid", None), doc.pop("list", None)
    combos = [{}]
    if lst:
        flat = _flatten(lst)
        n = {len(v) for v in flat.values()}
        assert len(n) == 1, "`list` entries must have the same length"
        combos = [{k: v[i] for k, v in flat.items()} for i in range(n.pop())]
    if grid:
        flat = _flatten(grid)
        keys = list(flat)
        combos = [dict(c, **dict(zip(keys, vals))) for c in combos
                  for vals in itertools.product(*(flat[k] for k in keys))]
    out = []
    for c in combos:
        d = copy.deepcopy(doc)
        for k, v in c.items():
            _set_path(d.setdefault("params", {}), k, v)
        if c:
            d["_suffix"] = "_".join("%s%s" % (k.split(".")[-1], v)
                                    for k, v in c.items())
        out.append(d)
    return out


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        key = prefix + k
        if isinstance(v, dict):
            out.update(_flatten(v, key + "."))
        else:
            out[key] = v
    return out


def load_experiments(path, exp_name=None):
    """All experiment documents of a cw2 YAML file (the format of the
    reference's mprl/config/*/*/*/{local,horeka,shared}.yaml), resolved the way
    cw2 resolves them: a document named DEFAULT is merged under every other
    one; ``import_path`` + ``import_exp`` pull an experiment from another file
    (relative to this one) underneath the importing document; ``grid`` /
    ``list`` sections expand into one document per combination.  YAML anchors
    are handled by the parser.  Slurm documents are skipped."""
    path = os.path.abspath(path)
    with open(path) as f:
        docs = [d for d in yaml.safe_load_all(f) if d]
    default = {}
    exps = []
    for d in docs:
        name = d.get("name")
        if name == "DEFAULT":
            default = d
        elif name == "SLURM":
            continue
        else:
            exps.append(d)
    out = []
    for d in exps:
        d = _deep_merge(default, d)
        imp = d.pop("import_path", None)
        imp_exp = d.pop("import_exp", None)
        if imp is not None:
            base = load_experiments(os.path.join(os.path.dirname(path), imp),
                                    imp_exp)
            assert base, "import_exp %r not found in %s" % (imp_exp, imp)
            d = _deep_merge(base[0], d)
        if exp_name is None or d.get("name") == exp_name:
            out.extend(_expand(d))
    return out


def load_config(path, exp_name=None):
    """The (first) experiment document of a cw2 YAML holding ``params``."""
    for d in load_experiments(path, exp_name):
        if "params" in d:
            return d
    raise ValueError("no document with a `params` block in %s" % path)


def main(argv):
    # under torch.distributed.run (WORLD_SIZE set): one process per GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # before anything initialises the HIP / HSA runtime (the host driver only
    # supports dmabuf IPC; RCCL's peer-to-peer setup needs it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(
            os.environ.get("TCE_BACKEND", "nccl"),
            **({"device_id": torch.device("cuda", local)}
               if os.environ.get("TCE_BACKEND", "nccl") == "nccl" else {}))
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
        if world == 1 or int(os.environ.get("RANK", "0")) == 0:
            print(n, keep, flush=True)
        exp.save_state(cfg, 0, n)


if __name__ == "__main__":
    main(sys.argv)
