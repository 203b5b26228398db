"""Agents: mirror of mprl/rl/agent/ (abstract_agent.py:12-255,
temporal_correlated_agent.py:12-753, black_box_agent.py:12-495).

``step()`` keeps the whole iteration on the device: rollout buffers, GAE /
segment advantages, critic and policy epochs.  Per-epoch scalars are collected
in device tensors and copied to the host ONCE at the end of the update (the
reference synchronises >= 20 times per policy epoch: ``.item()`` x4,
``to_np`` x12, NaN checks x3, grad-norm ``.item()`` per parameter).

Multi-GPU (env sharding): every rank holds ``num_env_train / world`` envs and a
replica of policy / critic; gradients are summed with one flat all-reduce per
optimizer step and divided by the world size (local losses are means over the
local shard, shards are equal, so this equals the reference's global mean);
advantage statistics are merged over ranks before normalisation.

Layout (as the reference's mprl/rl/agent/): ``abstract_agent.py``
(AbstractAgent), ``tce_agent.py`` (TemporalCorrelatedAgent), ``bb_agent.py``
(BlackBoxAgent), ``critic_epochs.py`` (the critic update of both on the
matrix-core epochs); this module re-exports them and holds the factory.  Which
implementation an update takes is decided in ONE place per update --
``critic_plan()`` / ``policy_plan()`` of the agent -- and returned as a named
plan that the update, the tests' path spies and the warnings read.
"""
from .abstract_agent import AbstractAgent  # noqa
from .bb_agent import BlackBoxAgent  # noqa
from .critic_epochs import CriticEpochs  # noqa
from .tce_agent import TemporalCorrelatedAgent  # noqa


def agent_factory(typ, **kwargs):
    return {"TemporalCorrelatedAgent": TemporalCorrelatedAgent,
            "BlackBoxAgent": BlackBoxAgent}[typ](**kwargs)
