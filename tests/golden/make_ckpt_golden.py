"""Reference-WRITTEN checkpoint files: a small BlackBoxPolicy,
ValueFunction, RunningMeanStd and the two torch Adam optimizers are built by the
reference's own classes and saved by the reference's own ``save_weights`` /
``save`` methods (mprl/util/util_nn.py:164-223,465-520,
mprl/util/util_numerical.py:339-350, mprl/rl/agent/abstract_agent.py:109-138)
into ``tests/golden/ckpt_ref/``, together with the outputs the reference
computes from them.  Runs ONLY in the build container (see make_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ckpt_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

OUT = os.path.join(HERE, "ckpt_ref")
EPOCH = 7
D_IN, K = 6, 5


def main():
    util, _, _, BBPolicy, _, ValueFunction = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(4321)
    common = dict(init_method="orthogonal", act_func_hidden="tanh",
                  act_func_last=None, dtype="float32", device="cpu")
    policy = BBPolicy(dim_in=D_IN, dim_out=K,
                      mean_net_args=dict(avg_neuron=16, num_hidden=2, shape=0.0),
                      variance_net_args=dict(std_only=False, contextual=False),
                      out_layer_gain=0.01, min_std=1e-4, **common)
    critic = ValueFunction(dim_in=D_IN, dim_out=1,
                           hidden=dict(avg_neuron=16, num_hidden=2, shape=0.0),
                           out_layer_gain=1, **common)
    # a few optimizer steps so that weights, biases, the variance variable and
    # the Adam moments are all non-trivial
    p_opt = torch.optim.Adam(policy.parameters, lr=3e-3, weight_decay=1e-5)
    c_opt = torch.optim.Adam(critic.parameters, lr=3e-3, weight_decay=1e-5)
    g = torch.Generator().manual_seed(99)
    obs = torch.randn(32, D_IN, generator=g)
    tgt = torch.randn(32, K, generator=g)
    for _ in range(3):
        mean, L = policy.policy(obs)
        loss = -policy.log_prob(tgt, mean, L).mean()
        p_opt.zero_grad()
        loss.backward()
        p_opt.step()
        closs = (critic.critic(obs).squeeze(-1) - tgt[:, 0]).pow(2).mean()
        c_opt.zero_grad()
        closs.backward()
        c_opt.step()
    rms = util.RunningMeanStd(name="obs_rms", shape=(D_IN,), dtype="float32",
                              device="cpu")
    for _ in range(2):
        rms.update(torch.randn(40, D_IN, generator=g) * 2 + 1)
    # --- the reference's own writers
    policy.save_weights(OUT, EPOCH)
    critic.save_weights(OUT, EPOCH)
    rms.save(OUT, EPOCH)
    for name, opt in (("policy_optimizer", p_opt), ("critic_optimizer", c_opt)):
        with open(util.get_training_state_save_path(OUT, name, EPOCH),
                  "wb") as f:
            torch.save(opt.state_dict(), f)
    # --- what the reference computes from that state
    with torch.no_grad():
        mean, L = policy.policy(obs)
        values = critic.critic(obs)
        lp = policy.log_prob(tgt, mean, L)
    np.savez_compressed(
        os.path.join(OUT, "expected.npz"), obs=obs.numpy(), tgt=tgt.numpy(),
        mean=mean.numpy(), L=L.numpy(), values=values.numpy(),
        log_prob=lp.numpy(), rms_mean=rms.mean.numpy(), rms_var=rms.var.numpy(),
        rms_count=np.float64(rms.count), epoch=EPOCH, d_in=D_IN, k=K)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
