"""f2: checkpoint files WRITTEN BY THE REFERENCE (tests/golden/
ckpt_ref/, produced by tests/golden/make_ckpt_golden.py through the reference's
own save methods) are read by the build's loaders and give the reference's
outputs."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CKPT = os.path.join(GOLDEN, "ckpt_ref")
EXP = np.load(os.path.join(CKPT, "expected.npz"))
EPOCH, D_IN, K = int(EXP["epoch"]), int(EXP["d_in"]), int(EXP["k"])


def test_reference_files_parse_on_the_host():
    """Structure files and state dicts as the reference pickled them: the
    build's MLP / TrainableVariable (host-resident here) accept them."""
    from tce_rl_amd import nn as tnn
    mlp = tnn.MLP("ValueFunction", D_IN, 1, [16, 16], "orthogonal", 1, "tanh",
                  None, dtype=torch.float32, device=torch.device("cpu"))
    mlp.load(CKPT, EPOCH)
    raw = torch.load(os.path.join(CKPT, "ValueFunction_mlp_weights_%d" % EPOCH))
    assert list(raw) == list(mlp.state_dict())        # same key names / order
    for k, v in mlp.state_dict().items():
        assert torch.equal(v, raw[k])
    n_var = K + K * (K - 1) // 2
    var = tnn.TrainableVariable("BlackBoxPolicy_variance",
                                torch.zeros(n_var))
    var.load(CKPT, EPOCH)
    raw_v = torch.load(os.path.join(
        CKPT, "BlackBoxPolicy_variance_variable_weights_%d" % EPOCH),
        weights_only=False)
    assert torch.equal(var.variable.data, raw_v.data)
    with open(os.path.join(CKPT, "BlackBoxPolicy_mean_mlp_parameters.pkl"),
              "rb") as f:
        p = pickle.load(f)
    assert p["hidden_layers"] == [16, 16] and p["dim_out"] == K
    # a mismatching structure is refused like in the reference
    bad = tnn.MLP("ValueFunction", D_IN, 1, [16, 8], "orthogonal", 1, "tanh",
                  None, dtype=torch.float32, device=torch.device("cpu"))
    with pytest.raises(AssertionError):
        bad.load(CKPT, EPOCH)


@pytest.mark.gpu
def test_reference_checkpoint_evaluates_to_the_reference_outputs():
    from tce_rl_amd.optim import FlatAdam
    from tce_rl_amd.rl import critic_factory, policy_factory
    from tce_rl_amd.rl.sampler import RunningMeanStd
    common = dict(init_method="orthogonal", act_func_hidden="tanh",
                  act_func_last=None, dtype="float32", device="cuda")
    policy = policy_factory(
        "BlackBoxPolicy", dim_in=D_IN, dim_out=K,
        mean_net_args=dict(avg_neuron=16, num_hidden=2, shape=0.0),
        variance_net_args=dict(std_only=False, contextual=False),
        out_layer_gain=0.01, min_std=1e-4, **common)
    critic = critic_factory("ValueFunction", dim_in=D_IN, dim_out=1,
                            hidden=dict(avg_neuron=16, num_hidden=2, shape=0.0),
                            out_layer_gain=1, **common)
    p_opt = FlatAdam(policy.parameters, lr=3e-3, weight_decay=1e-5)
    c_opt = FlatAdam(critic.parameters, lr=3e-3, weight_decay=1e-5)
    policy.load_weights(CKPT, EPOCH)
    critic.load_weights(CKPT, EPOCH)
    obs, tgt = (torch.as_tensor(EXP[k]).cuda() for k in ("obs", "tgt"))
    with torch.no_grad():
        mean, L = policy.policy(obs)
        values = critic.critic(obs)
        lp = policy.log_prob(tgt, params_mean=mean, params_L=L)
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mean.cpu().numpy(), EXP["mean"], **tol)
    from tce_rl_amd import ops
    np.testing.assert_allclose(ops.full_L(L, obs.shape[0]).cpu().numpy(),
                               EXP["L"], **tol)
    np.testing.assert_allclose(values.cpu().numpy(), EXP["values"], **tol)
    np.testing.assert_allclose(lp.cpu().numpy(), EXP["log_prob"], rtol=1e-5,
                               atol=1e-4)
    # optimizer states written by torch.optim.Adam
    for name, opt in (("policy_optimizer", p_opt), ("critic_optimizer", c_opt)):
        sd = torch.load(os.path.join(CKPT, "%s_state_%d" % (name, EPOCH)),
                        map_location="cuda")
        opt.load_state_dict(sd)
        assert opt.host_step == 3
        assert opt.param_groups[0]["lr"] == 3e-3
        for p, i in zip(opt._params, sd["param_groups"][0]["params"]):
            assert torch.equal(opt.state[p]["exp_avg"], sd["state"][i]["exp_avg"])
            assert torch.equal(opt.state[p]["exp_avg_sq"],
                               sd["state"][i]["exp_avg_sq"])
    # and the parameters are still the loaded ones (views of the flat buffer)
    raw = torch.load(os.path.join(CKPT, "ValueFunction_mlp_weights_%d" % EPOCH))
    for (k, v), p in zip(raw.items(), critic.net.parameters()):
        assert torch.equal(p.detach().cpu(), v)
    rms = RunningMeanStd(name="obs_rms", shape=(D_IN,), dtype="torch.float32",
                         device="cuda")
    rms.load(CKPT, EPOCH)
    np.testing.assert_array_equal(rms.mean.cpu().numpy(), EXP["rms_mean"])
    np.testing.assert_array_equal(rms.var.cpu().numpy(), EXP["rms_var"])
    assert float(rms.count) == float(EXP["rms_count"])
    assert rms.mean.is_cuda
