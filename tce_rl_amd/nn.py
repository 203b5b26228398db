"""MLP / TrainableVariable with the reference's parameter layout, initialisation
and checkpoint format (mprl/util/util_nn.py:28-246,449-520).  Parameters are
ordinary torch tensors (optimizer state / checkpoints stay interchangeable with
the reference); the forward / backward arithmetic of the hot loops runs in the
HIP kernels of ``tce_rl_amd.ops`` (fused MLP) -- see ``forward``.
"""
import pickle as pkl

import torch
from torch import nn

from . import util

_ACT_IDS = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3, None: -1}


def _act(key):
    import torch.nn.functional as F
    return {"tanh": torch.tanh, "relu": F.relu, "leaky_relu": F.leaky_relu,
            "softplus": F.softplus, None: None}[key]


def initialize_weights(model, initialization_type, scale=2 ** 0.5,
                       init_w=3e-3, activation="relu"):
    """util_nn.py:28-72."""
    for p in model.parameters():
        two_d = len(p.data.shape) >= 2
        if initialization_type == "normal":
            p.data.normal_(init_w, scale) if two_d else p.data.zero_()
        elif initialization_type == "uniform":
            p.data.uniform_(-init_w, init_w) if two_d else p.data.zero_()
        elif initialization_type == "xavier":
            nn.init.xavier_normal_(p.data, gain=scale) if two_d \
                else p.data.zero_()
        elif initialization_type in ("fan_in", "fan_out"):
            nn.init.kaiming_uniform_(p.data, mode=initialization_type,
                                     nonlinearity=activation) if two_d \
                else p.data.zero_()
        elif initialization_type == "orthogonal":
            nn.init.orthogonal_(p.data, gain=scale) if two_d \
                else p.data.zero_()
        else:
            raise ValueError("Not a valid initialization type.")


class MLP(nn.Module):
    def __init__(self, name, dim_in, dim_out, hidden_layers, init_method,
                 out_layer_gain, act_func_hidden, act_func_last,
                 dtype=torch.float32, device=torch.device("cpu")):
        super().__init__()
        self.mlp_name = name + "_mlp"
        self.dim_in, self.dim_out = dim_in, dim_out
        self.hidden_layers = list(hidden_layers)
        self.act_func_hidden_type = act_func_hidden
        self.act_func_last_type = act_func_last
        self.init_method, self.out_layer_gain = init_method, out_layer_gain
        self.dtype, self.device = dtype, device
        # Parameters are created on the host generator like the reference
        # (same seed -> same weights), then moved to the device.
        layers = nn.ModuleList()
        dims = [dim_in] + self.hidden_layers
        for i in range(len(self.hidden_layers)):
            lin = nn.Linear(dims[i], dims[i + 1], dtype=dtype)
            initialize_weights(lin, init_method)
            layers.append(lin)
        out = nn.Linear(dims[-1], dim_out, dtype=dtype)
        initialize_weights(out, init_method, scale=out_layer_gain)
        layers.append(out)
        setattr(self, self.mlp_name, layers.to(device))

    @property
    def layers(self):
        return getattr(self, self.mlp_name)

    def forward(self, data):
        from . import ops
        return ops.mlp_forward(self, data)

    def save(self, log_dir, epoch):
        s_path, w_path = util.get_nn_save_paths(log_dir, self.mlp_name, epoch)
        with open(s_path, "wb") as f:
            pkl.dump({"dim_in": self.dim_in, "dim_out": self.dim_out,
                      "hidden_layers": self.hidden_layers,
                      "act_func_hidden_type": self.act_func_hidden_type,
                      "act_func_last_type": self.act_func_last_type,
                      "dtype": self.dtype, "device": self.device}, f)
        with open(w_path, "wb") as f:
            torch.save({k: v.detach().clone()           # not the flat buffer
                        for k, v in self.state_dict().items()}, f)

    def load(self, log_dir, epoch):
        s_path, w_path = util.get_nn_save_paths(log_dir, self.mlp_name, epoch)
        with open(s_path, "rb") as f:
            p = pkl.load(f)
            assert self.dim_in == p["dim_in"] and self.dim_out == p["dim_out"] \
                and self.hidden_layers == p["hidden_layers"] \
                and self.act_func_hidden_type == p["act_func_hidden_type"] \
                and self.act_func_last_type == p["act_func_last_type"], \
                "NN structure parameters do not match"
        self.load_state_dict(torch.load(w_path, map_location=self.device))


class TrainableVariable:
    """util_nn.py:449-520."""

    def __init__(self, name, data):
        self.name = name
        self.variable_name = name + "_variable"
        self.variable = nn.Parameter(data=data)
        self.shape, self.dtype, self.device = data.shape, data.dtype, \
            data.device

    @property
    def data(self):
        return self.variable.data

    def parameters(self):
        return [self.variable]

    def save(self, log_dir, epoch):
        s_path, w_path = util.get_nn_save_paths(log_dir, self.variable_name,
                                                epoch)
        with open(s_path, "wb") as f:
            pkl.dump({"variable_name": self.variable_name,
                      "variable_shape": self.shape, "dtype": self.dtype,
                      "device": self.device}, f)
        with open(w_path, "wb") as f:
            torch.save(nn.Parameter(self.variable.detach().clone()), f)

    def load(self, log_dir, epoch):
        s_path, w_path = util.get_nn_save_paths(log_dir, self.variable_name,
                                                epoch)
        with open(s_path, "rb") as f:
            p = pkl.load(f)
            assert self.variable_name == p["variable_name"] \
                and self.shape == p["variable_shape"], \
                "Variable %s's parameters do not match" % self.variable_name
        loaded = torch.load(w_path, map_location=self.device,
                            weights_only=False)
        with torch.no_grad():                     # keep the optimizer's view
            self.variable.data.copy_(loaded.data)
