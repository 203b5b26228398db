"""State-value critic: mirror of mprl/rl/critic/ (abstract_critic.py:8-120,
value_function_critic.py:4-17)."""
from abc import ABC, abstractmethod

from .. import util
from ..nn import MLP


class AbstractCritic(ABC):
    def __init__(self, dim_in, dim_out, hidden, init_method, out_layer_gain,
                 act_func_hidden, act_func_last, dtype="torch.float32",
                 device="cpu", **kwargs):
        self.dim_in, self.dim_out = dim_in, dim_out
        self.hidden = hidden
        self.init_method, self.out_layer_gain = init_method, out_layer_gain
        self.act_func_hidden, self.act_func_last = act_func_hidden, \
            act_func_last
        self.dtype, self.device = util.parse_dtype_device(dtype, device)
        self.net = MLP(name=self.__class__.__name__, dim_in=dim_in,
                       dim_out=dim_out,
                       hidden_layers=util.mlp_arch_3_params(**hidden),
                       init_method=init_method, out_layer_gain=out_layer_gain,
                       act_func_hidden=act_func_hidden,
                       act_func_last=act_func_last, dtype=self.dtype,
                       device=self.device)

    @property
    def network(self):
        return self.net

    @property
    def parameters(self):
        return list(self.net.parameters())

    def save_weights(self, log_dir, epoch):
        self.net.save(log_dir, epoch)

    def load_weights(self, log_dir, epoch):
        self.net.load(log_dir, epoch)

    @abstractmethod
    def critic(self, *args, **kwargs):
        pass


class ValueFunction(AbstractCritic):
    def critic(self, state):
        return self.net(state)


def critic_factory(typ, **kwargs):
    return {"ValueFunction": ValueFunction}[typ](**kwargs)
