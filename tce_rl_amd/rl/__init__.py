from .agent import agent_factory, TemporalCorrelatedAgent, BlackBoxAgent  # noqa
from .critic import critic_factory, ValueFunction  # noqa
from .policy import policy_factory, BlackBoxPolicy, TemporalCorrelatedPolicy  # noqa
from .projection import projection_factory, KLProjectionLayer, \
    gaussian_kl_details  # noqa
from .sampler import sampler_factory, BlackBoxSampler, \
    TemporalCorrelatedSampler  # noqa
