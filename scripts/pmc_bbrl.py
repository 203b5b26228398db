"""Two steps of the BASELINE configs[3] shard (BBRL, 4096 envs, 3 + 3 epochs)
for rocprofv3 --pmc passes over the row kernels of csrc/smlp.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import bbrl_config
from tce_rl_amd.mp_exp import MPExperiment

exp = MPExperiment()
exp.initialize(bbrl_config(num_env=4096, epochs=3), 0, None)
for _ in range(2):
    exp.agent.step()
torch.cuda.synchronize()
