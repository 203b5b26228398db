"""cProfile of one policy update (host side), C2 shape."""
import sys, os, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
cfg = tce_config("metaworld", num_env=4096, num_basis=5, epochs=50, evaluation_interval=0)
cfg["params"]["agent"]["args"]["overlap_updates"] = False
exp = MPExperiment(); exp.initialize(cfg, 0, None)
ag = exp.agent
ag.step(); ag.step()
ds, _ = ag.sampler.run(training=True, policy=ag.policy, critic=ag.critic)
ds = ag.process_dataset(ds)
ag.update_policy(ds)
pr = cProfile.Profile()
pr.enable()
ag.update_policy(ds)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
