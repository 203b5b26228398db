"""Policy update alone (sequential mode), C2 shape: wall time per epoch and,
under rocprofv3 --kernel-trace, the kernel list of the policy epochs."""
import sys, os, time, torch
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
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ag.update_policy(ds)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"update_policy: host-return {1e3*(t1-t0):.1f} ms, synced {1e3*(t2-t0):.1f} ms", flush=True)
