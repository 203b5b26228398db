"""What the env-protocol adapter (envs/vec_adapter.py) costs per episode at the headline size,
WITHOUT an env behind it: a fake vec env hands back the same pre-built list of per-env
numpy dicts every time, so the time is the adapter's own -- the actions' device -> host
copy, the stacking of N per-env arrays into the pinned staging buffers, the uploads.
    python scripts/time_vec_adapter.py [num_env] [T] [D]"""
import os, sys, time, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tce_rl_amd.envs.vec_adapter import VecEnvAdapter

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
D = int(sys.argv[3]) if len(sys.argv) > 3 else 48
dof = 4


class Canned:
    num_envs = N
    observation_space = types.SimpleNamespace(shape=(D,))
    action_space = types.SimpleNamespace(shape=(2 * dof,))
    envs = [types.SimpleNamespace(dt=0.0125, spec=types.SimpleNamespace(max_episode_steps=T))]

    def __init__(self):
        rng = np.random.default_rng(0)
        trunc = np.zeros(T, dtype=bool); trunc[-1] = True
        self.infos = [{"step_states": rng.standard_normal((T, D)),       # float64, as MuJoCo
                       "step_rewards": rng.standard_normal(T),
                       "step_terminations": np.zeros(T, dtype=bool),
                       "step_truncations": trunc, "segment_length": T,
                       "success": np.zeros(T)} for _ in range(N)]
        self.obs = rng.standard_normal((N, D))
        self.rew = rng.standard_normal(N)
        self.done = np.ones(N, dtype=bool)

    def reset(self):
        return self.obs

    def step(self, a):
        return self.obs, self.rew, self.done, self.infos


ad = VecEnvAdapter(Canned(), dtype=torch.float32, device="cuda", last_element_keys=["success"])
acts = torch.randn(N, T, 2 * dof, device="cuda")
for _ in range(2):
    ad.step(acts)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter()
    out = ad.step(acts)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
dt = sorted(ts)[len(ts) // 2]
up = sum(v.numel() * v.element_size() for v in out[3].values() if torch.is_tensor(v))
down = acts.numel() * acts.element_size()
print("adapter alone, %d envs x T %d x D %d: %.1f ms per episode (median of 5; min %.1f) -- %.0f MB up, "
      "%.0f MB down; an env-steps/s ceiling of %.1f M before any simulator time"
      % (N, T, D, dt * 1e3, min(ts) * 1e3, up / 1e6, down / 1e6, N * T / dt / 1e6))
