import sys, time, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_agent_gpu import build_bbrl
agent, _ = build_bbrl(4096, 50)
agent.evaluation_interval = 0
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    res = agent.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{dt*1e3:.1f} ms  sampling {res['sampling_time']*1e3:.1f} update {res['update_time']*1e3:.1f}  -> {4096*500/dt/1e6:.1f} M env-steps/s", flush=True)
