import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rl_amd.config import tce_config
from tce_rl_amd.mp_exp import MPExperiment
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for wg in [int(a) for a in sys.argv[2:]] or (0, 208, 224, 240, 248):
    cfg = tce_config("metaworld", num_env=4096, epochs=50, num_basis=nb)
    a = cfg["params"]["agent"]["args"]
    a["overlap_updates"] = wg != 0
    a["critic_workgroups"] = wg or 256
    if os.environ.get("NO_GRAPH"):
        a["graph_policy_update"] = False
    if os.environ.get("GRAPH"):
        a["graph_policy_update"] = True
    if os.environ.get("CRITIC_ARITH"):
        a["critic_arith"] = os.environ["CRITIC_ARITH"]
    if wg < 0:                       # -n: CU-masked streams, n units per XCD for the critic
        a["critic_cus_per_xcd"] = -wg
    exp = MPExperiment(); exp.initialize(cfg, 0, None)
    for i in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        res = exp.iterate(cfg, 0, i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"critic_workgroups={wg or 'sequential'}: {dt*1e3:.1f} ms/step  critic {res['update_critic_time']*1e3:.0f} ms policy {res['update_policy_time']*1e3:.0f} ms (epochs on device {res['policy_epochs_device_time']*1e3:.0f} ms)", flush=True)
    del exp
