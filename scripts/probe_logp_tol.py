"""How far the fp32 pair log-prob kernels sit from the fp64 truth, in units of
max |truth| (north_star: 1e-5 on log-probs).  Prints one line per case; the bounds
of tests/test_prodmp_gpu.py are derived from this table."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_prodmp_gpu as P                                   # noqa: E402
from oracle import tce_oracle as O                            # noqa: E402
from oracle.prodmp_oracle import ProDMPOracle, pair_log_prob  # noqa: E402
from tce_rl_amd import ops                                    # noqa: E402
from tce_rl_amd._lib import call                              # noqa: E402
from tce_rl_amd.mp import ProDMP                              # noqa: E402

T_ = torch.as_tensor


def report(tag, lp32, truth, ref32=None):
    sc = np.abs(truth).max()
    e = np.abs(lp32 - truth).max()
    msg = f"{tag:46s} |truth| {sc:9.3e} err {e:9.3e} rel {e / sc:8.2e}"
    if ref32 is not None:
        er = np.abs(ref32 - truth).max()
        msg += f"   reference fp32: err {er:9.3e} rel {er / sc:8.2e}"
    print(msg, flush=True)


def golden():
    for tag in ("mw", "bp"):
        g = np.load(os.path.join(ROOT, "tests", "golden",
                                 "pair_logprob_plumbing.npz"))
        cfg = {k[len(tag) + 5:]: g[k].item() for k in g.files
               if k.startswith(tag + "_cfg_")}
        mp = ProDMP(dtype=torch.float32, device="cuda", **cfg)
        a = lambda k: T_(g[f"{tag}_{k}"]).cuda()
        times = P.affine(T_(g[f"{tag}_times"]))
        o64 = ProDMPOracle(dtype=torch.float64, **cfg)
        d = lambda k: T_(g[f"{tag}_{k}"]).double()
        truth = pair_log_prob(o64, d("traj"), d("mean"), d("L"), d("times"),
                              d("t0"), d("y0"), d("v0"),
                              T_(g[f"{tag}_pairs"])).numpy()
        lp = ops.pair_log_prob(mp, a("traj"), a("mean"), a("L"), times,
                               a("t0"), a("y0"), a("v0"), a("pairs"))
        report(f"golden {tag} per-env L", lp.cpu().numpy(), truth,
               g[f"{tag}_logp"])


def synthetic(name, N, shared, form, seed=3):
    dtype = torch.float32
    cfg = P.CFGS[name]
    mp = ProDMP(dtype=dtype, device="cuda", **cfg)
    o64 = ProDMPOracle(dtype=torch.float64, **cfg)
    T = P.HORIZON[name]
    mean, L, eps, t0, y0, v0 = P.inputs(name, N, dtype, seed, True)
    if shared:
        L = L[:1].expand(N, -1, -1).contiguous()
    times_cpu = O.get_times(t0, cfg["dt"], T)
    tg = P.affine(times_cpu)
    Lg = ops.expand_shared(L[0].cuda(), N) if shared else L.cuda()
    w = ops.mvn_rsample(mean.cuda(), Lg, eps.cuda())
    traj = ops.prodmp_traj(mp, tg, w, t0.cuda(), y0.cuda(), v0.cuda())
    torch.manual_seed(1)
    pairs = O.get_time_pairs(T, dict(num_select=25, fixed_interval=True))
    call("tce_pair_env_static", int(form == "static"))
    try:
        lp = ops.pair_log_prob(mp, traj, mean.cuda(), Lg, tg, t0.cuda(),
                               y0.cuda(), v0.cuda(), pairs.cuda())
    finally:
        call("tce_pair_env_static", 1)
    n = min(N, 64)
    dd = lambda x: x[:n].double()
    truth = pair_log_prob(o64, traj.cpu()[:n].double(), dd(mean), dd(L),
                          times_cpu[:n].double(), dd(t0), dd(y0), dd(v0),
                          pairs).numpy()
    report(f"{name} N {N} {'shared' if shared else 'per-env'} {form}",
           lp.cpu().numpy()[:n], truth)


if __name__ == "__main__":
    golden()
    for name in P._PL_NAMES:
        synthetic(name, 6, False, "static")
        synthetic(name, 6, True, "static")
        synthetic(name, 300, True, "static")
        synthetic(name, 300, True, "general")
        synthetic(name, 300, False, "static")
    synthetic("metaworld", 4096, True, "static", seed=4)
    synthetic("metaworld_nb5", 4096, True, "static", seed=4)
