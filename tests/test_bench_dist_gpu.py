"""`python bench.py --gpus 2` end to end on the one-GPU test box (VERDICT r5
item 2d): the launcher parent, two fresh ranks (both on cuda:0, process group
on gloo -- RCCL needs one device per rank), the sharded TCE step at the headline
size and the N > 1 JSON line -- once with the gradients on the in-library
exchange (HIP IPC between the two processes) and once with them on
torch.distributed all-reduces (``TCE_EXCHANGE=rccl``: the loud fallback of the
start-up self-test, which until round 6 had only ever run as a one-rank world).
The reference has no counterpart (no collectives under mprl/)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_two_ranks(exchange, **extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                        "MASTER_PORT", "TCE_EXCHANGE")}
    env.update(TCE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if exchange:
        env["TCE_EXCHANGE"] = exchange
    env.update(extra)
    r = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2",
         "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-configs",
         "--launch-timeout", "600"],
        env=env, capture_output=True, text=True, timeout=700, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("exchange", ["xgmi", "rccl"])
def test_two_rank_bench_line(exchange):
    line, err = _bench_two_ranks(exchange)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 2
    assert line["scaling"] == "weak" and line["backend"] == "gloo"
    assert line["launch_attempt"] == "1: as configured"
    assert line["hsa_ipc_mode_legacy"] == "0"
    assert len(line["ms_per_step_per_rank"]) == 2
    assert all(t > 0 for t in line["ms_per_step_per_rank"])
    # whole-job value: both ranks' env steps over the slowest rank's time
    want = 2 * 4096 * 500 * 2 / (line["ms_per_step"] * 2e-3)
    assert abs(line["value"] - want) <= 2e-3 * want
    x = line["exchange"]
    if exchange == "xgmi":
        assert line["gradient_exchange"] == "xgmi-oneshot"
        assert set(x) == {"critic", "policy", "aux"}
        for ch, rep in x.items():
            assert rep["kind"] == "xgmi-oneshot" and rep["self_test"] is True
            assert rep["collectives"] > 0
            # the device-side wait clock: finite, below the bounded-wait limit
            assert 0.0 <= rep["wait_us_mean_max_over_ranks"] <= \
                rep["wait_us_max_over_ranks"] < 20e6
        # 50 critic + 50 policy gradient exchanges per step on each rank
        assert x["critic"]["collectives"] >= 2 * 50
        assert x["policy"]["collectives"] >= 2 * 50
        assert line["collectives_per_step"] >= 100
    else:
        assert line["gradient_exchange"] == "rccl"
        assert all(rep["kind"] == "rccl" for rep in x.values()) or not x
        assert line["collectives_per_step"] >= 100
    assert "warmup done" in err


def test_launcher_replaces_child_sets_that_die_before_their_warmup():
    """VERDICT r5 item 2a, end to end with REAL children on the GPU box: the
    ranks of launch attempts 1 and 2 exit with code 3 after their process group
    and exchanges are up (TCE_BENCH_DIE_BEFORE_WARMUP, the rehearsal hook of
    bench.py); the launcher parent -- which never touches the GPU -- starts a
    fresh child set each time (new rendezvous port, new processes), the third
    one with the gradients on torch.distributed, and relays ITS line."""
    line, err = _bench_two_ranks(None, TCE_BENCH_DIE_BEFORE_WARMUP="1,2")
    assert line["launch_attempt"] == "3: TCE_EXCHANGE=rccl"
    assert line["gradient_exchange"] == "rccl" and line["n_gpus"] == 2
    assert line["hsa_ipc_mode_legacy"] == "0"       # (attempt 2 had tried "1")
    assert err.count("dying before the warm-up") >= 2
    assert err.count("starting a fresh child set") == 2
    assert "warmup done" in err
