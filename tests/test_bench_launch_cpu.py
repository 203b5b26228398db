"""bench.py as the driver starts it for N > 1: `python bench.py --gpus N` from a
plain shell must become a launcher (fresh torch.distributed.run children, no
exec, no GPU call in the parent) and relay rank 0's JSON line."""
import importlib.util
import json
import os
import subprocess
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location(
        "bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_builds_the_torchrun_command_and_relays_the_line(
        monkeypatch, capsys):
    bench = _bench()
    seen = {}
    record = {"metric": "env-steps/sec", "value": 1.0, "n_gpus": 4}

    class FakePopen:
        def __init__(self, cmd, env=None, stdout=None, start_new_session=False):
            seen["cmd"], seen["env"] = cmd, env
            seen["own_session"] = start_new_session
            self.returncode, self.pid = 0, 0

        def communicate(self, timeout=None):
            seen["timeout"] = timeout
            out = "banner from a library\n" + json.dumps(record) + "\n"
            return out.encode(), None
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    args = types.SimpleNamespace(gpus=4, steps=3, warmup=1,
                                 no_cpu_baseline=True, with_split_f16=False,
                                 no_configs=False, scaling="weak",
                                 launch_timeout=77.0)
    assert bench.self_launch(args) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1",
                                "--no-cpu-baseline"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["own_session"] and seen["timeout"] == 77.0
    out = capsys.readouterr()
    assert [json.loads(ln) for ln in out.out.splitlines()] == [record]
    assert "banner" in out.err


def test_self_launch_reports_a_failed_child(monkeypatch):
    bench = _bench()

    class FakePopen:
        returncode, pid = 3, 0

        def __init__(self, *a, **k):
            pass

        def communicate(self, timeout=None):
            return b"", None
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    args = types.SimpleNamespace(gpus=2, steps=1, warmup=0,
                                 no_cpu_baseline=False, with_split_f16=True)
    assert bench.self_launch(args) == 3


def test_self_launch_kills_a_hung_child_tree(tmp_path, monkeypatch):
    """A child that never finishes (a hung collective) is killed together with
    its own children after --launch-timeout and the launcher exits 124 -- no
    re-exec, no wait until the driver's limit."""
    import time
    bench = _bench()
    pidfile = tmp_path / "grandchild.pid"
    script = tmp_path / "hang.py"
    script.write_text(
        "import subprocess, sys, time\n"
        "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
        "open(%r, 'w').write(str(p.pid))\n"
        "time.sleep(600)\n" % str(pidfile))
    real_popen = subprocess.Popen

    def popen(cmd, **kw):                   # the launcher command -> the hanging script
        return real_popen([sys.executable, str(script)], **kw)
    monkeypatch.setattr(subprocess, "Popen", popen)
    args = types.SimpleNamespace(gpus=2, steps=1, warmup=0,
                                 no_cpu_baseline=True, with_split_f16=False,
                                 launch_timeout=3.0)
    t = time.time()
    assert bench.self_launch(args) == 124
    assert time.time() - t < 30
    pid = int(pidfile.read_text())
    for _ in range(50):                     # the grandchild is gone too
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("grandchild %d survived" % pid)


def test_plain_shell_multi_gpu_invocation_never_touches_the_gpu():
    """End to end in this GPU-less container: the parent launches two ranks,
    they stop at "needs a GPU", the parent returns their non-zero code -- and
    never raised the launch assertion the round-1 script died with."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():          # a GPU box: nothing to assert here
        return
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr
    assert "launch with torch.distributed.run" not in r.stderr


def test_checkpoint_paths_without_epoch():
    """util_file.py:293-317: no suffix when epoch is None."""
    from tce_rl_amd import util
    s, w = util.get_nn_save_paths("/x", "ValueFunction_mlp", None)
    assert (s, w) == ("/x/ValueFunction_mlp_parameters.pkl",
                      "/x/ValueFunction_mlp_weights")
    assert util.get_nn_save_paths("/x", "n", 7)[1] == "/x/n_weights_7"
    assert util.get_training_state_save_path("/x", "obs_rms", None) == \
        "/x/obs_rms_state"
    assert util.get_training_state_save_path("/x", "obs_rms", 3) == \
        "/x/obs_rms_state_3"
