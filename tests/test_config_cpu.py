"""cw2-style YAML resolution of tce_rl_amd.mp_exp (import_path / import_exp,
DEFAULT document, deep merge, grid / list expansion) -- SURVEY 8f(3)."""
import os

import pytest

from tce_rl_amd import mp_exp

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = os.path.join(HERE, "golden", "cfg")


def test_import_default_merge_and_expansion():
    exps = mp_exp.load_experiments(os.path.join(CFG, "local.yaml"))
    assert len(exps) == 4                     # 2 (list, zipped) x 2 (grid)
    combos = set()
    for e in exps:
        p = e["params"]
        assert e["name"] == "toy_tcp" and e["seed"] == "auto"        # DEFAULT
        assert e["iterations"] == 7                                  # override
        assert p["agent"]["args"]["total_iterations"] == 7
        assert p["agent"]["args"]["epochs_policy"] == 50             # imported
        assert p["agent"]["type"] == "TemporalCorrelatedAgent"
        assert p["sampler"]["args"]["num_env_train"] == 4
        assert p["policy"]["args"]["mp"]["args"]["num_dof"] == 4     # anchor
        combos.add((p["agent"]["args"]["lr_policy"],
                    p["mp"]["args"]["num_basis"], p["mp"]["args"]["tau"]))
    assert combos == {(1e-4, 5, 3), (3e-4, 5, 3), (1e-4, 8, 5), (3e-4, 8, 5)}
    cfg = mp_exp.load_config(os.path.join(CFG, "local.yaml"))
    assert cfg["params"]["mp"]["args"]["num_basis"] == 5


def test_named_experiment_and_plain_file():
    shared = os.path.join(CFG, "shared.yaml")
    assert mp_exp.load_config(shared, "other_exp")["params"]["agent"]["type"] \
        == "BlackBoxAgent"
    assert mp_exp.load_config(shared)["name"] == "toy_tcp"
    assert mp_exp.load_config(shared)["iterations"] == 100


REF = "/root/reference/mprl/config"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not here")
def test_reference_config_files_resolve():
    """The reference's own experiment files load unchanged (read as data)."""
    n = 0
    for root, _, files in os.walk(REF):
        for f in files:
            if f == "local.yaml":
                cfg = mp_exp.load_config(os.path.join(root, f))
                p = cfg["params"]
                for blk in ("agent", "policy", "critic", "sampler",
                            "projection", "mp"):
                    assert "type" in p[blk] and "args" in p[blk], (root, blk)
                assert p["agent"]["args"]["epochs_policy"] > 0
                mp_exp.dim_policy_out(p)
                # every class the file names exists behind the factories
                from tce_rl_amd.rl import agent, critic, policy, projection, \
                    sampler
                for mod, blk in ((agent, "agent"), (critic, "critic"),
                                 (policy, "policy"), (sampler, "sampler"),
                                 (projection, "projection")):
                    assert hasattr(mod, p[blk]["type"]), (root, p[blk]["type"])
                n += 1
    assert n >= 4


# ---------------------------------------------------------------------------
# tce_rl_amd.config (hyper-parameters typed by hand for the BASELINE points)
# against the reference's own resolved documents (tests/golden/resolved/*.json,
# written by tests/golden/make_resolved_cfg.py from mprl/config/*/*/entire/)
# ---------------------------------------------------------------------------
RES = os.path.join(HERE, "golden", "resolved")
# keys that are deployment choices, not hyper-parameters
_SKIP = {"device", "dtype", "seed", "total_iterations", "total_train_steps",
         "num_env_train", "num_env_test", "env_id", "evaluation_interval",
         "episodes_per_train_env", "episodes_per_test_env", "mp",
         # the synthetic env suite reports its own task metrics
         "task_specified_metrics",
         # accepted by get_mp, only False is supported (mp/prodmp.py)
         "learn_tau", "learn_delay"}


def _num(v):
    try:
        return float(v)                 # "3e-4" is a string for YAML 1.1
    except (TypeError, ValueError):
        return v


def _diff(ours, ref, path=""):
    out = []
    for k, rv in ref.items():
        if k in _SKIP:
            continue
        if k not in ours:
            out.append("%s%s missing (reference: %r)" % (path, k, rv))
        elif isinstance(rv, dict):
            out += _diff(ours[k], rv, path + k + ".")
        elif _num(ours[k]) != _num(rv):
            out.append("%s%s = %r, reference %r" % (path, k, ours[k], rv))
    return out


@pytest.mark.parametrize("fixture,env,nb", [
    ("metaworld_tcp", "metaworld", 8),
    ("box_push_random_init_tcp", "box_push", 8),
    ("table_tennis_4d_tcp", "table_tennis", 3)])
def test_tce_config_matches_the_reference_documents(fixture, env, nb):
    import json
    from tce_rl_amd.config import tce_config
    ref = json.load(open(os.path.join(RES, fixture + ".json")))["params"]
    ours = tce_config(env, num_basis=nb)["params"]
    problems = []
    for blk in ("agent", "mp", "policy", "critic", "projection", "sampler"):
        assert ours[blk]["type"] == ref[blk]["type"]
        problems += _diff(ours[blk]["args"], ref[blk]["args"], blk + ".")
    assert not problems, "\n".join(problems)


def test_bbrl_config_matches_the_reference_document():
    import json
    from tce_rl_amd.config import bbrl_config
    ref = json.load(open(os.path.join(RES, "metaworld_bbrl.json")))["params"]
    ours = bbrl_config()["params"]
    problems = []
    for blk in ("agent", "mp", "policy", "critic", "projection", "sampler"):
        assert ours[blk]["type"] == ref[blk]["type"]
        problems += _diff(ours[blk]["args"], ref[blk]["args"], blk + ".")
    assert not problems, "\n".join(problems)


def test_resolved_fixtures_are_current():
    """The committed fixtures equal what the generator writes from the
    reference tree today (runs where /root/reference exists)."""
    if not os.path.isdir(REF):
        pytest.skip("reference tree not here")
    import json
    for f in sorted(os.listdir(RES)):
        doc = json.load(open(os.path.join(RES, f)))
        cfg = mp_exp.load_config(os.path.join("/root/reference", doc["source"]))
        assert json.loads(json.dumps(cfg["params"])) == doc["params"], f
