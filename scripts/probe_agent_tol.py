"""Largest deviations GPU agent vs CPU oracle over every case of
tests/test_agent_gpu.py::_agent_vs_oracle (to derive the tolerances stated there)."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_agent_gpu as T
rec = T._DEVIATIONS = []
cases = [(o, f, g, "metaworld", 5, "float32", {}) for o, f, g in ((False, True, True), (True, True, True), (False, False, False), (True, False, True), (False, True, False), (True, True, False))]
cases += [(True, True, False, e, nb, dt, {}) for e, nb, dt in (("metaworld", 8, "float32"), ("box_push", 8, "float32"), ("box_push", 3, "float32"), ("table_tennis", 3, "float32"), ("table_tennis", 8, "float32"), ("box_push", 3, "float64"), ("metaworld", 5, "float64"))]
cases += [(o, True, False, "metaworld", 5, "float32", dict(critic_arith="f16x2")) for o in (False, True)]
names = ["actions", "rewards", "values", "returns", "advantages", "seg_adv", "logp"]
worst = {}
for o, f, g, env, nb, dt, kw in cases:
    rec.clear()
    try:
        T._agent_vs_oracle(o, f, g, env, nb, dt, **kw)
    except AssertionError as e:
        print("FAIL", env, nb, dt, kw, str(e)[:300])
    grp = {}
    for n, mx, sc in rec:
        a = grp.get(n, (0, 0)); grp[n] = (max(a[0], mx), max(a[1], sc))
    key = dt + ("/f16x2" if kw else "")
    print(env, nb, key, {n: "%.1e/%.1e" % v for n, v in grp.items()}, flush=True)
    w = worst.setdefault(key, {})
    for n, v in grp.items():
        a = w.get(n, (0, 0)); w[n] = (max(a[0], v[0]), max(a[1], v[1]))
for k, w in worst.items():
    print("WORST", k, {n: "%.1e (|ref| %.1e)" % v for n, v in w.items()})

for fn, args in ((T.test_deterministic_evaluation_matches_cpu_oracle, ("metaworld", 5)),
                 (T.test_deterministic_evaluation_matches_cpu_oracle, ("table_tennis", 8)),
                 ):
    rec.clear()
    try:
        fn(*args)
    except AssertionError as e:
        print("FAIL", str(e)[:200])
    print(fn.__name__, args, {n: "%.1e/%.1e" % (e, sc) for n, e, sc in rec})
