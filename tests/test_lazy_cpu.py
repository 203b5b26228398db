"""util.LazyMetrics: the mapping agent.step() returns (filled on first access)."""
import json

from tce_rl_amd.util import LazyMetrics


def test_lazy_metrics_resolve_once_on_any_access():
    calls = []

    def resolver():
        calls.append(1)
        return {"a": 1.0, "b": 2}
    m = LazyMetrics(resolver)
    assert m.pending and not calls
    assert m["a"] == 1.0 and calls == [1] and not m.pending
    assert m.get("b") == 2 and "b" in m and len(m) == 2 and calls == [1]
    assert dict(m) == {"a": 1.0, "b": 2} and {**m, "c": 3} == {"a": 1.0, "b": 2, "c": 3}
    assert sorted(m.keys()) == ["a", "b"] and sorted(m.items()) == [("a", 1.0), ("b", 2)]


def test_lazy_metrics_updates_land_on_top_of_the_resolved_values():
    m = LazyMetrics(lambda: {"x": 3})
    m["z"] = 5                               # resolves first
    m.update({"x": 4})
    assert dict(m) == {"x": 4, "z": 5}
    n = LazyMetrics(lambda: {"x": 3})
    assert json.dumps(n.resolve()) == '{"x": 3}'
    assert {k: v for k, v in LazyMetrics(lambda: {"q": 1}).items()} == {"q": 1}
