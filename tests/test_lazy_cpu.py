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
    m["z"] = 5                               # kept aside, no read yet
    m.update({"x": 4})
    assert m.pending
    assert dict(m) == {"x": 4, "z": 5} and not m.pending
    m["y"] = 1
    assert m.resolve() == {"x": 4, "z": 5, "y": 1}
    n = LazyMetrics(lambda: {"x": 3})
    assert json.dumps(n.resolve()) == '{"x": 3}'
    assert {k: v for k, v in LazyMetrics(lambda: {"q": 1}).items()} == {"q": 1}


def test_lazy_metrics_is_not_a_dict_and_fails_loudly_where_a_dict_is_required():
    """as a dict subclass the pending object printed "{}" through
    json's C encoder.  Now: a Mapping; json of the mapping itself raises, of
    resolve() works; pickling stores the plain dict."""
    import pickle
    import pytest
    m = LazyMetrics(lambda: {"x": 3})
    assert not isinstance(m, dict)
    with pytest.raises(TypeError):
        json.dumps(m)
    assert m.pending                         # (the failed dump read nothing)
    assert pickle.loads(pickle.dumps(m)) == {"x": 3}
    assert type(pickle.loads(pickle.dumps(m))) is dict
    assert m == {"x": 3} and {"x": 3} == m and (m | {"y": 1}) == {"x": 3, "y": 1}


def test_a_failing_read_keeps_failing():
    """The deferred read carries the NaN checks of the update: an exception must
    not leave a silently empty mapping behind."""
    import pytest
    n = [0]

    def resolver():
        n[0] += 1
        raise Exception("NAN surrogate_loss detected")
    m = LazyMetrics(resolver)
    for _ in range(2):
        with pytest.raises(Exception, match="NAN"):
            m["policy_loss_mean"]
    assert m.pending and n[0] == 2
