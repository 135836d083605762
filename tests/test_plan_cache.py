"""CPU: the launch-plan cache (lwdetr_amd/plan_cache.py) - bounded by the TOTAL number of resident plans, least recently used shape
first, all launch chains (slots) of a shape together, never the shape being asked for (advisor r4: the round-4 bound was 4 shapes x
all slots, i.e. 8 plans at two chains and 4 n at LWDETR_STREAMS=n)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location("plan_cache", os.path.join(ROOT, "lw-detr_amd", "plan_cache.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_lru_order_and_bound():
    pc = _mod().PlanCache(max_plans=4)
    built = []
    mk = lambda key: (lambda: built.append(key) or ("plan", key))
    for b in (1, 2, 3, 4):
        assert pc.get((b, 640, 640, 0), mk(b)) == ("plan", b)
    assert len(pc) == 4 and built == [1, 2, 3, 4]
    assert pc.get((1, 640, 640, 0), mk("again")) == ("plan", 1) and built == [1, 2, 3, 4]      # hit: no rebuild, now most recent
    pc.get((5, 640, 640, 0), mk(5))                                                              # evicts shape 2, the least recently used
    assert sorted(k[0] for k in pc.keys()) == [1, 3, 4, 5] and len(pc) == 4
    pc.get((2, 640, 640, 0), mk(2))                                                              # rebuilt; evicts 3
    assert built == [1, 2, 3, 4, 5, 2] and sorted(k[0] for k in pc.keys()) == [1, 2, 4, 5]


def test_slots_of_a_shape_are_evicted_together_and_never_the_requested_shape():
    pc = _mod().PlanCache(max_plans=4)
    mk = lambda: object()
    for slot in (0, 1):
        pc.get((16, 640, 640, slot), mk)            # a 32-image batch as two chains of 16
    for slot in (0, 1):
        pc.get((32, 640, 640, slot), mk)            # a 64-image batch as two chains of 32
    assert len(pc) == 4
    pc.get((1, 640, 640, 0), mk)                    # one more plan: the whole 16-image shape (both slots) goes
    assert sorted(pc.keys()) == [(1, 640, 640, 0), (32, 640, 640, 0), (32, 640, 640, 1)]
    # a shape with more chains than the bound keeps all of its slots (a forward needs them at the same time); the others go
    for slot in range(6):
        pc.get((8, 640, 640, slot), mk)
    assert sorted(k[3] for k in pc.keys() if k[0] == 8) == list(range(6)) and all(k[0] == 8 for k in pc.keys())
    # using slot 0 of a shape refreshes the shape
    pc2 = _mod().PlanCache(max_plans=3)
    pc2.get((16, 1, 1, 0), mk); pc2.get((16, 1, 1, 1), mk); pc2.get((4, 1, 1, 0), mk)
    pc2.get((16, 1, 1, 0), mk)                      # 16 is now more recent than 4
    pc2.get((2, 1, 1, 0), mk)
    assert sorted(pc2.keys()) == [(2, 1, 1, 0), (16, 1, 1, 0), (16, 1, 1, 1)]


def test_bound_from_environment(monkeypatch):
    m = _mod()
    monkeypatch.setenv("LWDETR_PLAN_CACHE", "3")
    assert m.PlanCache().max_plans == 3
    monkeypatch.setenv("LWDETR_PLAN_CACHE", "0")
    assert m.PlanCache().max_plans == 1
    monkeypatch.delenv("LWDETR_PLAN_CACHE")
    assert m.PlanCache().max_plans == m.DEFAULT_MAX_PLANS == 8
