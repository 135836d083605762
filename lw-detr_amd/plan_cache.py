"""LRU cache of launch plans, bounded by the TOTAL number of resident plans.

A ForwardPlan owns a full activation set of its batch shape (ViT taps, memory, the decoder's value tensors and buffers), so what the
cache holds is device memory. Keys are (batch, height, width, slot): a dense batch that runs as n launch chains holds n plans of the
per-chain batch, slot 0 .. n-1, and those are used - and evicted - together. Round 4 kept "4 batch shapes x all their slots"
(8 plans at the default of two chains, 4 n with LWDETR_STREAMS=n: the advisor's finding); the bound is now ``max_plans`` plans in
total (default 8, LWDETR_PLAN_CACHE=<n>), whatever the chain count: the least recently used SHAPE goes first, all of its slots at
once, and never the shape that is being asked for.
"""
import os
from collections import OrderedDict

DEFAULT_MAX_PLANS = 8


def max_plans_from_env():
    try:
        n = int(os.environ.get("LWDETR_PLAN_CACHE", DEFAULT_MAX_PLANS))
    except ValueError:
        n = DEFAULT_MAX_PLANS
    return max(1, n)


class PlanCache:
    def __init__(self, max_plans=None):
        self.max_plans = max_plans_from_env() if max_plans is None else max(1, int(max_plans))
        self._plans = OrderedDict()          # key -> plan, least recently used first

    def __len__(self):
        return len(self._plans)

    def __contains__(self, key):
        return key in self._plans

    def keys(self):
        return list(self._plans)

    def values(self):
        return list(self._plans.values())

    def __getitem__(self, key):          # no LRU update: inspection by tests / tools
        return self._plans[key]

    def clear(self):
        self._plans.clear()

    def shapes(self):
        """Resident shapes, least recently used first (a shape is as recent as its most recently used slot)."""
        order = OrderedDict()
        for k in self._plans:
            order.pop(k[:3], None)
            order[k[:3]] = True
        return list(order)

    def get(self, key, factory):
        """The plan of ``key``; built by ``factory()`` on a miss, after evicting least-recently-used shapes (all slots of a shape
        together, never ``key``'s own shape) until the new plan fits the bound. One shape alone may exceed the bound (more launch
        chains than ``max_plans``): its slots are all kept - a forward needs them at the same time."""
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        for shape in self.shapes():
            if len(self._plans) + 1 <= self.max_plans:
                break
            if shape == key[:3]:
                continue
            for k in [k for k in self._plans if k[:3] == shape]:
                del self._plans[k]
        plan = factory()
        self._plans[key] = plan
        return plan
