"""Random region graphs for RAT-SPNs (host-side structure generation).

Counterpart of the reference's model/spn/region_graph.py (RegionGraph.random_split :54-95,
make_layers :118-155).  The structure must be bit-identical to the reference's for a given
seed, because parameters are stored per region: it draws from the same legacy
`np.random.RandomState(seed).permutation` stream in the same order and keeps partitions in a
Python `set` that receives the same insertions, so the partition-layer order (which the
reference takes from CPython's set iteration, region_graph.py:141) is reproduced as well.
Pinned against the reference by tests/golden/g1_spn_structure.json.
"""
import numpy as np


class RegionGraph:
    def __init__(self, items, seed=12345):
        self._items = tuple(sorted(items))
        self._rng = np.random.RandomState(seed)
        self._regions = {self._items}
        self._partitions = set()
        self._child_partitions = {}
        self._layers = []

    # -- queries -------------------------------------------------------------------------
    def get_root_region(self):
        return self._items

    def get_num_items(self):
        return len(self._items)

    def get_regions(self):
        return self._regions

    def get_child_partitions(self, region):
        return self._child_partitions[region]

    def get_leaf_regions(self):
        return [r for r in self._regions if r not in self._child_partitions]

    # -- construction --------------------------------------------------------------------
    def random_split(self, num_parts, num_recursions=1, region=None):
        """Split `region` (default: the root) into `num_parts` random, equally sized parts and
        recurse `num_recursions - 1` more levels into every part."""
        if num_recursions < 1:
            return None
        if not region:
            region = self._items
        if region not in self._regions:
            raise LookupError('Trying to split non-existing region.')
        if len(region) == 1:
            return None

        shuffled = list(self._rng.permutation(list(region)))
        parts = min(len(shuffled), num_parts)
        base, extra = divmod(len(shuffled), parts)
        cuts = np.cumsum([0] + [base + (1 if k < extra else 0) for k in range(parts)])
        subs = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sub = tuple(sorted(shuffled[lo:hi]))
            subs.append(sub)
            self._regions.add(sub)
        partition = tuple(sorted(subs))
        if partition not in self._partitions:
            self._partitions.add(partition)
            self._child_partitions[region] = self._child_partitions.get(region, []) + [partition]
        if num_recursions > 1:
            for sub in partition:
                self.random_split(num_parts, num_recursions - 1, sub)
        return partition

    def make_layers(self):
        """layers[0] = leaf regions (lexicographic), odd layers = partitions whose parts are all
        known, even layers = regions whose partitions are all known (lexicographic)."""
        leaves = sorted(self.get_leaf_regions())
        self._layers = [leaves]
        if len(leaves) == 1 and self._items in leaves:
            return self._layers
        known_r, known_p = set(leaves), set()
        while len(known_r) != len(self._regions) or len(known_p) != len(self._partitions):
            ready_p = [p for p in self._partitions
                       if p not in known_p and all(r in known_r for r in p)]
            self._layers.append(ready_p)
            known_p.update(ready_p)
            ready_r = sorted(r for r in self._regions if r not in known_r
                             and all(p in known_p for p in self._child_partitions[r]))
            self._layers.append(ready_r)
            known_r.update(ready_r)
        return self._layers
