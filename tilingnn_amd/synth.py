"""Seeded synthetic candidate-placement super-graphs, numpy only.

The arrays have exactly the layout `BrickLayout.get_data_as_torch_tensor` hands to the
network (/root/reference/util/data_util.py:110-117,164-205):

  node_feature          float64 [N, tile_count+1]  one-hot tile id | area / max_area
  align_edge_index      int64   [2, Ea]   row 0 = src, row 1 = dst; pairs (u,v),(v,u)
                                          consecutive, NOT sorted (tile_graph.py:206-207)
  align_edge_features   float64 [Ea, Fe]  col0 = 0, col1 = align length / max, cols 2.. one-hot
                                          over the T "unique adjacency features"; symmetric per pair
  collide_edge_index    int64   [2, Ec]
  collide_edge_features float64 [Ec, Fe]  col0 = overlap area, rest 0 (never read by the network)

The graph is banded: v = u + delta, delta ~ U[1, B], B = ceil(8 sqrt(N)), which matches the
index distance of the reference's ring-ordered real graph (p90 245 / max ~500 at N = 1254)
and makes contiguous node ranges talk only to their neighbouring ranges.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class SuperGraph:
    node_feature: np.ndarray
    align_edge_index: np.ndarray
    align_edge_features: np.ndarray
    collide_edge_index: np.ndarray
    collide_edge_features: np.ndarray
    tile_count: int
    n_edge_types: int

    @property
    def n_nodes(self) -> int:
        return int(self.node_feature.shape[0])

    def to_torch(self, device):
        """Same conversion as util/data_util.py:110-117 (float64->float32, int64 kept)."""
        import torch
        return (torch.from_numpy(self.node_feature).float().to(device),
                torch.from_numpy(self.align_edge_index).long().to(device),
                torch.from_numpy(self.align_edge_features).float().to(device),
                torch.from_numpy(self.collide_edge_index).long().to(device),
                torch.from_numpy(self.collide_edge_features).float().to(device))


def _draw_pairs(rng: np.random.Generator, n: int, n_pairs: int, band: int,
                forbidden: Optional[np.ndarray] = None) -> np.ndarray:
    """`n_pairs` distinct undirected pairs (u < v <= u + band), returned as keys u * n + v
    in random order, none of them in `forbidden` (sorted key array)."""
    got = np.empty(0, dtype=np.int64)
    while got.size < n_pairs:
        need = n_pairs - got.size
        m = int(need * 1.3) + 64
        u = rng.integers(0, n, size=m, dtype=np.int64)
        v = u + rng.integers(1, band + 1, size=m, dtype=np.int64)
        ok = v < n
        key = (u * n + v)[ok]
        if forbidden is not None and forbidden.size:
            pos = np.searchsorted(forbidden, key)
            pos[pos >= forbidden.size] = forbidden.size - 1
            key = key[forbidden[pos] != key]
        # first occurrence order (np.unique sorts; restore the random draw order)
        uniq, first = np.unique(key, return_index=True)
        key = key[np.sort(first)]
        if got.size:
            key = key[~np.isin(key, got)]
        got = np.concatenate([got, key[:need]])
    return got


def _both_directions(key: np.ndarray, n: int) -> np.ndarray:
    u, v = key // n, key % n
    ei = np.empty((2, 2 * key.size), dtype=np.int64)
    ei[0, 0::2], ei[1, 0::2] = u, v
    ei[0, 1::2], ei[1, 1::2] = v, u
    return ei


def make_super_graph(n_nodes: int, n_adj_edges: int, n_col_edges: Optional[int] = None, tile_count: int = 2,
                     n_edge_types: int = 13, seed: int = 1, band: Optional[int] = None) -> SuperGraph:
    """n_adj_edges / n_col_edges are DIRECTED edge counts (both directions stored), so they are
    rounded down to even; n_col_edges defaults to ceil(1.25 * n_adj_edges), the real
    ratio (10472 / 8502) of data/labyrinth/complete_graph_ring9.pkl."""
    rng = np.random.default_rng(seed)
    n = int(n_nodes)
    if n_col_edges is None:
        n_col_edges = int(np.ceil(1.25 * n_adj_edges))
    band = int(band if band is not None else max(2, int(np.ceil(8.0 * np.sqrt(n)))))
    band = min(band, max(1, n - 1))
    pa, pc = n_adj_edges // 2, n_col_edges // 2
    max_pairs = band * (n - band) + band * (band - 1) // 2      # sum_u min(band, n-1-u)
    if pa + pc > max_pairs:
        raise ValueError(f"cannot place {pa}+{pc} distinct pairs in a band-{band} graph on {n} nodes")

    # nodes: one-hot tile id | per-type area ratio in (0, 1]   (data_util.py:185-189)
    node_type = rng.integers(0, tile_count, size=n)
    area_ratio = np.linspace(1.0, 0.5, tile_count) if tile_count > 1 else np.ones(1)
    node_feature = np.zeros((n, tile_count + 1), dtype=np.float64)
    node_feature[np.arange(n), node_type] = 1.0
    node_feature[:, -1] = area_ratio[node_type]

    adj_key = _draw_pairs(rng, n, pa, band)
    col_key = _draw_pairs(rng, n, pc, band, forbidden=np.sort(adj_key))
    adj_ei = _both_directions(adj_key, n)
    col_ei = _both_directions(col_key, n)

    fe = 2 + n_edge_types
    t_pair = rng.integers(0, n_edge_types, size=pa)
    lengths = np.where(np.arange(n_edge_types) % 2 == 0, 0.57735, 1.0)
    adj_attr = np.zeros((2 * pa, fe), dtype=np.float64)
    t_dir = np.repeat(t_pair, 2)
    adj_attr[:, 1] = lengths[t_dir]
    adj_attr[np.arange(2 * pa), 2 + t_dir] = 1.0

    col_attr = np.zeros((2 * pc, fe), dtype=np.float64)
    col_attr[:, 0] = np.repeat(rng.choice(np.array([0.018, 0.036, 0.054]), size=pc), 2)

    return SuperGraph(node_feature, adj_ei, adj_attr, col_ei, col_attr, tile_count, n_edge_types)


def make_super_graph_on_device(n_nodes: int, n_adj_edges: int, n_col_edges: Optional[int], device, tile_count: int = 2,
                               n_edge_types: int = 13, seed: int = 1):
    """The same family of graphs drawn with torch on `device` (seeded torch generator, so NOT the same graph as
    make_super_graph for a given seed): banded distinct undirected pairs, both directions stored consecutively in
    random (unsorted) order, adjacency and collision sets disjoint, one-hot edge types.  For layouts too large to draw
    on the host in reasonable time (2M nodes / 45M edges: ~70 s with numpy).  Returns the five tensors of
    `SuperGraph.to_torch` (float32 / int64)."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    n = int(n_nodes)
    if n_col_edges is None:
        n_col_edges = int(np.ceil(1.25 * n_adj_edges))
    band = min(max(2, int(np.ceil(8.0 * np.sqrt(n)))), max(1, n - 1))
    pa, pc = n_adj_edges // 2, n_col_edges // 2

    def draw(n_pairs, forbidden=None):
        got = torch.empty(0, dtype=torch.int64, device=dev)
        while got.numel() < n_pairs:
            need = n_pairs - got.numel()
            m = int(need * 1.3) + 64
            u = torch.randint(0, n, (m,), generator=g, device=dev, dtype=torch.int64)
            v = u + torch.randint(1, band + 1, (m,), generator=g, device=dev, dtype=torch.int64)
            key = (u * n + v)[v < n]
            key = torch.unique(key)
            if forbidden is not None:
                key = key[~torch.isin(key, forbidden)]
            if got.numel():
                key = key[~torch.isin(key, got)]
            key = key[torch.randperm(key.numel(), generator=g, device=dev)]
            got = torch.cat([got, key[:need]])
        return got

    def both(key):
        u, v = key // n, key % n
        return torch.stack([torch.stack([u, v], 1).reshape(-1), torch.stack([v, u], 1).reshape(-1)])

    adj_key = draw(pa)
    col_key = draw(pc, adj_key)
    node_type = torch.randint(0, tile_count, (n,), generator=g, device=dev)
    area = torch.linspace(1.0, 0.5, tile_count, device=dev) if tile_count > 1 else torch.ones(1, device=dev)
    x = torch.zeros(n, tile_count + 1, device=dev)
    x[torch.arange(n, device=dev), node_type] = 1.0
    x[:, -1] = area[node_type]
    fe = 2 + n_edge_types
    t_dir = torch.randint(0, n_edge_types, (pa,), generator=g, device=dev).repeat_interleave(2)
    lengths = torch.where(torch.arange(n_edge_types, device=dev) % 2 == 0, 0.57735, 1.0)
    adj_attr = torch.zeros(2 * pa, fe, device=dev)
    adj_attr[:, 1] = lengths[t_dir]
    adj_attr[torch.arange(2 * pa, device=dev), 2 + t_dir] = 1.0
    col_attr = torch.zeros(2 * pc, fe, device=dev)
    return x, both(adj_key), adj_attr, both(col_key), col_attr
