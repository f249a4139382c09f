"""Loaders for the committed golden fixtures (tests/golden/*.npz, made by generate_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_labyrinth_graph():
    """The real data/labyrinth/complete_graph_ring9 graph, expanded to the arrays
    BrickLayout.get_data_as_torch_tensor would produce (float64 / int64 numpy)."""
    z = np.load(os.path.join(GOLDEN, "labyrinth_ring9_graph.npz"))
    n, tc = z["node_type"].shape[0], int(z["tile_count"])
    x = np.zeros((n, tc + 1))
    x[np.arange(n), z["node_type"]] = 1.0
    x[:, -1] = z["node_area"]
    adj_attr = z["adj_rows"][z["adj_type"]]
    col_attr = np.zeros((z["col"].shape[1], adj_attr.shape[1]))
    col_attr[:, 0] = z["col_areas"][z["col_area_type"]]
    return dict(x=x, adj=z["adj"].astype(np.int64), adj_attr=adj_attr, col=z["col"].astype(np.int64),
                col_attr=col_attr, tile_count=tc, adj_type=z["adj_type"].astype(np.int64))


def graph_tensors(g, dtype=torch.float32, device="cpu"):
    return (torch.from_numpy(np.asarray(g["x"])).to(dtype).to(device),
            torch.from_numpy(np.asarray(g["adj"]).astype(np.int64)).to(device),
            torch.from_numpy(np.asarray(g["adj_attr"])).to(dtype).to(device),
            torch.from_numpy(np.asarray(g["col"]).astype(np.int64)).to(device),
            torch.from_numpy(np.asarray(g["col_attr"])).to(dtype).to(device))


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))
