#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container (needs the read-only mount /root/reference); nothing here
travels to the GPU box except the .npz files it writes.  Recipe = SURVEY.md Appendix A:

  * the reference's graph_networks/{networks/TilinGNN,layers/*,network_utils}.py are imported
    UNCHANGED from /root/reference (sys.path), never copied;
  * `inputs.config` (shapely + file reads at import) is replaced by a stub exposing
    environment.tile_count / network_depth / network_width;
  * torch_geometric is absent from the image, so `torch_geometric.nn.conv.nn_conv.NNConv` and
    `torch_geometric.nn.GINConv` are stood in by the two small classes below, written from
    PyG 1.3.2's published semantics (parameter names `root` [in,out], `bias`, submodule `nn`,
    buffer `eps`, so that the state-dict keys are the real ones);
  * data/labyrinth/complete_graph_ring9.pkl (the only real graph in the checkout) is read with a
    stub Unpickler; tile areas come from the polygons' WKB bytes (shoelace formula).

Outputs (all small):
  labyrinth_ring9_graph.npz   the real graph as compact arrays
  ref_forward_labyrinth.npz   reference fp64 / fp32 forward on it with the seeded recipe weights:
                              probs, per-layer column means / rms of every intermediate, sampled rows
  ref_ops_small.npz           per-op teacher-forced input/output pairs (fp64 results of fp32 inputs, stored as fp32)
                              from the reference on a 256-node induced sub-graph
  tiny_graph.npz              6-node hand-checkable graph (zero in-degree node, collision self loop)
  ref_state_dict_keys.json    the reference's state_dict keys and shapes
"""
import io
import json
import os
import pickle
import struct
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from tilingnn_amd.weights import make_state_dict, state_dict_spec  # noqa: E402


# ----------------------------------------------------------------------------- PyG stand-ins
class NNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, nn, aggr="add", root_weight=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.nn, self.aggr = in_channels, out_channels, nn, aggr
        self.root = torch.nn.Parameter(torch.empty(in_channels, out_channels).uniform_(-1, 1) / in_channels ** 0.5)
        self.bias = torch.nn.Parameter(torch.empty(out_channels).uniform_(-1, 1) / in_channels ** 0.5)

    def forward(self, x, edge_index, edge_attr):
        src, dst = edge_index[0], edge_index[1]
        weight = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
        msg = torch.matmul(x[src].unsqueeze(1), weight).squeeze(1)
        out = torch.zeros(x.shape[0], self.out_channels, dtype=x.dtype).index_add_(0, dst, msg)
        if self.aggr == "mean":
            cnt = torch.bincount(dst, minlength=x.shape[0]).clamp(min=1).to(x.dtype)
            out = out / cnt.unsqueeze(1)
        return out + torch.mm(x, self.root) + self.bias


class GINConv(torch.nn.Module):
    def __init__(self, nn, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn
        self.register_buffer("eps", torch.Tensor([eps]))

    def forward(self, x, edge_index):
        keep = edge_index[0] != edge_index[1]
        src, dst = edge_index[0][keep], edge_index[1][keep]
        agg = torch.zeros_like(x).index_add_(0, dst, x[src])
        return self.nn((1 + self.eps) * x + agg)


def import_reference(tile_count):
    for name in list(sys.modules):
        if name.split(".")[0] in ("graph_networks", "inputs", "torch_geometric"):
            del sys.modules[name]
    inputs = types.ModuleType("inputs"); inputs.__path__ = []
    cfg = types.ModuleType("inputs.config")
    cfg.environment = types.SimpleNamespace(tile_count=tile_count)
    cfg.network_depth, cfg.network_width = 20, 32
    inputs.config = cfg
    sys.modules["inputs"], sys.modules["inputs.config"] = inputs, cfg
    tg = types.ModuleType("torch_geometric"); tg.__path__ = []
    tgnn = types.ModuleType("torch_geometric.nn"); tgnn.__path__ = []
    tgconv = types.ModuleType("torch_geometric.nn.conv"); tgconv.__path__ = []
    tgnc = types.ModuleType("torch_geometric.nn.conv.nn_conv")
    tgnn.GINConv, tgnn.NNConv, tgnc.NNConv = GINConv, NNConv, NNConv
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn,
                        "torch_geometric.nn.conv": tgconv, "torch_geometric.nn.conv.nn_conv": tgnc})
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from graph_networks.networks.TilinGNN import TilinGNN
    return TilinGNN


# ----------------------------------------------------------------------------- real graph
class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.state = st


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("shapely") or module.startswith("tiling"):
            return type(name, (_Stub,), {})
        return super().find_class(module, name)


def _wkb_polygon_area(wkb: bytes) -> float:
    bo = "<" if wkb[0] == 1 else ">"
    gtype, nrings = struct.unpack(bo + "II", wkb[1:9])
    assert gtype == 3, gtype
    off, area = 9, 0.0
    for r in range(nrings):
        (npts,) = struct.unpack(bo + "I", wkb[off:off + 4]); off += 4
        pts = np.frombuffer(wkb[off:off + 16 * npts], dtype=bo + "f8").reshape(npts, 2); off += 16 * npts
        a = 0.5 * abs(np.dot(pts[:-1, 0], pts[1:, 1]) - np.dot(pts[1:, 0], pts[:-1, 1]))
        area += a if r == 0 else -a
    return area


def load_labyrinth():
    d = _Unpickler(open(os.path.join(REF, "data/labyrinth/complete_graph_ring9.pkl"), "rb")).load()
    tiles = d["tiles"]
    ids = np.array([t.state["id"] for t in tiles], dtype=np.int64)
    areas = []
    for t in tiles:
        st = t.state["tile_poly"].state
        wkb = st if isinstance(st, (bytes, bytearray)) else (st[0] if isinstance(st, tuple) else st)
        areas.append(_wkb_polygon_area(bytes(wkb)))
    areas = np.array(areas)
    tile_count = int(ids.max()) + 1
    ef = d["edges_features"]
    adj = np.array(d["adj_edges"], dtype=np.int64).T
    col = np.array(d["colli_edges"], dtype=np.int64).T
    adj_attr = np.array([ef[u][v] for u, v in d["adj_edges"]], dtype=np.float64)
    col_attr = np.array([ef[u][v] for u, v in d["colli_edges"]], dtype=np.float64)
    adj_attr[:, 1] /= d["max_align_length"]                       # util/data_util.py:168-169
    x = np.zeros((len(tiles), tile_count + 1))                    # util/data_util.py:185-189
    x[np.arange(len(tiles)), ids] = 1
    x[:, -1] = areas / d["max_area"]
    # util/data_util.py:110-117 (to_torch_tensor): the network only ever sees the .float() of these
    # arrays.  Round here once, so that the fp64 reference runs on exactly the values the fp32
    # pipeline would feed it (in float64 the align-length column carries ~1e-9 noise: 771 distinct
    # edge-attribute rows; after .float() exactly 13 remain).
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    return dict(x=f32(x), adj=adj, adj_attr=f32(adj_attr), col=col, col_attr=f32(col_attr), tile_count=tile_count,
                max_area=d["max_area"], max_align_length=d["max_align_length"])


def induced_subgraph(g, n_keep):
    def cut(ei, attr):
        keep = (ei[0] < n_keep) & (ei[1] < n_keep)
        return ei[:, keep], attr[keep]
    adj, adj_attr = cut(g["adj"], g["adj_attr"])
    col, col_attr = cut(g["col"], g["col_attr"])
    return dict(x=g["x"][:n_keep], adj=adj, adj_attr=adj_attr, col=col, col_attr=col_attr, tile_count=g["tile_count"])


def to_t(g, dtype):
    return (torch.from_numpy(g["x"]).to(dtype), torch.from_numpy(g["adj"]), torch.from_numpy(g["adj_attr"]).to(dtype),
            torch.from_numpy(g["col"]), torch.from_numpy(g["col_attr"]).to(dtype))


def build_reference_net(TilinGNN, fe, fx, dtype, seed=0, depth=20, width=32):
    net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=width, node_features_dim=fx)
    sd = make_state_dict(fe, depth, width, 1, fx, seed=seed)
    net.load_state_dict(sd, strict=True)
    net = net.to(dtype)
    net.train()                                                   # ml_solver.py:131
    return net


def run_with_hooks(net, inputs):
    cap = {}

    def hook(name):
        def fn(mod, inp, out):
            cap[name + ".in"] = [t.detach().clone() for t in inp if torch.is_tensor(t)]
            cap[name + ".out"] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return fn
    hs = [net.init_node_feature_trans.register_forward_hook(hook("init")),
          net.final_mlp.register_forward_hook(hook("final"))]
    for i, (l1, l2) in enumerate(zip(net.brch_1_graph_conv_layers, net.brch_2_coll_conv_layers)):
        hs += [l1.nnConv.register_forward_hook(hook(f"nnconv.{i}")), l1.register_forward_hook(hook(f"gconv.{i}")),
               l2.ginConv.register_forward_hook(hook(f"gin.{i}")), l2.register_forward_hook(hook(f"cconv.{i}"))]
    x, adj, adj_attr, col, col_attr = inputs
    with torch.no_grad():
        probs, _ = net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col, col_e_features=col_attr)
    for h in hs:
        h.remove()
    cap["probs"] = probs.detach().clone()
    return cap


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)                                      # deterministic summation order
    g = load_labyrinth()
    n, fe, fx = g["x"].shape[0], g["adj_attr"].shape[1], g["x"].shape[1]
    print(f"labyrinth: N={n} Ea={g['adj'].shape[1]} Ec={g['col'].shape[1]} Fe={fe} Fx={fx}")

    # ---- fixture 1: the real graph, compact
    uniq, type_id = np.unique(g["adj_attr"], axis=0, return_inverse=True)
    col_uniq, col_type = np.unique(g["col_attr"][:, 0], return_inverse=True)
    assert len(uniq) < 65536 and len(col_uniq) < 65536
    print(f"distinct adj edge-attribute rows (after .float()): {len(uniq)}; distinct collision areas: {len(col_uniq)}")
    np.savez_compressed(os.path.join(HERE, "labyrinth_ring9_graph.npz"),
                        node_type=np.argmax(g["x"][:, :-1], axis=1).astype(np.uint8),
                        node_area=g["x"][:, -1].astype(np.float32),
                        adj=g["adj"].astype(np.int32), adj_type=type_id.reshape(-1).astype(np.uint16),
                        adj_rows=uniq.astype(np.float32),
                        col=g["col"].astype(np.int32), col_area_type=col_type.reshape(-1).astype(np.uint16),
                        col_areas=col_uniq.astype(np.float32),
                        tile_count=np.int64(g["tile_count"]))

    TilinGNN = import_reference(g["tile_count"])

    # ---- state-dict keys of the reference vs the build's spec
    net64 = build_reference_net(TilinGNN, fe, fx, torch.float64)
    ref_keys = {k: list(v.shape) for k, v in net64.state_dict().items()}
    spec = {k: list(v) for k, v in state_dict_spec(fe, 20, 32, 1, fx).items()}
    assert ref_keys == spec, "state-dict layout mismatch"
    assert list(ref_keys) == list(spec), "state-dict order mismatch"
    json.dump(ref_keys, open(os.path.join(HERE, "ref_state_dict_keys.json"), "w"), indent=0)
    print(f"state dict: {len(ref_keys)} entries, layout == spec")

    # ---- fixture 2: reference forward on the real graph, fp64 and fp32
    cap64 = run_with_hooks(net64, to_t(g, torch.float64))
    net32 = build_reference_net(TilinGNN, fe, fx, torch.float32)
    cap32 = run_with_hooks(net32, to_t(g, torch.float32))
    rows = np.linspace(0, n - 1, 16).astype(np.int64)
    out = {"probs_fp64": cap64["probs"].numpy(), "probs_fp32": cap32["probs"].numpy(), "sample_rows": rows}
    for i in range(20):
        for name in ("nnconv", "gconv", "gin", "cconv"):
            t = cap64[f"{name}.{i}.out"].numpy()
            out[f"{name}.{i}.colmean"] = t.mean(0)
            out[f"{name}.{i}.colrms"] = np.sqrt((t ** 2).mean(0))
            out[f"{name}.{i}.rows"] = t[rows]
    out["init.colmean"] = cap64["init.out"].numpy().mean(0)
    out["init.rows"] = cap64["init.out"].numpy()[rows]
    # running statistics after ONE train-mode forward (momentum 0.1, unbiased variance)
    sd_after = net64.state_dict()
    for k in ("init_node_feature_trans.mlp.0.batch_norm", "brch_1_graph_conv_layers.0.batch_norm",
              "brch_2_coll_conv_layers.19.batch_norm", "final_mlp.0.mlp.3.batch_norm"):
        out[k + ".running_mean"] = sd_after[k + ".running_mean"].numpy()
        out[k + ".running_var"] = sd_after[k + ".running_var"].numpy()
        out[k + ".num_batches_tracked"] = sd_after[k + ".num_batches_tracked"].numpy()
    np.savez_compressed(os.path.join(HERE, "ref_forward_labyrinth.npz"), **out)
    d = (cap64["probs"] - cap32["probs"].double()).abs().max().item()
    print(f"reference fp32 vs fp64 end-to-end max abs diff: {d:.3e}")

    # ---- fixture 3: per-op teacher-forced pairs on a 256-node induced sub-graph
    gs = induced_subgraph(g, 256)
    net64s = build_reference_net(TilinGNN, fe, fx, torch.float64)
    caps = run_with_hooks(net64s, to_t(gs, torch.float64))
    ops = {"x": gs["x"].astype(np.float32), "adj": gs["adj"].astype(np.int32), "adj_attr": gs["adj_attr"].astype(np.float32),
           "col": gs["col"].astype(np.int32), "col_attr": gs["col_attr"].astype(np.float32)}
    xs, adjs, adj_attrs, cols, _ = to_t(gs, torch.float64)
    adj_attr32 = adj_attrs.float().double()
    with torch.no_grad():
        # init MLP on the fp32-rounded x
        f32 = lambda t: t.numpy().astype(np.float32)      # expected values: fp64 results rounded once to fp32
        ops["init.out"] = f32(net64s.init_node_feature_trans(xs.float().double()))
        for i in (0, 2, 19):
            l1, l2 = net64s.brch_1_graph_conv_layers[i], net64s.brch_2_coll_conv_layers[i]
            h1 = caps[f"gconv.{i}.in"][0].float()               # fp32-rounded teacher-forced inputs
            h2 = caps[f"cconv.{i}.in"][0].float()
            ops[f"h1_in.{i}"], ops[f"h2_in.{i}"] = h1.numpy(), h2.numpy()
            ops[f"nnconv.{i}.out"] = f32(l1.nnConv(h1.double(), adjs, adj_attr32))
            ops[f"gconv.{i}.out"] = f32(l1(h1.double(), adjs, adj_attr32)[0])
            ops[f"gin.{i}.out"] = f32(l2.ginConv(h2.double(), cols))
            ops[f"cconv.{i}.out"] = f32(l2(h2.double(), cols)[0])
        cat = caps["final.in"][0][:96].float()                  # 96 rows keep the fixture small
        ops["final.in"] = cat.numpy()
        ops["final.out"] = f32(net64s.final_mlp(cat.double()))
    np.savez_compressed(os.path.join(HERE, "ref_ops_small.npz"), **ops)

    # ---- fixture 4: tiny hand-checkable graph; node 5 has no in-edges in either set,
    # the collision set holds a self loop (3,3) that GINConv must drop, NNConv keeps (2,2).
    tiny = dict(
        x=np.array([[1, 0, 1.0], [0, 1, 0.5], [1, 0, 1.0], [0, 1, 0.5], [1, 0, 1.0], [0, 1, 0.5]], dtype=np.float64),
        adj=np.array([[0, 1, 1, 2, 2, 3, 0, 4, 5, 2], [1, 0, 2, 1, 3, 2, 4, 0, 0, 2]], dtype=np.int64),
        col=np.array([[0, 2, 1, 3, 3, 4, 3, 5], [2, 0, 3, 1, 4, 3, 3, 1]], dtype=np.int64), tile_count=2)
    rng = np.random.default_rng(7)
    t = rng.integers(0, 4, size=tiny["adj"].shape[1])
    tiny["adj_attr"] = np.zeros((tiny["adj"].shape[1], 6)); tiny["adj_attr"][:, 1] = np.where(t % 2 == 0, 0.57735, 1.0)
    tiny["adj_attr"][np.arange(t.size), 2 + t] = 1.0
    tiny["col_attr"] = np.zeros((tiny["col"].shape[1], 6)); tiny["col_attr"][:, 0] = 0.036
    TilinGNN2 = import_reference(2)
    nett = build_reference_net(TilinGNN2, 6, 3, torch.float64, seed=3, depth=3, width=32)
    capt = run_with_hooks(nett, to_t(tiny, torch.float64))
    np.savez_compressed(os.path.join(HERE, "tiny_graph.npz"), x=tiny["x"], adj=tiny["adj"], adj_attr=tiny["adj_attr"],
                        col=tiny["col"], col_attr=tiny["col_attr"], probs_fp64=capt["probs"].numpy(),
                        nnconv0=capt["nnconv.0.out"].numpy(), gin0=capt["gin.0.out"].numpy(),
                        gconv2=capt["gconv.2.out"].numpy(), cconv2=capt["cconv.2.out"].numpy(),
                        depth=np.int64(3), width=np.int64(32), seed=np.int64(3))
    for f in sorted(os.listdir(HERE)):
        if f.endswith((".npz", ".json")):
            print(f"  {f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
