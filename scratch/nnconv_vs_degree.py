"""The column NNConv (fp16-pair kernel, isolated, HIP events) against the adjacency in-degree at 100k nodes, 13 types: columns per
tile go with the degree, type runs per tile do not (all 13 types occur in every 16-row tile from degree ~3 on) -- what a densely
packed column stream (16 entries of one type per column, whatever their rows) could cost at degree 10 is what THIS kernel costs at
the degree that gives the same number of columns."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = 100_000
for deg in (2, 3, 4, 5, 6, 8, 10, 14):
    ea = deg * n
    sg = make_super_graph(n, ea, 125_000, tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    torch.manual_seed(0)
    h = torch.randn(n, 32, device=dev)
    wtab = torch.randn(g.n_types, 32, 32, device=dev) * 0.2
    root = torch.randn(32, 32, device=dev) * 0.2
    bias = torch.randn(32, device=dev)
    part = ops.new_partials(32, dev)
    kw = dict(kernel="cols_f16", max_in_degree=g.max_in_degree)
    for _ in range(5): ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, **kw)
    torch.cuda.synchronize(); ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, **kw)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ncols = int(g.cols.tile_col_ptr[(n + 15) // 16].item()) if g.cols is not None else -1
    print(f"in-degree {deg:2d}: {ncols / ((n + 15) // 16):5.1f} columns per tile, {ea / n * 16 + 16:5.0f} entries per tile; "
          f"us per call (weight image + bounds + kernel: ~10 us of launches around the kernel) min {min(ts):.1f} median {sorted(ts)[2]:.1f}", flush=True)
