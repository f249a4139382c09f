"""Edge-group NNConv (csrc/nnconv_eg.hip) against an fp64 evaluation and against the column kernel; times both ops with
HIP events (argv: n ea [types]).  Run under rocprofv3 --kernel-trace --stats for the kernels alone."""
import sys, os, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ea = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 13
dev = torch.device('cuda:0')
sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=nt, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
torch.manual_seed(0)
h = torch.randn(n, 32, device=dev)
wtab = torch.rand(g.n_types, 32, 32, device=dev)
root = torch.randn(32, 32, device=dev) * 0.2
bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
part2 = ops.new_partials(32, dev)

# fp64 evaluation
rp = g.adj_rowptr.long()
deg = (rp[1:] - rp[:-1])
dst = torch.repeat_interleave(torch.arange(n, device=dev), deg)
src = g.adj_src.long()[: dst.numel()]
typ = g.adj_type.long()[: dst.numel()]
h64, w64 = h.double(), wtab.double()
acc = torch.zeros(n, 32, dtype=torch.float64, device=dev)
for t in range(g.n_types):
    sel = typ == t
    acc.index_add_(0, dst[sel], h64[src[sel]] @ w64[t])
ref = acc / deg.clamp(min=1).double()[:, None] + h64 @ root.double() + bias.double()
ref = torch.where(ref >= 0, ref, ref * 0.01)

out_c, np_c = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel="cols_f16")
out_e, np_e = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part2, kernel="eg")
torch.cuda.synchronize()
scale = float(ref.abs().max())
print(f"n={n} ea={ea} T={g.n_types} |ref|max {scale:.3f}")
print(f"cols_f16 vs fp64: max abs {float((out_c.double() - ref).abs().max()):.3e}")
print(f"eg       vs fp64: max abs {float((out_e.double() - ref).abs().max()):.3e}")
grp = ops.graph_groups(g)
ng = int(grp.tile_grp_ptr[-1])
nc = int(ops.graph_columns(g).tile_col_ptr[-1])
ntile = (n + 15) // 16
print(f"groups {ng} ({ng / ntile:.2f} per tile, fill {dst.numel() / max(ng - ntile, 1) / 16:.3f}); columns {nc} ({nc / ntile:.2f} per tile)")
pc = part[:np_c].double().sum(0)
pe = part2[:np_e].double().sum(0)
print(f"BN partial sums: rel diff {float(((pc - pe).abs() / pc.abs().clamp(min=1e-9)).max()):.3e}  (blocks {np_c} / {np_e})")

for name in ("cols_f16", "eg"):
    for _ in range(5):
        ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=name)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=name)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{name:9s} us/op (incl. bounds + weight image): min {min(ts):.1f} median {sorted(ts)[2]:.1f}")
