import sys, torch
sys.path.insert(0, '.')
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tilingnn_amd import ops
dev = torch.device('cuda:0')
x, adj, attr, col, _ = graph_tensors(load_labyrinth_graph(), torch.float32, dev)
for _ in range(3): g = ops.prepare_graph(1254, adj, attr, col)
torch.cuda.synchronize()
# the result words sit at the end of the one int32 buffer the small path allocates: 32 ints after tmp
base = g.adj_rowptr
buf = base.untyped_storage()
allv = torch.tensor([], dtype=torch.int32, device=dev).set_(buf)
res = allv[-32:].cpu().tolist()
names = ["zero + barrier", "count + barrier", "row starts (scan)", "arrival fill + barrier", "rank pass", "dedup + barrier", "numbering, types + barrier", "columns"]
for nm, t in zip(names, res[8:16]): print(f"{nm:28s} {t * 0.01:7.2f} us")
