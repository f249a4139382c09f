import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
net.cache_graph = True
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
def run(stream, n=20):
    for _ in range(3):
        with torch.cuda.stream(stream): net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(stream): net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("default stream      ", run(torch.cuda.current_stream()))
hi = torch.cuda.Stream(priority=-1)
print("high-priority stream", run(hi))
lo = torch.cuda.Stream(priority=0)
print("normal new stream   ", run(lo))
print("default stream again", run(torch.cuda.current_stream()))
