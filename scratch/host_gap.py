"""Host time between the preparation's one synchronisation returning and tgnn_forward_resume being entered / returning
(new layout every forward, 100 000 nodes): the device idles from the end of the NNConv structure's last launch until the
forward's first kernel arrives (profiles/r06_step_trace_100000.txt: ~31 us)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
n = 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net.cache_graph = False
T = {}
lib = _lib.lib
w0, r0, p0, b0 = lib.tgnn_graph_prep_wait, lib.tgnn_forward_resume, lib.tgnn_graph_prep, lib.tgnn_forward_begin
def wait(*a):
    T['wait_in'] = time.perf_counter(); rc = w0(*a); T['wait_out'] = time.perf_counter(); return rc
def resume(*a):
    T['res_in'] = time.perf_counter(); rc = r0(*a); T['res_out'] = time.perf_counter(); return rc
def prep(*a):
    T['prep_in'] = time.perf_counter(); rc = p0(*a); T['prep_out'] = time.perf_counter(); return rc
def begin(*a):
    T['beg_in'] = time.perf_counter(); rc = b0(*a); T['beg_out'] = time.perf_counter(); return rc
class Proxy:
    def __getattr__(self, k):
        return {'tgnn_graph_prep_wait': wait, 'tgnn_forward_resume': resume, 'tgnn_graph_prep': prep, 'tgnn_forward_begin': begin}.get(k) or getattr(lib, k)
import tilingnn_amd.ops as ops
import tilingnn_amd.graph_networks.networks.TilinGNN as M
ops.lib = Proxy(); M.lib = Proxy()
for _ in range(10): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
rows = []
for _ in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    t1 = time.perf_counter()
    rows.append([(T[k] - t0) * 1e6 for k in ('beg_in', 'beg_out', 'prep_in', 'prep_out', 'wait_in', 'wait_out', 'res_in', 'res_out')] + [(t1 - t0) * 1e6])
rows.sort(key=lambda r: r[-1])
m = rows[len(rows) // 2]
print("median call (us from entry): begin %.0f-%.0f  prep %.0f-%.0f  wait %.0f-%.0f  resume %.0f-%.0f  return %.0f" % tuple(m))
print("wait_out -> resume_in: %.0f us; inside resume: %.0f us" % (m[6] - m[5], m[7] - m[6]))
