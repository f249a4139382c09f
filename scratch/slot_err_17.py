"""n = 17: every slot of the skip buffer against the fp64 oracle for the small kernel, the general schedule on type columns and on
edge groups (BatchNorm over 17 rows: the ill-conditioned corner of tests/test_small_layout.py)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_small_layout as T
from tilingnn_amd import ops
from tilingnn_amd.graph_networks import _graph_cache
dev = torch.device('cuda:0')
for n in (17, 300):
    inputs = T._synthetic(n, dev)
    inputs64 = tuple(t.cpu().double() if t.is_floating_point() else t.cpu() for t in inputs)
    for name, limit, groups in (("small", 4096, True), ("general/columns", 0, False), ("general/groups", 0, True)):
        ops.GROUPS = groups
        _graph_cache.clear()
        errs = T._slot_errors(dev, inputs64, inputs, n, 3, limit)
        print(n, name, errs)
