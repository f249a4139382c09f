import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
dev = torch.device('cuda:0')
N = 100_000
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (k, m) in ((672, 256), (256, 128), (128, 64), (64, 32)):
    a = torch.randn(N, k, device=dev); w = torch.randn(m, k, device=dev) * 0.05; b = torch.randn(m, device=dev)
    parts = ops.new_partials(m, dev)
    t = timeit(lambda: ops.dense_act(a, w, b, 1, partials=parts))
    stat = torch.randn(4, k, device=dev).contiguous()
    t2 = timeit(lambda: ops.dense_act(a, w, b, 1, partials=parts, in_stat=stat))
    print(f"dense {k}->{m}: {t:.1f} us  {2*N*k*m/t/1e6:.1f} TFLOP/s   with BN-on-load: {t2:.1f} us")
