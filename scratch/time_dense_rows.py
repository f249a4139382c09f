"""672 -> 256 over the slot-major skip buffer: the rows-per-wave fp16-pair kernel against the block-tile one (both through
tgnn_dense_act_slots_f16_fwd; the op also computes the 22 bounds and, for the rows kernel, the operand image: timed with rocprofv3
per kernel, here end to end and relative)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N in (10_000, 32_000, 100_000, 300_000):
    mid = torch.randn(21, N, 32, device=dev); w = torch.randn(256, 672, device=dev) * 0.05; b = torch.randn(256, device=dev)
    parts = ops.new_partials(256, dev)
    o1, n1 = ops.dense_act(mid, w, b, 1, slot_major=True, f16_split="tile", partials=parts); p1 = parts.clone()
    o2, n2 = ops.dense_act(mid, w, b, 1, slot_major=True, f16_split=True, partials=parts); p2 = parts.clone()
    s1 = p1.view(-1)[:n1 * 512].view(n1, 512).sum(0); s2 = p2.view(-1)[:n2 * 512].view(n2, 512).sum(0)
    print(N, "max rel diff of outputs", float(((o1 - o2).abs().max() / o1.abs().max())), "partials", n1, n2, "max rel diff of column sums",
          float(((s1 - s2).abs() / s1.abs().clamp(min=1e-30)).max()))
    t1 = timeit(lambda: ops.dense_act(mid, w, b, 1, slot_major=True, f16_split="tile", partials=parts))
    t2 = timeit(lambda: ops.dense_act(mid, w, b, 1, slot_major=True, f16_split=True, partials=parts))
    print(f"  N={N}: op with the block-tile kernel {t1:.1f} us, with the rows kernel {t2:.1f} us (both incl. 22 bound launches)")
