#!/usr/bin/env python3
"""bench.py -- scored tile-nodes/s of the TilinGNN graph-conv scoring forward on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full TilinGNN.forward (ML_Solver.predict's network call,
/root/reference/solver/ml_solver/ml_solver.py:39-43) over one synthetic super-graph whose
inputs are already resident in HBM; graph preparation (int64 COO -> CSR, edge-type de-dup) is
INSIDE the timed step (every greedy round of the reference presents a new sub-layout).
Workload at N=1: BASELINE.json's metric configuration -- 100 000 nodes / 1 000 000 adjacency
edges / 1 250 000 collision edges, 30-60-90 tile set (tile_count 2 -> Fx 3), T = 13 edge types
(Fe 15), network_width 32, depth 20, fp32, seeded (BASELINE.json
north_star "100k-node/1M-edge 30-60-90 super-graph").  At N GPUs the graph has N x 100 000
nodes, node-range sharded (weak scaling); value = all nodes scored / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured with HIP events on the
launch stream in a second instrumented pass) and `cpu_baseline` (the CPU oracle = a port of the
reference's op sequence, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch

NODES_PER_GPU = 100_000
ADJ_PER_GPU = 1_000_000
COL_PER_GPU = 1_250_000
TILE_COUNT, N_TYPES, WIDTH, DEPTH = 2, 13, 32, 20
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak = fp32 vector peak
F16_MFMA_PEAK_TFLOPS = 2516.0  # v_mfma_f32_*_f16 dense peak (256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz)


def nnconv_bytes(n, ea, t, c=32, s=4):
    """ALGORITHMIC bytes of one NNConv-mean launch, SURVEY.md section 8d verbatim:
    B_nn = (N+1)*4 [rowptr] + Ea*4 [src idx] + Ea*1 [type id] + N*C*s [read h once] + N*C*s [write out] + T*C^2*s."""
    return (n + 1) * 4 + ea * 4 + ea * 1 + n * c * s + n * c * s + t * c * c * s


def merge_bytes(n, c=32, s=4):
    return 4 * n * c * s


def nnconv_flops(n, ea, c=32):
    return 2.0 * c * c * (ea + n)      # per edge one [1,C]x[C,C] product, per node the root term


def gin_bytes(n, ec, c=32, s=4):
    return (n + 1) * 4 + ec * 4 + 2 * n * c * s


def forward_bytes(n, ea, ec, t, fe=15, fx=3, c=32, d=20, s=4):
    """B_fwd of SURVEY.md section 8d (compulsory traffic of the whole forward)."""
    api = 2 * 8 * (ea + ec) + ea * fe * 4 + n * fx * 4
    b_nn = (n + 1) * 4 + ea * 4 + ea * 1 + 2 * n * c * s + t * c * c * s
    b_gin = gin_bytes(n, ec, c, s)
    b_mrg = 4 * n * c * s
    return api + d * (b_nn + b_gin + b_mrg + n * c * s) + n * (d + 1) * c * s + n * 4


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        return ""


def cpu_baseline():
    """The reference's CPU op sequence (oracle = port, incl. the materialised [Ea, C*C] tensor), fp32, no_grad,
    train-mode BN, on this box's host cores: median of 3 forwards at BASELINE config 2 (10 000 nodes / 80 000 + 100 000
    edges) = `value`, plus one forward of a 20 000-node / 200 000 + 250 000-edge sample of the benchmark's
    own generator (1/5 of the GPU workload; the cost is linear in Ea).  Also the per-op parity probe of smoke()."""
    from oracle import tilingnn_oracle as orc
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    # torch CPU ops stop scaling (and then slow down) well before a 256-thread host is full: measured on
    # the MI355X box (2 x EPYC 9575F): 8 thr 1121, 32 thr 1129, 64 thr 830, 128 thr 448, 256 thr 65 nodes/s.
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = make_state_dict(2 + N_TYPES, DEPTH, WIDTH, 1, TILE_COUNT + 1, seed=0)

    def run(n, ea, ec, seed, reps, reference_ops=True):
        sg = make_super_graph(n, ea, ec, tile_count=TILE_COUNT, n_edge_types=N_TYPES, seed=seed)
        x, adj, adj_attr, col, _ = sg.to_torch("cpu")
        ts = []
        with torch.no_grad():
            for _ in range(reps):
                t0 = time.perf_counter()
                if reference_ops:
                    # the reference's modules -- torch.nn.Sigmoid, nn.LeakyReLU, nn.BatchNorm1d(train): one fused pass each
                    # (edge_conv.py:12, TilinGNN.py:31, layers/util.py:28-37) -- not the decomposed forms the fp64 checker keeps
                    with orc.reference_ops():
                        probs, _ = orc.tilingnn_forward(sd, x, adj, adj_attr, col)
                else:
                    probs, _ = orc.tilingnn_forward(sd, x, adj, adj_attr, col)
                ts.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(probs).all())
        return sorted(ts)[len(ts) // 2], ts

    t2, all2 = run(10_000, 80_000, 100_000, 1, 3)
    t2_dec, _ = run(10_000, 80_000, 100_000, 1, 1, reference_ops=False)      # once, so that the factor is on file
    t20, _ = run(20_000, 200_000, 250_000, 11, 1)
    # BASELINE config 0: the reference's own example layout (what its greedy solver scores every round)
    real = None
    try:
        from tests.golden_util import graph_tensors, load_labyrinth_graph
        xl, adjl, attrl, coll, _ = graph_tensors(load_labyrinth_graph(), torch.float32)
        sdl = make_state_dict(int(attrl.shape[1]), DEPTH, WIDTH, 1, int(xl.shape[1]), seed=0)
        tl = []
        with torch.no_grad():
            for _ in range(5):
                t0 = time.perf_counter()
                with orc.reference_ops():
                    orc.tilingnn_forward(sdl, xl, adjl, attrl, coll)
                tl.append(time.perf_counter() - t0)
        real = {"ms_per_forward": sorted(tl)[2] * 1e3, "what": "median of 5 forwards of the labyrinth layout (1254 nodes, 8502 + 10472 edges)"}
    except FileNotFoundError:
        pass
    return {"config0_real_layout": real, "value": 10_000 / t2, "unit": "nodes/s", "cores": cores, "kind": "port",
            "sample": f"median of 3 forwards at BASELINE config 2 (N=10000 Ea=80000 Ec=100000, seed 1): "
                      f"{', '.join(f'{t:.1f}' for t in all2)} s; the reference's op sequence with its own fused modules (torch.sigmoid, "
                      f"F.leaky_relu, F.batch_norm(training=True): oracle.reference_ops); torch {torch.__version__} CPU, {_cpu_model()}",
            "decomposed_checker_forms": {"value": 10_000 / t2_dec, "seconds": t2_dec,
                                         "what": "the same forward with the fp64 checker's decomposed sigmoid / LeakyReLU / BatchNorm (what "
                                                 "rounds 1-5 timed: 3.4 x slower on the build box) -- on file for the factor, not the baseline"},
            "sample_20k": {"value": 20_000 / t20, "seconds": t20,
                           "what": "1 forward, N=20000 Ea=200000 Ec=250000 (the benchmark's generator at 1/5 of its size)"}}


def parity_probe(dev):
    """Per-op, teacher-forced max-norm relative error against the fp64 oracle on the real labyrinth graph (what smoke()
    asserts): GraphConv (NNConv + LeakyReLU + BN) and CollConv (GIN + LeakyReLU + BN) of layer 0."""
    from oracle import tilingnn_oracle as orc
    from tests.golden_util import graph_tensors, load_labyrinth_graph
    from tilingnn_amd import TilinGNN
    from tilingnn_amd.weights import make_state_dict
    g = load_labyrinth_graph()
    sd = make_state_dict(15, 20, 32, 1, 3, seed=0)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    x, adj, adj_attr, col, _ = graph_tensors(g, torch.float32, dev)
    sd64 = orc.cast_sd(sd, torch.float64)
    xc, adjc, attrc, colc, _ = graph_tensors(g, torch.float64)
    with torch.no_grad():
        h0_32 = orc.init_node_feature_trans(xc, sd64).float()
        want_g = orc.graph_conv(h0_32.double(), adjc, attrc, sd64, "brch_1_graph_conv_layers.0")
        want_c = orc.coll_conv(h0_32.double(), colc, sd64, "brch_2_coll_conv_layers.0")
    got_g = net.brch_1_graph_conv_layers[0](h0_32.to(dev), adj, adj_attr)[0]
    got_c = net.brch_2_coll_conv_layers[0](h0_32.to(dev), col)[0]
    return {"graph": "labyrinth ring-9 (1254 nodes), layer 0, teacher forced, vs fp64 oracle, max-norm relative",
            "graphconv_bn": orc.rel_max_err(got_g.cpu(), want_g), "collconv_bn": orc.rel_max_err(got_c.cpu(), want_c),
            "bars": "north_star 1e-5; SURVEY 8c allows 2e-4 for the ill-conditioned collision-branch BatchNorm"}


CLASS_NAMES = ["edge_weights", "dense_init", "nnconv", "gin", "bn_finalize", "merge", "dense_final", "-"]


def profiled_classes(net, x, adj, adj_attr, col, steps):
    """Per-kernel-class time of the forward, HIP events on the launch stream (tgnn_forward_profiled: the single-stream
    schedule with an event pair around every launch) -> ({class: {ms_per_forward, launches_per_forward}}, n_types)."""
    from tilingnn_amd import ops
    from tilingnn_amd._lib import check, lib, ptr
    dev = x.device
    n = int(x.shape[0])
    graph = ops.prepare_graph(n, adj, adj_attr, col)
    dims = net._dims()
    table, _ = net._param_table()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, 1, dtype=torch.float32, device=dev)
    ms = (C.c_float * 8)()
    cnt = (C.c_int32 * 8)()
    g = graph.c_struct()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(steps):
        check(lib.tgnn_forward_profiled(C.byref(dims), table, ptr(x), ptr(adj_attr), C.byref(g), 1, 0, ptr(probs),
                                        ptr(ws), ws_bytes, stream, ms, cnt))
    return ({CLASS_NAMES[i]: {"ms_per_forward": ms[i] / steps, "launches_per_forward": cnt[i] // steps} for i in range(7)},
            graph.n_types)


def in_forward_classes(net, x, adj, adj_attr, col, steps):
    """Average launch duration of the adjacency chain's kernels INSIDE the production (two-stream) forward: HIP events on
    the stream they are launched on, the collision chain running beside them (tgnn_forward_profiled_two_stream)."""
    from tilingnn_amd import _lib, ops
    from tilingnn_amd._lib import check, lib, ptr
    dev = x.device
    n = int(x.shape[0])
    graph = ops.prepare_graph(n, adj, adj_attr, col)
    dims = net._dims()
    table, _ = net._param_table()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, 1, dtype=torch.float32, device=dev)
    ms = (C.c_float * 8)()
    cnt = (C.c_int32 * 8)()
    g = graph.c_struct()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    side = _lib.side_stream(dev)
    if not side.value:
        return None
    for _ in range(steps):
        check(lib.tgnn_forward_profiled_two_stream(C.byref(dims), table, ptr(x), ptr(adj_attr), C.byref(g), 1, ptr(probs),
                                                   ptr(ws), ws_bytes, stream, side, ms, cnt))
    return {CLASS_NAMES[i]: {"ms_per_forward": ms[i] / steps, "launches_per_forward": cnt[i] // steps} for i in (2, 5)}


def stamped_nnconv_us(net, x, adj, adj_attr, col, steps):
    """Average duration of the column NNConv launches INSIDE the production forward (two chains, no event, no profiler): the
    kernel stamps the device's wall clock in its first and its last block (tgnn_forward_stamped) -- what a kernel trace
    reports for the launch.  -> (average us, launches averaged) or None."""
    from tilingnn_amd import _lib, ops
    from tilingnn_amd._lib import check, lib, ptr
    dev = x.device
    n = int(x.shape[0])
    graph = ops.prepare_graph(n, adj, adj_attr, col)
    dims = net._dims()
    table, _ = net._param_table()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, 1, dtype=torch.float32, device=dev)
    g = graph.c_struct()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    side = _lib.side_stream(dev)
    depth = int(dims.network_depth)
    us = (C.c_float * depth)()
    tot, cnt = 0.0, 0
    for k in range(steps + 2):
        check(lib.tgnn_forward_stamped(C.byref(dims), table, ptr(x), ptr(adj_attr), C.byref(g), 1, ptr(probs), ptr(ws), ws_bytes,
                                       stream, side, us))
        if k >= 2:                                           # (two warm-up forwards)
            vals = [float(v) for v in us if v > 0]
            tot, cnt = tot + sum(vals), cnt + len(vals)
    return (tot / cnt, cnt) if cnt else None


def gather_ceiling(dev, n_rows=100_000):
    """The measured ceiling the two gather kernels run against (csrc/ubench.hip; profiles/r05_gather_ceiling.txt,
    r05_nnconv_study.txt): GB/s of 128-byte row gathers from a cache-resident table, per access shape, on THIS device."""
    from tilingnn_amd._lib import check, lib, ptr
    table = torch.zeros(n_rows * 32, dtype=torch.float32, device=dev)
    sink = torch.empty(256 * 1024, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for name, shape in (("nnconv_lane_map_16rows_x_64B", 0), ("whole_rows_8lanes_x_16B", 1)):
        rate = C.c_double(0.0)
        check(lib.tgnn_ubench_row_gather(shape, ptr(table), n_rows, ptr(sink), 400, 3, C.byref(rate), stream))
        out[name] = rate.value * 128.0 / 1e9
    return out


def kernel_roofline(class_ms, n, ea, ec, n_types, edge_groups=True):
    """`roofline` of the NNConv column kernel (the path's scatter-add) + the GIN pair and merge, all against the HBM
    bound with SURVEY 8d's algorithmic bytes; `achieved` = bytes / the average launch duration of the events above."""
    def per_launch_s(k):
        return class_ms[k]["ms_per_forward"] / max(1, class_ms[k]["launches_per_forward"]) * 1e-3
    t_nn = per_launch_s("nnconv")
    b_alg = nnconv_bytes(n, ea, n_types)
    out = {"kernel": "nnconv32_eg_kernel (NNConv mean over edge groups: 16 source rows per gather, two chained MFMA products, per layer)"
                     if edge_groups else "nnconv32_cols_kernel (NNConv mean as a type-column MFMA product, per layer)",
           "bound": "hbm",
           "achieved": b_alg / t_nn / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_alg / t_nn / 1e9 / HBM_PEAK_GBS,
           "traffic": None, "algorithmic_bytes_per_launch": b_alg, "avg_launch_us": t_nn * 1e6,
           "timing": "HIP events on the launch stream, instrumented single-stream pass of this run",
           "flops_per_launch": nnconv_flops(n, ea)}
    t_gin = per_launch_s("gin")           # aggregate + MLP kernels of one layer (one class in the profiled pass)
    b_gin = gin_bytes(n, ec) + 2 * n * 32 * 4
    out["gin_kernel"] = {"avg_launch_us": t_gin * 1e6, "algorithmic_bytes_per_launch": b_gin,
                         "what": "gin32_aggregate_kernel + gin32_mlp_kernel of one layer (B_gin + the MLP's read and write)",
                         "achieved": b_gin / t_gin / 1e9, "frac": b_gin / t_gin / 1e9 / HBM_PEAK_GBS}
    t_m = per_launch_s("merge")
    out["merge_kernel"] = {"avg_launch_us": t_m * 1e6, "algorithmic_bytes_per_launch": merge_bytes(n),
                           "achieved": merge_bytes(n) / t_m / 1e9, "frac": merge_bytes(n) / t_m / 1e9 / HBM_PEAK_GBS}
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher around it: N ranks of this script through torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1, a free port), their stdout -- rank 0's ONE JSON line -- passed through.  Fails loudly, and
    before anything is started, when fewer than N GPUs are visible; a run that outlives --launch-timeout is killed as a whole."""
    import signal
    import socket
    import subprocess
    n = int(args.gpus)
    if not args.launcher_selftest:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have}", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {n}-rank run did not finish within {args.launch_timeout:.0f} s: killed", file=sys.stderr)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        return 3


def launcher_selftest(rank: int, world: int) -> None:
    """The launcher's own path without a GPU: gloo rendezvous, one all-reduce on the CPU, rank 0 reports how many ranks it saw."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    ones = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ones)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "ranks_seen": int(ones.item()), "n_gpus": world}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nodes-per-gpu", type=int, default=NODES_PER_GPU)
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step timing")
    ap.add_argument("--no-extra-sizes", action="store_true", help="skip the 500k / 2M-node single-GPU lines")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded (RCCL) schedule even at world size 1 (exercises the multi-GPU code path)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's mode): the workload PER GPU is fixed; strong: the TOTAL is fixed and split over the ranks")
    ap.add_argument("--config", choices=["headline", "4", "5"], default="headline",
                    help="headline: 100k nodes / 1M + 1.25M edges; 4 / 5: SURVEY 8d's fixed totals (500k / 6M + 7.5M, tile_count 2, seed 3; "
                         "2M / 20M + 25M, tile_count 1, seed 4) -- meant for --scaling strong over 4 / 8 GPUs")
    ap.add_argument("--spawn", action="store_true",
                    help="start the N ranks through torch.distributed.run from here even for N = 1 (N > 1 without WORLD_SIZE in the "
                         "environment does so by itself)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="the launcher alone: every rank joins a gloo group on the CPU, all-reduces a one, rank 0 prints "
                         '{"launcher_selftest": true, "ranks_seen": N}; needs no GPU')
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="seconds after which a self-launched run is killed")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N "
                         "(or leave WORLD_SIZE unset: bench.py then starts them itself)")
    if args.launcher_selftest:
        return launcher_selftest(rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sharded = world > 1 or args.force_sharded
    saved_stdout_fd = None
    if sharded:
        # RCCL prints a version banner to fd 1 when the communicator is created; keep stdout = the one JSON line
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from tilingnn_amd import TilinGNN
    from tilingnn_amd._lib import ModelDims, check, lib, ptr
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict

    scale = args.nodes_per_gpu / NODES_PER_GPU
    # the workload: (nodes, adjacency edges, collision edges) of ONE unit, tile_count, seed; weak scaling: one unit per GPU,
    # strong scaling: one unit in all, node-range sharded over the ranks
    unit = {"headline": (args.nodes_per_gpu, int(ADJ_PER_GPU * scale), int(COL_PER_GPU * scale), TILE_COUNT, 2),
            "4": (500_000, 6_000_000, 7_500_000, 2, 3), "5": (2_000_000, 20_000_000, 25_000_000, 1, 4)}[args.config]
    mult = world if args.scaling == "weak" else 1
    n_total, ea_total, ec_total = unit[0] * mult, unit[1] * mult, unit[2] * mult
    tile_count = unit[3]
    if args.config != "headline":
        args.no_extra_sizes = args.no_train_step = True            # (the side measurements belong to the headline workload)
    sg = make_super_graph(n_total, ea_total, ec_total, tile_count=tile_count, n_edge_types=N_TYPES, seed=unit[4])
    fe, fx = 2 + N_TYPES, tile_count + 1
    net = TilinGNN(adj_edge_features_dim=fe, network_depth=DEPTH, network_width=WIDTH, node_features_dim=fx)
    net.load_state_dict(make_state_dict(fe, DEPTH, WIDTH, 1, fx, seed=0), strict=True)
    net = net.to(dev).train()               # inference in train mode, as ml_solver.py:131 leaves it
    net.cache_graph = False                 # graph preparation is part of every timed step

    if not sharded:
        x, adj, adj_attr, col, col_attr = sg.to_torch(dev)

        def step():
            return net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col, col_e_features=col_attr)[0]
        n_local, ea_local, ec_local = n_total, ea_total, ec_total
    else:
        from tilingnn_amd.dist import ShardedTilinGNN
        shard_runner = ShardedTilinGNN(net, sg, rank, world, dev)
        step = shard_runner.step
        n_local, ea_local, ec_local = shard_runner.n_local, shard_runner.ea_local, shard_runner.ec_local

    def barrier():
        if sharded:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    ranks_seen = 1
    if sharded:
        import torch.distributed as dist
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)                                # (every rank that actually joined adds its one)
        ranks_seen = int(ones.item())

    for _ in range(args.warmup):
        step()
    barrier()
    # EXACTLY K steps between two barriers; one event behind every step: the spacing of consecutive events is the
    # step time the stream saw (host gaps included), its median is what SURVEY 8d asks to be reported
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(args.steps):
        out = step()
        evs[k + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step_ms = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps))
    median_ms = per_step_ms[len(per_step_ms) // 2] if args.steps % 2 else \
        0.5 * (per_step_ms[args.steps // 2 - 1] + per_step_ms[args.steps // 2])
    if sharded:
        import torch.distributed as dist
        tmax = torch.tensor([dt, median_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, median_ms = float(tmax[0].item()), float(tmax[1].item())
    assert bool(torch.isfinite(out).all())
    mean_ms = dt / args.steps * 1e3
    ms_per_step = median_ms
    value = n_total / (median_ms * 1e-3)

    # ---- cached-layout variant (prep amortised, e.g. repeated predict on one BrickLayout)
    cached_ms = None
    if not sharded:
        net.cache_graph = True
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        cached_ms = (time.perf_counter() - t1) / args.steps * 1e3
        net.cache_graph = False

    # ---- roofline of the dominant kernel: second, instrumented pass (HIP events on the launch stream)
    roofline, class_ms = None, None
    if not sharded:
        class_ms, n_types_seen = profiled_classes(net, x, adj, adj_attr, col, args.steps)
        dom = max(("nnconv", "gin", "dense_final"), key=lambda k: class_ms[k]["ms_per_forward"])
        from tilingnn_amd import ops as _ops
        roofline = kernel_roofline(class_ms, n_total, ea_total, ec_total, n_types_seen,
                                   edge_groups=bool(_ops.GROUPS and _ops.runs_general_schedule(n_total)))
        roofline["slowest_class"] = dom
        # THE headline fraction is the kernel's average launch duration inside the production two-stream forward (the
        # collision chain competes for the CUs), first block in -> last block out on the device clock -- the quantity a
        # rocprofv3 kernel trace of the same command averages; the event-bracketed figures stay beside it
        stamped = stamped_nnconv_us(net, x, adj, adj_attr, col, args.steps)
        infwd = in_forward_classes(net, x, adj, adj_attr, col, args.steps)
        roofline["single_stream"] = {k: roofline[k] for k in ("avg_launch_us", "achieved", "frac", "timing")}
        # the bound this formulation REALLY has: neither kernel moves HBM bytes at its limit -- both gather
        # 128-byte rows out of L2 through the CU's vector-memory path, whose ceiling is measured here, per access shape
        try:
            ceil = gather_ceiling(dev)
            t_ss = roofline["single_stream"]["avg_launch_us"] * 1e-6
            nn_rows_bytes = (ea_total + n_total) * 128           # every in-edge's source row + the tile's own rows (root term)
            roofline["gather_bound"] = {
                "what": "rows gathered per launch x 128 B / the chip-wide row-gather rate of the kernel's own access shape, measured on "
                        "this device by tgnn_ubench_row_gather (cache-resident 12.8 MB table, band-local rows, 16 waves per CU)",
                "bytes": nn_rows_bytes, "peak_GBs": ceil["nnconv_lane_map_16rows_x_64B"],
                "floor_us": nn_rows_bytes / (ceil["nnconv_lane_map_16rows_x_64B"] * 1e9) * 1e6,
                "frac_single_stream": nn_rows_bytes / t_ss / 1e9 / ceil["nnconv_lane_map_16rows_x_64B"],
                "whole_row_peak_GBs": ceil["whole_rows_8lanes_x_16B"],
                "note": "the floor assumes every gather instruction full; the edge groups of the benchmark layout are 68 % full (15.6 groups "
                        "of 16 slots per 16-row tile; the type columns of rounds 1-4 were 29 % full: profiles/r05_nnconv_study.txt, "
                        "r05_nnconv_eg.txt)"}
            gk = roofline["gin_kernel"]
            gin_rows_bytes = (ec_total + n_total) * 128
            gk["gather_bound"] = {"bytes": gin_rows_bytes, "peak_GBs": ceil["whole_rows_8lanes_x_16B"],
                                  "floor_us": gin_rows_bytes / (ceil["whole_rows_8lanes_x_16B"] * 1e9) * 1e6,
                                  "what": "gin32_aggregate_kernel's neighbourhood rows (whole-row gathers) against the same measured ceiling"}
        except Exception as exc:                                  # (an older library without the helper)
            roofline["gather_bound"] = {"error": str(exc)}
        if stamped:
            t_in = stamped[0] * 1e-6
            roofline.update({"avg_launch_us": stamped[0], "achieved": roofline["algorithmic_bytes_per_launch"] / t_in / 1e9,
                             "frac": roofline["algorithmic_bytes_per_launch"] / t_in / 1e9 / HBM_PEAK_GBS,
                             "timing": f"device wall clock stamped by the kernel's first and last block INSIDE the production "
                                       f"two-stream forward of this run (tgnn_forward_stamped, {stamped[1]} launches): the "
                                       f"duration a kernel trace reports, no event or profiler in the schedule"})
        # the matrix pipe's share, on the SAME time base as `frac` (the in-forward duration) and against the pipe the kernel runs on:
        # v_mfma_f32_16x16x32_f16, three fp16-pair terms per algorithmic product (the old fields divided by the
        # single-stream time and the fp32 vector peak)
        t_mm = roofline["avg_launch_us"] * 1e-6
        roofline["matrix_pipe"] = {"algorithmic_tflops": roofline["flops_per_launch"] / t_mm / 1e12,
                                   "issued_tflops_fp16_x3_terms": 3 * roofline["flops_per_launch"] / t_mm / 1e12,
                                   "frac_of_fp16_dense_peak": 3 * roofline["flops_per_launch"] / t_mm / 1e12 / F16_MFMA_PEAK_TFLOPS,
                                   "peak_tflops": F16_MFMA_PEAK_TFLOPS,
                                   "what": "2 C^2 (Ea + N) flops per launch x 3 fp16-pair terms / the in-forward launch duration / the dense fp16 "
                                           "MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s)"}
        if stamped and "peak_GBs" in roofline.get("gather_bound", {}):
            gb = roofline["gather_bound"]
            gb["frac"] = gb["bytes"] / (stamped[0] * 1e-6) / 1e9 / gb["peak_GBs"]   # in the production forward, like roofline.frac
        if infwd and infwd["nnconv"]["launches_per_forward"]:
            t_ev = infwd["nnconv"]["ms_per_forward"] / infwd["nnconv"]["launches_per_forward"] * 1e-3
            roofline["in_forward_events"] = {"avg_launch_us": t_ev * 1e6, "frac": roofline["algorithmic_bytes_per_launch"] / t_ev / 1e9 / HBM_PEAK_GBS,
                                             "timing": "HIP events around the launch on its stream inside the two-stream forward "
                                                       "(tgnn_forward_profiled_two_stream): includes the wait for CUs the collision "
                                                       "chain holds"}
            t_m = infwd["merge"]["ms_per_forward"] / max(1, infwd["merge"]["launches_per_forward"]) * 1e-3
            roofline["merge_kernel"]["in_forward"] = {"avg_launch_us": t_m * 1e6, "frac": merge_bytes(n_total) / t_m / 1e9 / HBM_PEAK_GBS}
            if infwd.get("gin") and infwd["gin"]["launches_per_forward"]:
                t_g = infwd["gin"]["ms_per_forward"] / infwd["gin"]["launches_per_forward"] * 1e-3
                roofline["gin_kernel"]["in_forward_events"] = {
                    "avg_launch_us": t_g * 1e6, "frac": roofline["gin_kernel"]["algorithmic_bytes_per_launch"] / t_g / 1e9 / HBM_PEAK_GBS,
                    "timing": "HIP events around the pair on the side stream inside the two-stream forward of this run (includes the "
                              "wait for CUs the NNConv holds)"}
        # quoted, not measured here: rocprofv3 of the same command (cannot run inside this process) and the PMC passes;
        # only for the workload they were taken on, with the file they come from
        prof_file = os.path.join(REPO, "profiles", "r06_nnconv.json")
        if not os.path.exists(prof_file):
            prof_file = os.path.join(REPO, "profiles", "r05_nnconv.json")
        if os.path.exists(prof_file) and (n_total, ea_total, n_types_seen) == (100_000, 1_000_000, 13):
            with open(prof_file) as fh:
                q = json.load(fh)
            roofline["traffic"] = q.get("hbm_bytes_per_launch_corrected")
            roofline["traffic_source"] = q.get("traffic_source")
            if q.get("rocprof_avg_us_in_forward"):
                if q.get("gin_rocprof_avg_us_in_forward") and "gin_kernel" in roofline:
                    # the GIN pair INSIDE the two-stream forward (the single-stream event timing above flatters it: the aggregate
                    # nearly doubles when it shares CUs with the NNConv)
                    tg = q["gin_rocprof_avg_us_in_forward"] * 1e-6
                    gk = roofline["gin_kernel"]
                    gk["single_stream"] = {"avg_launch_us": gk["avg_launch_us"], "achieved": gk["achieved"], "frac": gk["frac"]}
                    gk.update({"avg_launch_us": q["gin_rocprof_avg_us_in_forward"], "achieved": gk["algorithmic_bytes_per_launch"] / tg / 1e9,
                               "frac": gk["algorithmic_bytes_per_launch"] / tg / 1e9 / HBM_PEAK_GBS,
                               "timing": "in the production two-stream forward, quoted: " + q.get("gin_rocprof_source", "")})
                roofline["rocprof"] = {"avg_launch_us_in_forward": q["rocprof_avg_us_in_forward"],
                                       "frac": roofline["algorithmic_bytes_per_launch"] / (q["rocprof_avg_us_in_forward"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       "source": q.get("rocprof_source")}
        roofline["whole_forward"] = {"algorithmic_bytes": forward_bytes(n_total, ea_total, ec_total, n_types_seen),
                                     "frac_of_hbm_peak": forward_bytes(n_total, ea_total, ec_total, n_types_seen)
                                     / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if os.path.exists(prof_file) and (n_total, ea_total, n_types_seen) == (100_000, 1_000_000, 13):
            with open(prof_file) as fh:
                q = json.load(fh)
            if q.get("whole_forward_hbm_bytes_pmc"):      # quoted: HBM bytes of one cached-layout forward by PMC counters
                roofline["whole_forward"]["traffic"] = q["whole_forward_hbm_bytes_pmc"]
                roofline["whole_forward"]["traffic_source"] = q.get("whole_forward_hbm_source")

    # ---- larger single-GPU layouts (BASELINE configs 4 / 5 are 500k / 2M nodes over 4 / 8 GPUs; here on ONE GPU, drawn
    #      on the device): step time with graph preparation, kernel classes, fractions of the HBM bound per kernel
    extras = None
    if not sharded and not args.no_extra_sizes:
        from tilingnn_amd.synth import make_super_graph_on_device
        extras = []
        for n_big, seed in ((500_000, 3), (2_000_000, 4)):
            xb, adjb, attrb, colb, _ = make_super_graph_on_device(n_big, 10 * n_big, 10 * n_big // 4 * 5, dev,
                                                                  tile_count=TILE_COUNT, n_edge_types=N_TYPES, seed=seed)
            torch.cuda.synchronize()
            for _ in range(2):
                net(x=xb, adj_e_index=adjb, adj_e_features=attrb, col_e_idx=colb)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                tb = time.perf_counter()
                net(x=xb, adj_e_index=adjb, adj_e_features=attrb, col_e_idx=colb)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - tb) * 1e3)
            med = sorted(ts)[2]
            cls, nt = profiled_classes(net, xb, adjb, attrb, colb, 3)
            rf = kernel_roofline(cls, n_big, int(adjb.shape[1]), int(colb.shape[1]), nt,
                                 edge_groups=bool(_ops.GROUPS and _ops.runs_general_schedule(n_big)))
            extras.append({"n_nodes": n_big, "n_adj_edges": int(adjb.shape[1]), "n_col_edges": int(colb.shape[1]),
                           "ms_per_step": med, "value": n_big / (med * 1e-3), "data": "synthetic, drawn on the device",
                           "nnconv": {k: rf[k] for k in ("avg_launch_us", "achieved", "frac")},
                           "gin_aggregate_plus_mlp": rf["gin_kernel"], "merge": rf["merge_kernel"],
                           "whole_forward_frac_of_hbm_peak": forward_bytes(n_big, int(adjb.shape[1]), int(colb.shape[1]), nt)
                           / (med * 1e-3) / 1e9 / HBM_PEAK_GBS})
            del xb, adjb, attrb, colb
            torch.cuda.empty_cache()

    # ---- BASELINE config 3: 30-60-90 + equilateral (tile_count 4), the same 100k / 1M / 1.25M graph shape, network_width 64,
    #      bf16 activation storage (csrc/bf16_path.hip); fp32 accumulate, fp64 BatchNorm sums.  Step = forward incl. graph prep.
    config3 = None
    if not sharded and not args.no_extra_sizes:
        sg3 = make_super_graph(args.nodes_per_gpu, int(ADJ_PER_GPU * scale), int(COL_PER_GPU * scale), tile_count=4,
                               n_edge_types=N_TYPES, seed=2)
        x3, adj3, attr3, col3, _ = sg3.to_torch(dev)
        net3 = TilinGNN(adj_edge_features_dim=fe, network_depth=DEPTH, network_width=64, node_features_dim=5)
        net3.load_state_dict(make_state_dict(fe, DEPTH, 64, 1, 5, seed=0), strict=True)
        net3 = net3.to(dev).train()
        net3.activation_dtype = torch.bfloat16
        res3 = {}
        for cached in (False, True):
            net3.cache_graph = cached
            for _ in range(3):
                net3(x=x3, adj_e_index=adj3, adj_e_features=attr3, col_e_idx=col3)
            torch.cuda.synchronize()
            ts = []
            for _ in range(max(5, args.steps // 2)):
                tb = time.perf_counter()
                p3 = net3(x=x3, adj_e_index=adj3, adj_e_features=attr3, col_e_idx=col3)[0]
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - tb) * 1e3)
            res3["cached_layout_ms" if cached else "ms_per_step"] = sorted(ts)[len(ts) // 2]
        assert bool(torch.isfinite(p3).all())
        n3, ea3, ec3 = int(x3.shape[0]), int(adj3.shape[1]), int(col3.shape[1])
        b3 = forward_bytes(n3, ea3, ec3, N_TYPES, fe=fe, fx=5, c=64, d=DEPTH, s=2)
        config3 = {"workload": f"TilinGNN.forward, {n3} nodes / {ea3} + {ec3} edges, tile_count 4 (Fx 5), T=13, width 64, depth 20, "
                               "bf16 storage of the skip buffer and branch outputs, fp32 accumulate, graph prep included",
                   "dtype": "bf16 storage / f32 accumulate", "ms_per_step": res3["ms_per_step"],
                   "value": n3 / (res3["ms_per_step"] * 1e-3), "cached_layout_ms": res3["cached_layout_ms"],
                   "whole_forward": {"algorithmic_bytes": b3, "frac_of_hbm_peak": b3 / (res3["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                   "nnconv_algorithmic_bytes_per_launch": nnconv_bytes(n3, ea3, N_TYPES, c=64, s=2),
                   "profile": "profiles/r06_config3_kernel_stats.txt (rocprofv3 --kernel-trace --stats of scratch/run_config3_only.py), "
                              "profiles/r06_config3_pmc.txt (FETCH / WRITE, SQ_INSTS_VALU / MFMA, MFMA-busy, wait cycles of nnconv64_bf16_eg_kernel -- the "
                              "kernel that runs -- and the other kernels of a layer; scratch/pmc_config3.sh)"}
        del net3, x3, adj3, attr3, col3
        torch.cuda.empty_cache()

    # ---- BASELINE config 0: the reference's own example, the real labyrinth layout (1 254 nodes, 8 502 + 10 472 edges, T = 13):
    #      what a greedy round scores.  Small layouts run the forward as one persistent kernel (csrc/forward_small.hip); the
    #      general launch schedule is timed beside it (tgnn_set_small_layout_limit(0)).
    real_layout = None
    if not sharded and not args.no_extra_sizes:
        try:
            from tests.golden_util import graph_tensors, load_labyrinth_graph
            from tilingnn_amd import _lib
            gl = load_labyrinth_graph()
            xl, adjl, attrl, coll, _ = graph_tensors(gl, torch.float32, dev)
            netl = TilinGNN(adj_edge_features_dim=int(attrl.shape[1]), network_depth=DEPTH, network_width=WIDTH,
                            node_features_dim=int(xl.shape[1]))
            netl.load_state_dict(make_state_dict(int(attrl.shape[1]), DEPTH, WIDTH, 1, int(xl.shape[1]), seed=0), strict=True)
            netl = netl.to(dev).train()
            resl = {}
            limit0 = _lib.lib.tgnn_get_small_layout_limit()
            for name, limit, cached in (("persistent_kernel_ms", limit0, False), ("persistent_kernel_cached_layout_ms", limit0, True),
                                        ("general_schedule_ms", 0, False), ("general_schedule_cached_layout_ms", 0, True)):
                _lib.lib.tgnn_set_small_layout_limit(limit)
                netl.cache_graph = cached
                for _ in range(5):
                    netl(x=xl, adj_e_index=adjl, adj_e_features=attrl, col_e_idx=coll)
                torch.cuda.synchronize()
                ts = []
                for _ in range(30):
                    tb = time.perf_counter()
                    netl(x=xl, adj_e_index=adjl, adj_e_features=attrl, col_e_idx=coll)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - tb) * 1e3)
                resl[name] = sorted(ts)[len(ts) // 2]
            _lib.lib.tgnn_set_small_layout_limit(limit0)
            real_layout = {"workload": "TilinGNN.forward on data/labyrinth's complete graph (1254 nodes, 8502 + 10472 edges, T=13), "
                                       "width 32, depth 20, fp32; median of 30, host-synchronised per forward",
                           **resl, "value": 1254 / (resl["persistent_kernel_ms"] * 1e-3), "unit": "tile-nodes/s",
                           "design": "DESIGN.md section 12"}
            del netl
        except FileNotFoundError:
            real_layout = None

    # ---- BASELINE config 2: 30-60-90, 10 000 nodes / 80 000 + 100 000 edges, fp32, 1 GPU (the CPU leg below runs
    #      the very same layout).  Mid-size layouts run their 20 layers as ONE persistent kernel (csrc/forward_mid.hip); the
    #      general launch schedule is timed beside it (tgnn_set_mid_layout_limit(0)).  Also 20 000 and 50 000 nodes of the
    #      headline generator (what a greedy round of a large solve, or a rank of a strong-scaled 100k layout, scores).
    config2 = None
    if not sharded and not args.no_extra_sizes:
        from tilingnn_amd import _lib
        from tilingnn_amd.graph_networks import _graph_cache
        mid0 = _lib.lib.tgnn_get_mid_layout_limit()

        def time_layout(n2, ea2, ec2, seed2, reps=40):
            sg2 = make_super_graph(n2, ea2, ec2, tile_count=TILE_COUNT, n_edge_types=N_TYPES, seed=seed2)
            x2, adj2, attr2, col2, _ = sg2.to_torch(dev)
            net2 = TilinGNN(adj_edge_features_dim=fe, network_depth=DEPTH, network_width=WIDTH, node_features_dim=fx)
            net2.load_state_dict(make_state_dict(fe, DEPTH, WIDTH, 1, fx, seed=0), strict=True)
            net2 = net2.to(dev).train()
            res = {}
            for name, limit, cached in (("persistent_layer_loop_ms", mid0, False), ("persistent_layer_loop_cached_layout_ms", mid0, True),
                                        ("general_schedule_ms", 0, False), ("general_schedule_cached_layout_ms", 0, True)):
                _lib.lib.tgnn_set_mid_layout_limit(limit)
                _graph_cache.clear()
                net2.cache_graph = cached
                for _ in range(5):
                    net2(x=x2, adj_e_index=adj2, adj_e_features=attr2, col_e_idx=col2)
                torch.cuda.synchronize()
                evs2 = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
                evs2[0].record()
                for k in range(reps):                        # back to back, one event behind every forward (as the headline)
                    p2 = net2(x=x2, adj_e_index=adj2, adj_e_features=attr2, col_e_idx=col2)[0]
                    evs2[k + 1].record()
                torch.cuda.synchronize()
                ts = sorted(evs2[k].elapsed_time(evs2[k + 1]) for k in range(reps))
                res[name] = 0.5 * (ts[reps // 2 - 1] + ts[reps // 2])
                assert bool(torch.isfinite(p2).all())
            _lib.lib.tgnn_set_mid_layout_limit(mid0)
            _graph_cache.clear()
            return res
        r2 = time_layout(10_000, 80_000, 100_000, 1)
        config2 = {"workload": "TilinGNN.forward, BASELINE config 2: 10000 nodes / 80000 adjacency + 100000 collision edges, 30-60-90 "
                               "(tile_count 2), T=13, width 32, depth 20, fp32, train-mode BatchNorm; median of 40 back-to-back forwards",
                   "ms_per_step": r2["persistent_layer_loop_ms"], "value": 10_000 / (r2["persistent_layer_loop_ms"] * 1e-3),
                   "unit": "tile-nodes/s", "graph_prep": "included in ms_per_step / value; *_cached_layout_ms: prepared once",
                   **r2, "design": "DESIGN.md section 14 (csrc/forward_mid.hip)",
                   "other_mid_sizes": [dict(n_nodes=nm, **time_layout(nm, 10 * nm, 12 * nm + nm // 2, 1, reps=20)) for nm in (20_000, 50_000)]}

    # ---- the loss ML_Solver.predict evaluates on the probabilities: two launches, HBM bound
    loss_info = None
    if not sharded:
        from tilingnn_amd.solver.ml_solver.losses import Losses
        p1 = torch.rand(n_total, 1, device=dev)
        for _ in range(3):
            Losses.unsupervised_losses(p1, x, col, adj, adj_attr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            Losses.unsupervised_losses(p1, x, col, adj, adj_attr)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        # algorithmic bytes: both index rows of both edge sets (int64), one length per adjacency edge, the
        # probability and area columns once (the per-edge probability gathers are served by L2)
        b_loss = (ec_total + ea_total) * 16 + ea_total * 4 + n_total * 8
        loss_info = {"us_per_call": us, "algorithmic_bytes": b_loss, "achieved_GBs": b_loss / (us * 1e-6) / 1e9,
                     "frac_of_hbm_peak": b_loss / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}

    # ---- one round of the greedy loop's layout re-indexing: stream compaction on the device,
    #      half of the nodes still unlabelled; the CPU figure is the numpy restatement of compute_sub_layout (the
    #      reference itself runs Python comprehensions with dict look-ups over all edges)
    sub_info = None
    if not sharded:
        from tilingnn_amd.util.algorithms import DeviceLayout, SubLayoutBuilder
        rng = np.random.default_rng(0)
        alive_h = (rng.uniform(size=n_total) < 0.5).astype(np.int32)
        alive_d = torch.from_numpy(alive_h).to(dev)
        builder = SubLayoutBuilder(DeviceLayout(x, adj, adj_attr, col))
        for _ in range(3):
            builder.build(alive_d)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(10):
            sub = builder.build(alive_d)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t2) / 10 * 1e6
        fe_ = int(adj_attr.shape[1])
        kept_a, kept_c, kept_n = int(sub.align_edge_index.shape[1]), int(sub.collide_edge_index.shape[1]), int(sub.node_feature.shape[0])
        b_sub = (ea_total + ec_total) * 16 + ea_total * 4 * fe_ * (kept_a / max(ea_total, 1)) + n_total * 4 \
            + kept_a * (16 + 4 * fe_) + kept_c * 16 + kept_n * (8 + 4 * int(x.shape[1]))
        sub_info = {"us_per_round": us, "alive_fraction": 0.5, "kept": [kept_n, kept_a, kept_c],
                    "algorithmic_bytes": int(b_sub), "achieved_GBs": b_sub / (us * 1e-6) / 1e9}
        if not args.no_cpu_baseline:
            from oracle import greedy_oracle as go
            xh, ah, aah, ch = x.cpu().numpy(), adj.cpu().numpy(), adj_attr.cpu().numpy(), col.cpu().numpy()
            cah = np.zeros((ch.shape[1], 1), dtype=np.float32)
            t3 = time.perf_counter()
            go.compute_sub_layout(xh, ah, aah, ch, cah, np.flatnonzero(alive_h))
            sub_info["cpu_numpy_us"] = (time.perf_counter() - t3) * 1e6

    # ---- a whole greedy solve of the headline layout with the acceptance step on the device (csrc/greedy.hip: the documented
    #      substitute for the reference's sequential host sweep): forwards of the shrinking sub-layouts, compaction and
    #      acceptance per round, nothing but three counts per round on the host
    if sub_info is not None and not args.no_extra_sizes:
        from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
        from tilingnn_amd.util import algorithms as alg
        cg = net.cache_graph
        net.cache_graph = False
        solver = ML_Solver(None, dev, None, net, num_prob_maps=1)
        lay = alg.DeviceLayout(x, adj, adj_attr, col)
        alg.solve_by_device_greedy(solver, lay, seed=1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        sel, _, _ = alg.solve_by_device_greedy(solver, lay, seed=1)
        torch.cuda.synchronize()
        sub_info["device_greedy_solve"] = {"ms": (time.perf_counter() - t2) * 1e3, "rounds": int(alg.solve_by_device_greedy.last_rounds),
                                           "tiles_selected": int(sel.sum()), "n_nodes": int(n_total),
                                           "what": "solve_by_device_greedy: every round a forward of the remaining sub-layout; the "
                                                   "reference's host sweep takes 244 rounds / 0.62 s at this size "
                                                   "(profiles/r04_greedy_solve.txt)"}
        net.cache_graph = cg

    # ---- one training step at the same shape: forward keeping activations, loss, backward through the
    #      adjoint kernels, Adam (torch's, the caller's optimizer in the reference: network_train.py)
    train_info = None
    if not sharded and not args.no_train_step:
        from tilingnn_amd.solver.ml_solver.losses import Losses
        net.cache_graph = True              # the transposed CSR is built once per layout, like the forward structures
        net.autograd = True
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)

        def train_step():
            probs, _ = net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
            opt.zero_grad()
            loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, adj_attr)
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(5):
            train_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t4) / 5 * 1e3
        for _ in range(2):                  # the allocator sees this pattern (no backward frees the kept buffers) for the first time
            probs, _ = net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
            del probs
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(5):
            probs, _ = net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
            del probs
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t5) / 5 * 1e3
        train_info = {"ms_per_step": ms, "forward_keeping_activations_ms": fwd_ms, "nodes_per_s": n_total / (ms * 1e-3),
                      "optimizer": "torch.optim.Adam", "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
        net.autograd = False
        net.cache_graph = False

    if saved_stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        os.close(saved_stdout_fd)
    if rank == 0:
        line = {
            "metric": "scored tile-nodes/sec (GNN forward)", "value": value, "unit": "nodes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "timing": {"value_from": "median step time (event spacing over the K timed steps, max over ranks)",
                       "ms_per_step_median": median_ms, "ms_per_step_mean": mean_ms, "value_mean": n_total / (mean_ms * 1e-3),
                       "ms_per_step_min": per_step_ms[0], "ms_per_step_max": per_step_ms[-1]},
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"TilinGNN.forward on a seeded banded super-graph: {unit[0]} nodes / "
                                   f"{unit[1]} adjacency edges / {unit[2]} collision edges "
                                   f"{'per GPU' if args.scaling == 'weak' else 'IN ALL (strong scaling: split over the ranks)'}, "
                                   f"tile_count {tile_count}, T=13 edge types, width 32, depth 20, "
                                   f"train-mode BatchNorm, graph prep included"
                                   + ("" if args.config == "headline" else f" (BASELINE config {args.config}, SURVEY 8d)"),
                       "n_nodes": n_total, "n_adj_edges": ea_total, "n_col_edges": ec_total,
                       "parallelism": "single GPU" if not sharded else f"node-range shards x{world}, one all-to-all per layer (halo rows + BN sums) over RCCL"},
            "roofline": roofline,
        }
        if cached_ms is not None:
            line["cached_layout"] = {"ms_per_step": cached_ms, "value": n_total / (cached_ms * 1e-3)}
        if loss_info is not None:
            line["predict_loss"] = loss_info
        if sub_info is not None:
            line["greedy_sublayout"] = sub_info
        if train_info is not None:
            line["train_step"] = train_info
        if class_ms is not None:
            line["kernel_classes"] = class_ms
        if extras is not None:
            line["larger_layouts_single_gpu"] = extras
        if real_layout is not None:
            line["config0_real_layout"] = real_layout
        if config2 is not None:
            line["config2_10k_nodes"] = config2
        if config3 is not None:
            line["config3_width64_bf16"] = config3
        if sharded:
            import torch.distributed as dist
            line["collectives"] = {"backend": dist.get_backend(), "ranks_seen": ranks_seen,
                                   "per_forward": shard_runner.collectives_per_forward}
        if not args.no_cpu_baseline and world == 1:           # the CPU leg is a 1-GPU (rank 0, N = 1) measurement
            line["cpu_baseline"] = cpu_baseline()
            # GPU and CPU at the SAME configuration (BASELINE config 2, graph preparation included on the GPU side); the headline's
            # nodes/s (100 000-node layout) over the CPU's nodes/s at 10 000 nodes is quoted beside it under its own name
            if config2 is not None:
                line["speedup_vs_cpu_baseline"] = config2["value"] / line["cpu_baseline"]["value"]
                line["speedup_vs_cpu_baseline_what"] = "config 2 on the GPU (prep included) / config 2 on the CPU, nodes/s"
            line["headline_nodes_per_s_over_cpu_config2_nodes_per_s"] = value / line["cpu_baseline"]["value"]
            line["parity_probe"] = parity_probe(dev)
        print(json.dumps(line), flush=True)
    if sharded:
        os.dup2(2, 1)                      # teardown chatter, if any, goes to stderr
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
