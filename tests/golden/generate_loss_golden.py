"""Golden vectors for the loss on the predict path, produced by the REFERENCE itself.

Run in the build container only (needs /root/reference):   python tests/golden/generate_loss_golden.py

`solver/ml_solver/losses.py` is imported UNCHANGED (namespace packages: the reference has no __init__.py there) with
the stub `inputs.config` of generate_golden.py extended by the three weights of `inputs/config.py:49-51`.
Stored: ref_losses.npz -- per case the probability maps fed in, the loss vector, the arg-min and the scalar the
reference returns, in fp64 and fp32, for
  * the real labyrinth graph with the reference's own fp64 probabilities (1 map) and 3 seeded random maps,
  * the tiny 6-node graph (self loops, a zero-in-degree node),
  * corner cases: no collision edges / no adjacency edges ([2, 0] index tensors -> that term is 0.0).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import generate_golden as gg                                   # noqa: E402  (helpers only; its main() is not run)


def main():
    torch.set_num_threads(1)
    gg.import_reference(2)
    cfg = sys.modules["inputs.config"]
    cfg.COLLISION_WEIGHT = 1 / math.log(1 + 1e-1)              # inputs/config.py:49
    cfg.ALIGN_LENGTH_WEIGHT = 0.02                             # inputs/config.py:50
    cfg.AVG_AREA_WEIGHT = 1                                    # inputs/config.py:51
    from solver.ml_solver.losses import Losses                 # the reference's own code

    out = {}

    def case(name, probs, x, col, adj, adj_attr):
        for tag, dt in (("fp64", torch.float64), ("fp32", torch.float32)):
            loss, min_index, losses = Losses.calculate_unsupervised_loss(
                torch.from_numpy(probs).to(dt), torch.from_numpy(x).to(dt), torch.from_numpy(col).long(),
                torch.from_numpy(adj).long(), torch.from_numpy(adj_attr).to(dt))
            out[f"{name}.losses_{tag}"] = np.asarray(losses, dtype=np.float64)
            out[f"{name}.min_index_{tag}"] = np.int64(min_index)
            out[f"{name}.loss_{tag}"] = np.float64(loss.item())
        out[f"{name}.probs"] = probs

    g = gg.load_labyrinth()
    n = g["x"].shape[0]
    fwd = np.load(os.path.join(HERE, "ref_forward_labyrinth.npz"))
    case("laby_ref_probs", fwd["probs_fp64"].reshape(n, 1).astype(np.float64), g["x"], g["col"], g["adj"], g["adj_attr"])
    rng = np.random.default_rng(21)
    case("laby_3maps", rng.uniform(0.02, 0.98, size=(n, 3)), g["x"], g["col"], g["adj"], g["adj_attr"])
    case("laby_extreme", np.stack([np.full(n, 1e-9), np.full(n, 1.0 - 1e-9), rng.uniform(0, 1, n)], axis=1),
         g["x"], g["col"], g["adj"], g["adj_attr"])

    tiny = np.load(os.path.join(HERE, "tiny_graph.npz"))
    case("tiny_2maps", rng.uniform(0.1, 0.9, size=(6, 2)), tiny["x"], tiny["col"], tiny["adj"], tiny["adj_attr"])
    empty = np.zeros((2, 0), dtype=np.int64)
    case("tiny_no_col", rng.uniform(0.1, 0.9, size=(6, 2)), tiny["x"], empty, tiny["adj"], tiny["adj_attr"])
    case("tiny_no_adj", rng.uniform(0.1, 0.9, size=(6, 2)), tiny["x"], tiny["col"], empty,
         np.zeros((0, tiny["adj_attr"].shape[1])))
    np.savez_compressed(os.path.join(HERE, "ref_losses.npz"), **out)
    for k in sorted(out):
        if "losses_fp64" in k:
            print(k, out[k])
    print(f"ref_losses.npz: {os.path.getsize(os.path.join(HERE, 'ref_losses.npz')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
