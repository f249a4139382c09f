"""`ML_Solver` -- the boundary caller of the scoring path, mirroring the part of
/root/reference/solver/ml_solver/ml_solver.py that sits on it:

    predict(brick_layout)            ml_solver.py:29-49   (empty-edge early-out, forward, best-map pick)
    get_unsupervised_losses_from_layout(layout, probs)    ml_solver.py:51-62 (losses.py on the GPU, csrc/loss.hip)
    get_predict_probs(brick_layout)  ml_solver.py:69-81
    load_saved_network(path)         ml_solver.py:129-131 (load_state_dict + network.train())

    solve(brick_layout)              ml_solver.py:64-73   (greedy assembly loop: tilingnn_amd/util/algorithms.py keeps the
                                                           layout on the GPU; `greedy_solver=` swaps in the reference's)
The debug dumps stay with the reference.

`brick_layout` is duck-typed exactly as the reference uses it: `.node_feature`,
`.align_edge_index`, `.collide_edge_index` (numpy) and `.get_data_as_torch_tensor(device)`.
"""
from copy import deepcopy

import numpy as np
import torch

from ...graph_networks.network_utils import get_network_prediction


class LayoutArrays:
    """Minimal stand-in for tiling.brick_layout.BrickLayout's data side (brick_layout.py:242-246):
    the five numpy arrays and the float32/int64 conversion of util/data_util.py:110-117."""

    def __init__(self, node_feature, align_edge_index, align_edge_features, collide_edge_index,
                 collide_edge_features):
        self.node_feature = node_feature
        self.align_edge_index = align_edge_index
        self.align_edge_features = align_edge_features
        self.collide_edge_index = collide_edge_index
        self.collide_edge_features = collide_edge_features

    def get_data_as_torch_tensor(self, device):
        return (torch.from_numpy(self.node_feature).float().to(device),
                torch.from_numpy(self.align_edge_index).long().to(device),
                torch.from_numpy(self.align_edge_features).float().to(device),
                torch.from_numpy(self.collide_edge_index).long().to(device),
                torch.from_numpy(self.collide_edge_features).float().to(device))


class ML_Solver:
    def __init__(self, debugger, device, complete_graph, network, num_prob_maps, greedy_solver=None, score_fn=None):
        self.debugger = debugger
        self.device = device
        self.complete_graph = complete_graph
        self.network = network
        self.random_network = deepcopy(self.network)            # ml_solver.py:26
        self.num_prob_maps = num_prob_maps
        self._greedy_solver = greedy_solver
        self._score_fn = score_fn                               # (selection, layout) -> float; default: Losses.solution_score
        # None (default): solve() runs the reference's sweep with numpy's RNG stream (seeded parity) at every size; a node count:
        # layouts at least that large take tilingnn_amd.util.algorithms.solve_by_device_greedy (batched acceptance on the GPU)
        self.device_greedy_min_nodes = None
        self.device_greedy_seed = 0

    @staticmethod
    def _no_edges(index) -> bool:
        """`len(index) == 0` of the reference (an empty edge set is `np.array([]).T` there); a device-resident
        sub-layout holds an empty [2, 0] tensor instead, and so may a caller's numpy layout."""
        return index.numel() == 0 if torch.is_tensor(index) else np.asarray(index).size == 0

    def predict(self, brick_layout):
        if self._no_edges(brick_layout.collide_edge_index) or self._no_edges(brick_layout.align_edge_index):
            # only one edge set left: select every remaining tile (ml_solver.py:31-32)
            predictions = torch.ones((brick_layout.node_feature.shape[0], self.num_prob_maps)).float().to(self.device)
        else:
            x, adj_edge_index, adj_edge_features, collide_edge_index, collide_edge_features = \
                brick_layout.get_data_as_torch_tensor(self.device)
            # the reference's call (ml_solver.py:39-43); tilingnn_amd.TilinGNN also checks the health word of its persistent
            # kernels here (forward_checked: the probabilities travel to the host below anyway)
            run = getattr(self.network, "forward_checked", None) or self.network
            predictions, *_ = run(x=x, adj_e_index=adj_edge_index, adj_e_features=adj_edge_features,
                                  col_e_idx=collide_edge_index, col_e_features=collide_edge_features)
        best_map_index = self._best_prob_map(predictions, brick_layout)
        return predictions[:, best_map_index].detach().cpu().numpy()

    def predict_on_device(self, brick_layout):
        """predict() for a layout that lives on the GPU, the probabilities staying there: [N] float32 on self.device (the loop of
        tilingnn_amd.util.algorithms.solve_by_device_greedy feeds them straight into the acceptance kernels)."""
        if self._no_edges(brick_layout.collide_edge_index) or self._no_edges(brick_layout.align_edge_index):
            return torch.ones(brick_layout.node_feature.shape[0], dtype=torch.float32, device=self.device)     # ml_solver.py:31-32
        x, adj_edge_index, adj_edge_features, collide_edge_index, collide_edge_features = \
            brick_layout.get_data_as_torch_tensor(self.device)
        run = getattr(self.network, "forward_checked", None) or self.network
        predictions, *_ = run(x=x, adj_e_index=adj_edge_index, adj_e_features=adj_edge_features,
                              col_e_idx=collide_edge_index, col_e_features=collide_edge_features)
        return predictions[:, self._best_prob_map(predictions, brick_layout)].detach().contiguous()

    def _best_prob_map(self, predictions, brick_layout):
        """get_best_prob_map (ml_solver.py:133-136): argsort of the per-map unsupervised loss.
        With one probability map -- the only configuration the reference ever constructs
        (Tiling-Shape.py:37, Tiling-GUI.py:570) -- the answer is 0 and the loss (a device round trip) is skipped."""
        if predictions.shape[1] == 1:
            return 0
        losses = self.get_unsupervised_losses_from_layout(brick_layout, predictions)
        return int(np.argsort(losses)[0])

    def get_unsupervised_losses_from_layout(self, brick_layout, probs):
        """ml_solver.py:51-62: the per-map losses of `probs` on a layout, as numpy."""
        from .losses import Losses
        x, adj_edge_index, adj_edge_features, collide_edge_index, collide_edge_features = \
            brick_layout.get_data_as_torch_tensor(self.device)
        _, _, losses = Losses.calculate_unsupervised_loss(probs, x, collide_edge_index,
                                                          adj_edges_index=adj_edge_index,
                                                          adj_edge_features=adj_edge_features)
        return losses

    def get_predict_probs(self, brick_layout):
        x, adj_edge_index, adj_edge_features, collide_edge_index, collide_edge_features = \
            brick_layout.get_data_as_torch_tensor(self.device)
        return get_network_prediction(network=self.network, x=x, adj_e_index=adj_edge_index,
                                      adj_e_features=adj_edge_features, col_e_idx=collide_edge_index,
                                      col_e_features=collide_edge_features)

    def solve(self, brick_layout):
        """ml_solver.py:64-73.  Default loop: tilingnn_amd.util.algorithms.solve_by_probablistic_greedy (layout resident
        on the GPU; score = Losses.solution_score when the layout carries its complete graph and super-contour area, or
        `score_fn`, else None); `greedy_solver=` swaps in another one, e.g. the reference's own."""
        if self._greedy_solver is None:
            from ...util.algorithms import solve_by_device_greedy, solve_by_probablistic_greedy
            n_nodes = int(brick_layout.node_feature.shape[0])
            if self.device_greedy_min_nodes is not None and n_nodes >= self.device_greedy_min_nodes:
                # large layouts: the acceptance batched on the device (a documented substitute of the sequential sweep)
                output_solution, score, predict_order = solve_by_device_greedy(self, brick_layout, seed=self.device_greedy_seed,
                                                                               score_fn=self._score_fn)
            else:
                output_solution, score, predict_order = solve_by_probablistic_greedy(self, brick_layout, score_fn=self._score_fn)
        else:
            output_solution, score, predict_order = self._greedy_solver(self, brick_layout)
        output_layout = deepcopy(brick_layout)
        output_layout.predict_order = predict_order
        output_layout.predict = output_solution
        output_layout.predict_probs = self.predict(brick_layout)
        return output_layout, score

    def load_saved_network(self, net_path):
        self.network.load_state_dict(torch.load(net_path, map_location=self.device))
        self.network.train()                                     # ml_solver.py:131: inference stays in train mode
