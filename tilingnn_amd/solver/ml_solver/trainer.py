"""`Trainer` -- the training loop (/root/reference/solver/ml_solver/trainer.py:22-124).

What the reference does per epoch (:68-124): for every layout of `<data_path>/train/raw/*.pkl` (a PyG Dataset/DataLoader,
batch_size 1, inputs/config.py:44): forward in train mode, `Losses.calculate_unsupervised_loss`, `loss.backward()`,
`optimizer.step()`; then the average loss over the training and the testing split (`Losses.cal_avg_loss`), and a
checkpoint of network + optimizer when the test loss improved or every `save_model_per_epoch` epochs, which
`ml_solver.load_saved_network` reads back.

Here: the same loop with
  * the layout files read by the shapely-free loader (util/data_util.py) ONCE and kept on the GPU (a split of 20 000
    layouts of a few thousand nodes is a few GB; the reference re-reads `data_<i>.pt` from disk every step),
  * forward + backward through the adjoint kernels (tilingnn_amd/train.py; `network.autograd` is switched on around the
    steps), the loss and its gradient on the GPU (solver/ml_solver/losses.py),
  * the caller's optimizer untouched (`optimizer.step()` on the `.grad`s, as in network_train.py).
Not mirrored: `create_data` (random target shapes cut out of the complete graph with shapely, trainer.py:39-50, :126-165)
and the per-checkpoint debug plots (`ml_solver.save_debug_info`, trainer.py:113-121).  batch_size must be 1, the only
value the reference configures; a larger one would need PyG's disjoint-union batching.
"""
import glob
import os
import traceback

import numpy as np
import torch

from ...util import data_util
from ...util.algorithms import DeviceLayout
from .losses import Losses


class LayoutDataset:
    """`<root>/raw/*.pkl` (GraphDataset.raw_file_names, solver/ml_solver/data_util.py:15-17) resident on the device."""

    def __init__(self, root, device):
        self.files = sorted(glob.glob(os.path.join(root, "raw", "*.pkl")))
        self.layouts = []
        for f in self.files:
            _, x, col_idx, col_feat, adj_idx, adj_feat, *_ = data_util.load_brick_layout_data(f)
            if x is None or col_idx is None or adj_idx is None or adj_feat is None:
                raise ValueError(f"{f}: a training layout file must carry its features (write_bricklayout(with_features=True))")
            self.layouts.append(DeviceLayout.upload(_Arrays(x, adj_idx, adj_feat, col_idx), device))
        from ...graph_networks import _graph_cache
        _graph_cache.reserve(len(_graph_cache._entries) + len(self.layouts))    # every layout's prepared graph stays cached

    def __len__(self):
        return len(self.layouts)

    def __getitem__(self, i):
        return self.layouts[i]


class _Arrays:
    def __init__(self, node_feature, align_edge_index, align_edge_features, collide_edge_index):
        self.node_feature, self.align_edge_index = node_feature, align_edge_index
        self.align_edge_features, self.collide_edge_index = align_edge_features, collide_edge_index


def cal_avg_loss(network, layouts):
    """Losses.cal_avg_loss (losses.py:14-45), first return value: the mean loss over a split."""
    losses = []
    was = network.autograd
    network.autograd = False
    try:
        for lay in layouts:
            if lay.align_edge_index.numel() == 0 or lay.collide_edge_index.numel() == 0:
                continue
            with torch.no_grad():
                probs, _ = network(lay.node_feature, lay.align_edge_index, lay.align_edge_features, lay.collide_edge_index)
                loss, _, _ = Losses.calculate_unsupervised_loss(probs, lay.node_feature, lay.collide_edge_index,
                                                                lay.align_edge_index, lay.align_edge_features)
            losses.append(float(loss))
    finally:
        network.autograd = was
    return float(np.mean(losses)) if losses else float("nan")


class Trainer:
    def __init__(self, debugger, plotter, device, network, data_path, model_save_path=None):
        self.debugger, self.plotter, self.device, self.network = debugger, plotter, device, network
        self.data_path = data_path
        self.training_path = os.path.join(data_path, "train")
        self.testing_path = os.path.join(data_path, "test")
        if model_save_path is None:                                      # trainer.py:27, :34-35
            model_save_path = debugger.file_path("model") if debugger is not None else os.path.join(data_path, "model")
        self.model_save_path = model_save_path
        os.makedirs(self.model_save_path, exist_ok=True)

    def create_data(self, *args, **kwargs):
        raise NotImplementedError("create_data cuts random target shapes with shapely (trainer.py:39-50): generate the "
                                  "layout files with the reference, they load here")

    def train_step(self, layout, optimizer):
        """trainer.py:69-84 for one layout; returns the loss (a 0-dim tensor) or None when the layout has an empty edge
        set (the reference's forward cannot run on one either)."""
        if layout.align_edge_index.numel() == 0 or layout.collide_edge_index.numel() == 0:
            return None
        probs, _ = self.network(layout.node_feature, layout.align_edge_index, layout.align_edge_features,
                                layout.collide_edge_index)
        optimizer.zero_grad()
        loss, *_ = Losses.calculate_unsupervised_loss(probs, layout.node_feature, layout.collide_edge_index,
                                                      adj_edges_index=layout.align_edge_index,
                                                      adj_edge_features=layout.align_edge_features)
        loss.backward()
        optimizer.step()
        return loss.detach()

    def train(self, ml_solver, optimizer, batch_size=1, training_epoch=10000, save_model_per_epoch=5, shuffle_seed=None,
              log=print):
        if batch_size != 1:
            raise NotImplementedError("batch_size 1 only (inputs/config.py:44)")
        train_set = LayoutDataset(self.training_path, self.device)
        test_set = LayoutDataset(self.testing_path, self.device)
        rng = np.random.default_rng(shuffle_seed)
        log("Training Start!!!")
        min_test_loss = float("inf")
        history = []
        for epoch in range(training_epoch):
            self.network.train()
            self.network.autograd = True
            try:
                for i in rng.permutation(len(train_set)):               # DataLoader(shuffle=True), trainer.py:61
                    try:
                        self.train_step(train_set[int(i)], optimizer)
                    except Exception:                                    # trainer.py:82-84: report and go on
                        log(traceback.format_exc())
            finally:
                self.network.autograd = False
            loss_train = cal_avg_loss(self.network, train_set)
            log(f"epoch {epoch}: training loss: {loss_train}")
            loss_test = cal_avg_loss(self.network, test_set)
            log(f"epoch {epoch}: testing loss: {loss_test}")
            history.append((loss_train, loss_test))
            if loss_test < min_test_loss or epoch % save_model_per_epoch == 0:      # trainer.py:96-108
                min_test_loss = min(min_test_loss, loss_test)
                model_file = os.path.join(self.model_save_path, f"model_{epoch}_{loss_test}.pth")
                torch.save(self.network.state_dict(), model_file)
                torch.save(optimizer.state_dict(), os.path.join(self.model_save_path, f"optimizer_{epoch}_{loss_test}.pth"))
                log(f"model saved at epoch {epoch}")
                if ml_solver is not None:
                    ml_solver.load_saved_network(model_file)
        log("Training Done!!!")
        return history
