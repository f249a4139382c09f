"""Structure epoch: lets TilinGNN cache its host-side table of device pointers safely.

Any event that can change WHICH storage a parameter/buffer lives in -- assigning a Parameter,
Tensor or Module attribute on one of the package's modules, or Module._apply (.to/.cuda/.float)
-- bumps a process-wide counter.  In-place updates (load_state_dict's copy_, optimizers, the
kernels' own running-stat updates) keep the pointers and need no invalidation."""
import torch

_EPOCH = 0


def epoch() -> int:
    return _EPOCH


def bump() -> None:
    global _EPOCH
    _EPOCH += 1


class Tracked:
    """Mixin for nn.Module subclasses (must precede nn.Module in the MRO)."""

    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, torch.nn.Module)) or value is None:
            bump()
        super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):
        bump()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        bump()                      # load_state_dict(assign=True) re-binds parameters and buffers in place of copying
        return super()._load_from_state_dict(*args, **kwargs)


class Linear(Tracked, torch.nn.Linear):
    pass


class BatchNorm1d(Tracked, torch.nn.BatchNorm1d):
    pass
