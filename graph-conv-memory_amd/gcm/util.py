"""Helpers of the hot path (reference: src/gcm/util.py:9-26)."""
import torch


class STEFunction(torch.autograd.Function):
    """util.py:9-18 - forward (x > 0) as float, backward passes the gradient through."""

    @staticmethod
    def forward(ctx, input):
        return (input > 0).float()

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class StraightThroughEstimator(torch.nn.Module):
    """util.py:21-26."""

    def forward(self, x):
        return STEFunction.apply(x)


class Spardmax(torch.nn.Module):
    """util.py:29-42 - unusable at the reference HEAD too (its `sparsemax`
    import is commented out, util.py:5 -> NameError at util.py:36)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            "Spardmax needs the `sparsemax` package, which the reference itself no longer "
            "imports (util.py:5); deterministic=True selectors are out of scope")
