"""Loss-side image scalings with the reference's class names and state-dict keys (sunerf/train/scaling.py:6-28).

The constants are frozen, non-trainable ``nn.Parameter`` scalars (that is what puts ``normalization`` / ``a`` / ``vmax`` /
``vmin`` into reference checkpoints).  On the fused training path the asinh scaling is evaluated inside ``loss_kernel``
(csrc/train_step.hip) from ``(vmax, a)``; the ``forward`` methods here serve validation / plotting code that calls the
modules directly, on any device.
"""
import math

import torch
from torch import nn


def _frozen_scalars(module: nn.Module, **values) -> None:
    """Registers every keyword as a 0-d fp32 parameter that takes no gradient."""
    for name, value in values.items():
        module.register_parameter(name, nn.Parameter(torch.tensor(float(value), dtype=torch.float32), requires_grad=False))


class ImageLogScaling(nn.Module):
    """``(log(image) - vmin) / (vmax - vmin)`` (scaling.py:6-14)."""

    def __init__(self, vmin, vmax):
        super().__init__()
        _frozen_scalars(self, vmin=vmin, vmax=vmax)

    def forward(self, image):
        return torch.log(image).sub(self.vmin).div(self.vmax - self.vmin)


class ImageAsinhScaling(nn.Module):
    """``asinh(image / vmax / a) / asinh(1 / a)`` (scaling.py:17-28): three fp32 roundings in that order, the normalisation
    constant being ``asinh(1 / a)`` evaluated in double precision and stored as fp32."""

    def __init__(self, vmax=1, a=0.005):
        super().__init__()
        _frozen_scalars(self, normalization=math.asinh(1 / a), a=a, vmax=vmax)

    def forward(self, image):
        return torch.asinh(image.div(self.vmax).div(self.a)).div(self.normalization)
