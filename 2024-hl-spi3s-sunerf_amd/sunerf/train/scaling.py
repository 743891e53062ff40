"""Mirror of the reference's ``sunerf/train/scaling.py`` (loss-side image scaling; "next" row of SURVEY.md 8f)."""
import numpy as np
import torch
from torch import nn


class ImageLogScaling(nn.Module):
    def __init__(self, vmin, vmax):
        super().__init__()
        self.vmin = nn.Parameter(torch.tensor(vmin, dtype=torch.float32), requires_grad=False)
        self.vmax = nn.Parameter(torch.tensor(vmax, dtype=torch.float32), requires_grad=False)

    def forward(self, image):
        return (torch.log(image) - self.vmin) / (self.vmax - self.vmin)


class ImageAsinhScaling(nn.Module):
    """scaling.py:17-28: asinh(x / vmax / a) / asinh(1 / a)."""

    def __init__(self, vmax=1, a=0.005):
        super().__init__()
        self.normalization = nn.Parameter(torch.tensor(np.arcsinh(1 / a), dtype=torch.float32), requires_grad=False)
        self.a = nn.Parameter(torch.tensor(a, dtype=torch.float32), requires_grad=False)
        self.vmax = nn.Parameter(torch.tensor(vmax, dtype=torch.float32), requires_grad=False)

    def forward(self, image):
        image = image / self.vmax
        return torch.asinh(image / self.a) / self.normalization
