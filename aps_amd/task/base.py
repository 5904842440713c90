"""Task: the module that glues the network forward and the loss (aps/task/base.py:14-31)."""
from typing import Optional

import torch.nn as nn


class Task(nn.Module):
    """nnet: network instance; ctx: context module for the loss (e.g. an STFT layer)"""

    def __init__(self, nnet: nn.Module, ctx: Optional[nn.Module] = None,
                 description: str = "unknown") -> None:
        super(Task, self).__init__()
        self.nnet = nnet
        self.ctx = ctx
        self.description = description
