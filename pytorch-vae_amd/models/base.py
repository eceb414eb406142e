# -*- coding: utf-8 -*-
"""BaseVAE plugin surface (reference: models/base.py:5-28): encode / decode / sample / generate /
forward / loss_function.  Kept importable for drop-in compatibility; the reference's own VQVAE does
not subclass it and neither does ours."""
from abc import abstractmethod

from torch import nn

from .types_ import Any, List, Tensor


class BaseVAE(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def encode(self, input: Tensor) -> List[Tensor]:
        raise NotImplementedError

    def decode(self, input: Tensor) -> Any:
        raise NotImplementedError

    def sample(self, batch_size: int, current_device: int, **kwargs) -> Tensor:
        raise NotImplementedError

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        raise NotImplementedError

    @abstractmethod
    def forward(self, *inputs: Tensor) -> Tensor:
        pass

    @abstractmethod
    def loss_function(self, *inputs: Any, **kwargs) -> Tensor:
        pass
