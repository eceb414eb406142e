from typing import Any, Callable, List, Tuple, TypeVar, Union

Tensor = TypeVar("torch.tensor")
