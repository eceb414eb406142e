import os as _os
import sys as _sys

_PKG = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _PKG not in _sys.path:
    _sys.path.insert(0, _PKG)

from .base import *            # noqa: F401,F403
from .vq_vae import *          # noqa: F401,F403

# registry used by the reference's scripts: vae_models['VQVAE'](**model_params)
vae_models = {"VQVAE": VQVAE}
