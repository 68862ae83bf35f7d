"""MI355X-native (gfx950) implementation of the MiViT hot path -- see DESIGN.md.

    from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead

Importing the package loads libmivit_hip.so through ctypes and raises if it has not been built
(`python -m moleculardiffusion_mivit_amd.csrc.build`): there is no CPU or PyTorch fallback.
"""
from . import _native  # noqa: F401  (fails loudly when the HIP library is missing)

__all__ = ["_native"]
__version__ = "0.1.0"
