"""MI355X-native (gfx950) implementation of the MiViT hot path -- see DESIGN.md.

    from moleculardiffusion_mivit_amd.helpers.models import GeneralTransformer, LinearProjectionEmbedding, MLPHead

Importing `helpers.models`, `ops` or `engine` loads libmivit_hip.so through ctypes (`_native.py`) and raises if it
has not been built (`python -m moleculardiffusion_mivit_amd.csrc.build`): there is no CPU or PyTorch fallback.
(The package root itself imports nothing, so the build module can run before the library exists.)
"""
__version__ = "0.1.0"
