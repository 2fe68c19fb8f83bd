"""plainlm_amd — MI355X-native (gfx950) implementation of plainLM's training hot path.

Host-side mirror of the reference's module API (models/, engine/) over hand-written HIP
kernels behind a C ABI (include/plainlm_hip.h -> plainlm_amd/libplainlm_hip.so).
Importing the package does not require a GPU; running a model does.
"""

from .construct import construct_model, get_param_groups  # noqa: F401
from .transformer import ModelConfig, Transformer  # noqa: F401
from .engine import HipEngine, TorchEngine  # noqa: F401

__all__ = ['construct_model', 'get_param_groups', 'ModelConfig', 'Transformer', 'HipEngine', 'TorchEngine']
