"""plainlm_amd — MI355X-native (gfx950) implementation of plainLM's training hot path.

Host-side mirror of the reference's module API (models/, engine/) over hand-written HIP
kernels behind a C ABI (include/plainlm_hip.h -> plainlm_amd/libplainlm_hip.so).
Importing the package does not require a GPU; running a model does.
"""

import os

# RCCL between processes needs dmabuf IPC on this stack (hipIpcGetMemHandle fails under the legacy mode).  Read when HIP initialises: effective
# when the package is imported before the first GPU call, harmless otherwise (INTEGRATION.md lists it with the launch environment).
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

from .construct import construct_model, get_param_groups  # noqa: F401
from .transformer import ModelConfig, Transformer  # noqa: F401
from .engine import HipEngine, TorchEngine  # noqa: F401

__all__ = ['construct_model', 'get_param_groups', 'ModelConfig', 'Transformer', 'HipEngine', 'TorchEngine']
