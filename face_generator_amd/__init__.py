"""face_generator_amd -- MI355X-native (gfx950) GAN training hot path of aleju/face-generator.

The product is `libfacegen_hip.so` (hand-written HIP kernels behind the C ABI of include/facegen_hip.h).
This package is the host side above that ABI: it mirrors the reference's Lua-side surface
(models.lua, utils/nn_utils.lua, interruptable_optimizers.lua, adversarial.lua) in Python because the
image has no Lua; lua/ holds the LuaJIT-FFI binding of the same entry points (not executable here).
PyTorch is used only for device memory, streams and torch.distributed.
"""
from ._lib import FgError, lib_path, load_library  # noqa: F401

__all__ = ["FgError", "lib_path", "load_library"]
