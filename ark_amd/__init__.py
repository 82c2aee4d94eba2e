"""ark_amd: MI355X-native (gfx950) kernels + host engine for the ARK / SAIL training hot path."""
from ._lib import ArkError, LIB_PATH, PREC_F32, PREC_BF16  # noqa: F401
