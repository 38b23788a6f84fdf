"""genozip_amd -- MI355X (gfx950) implementation of Genozip's context entropy-coding hot path.

See DESIGN.md. The compute path is libgenozip_amd.so (hand-written HIP, C-ABI in include/genozip_amd.h);
this package is the thin host-side mirror of the reference's codec/context interface over that ABI.
"""
from .lib import (CODEC_NONE, CODEC_RANB, CODEC_RANW, CODEC_RANb, CODEC_RANw, CODEC_ARTB, CODEC_ARTW, CODEC_ARTb,  # noqa: F401
                  CODEC_ARTw, SIMPLE_CODECS, CODEC_NAMES, SEC_B250, SEC_LOCAL)

__all__ = ["Engine"]


def __getattr__(name):
    if name in ("Engine", "Section", "VBlock", "GenozipAMDError"):
        from . import codec
        return getattr(codec, name)
    raise AttributeError(name)
