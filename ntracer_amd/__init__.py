"""ntracer_amd -- MI355X-native replacement for NTracer's per-pixel ray-cast path.

Same top-level names as the reference's ``ntracer`` package (lib/ntracer/__init__.py) for the
path in scope: ``NTracer``, ``Material``, ``ImageFormat``, ``Channel``, ``BlockingRenderer``,
``CallbackRenderer``, ``Color``, ``CUBE``, ``SPHERE``.  All rendering goes through
libntracer_hip.so (include/ntracer_hip.h); there is no CPU fallback.
"""
from .render import (BlockingRenderer, CallbackRenderer, Channel, Color, ImageFormat, LockedError, Material,  # noqa: F401
                     Scene)
from .tracern import CUBE, SPHERE  # noqa: F401
from .wrapper import NTracer  # noqa: F401

__all__ = ["NTracer", "Material", "ImageFormat", "Channel", "BlockingRenderer", "CallbackRenderer", "Color",
           "LockedError", "Scene", "CUBE", "SPHERE"]
