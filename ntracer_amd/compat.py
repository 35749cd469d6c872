"""Running code written against the reference package unchanged.

``alias_reference_modules()`` registers this package under the reference's module names (``ntracer``,
``ntracer.render``, ``ntracer.wrapper``, ``ntracer.tracern``, ``ntracer.tracer3`` .. ``tracer8``,
``ntracer.pygame_render``), so that ``from ntracer import NTracer`` picks up the HIP path and pickles written by
the reference -- which name ``ntracer.render._vector_unpickle`` etc. (src/render.cpp:1662-1673) -- load here.
Nothing is aliased unless this is called; it refuses to shadow a real ``ntracer`` that is already imported."""
import sys


def alias_reference_modules(force=False):
    import ntracer_amd
    from ntracer_amd import render, tracern, wrapper
    names = {"ntracer": ntracer_amd, "ntracer.render": render, "ntracer.wrapper": wrapper, "ntracer.tracern": tracern}
    for n in range(3, 9):
        names["ntracer.tracer%d" % n] = tracern          # get_optimized_tracern: one module serves every dimension here
    try:
        from ntracer_amd import pygame_render
        names["ntracer.pygame_render"] = pygame_render
    except Exception:                                     # pygame absent: the renderer module is optional
        pass
    if not force:
        for k in names:
            m = sys.modules.get(k)
            if m is not None and m is not names[k]:
                raise RuntimeError("%s is already imported from %s" % (k, getattr(m, "__file__", "?")))
    sys.modules.update(names)
    return sorted(names)


def remove_aliases():
    import ntracer_amd
    for k in list(sys.modules):
        if (k == "ntracer" or k.startswith("ntracer.")) and getattr(sys.modules[k], "__name__", "").startswith("ntracer_amd"):
            del sys.modules[k]
