"""Mirror of the reference's ``ntracer.pygame_render`` (lib/ntracer/pygame_render.py): render onto a pygame
Surface and get a pygame event when the frame is done.  pygame is imported lazily -- it is not needed for
anything else in this package."""
import weakref

from . import render


def channels_from_surface(surface):
    """render.Channel list describing ``surface``'s pixel layout, most significant bit first
    (reference: lib/ntracer/pygame_render.py:8-46).  Works with any object offering pygame.Surface's
    get_bytesize / get_losses / get_shifts / get_masks.  Indexed (8-bit) modes are not supported."""
    nbytes = surface.get_bytesize()
    if nbytes == 1:
        raise TypeError("indexed color modes are not supported")
    total_bits = nbytes * 8
    fields = []
    for loss, shift, name in zip(surface.get_losses(), surface.get_shifts(), "RGBA"):
        width = 8 - loss
        if width:
            # position of the field's top bit, counted from the pixel's most significant bit
            fields.append((total_bits - (shift + width), width, name))
    fields.sort()
    channels = []
    cursor = 0
    for start, width, name in fields:
        if start < cursor:
            raise ValueError("overlapping colour masks")
        if start > cursor:
            channels.append(render.Channel(start - cursor, 0, 0, 0))          # padding bits
        channels.append(render.Channel(width, name == "R", name == "G", name == "B", name == "A"))
        cursor = start + width
    if cursor > total_bits:
        raise ValueError("colour masks exceed the pixel size")
    return channels


class PygameRenderer(render.CallbackRenderer):
    """A CallbackRenderer that draws onto a pygame.Surface and posts ``ON_COMPLETE`` (``source``, ``surface``,
    ``scene`` attributes) when the frame is finished (reference: lib/ntracer/pygame_render.py:51-117)."""
    ON_COMPLETE = None          # defaults to pygame.USEREVENT at first use
    instances = weakref.WeakSet()
    _quit_registered = False

    def __init__(self, threads=0, device=-1):
        super(PygameRenderer, self).__init__(threads, device)
        PygameRenderer.instances.add(self)
        self._layout_key = None
        self._channels = None

    def begin_render(self, surface, scene):
        import pygame
        if PygameRenderer.ON_COMPLETE is None:
            PygameRenderer.ON_COMPLETE = pygame.USEREVENT
        if not PygameRenderer._quit_registered:
            # pygame destroys surfaces at shutdown regardless of references: stop renders first
            pygame.register_quit(lambda: [r.abort_render() for r in list(PygameRenderer.instances)])
            PygameRenderer._quit_registered = True
        key = (surface.get_bitsize(), surface.get_masks())
        if key != self._layout_key:
            self._layout_key, self._channels = key, channels_from_surface(surface)
        fmt = render.ImageFormat(surface.get_width(), surface.get_height(), self._channels, surface.get_pitch(),
                                 pygame.get_sdl_byteorder() == pygame.LIL_ENDIAN)
        target = surface.get_view() if hasattr(surface, "get_view") else surface.get_buffer()

        def done(_renderer):
            pygame.event.post(pygame.event.Event(self.ON_COMPLETE, source=self, scene=scene, surface=surface))

        super(PygameRenderer, self).begin_render(target, fmt, scene, done)
