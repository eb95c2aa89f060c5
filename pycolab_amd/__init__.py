"""pycolab_amd: an MI355X-native batched gridworld step engine with pycolab's API.

`ascii_art.ascii_art_to_game()` / `Engine.its_showtime()` / `Engine.play()` and
the `things.Sprite`-`Drape`-`Backdrop` classes keep the reference's surface;
stepping happens in hand-written HIP kernels (csrc/) reached through the C ABI
declared in include/pcx.h.
"""

__version__ = '0.1.0'
