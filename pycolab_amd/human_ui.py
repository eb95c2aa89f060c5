"""Placeholder for the reference's curses UI (pycolab/human_ui.py).

Interactive terminal play is outside this framework's scope (SURVEY.md 2,
row 13); the module exists so that game files which import it at module top
(e.g. examples/scrolly_maze.py:39) still load.
"""


class CursesUi(object):

  def __init__(self, *unused_args, **unused_kwargs):
    raise NotImplementedError(
        'pycolab_amd has no curses UI; step the Engine programmatically')
