"""Template compiler: a constructed game -> plain data for the HIP engine.

`GameTemplate` is the host image of `pcx_template` (include/pcx.h): what the
constructors of the user's Sprite/Drape/Backdrop classes left behind
(reference: ascii_art.py:243-289, engine.py:248-518), plus the device program
chosen for each entity.  It round-trips through `.npz` so that benchmark and
GPU tests can run where the game files themselves are absent.
"""

import ctypes
import json

import numpy as np

from pycolab_amd import _native as N
from pycolab_amd import programs
from pycolab_amd import things
from pycolab_amd.prefab_parts import drapes as prefab_drapes
from pycolab_amd.prefab_parts import sprites as prefab_sprites


def _impassable_bits(chars):
  bits = bytearray(16)
  for ch in chars:
    o = ord(ch)
    if o > 127:
      raise ValueError('impassable characters must be ASCII')
    bits[o >> 3] |= 1 << (o & 7)
  return bytes(bits)


def _directive(ch, selector, call, things_by_char, float_rewards=False):
  """One ('add_reward', r) / ('terminate_episode'[, d]) / ('change_z_order', a, b) /
  ('next_chapter', key) entry as a pcx_directive tuple, checked the way plot.py and engine.py check it."""
  name, args = call[0], tuple(call[1:])
  if name == 'add_reward':
    (reward,) = args
    if float_rewards:  # pcx_template::reward_is_float: the table holds the bits of a float32
      return (ch, N.DIR_ADD_REWARD, 0, 0, selector, int(np.array([reward], np.float32).view(np.int32)[0]), 0.0)
    if int(reward) != reward:
      raise ValueError('device rewards are integers (float32 ones: `pcx_float_rewards = True` on an entity class of the game)')
    return (ch, N.DIR_ADD_REWARD, 0, 0, selector, int(reward), 0.0)
  if name == 'terminate_episode':
    discount = float(args[0]) if args else 0.0
    if not 0.0 <= discount <= 1.0:  # plot.py:192-193
      raise ValueError('Discount must be in range [0,1].')
    return (ch, N.DIR_TERMINATE, 0, 0, selector, 0, discount)
  if name == 'change_z_order':
    move_this, in_front_of = args
    if move_this not in things_by_char:  # engine.py:804-808 raises when the directive is applied
      raise RuntimeError(
          'A z-order change directive said to move a Sprite or Drape '
          'corresponding to character {}, but no such Sprite or Drape '
          'exists'.format(repr(move_this)))
    if in_front_of is not None and in_front_of not in things_by_char:  # engine.py:809-814
      raise RuntimeError(
          'A z-order change directive said to move a Sprite or Drape in '
          'front of a Sprite or Drape corresponding to character {}, but '
          'no such Sprite or Drape exists'.format(repr(in_front_of)))
    if move_this == in_front_of:
      raise ValueError('a z-order change directive cannot move {} in front of itself'.format(repr(move_this)))
    return (ch, N.DIR_Z_ORDER, ord(move_this), 0 if in_front_of is None else ord(in_front_of), selector, 0, 0.0)
  if name == 'next_chapter':  # the_plot.next_chapter = key (plot.py:299-324; examples/ordeal.py:177-235)
    (key,) = args
    if key is not None and (isinstance(key, bool) or int(key) != key or int(key) < 0):
      raise ValueError('on the device a next_chapter is a chapter index (a non-negative integer key) or None')
    return (ch, N.DIR_NEXT_CHAPTER, 0, 0, selector, N.CHAPTER_NONE if key is None else int(key), 0.0)
  raise ValueError('unknown plot directive {!r}'.format(name))


class GameTemplate(object):
  """Plain-data image of a built (not yet started) game."""

  def __init__(self):
    self.game = 0
    self.rows = self.cols = 0
    self.occlusion_in_layers = True
    self.chars = b''            # sorted characters = layer plane order
    self.backdrop = None        # uint8 [rows, cols]
    self.sprites = []           # dicts, engine insertion order
    self.drapes = []            # dicts
    self.z_order = b''
    self.schedule = b''
    self.group_of = []
    self.n_groups = 0
    self.n_actions = 0
    self.param = [0] * 8
    self.directives = []        # (ch, kind, move_this, in_front_of, selector, reward, discount)
    self.reward_is_float = False  # include/pcx.h pcx_template::reward_is_float
    self.n_plot_words = 0         # ... n_plot_words
    self.chapter_keys = None      # host only: the Story keys behind the chapter codes the programs assign to next_chapter

  # -- construction from a host Engine ---------------------------------------
  @classmethod
  def from_engine(cls, eng):
    t = cls()
    t.rows, t.cols = eng.rows, eng.cols
    t.occlusion_in_layers = bool(eng._occlusion_in_layers)
    if eng.backdrop is None:
      raise RuntimeError('an Engine needs a Backdrop before its_showtime()')
    t.backdrop = np.ascontiguousarray(eng.backdrop.curtain, dtype=np.uint8)
    chars = set(eng._sprites_and_drapes.keys()).union(eng.backdrop.palette)
    t.chars = bytes(sorted(ord(c) for c in chars))
    if len(t.chars) > N.MAX_CHARS:
      raise ValueError('at most {} distinct characters'.format(N.MAX_CHARS))
    progs = [programs.resolve(eng.backdrop)]
    if progs[0] != N.PROG_STATIC:
      raise programs.UnsupportedEntityError('only static Backdrops are supported')
    # scrolling groups (protocols/scrolling.py): indices into the sorted distinct names
    group_names = sorted({getattr(ent, '_scrolling_group', '') for ent in eng._sprites_and_drapes.values()})
    if len(group_names) > N.MAX_SCROLL_GROUPS:
      raise programs.UnsupportedEntityError('at most {} scrolling groups per game'.format(N.MAX_SCROLL_GROUPS))
    group_index = {name: i for i, name in enumerate(group_names)} if group_names != [''] else {'': 0}
    for ch, ent in eng._sprites_and_drapes.items():
      prog = programs.resolve(ent)
      progs.append(prog)
      if isinstance(ent, things.Sprite):
        walker = isinstance(ent, prefab_sprites.MazeWalker)
        pos = ent.position
        vpos = ent.virtual_position if walker else pos
        t.sprites.append(dict(
            ch=ord(ch), is_walker=int(walker), visible=int(bool(ent.visible)),
            prior_visible=int(bool(getattr(ent, '_prior_visible', None))),
            confined=int(bool(getattr(ent, '_confined_to_board', False))),
            egocentric=int(bool(getattr(ent, '_egocentric_scroller', False))),
            scrolling_group=group_index[getattr(ent, '_scrolling_group', '')],
            program=prog, row=int(pos[0]), col=int(pos[1]),
            vrow=int(vpos[0]), vcol=int(vpos[1]),
            impassable=_impassable_bits(getattr(ent, '_impassable', ())),
            param=programs.extract_params(ent, prog)))
      else:
        scrolly = isinstance(ent, prefab_drapes.Scrolly)
        d = dict(ch=ord(ch), is_scrolly=int(scrolly), have_margins=0, program=prog,
                 scrolling_group=group_index[getattr(ent, '_scrolling_group', '')],
                 curtain=np.ascontiguousarray(ent.curtain, dtype=np.uint8),
                 pattern=None, corner=(0, 0), margins=(0, 0),
                 param=programs.extract_params(ent, prog))
        if scrolly:
          d['pattern'] = np.ascontiguousarray(ent.whole_pattern, dtype=np.uint8)
          d['corner'] = (int(ent._northwest_corner[0]), int(ent._northwest_corner[1]))
          d['have_margins'] = int(ent._have_margins)
          if ent._have_margins:
            d['margins'] = (int(ent._scroll_margins[0]), int(ent._scroll_margins[1]))
        t.drapes.append(d)
    # examples/ordeal.py: float rewards, Plot entries in the plot words, the player told which chapter it is in
    if any(p in programs.ORDEAL_PROGRAMS for p in progs):
      t.reward_is_float, t.n_plot_words, t.chapter_keys = True, 3, list(programs.ORDEAL_CHAPTERS)
      this = eng.the_plot.this_chapter
      for s in t.sprites:
        if s['program'] == N.PROG_OD_PLAYER:
          s['param'] = [t.chapter_keys.index(this) if this in t.chapter_keys else -1, 0, 0, 0]
    t.reward_is_float = t.reward_is_float or any(
        getattr(ent, 'pcx_float_rewards', False) for ent in eng._sprites_and_drapes.values())
    # plot directives of tabled entities (prefab_parts/tabled.py), in z-order of the entities
    for ch, ent in eng._sprites_and_drapes.items():
      for selector, calls in sorted(getattr(ent, 'pcx_directives', {}).items()):
        if int(selector) <= 0:
          raise ValueError('directive selector values must be positive (0 means "no directive")')
        for call in calls:
          t.directives.append(_directive(ord(ch), int(selector), call, eng._sprites_and_drapes, t.reward_is_float))
    if len(t.directives) > N.MAX_DIRECTIVES:
      raise ValueError('at most {} plot directives per game'.format(N.MAX_DIRECTIVES))
    if len(t.sprites) > N.MAX_SPRITES or len(t.drapes) > N.MAX_DRAPES:
      raise ValueError('too many sprites or drapes for the device engine')
    t.z_order = bytes(ord(c) for c in eng._sprites_and_drapes.keys())
    sched, group_of = [], []
    for gi, (_, entities) in enumerate(eng._frozen_update_groups()):
      for ent in entities:
        sched.append(ord(ent.character))
        group_of.append(gi)
      t.n_groups = gi + 1
    t.schedule = bytes(sched)
    t.group_of = group_of
    t.game = programs.infer_game(progs)
    t.n_actions = programs.N_ACTIONS[t.game]
    if t.n_plot_words:
      t.n_actions = programs.ORDEAL_N_ACTIONS
    if t.game == N.GAME_WALKERS and not t.n_plot_words:
      # tabled entities read bit fields of the action: "ordinary" actions (what
      # the benchmark / hashed-action tests draw from) cover every field
      top = 0
      for ent in list(t.sprites) + list(t.drapes):
        prm = ent['param']
        for shift, mask in ((prm[0], prm[1]), (prm[2], prm[3])):
          if mask:
            top = max(top, shift + int(mask).bit_length())
      if top:
        t.n_actions = 1 << min(top, 30)
    return t

  # -- (de)serialisation -------------------------------------------------------
  def save(self, path):
    meta = dict(game=self.game, rows=self.rows, cols=self.cols,
                occlusion_in_layers=self.occlusion_in_layers,
                chars=list(self.chars), z_order=list(self.z_order),
                schedule=list(self.schedule), group_of=list(self.group_of),
                n_groups=self.n_groups, n_actions=self.n_actions,
                param=list(self.param), directives=[list(d) for d in self.directives], sprites=[], drapes=[])
    if self.reward_is_float or self.n_plot_words:  # (only where they say something: the earlier fixtures stay byte-identical)
      meta.update(reward_is_float=bool(self.reward_is_float), n_plot_words=int(self.n_plot_words), chapter_keys=self.chapter_keys)
    arrays = {'backdrop': self.backdrop}
    for s in self.sprites:
      m = dict(s)
      m['impassable'] = list(s['impassable'])
      meta['sprites'].append(m)
    for i, d in enumerate(self.drapes):
      m = {k: v for k, v in d.items() if k not in ('curtain', 'pattern')}
      m['has_pattern'] = d['pattern'] is not None
      meta['drapes'].append(m)
      arrays['curtain_%d' % i] = np.packbits(d['curtain'].astype(bool))
      if d['pattern'] is not None:
        arrays['pattern_%d' % i] = np.packbits(d['pattern'].astype(bool))
        arrays['pattern_shape_%d' % i] = np.array(d['pattern'].shape, np.int32)
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)

  @classmethod
  def load(cls, path):
    z = np.load(path)
    meta = json.loads(bytes(z['meta']).decode())
    t = cls()
    t.game, t.rows, t.cols = meta['game'], meta['rows'], meta['cols']
    t.occlusion_in_layers = meta['occlusion_in_layers']
    t.chars = bytes(meta['chars'])
    t.z_order, t.schedule = bytes(meta['z_order']), bytes(meta['schedule'])
    t.group_of, t.n_groups = meta['group_of'], meta['n_groups']
    t.n_actions, t.param = meta['n_actions'], meta['param']
    t.directives = [tuple(d) for d in meta.get('directives', [])]
    t.reward_is_float, t.n_plot_words = bool(meta.get('reward_is_float', False)), int(meta.get('n_plot_words', 0))
    t.chapter_keys = meta.get('chapter_keys')
    t.backdrop = np.ascontiguousarray(z['backdrop'], dtype=np.uint8)
    n = t.rows * t.cols
    for s in meta['sprites']:
      s = dict(s)
      s['impassable'] = bytes(s['impassable'])
      t.sprites.append(s)
    for i, d in enumerate(meta['drapes']):
      d = dict(d)
      has_pattern = d.pop('has_pattern')
      d['corner'], d['margins'] = tuple(d['corner']), tuple(d['margins'])
      d['curtain'] = np.unpackbits(z['curtain_%d' % i])[:n].reshape(t.rows, t.cols).astype(np.uint8)
      d['pattern'] = None
      if has_pattern:
        shape = tuple(int(v) for v in z['pattern_shape_%d' % i])
        d['pattern'] = np.unpackbits(z['pattern_%d' % i])[:shape[0] * shape[1]].reshape(shape).astype(np.uint8)
      t.drapes.append(d)
    return t

  def __eq__(self, other):
    if not isinstance(other, GameTemplate):
      return NotImplemented
    def norm(t):
      return (t.game, t.rows, t.cols, bool(t.occlusion_in_layers), t.chars,
              t.backdrop.tobytes(), t.z_order, t.schedule, list(t.group_of),
              t.n_groups, t.n_actions, list(t.param), [tuple(d) for d in t.directives], bool(t.reward_is_float), int(t.n_plot_words),
              [sorted((k, (v if not isinstance(v, (list, tuple)) else tuple(v)))
                      for k, v in s.items()) for s in t.sprites],
              [sorted((k, (v.tobytes() if isinstance(v, np.ndarray) else
                           (tuple(v) if isinstance(v, (list, tuple)) else v)))
                      for k, v in d.items()) for d in t.drapes])
    return norm(self) == norm(other)

  def thing_chars(self):
    return [chr(c) for c in self.z_order]

  # -- the table-driven kernel's run-time build -----------------------------------
  def prebuild(self):
    """Compile (or find in the cache) the build of `pcx_generic_step` specialised for this template -- what an engine
    of 4,096 environments or more would do at `its_showtime()` (1-3 s once per template; include/pcx.h
    pcx_generic_specialise_check, csrc/pcx_generic.hip) -- or, for a scrolly_maze level of one's own on the example's 10x30
    board, the two instances of `pcx_scrolly_maze_step` with the level's constants compiled in (~25 s once per level).  Needs no GPU: a deployment can fill `$PCX_JIT_CACHE` at
    build time, or rank 0 for the others.  Returns the code object's size in bytes; raises `NotImplementedError` for
    templates only the hand-written kernels step (scrolly_maze's programs) and `RuntimeError` with the compiler's
    words if the build fails."""
    ct, _keep = self.to_ctypes()
    log, size = ctypes.create_string_buffer(1 << 16), N.c_i64(0)
    rc = N.lib().pcx_generic_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(size))
    kernel = 'pcx_generic_step'
    if rc == N.E_UNSUPPORTED and not log.value:
      # not a template of the table-driven kernel: a scrolly_maze level of one's own on the example's board gets the
      # run-time instances of pcx_scrolly_maze_step (round 6: include/pcx.h pcx_scrolly_maze_specialise_check)
      why = N.lib().pcx_last_error().decode()
      rc = N.lib().pcx_scrolly_maze_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(size))
      kernel = 'pcx_scrolly_maze_step'
      if rc == N.E_UNSUPPORTED and not log.value:
        raise NotImplementedError(why + '; ' + N.lib().pcx_last_error().decode())
    if rc != 0:
      raise RuntimeError('no specialised build of %s: ' % kernel + (log.value.decode() or N.lib().pcx_last_error().decode()))
    return int(size.value)

  # -- ctypes image --------------------------------------------------------------
  def to_ctypes(self):
    """Returns (Template, keepalive) -- keepalive owns the pointed-to arrays."""
    keep = []

    def ptr(arr):
      arr = np.ascontiguousarray(arr, dtype=np.uint8)
      keep.append(arr)
      return arr.ctypes.data_as(N.c_u8_p)

    ct = N.Template()
    ct.abi_version = N.ABI_VERSION
    ct.game = self.game
    ct.rows, ct.cols = self.rows, self.cols
    ct.occlusion_in_layers = int(bool(self.occlusion_in_layers))
    ct.n_chars = len(self.chars)
    for i, c in enumerate(self.chars):
      ct.chars[i] = c
    ct.backdrop = ptr(self.backdrop)
    ct.n_sprites = len(self.sprites)
    for i, s in enumerate(self.sprites):
      cs = ct.sprites[i]
      for k in ('ch', 'is_walker', 'visible', 'prior_visible', 'confined',
                'egocentric', 'program', 'row', 'col', 'vrow', 'vcol'):
        setattr(cs, k, s[k])
      cs.scrolling_group = s.get('scrolling_group', 0)
      for j, b in enumerate(s['impassable']):
        cs.impassable[j] = b
      for j, v in enumerate(s['param']):
        cs.param[j] = v
    ct.n_drapes = len(self.drapes)
    for i, d in enumerate(self.drapes):
      cd = ct.drapes[i]
      cd.ch, cd.is_scrolly, cd.have_margins = d['ch'], d['is_scrolly'], d['have_margins']
      cd.program = d['program']
      cd.scrolling_group = d.get('scrolling_group', 0)
      cd.curtain = ptr(d['curtain'])
      if d['pattern'] is not None:
        cd.pattern = ptr(d['pattern'])
        cd.pattern_rows, cd.pattern_cols = d['pattern'].shape
      cd.corner_row, cd.corner_col = d['corner']
      cd.margin_rows, cd.margin_cols = d['margins']
      for j, v in enumerate(d['param']):
        cd.param[j] = v
    ct.n_things = len(self.z_order)
    for i in range(ct.n_things):
      ct.z_order[i] = self.z_order[i]
      ct.schedule[i] = self.schedule[i]
      ct.group_of[i] = self.group_of[i]
    ct.n_groups = self.n_groups
    ct.n_actions = self.n_actions
    for i, v in enumerate(self.param):
      ct.param[i] = v
    ct.n_directives = len(self.directives)
    for i, (ch, kind, move_this, in_front_of, selector, reward, discount) in enumerate(self.directives):
      d = ct.directives[i]
      d.ch, d.kind, d.move_this, d.in_front_of = ch, kind, move_this, in_front_of
      d.selector, d.reward, d.discount = selector, reward, discount
    ct.reward_is_float, ct.n_plot_words = int(bool(self.reward_is_float)), int(self.n_plot_words)
    return ct, keep
