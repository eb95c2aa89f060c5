"""The fixtures have teeth: each mutant of the oracle (oracle/mutants.py -- one plausible mis-reading of the reference
each, compiled from mutated source text) must FAIL the fixtures named for it.  The whole matrix (every mutant against
every fixture) is `python -m oracle.mutants`; DESIGN.md section 2 has its result.  Mutants that survived every fixture
when the matrix was first run (round 5) had fixtures recorded from the reference for them (`walkers_hidden`,
`better_scrolly_custom_C`, raise fixtures `walkers_scroll_disagree`, `warehouse_open_A/B`, the CPU test of
`fixed_crop_overhang`) or are marked `equivalent` with the reason why nothing the reference's own entities can do tells
them apart; one mutant cannot be told apart on the shipped levels and must SURVIVE them."""
import pytest

from oracle import mutants


@pytest.fixture(scope='module')
def scratch(tmp_path_factory):
  return str(tmp_path_factory.mktemp('pcx_mutants'))


def test_the_unmutated_oracle_passes_every_fixture_a_mutant_is_judged_by():
  from oracle import ref_live
  for f in sorted({f for m in mutants.MUTANTS for f in m.killed_by + m.survives}):
    if f.startswith('live:') and ref_live.reference_path() is None:
      continue
    assert mutants.fixture_passes(f), f


@pytest.mark.parametrize('mutant', mutants.MUTANTS, ids=[m.name for m in mutants.MUTANTS])
def test_fixtures_kill_the_mutant(mutant, scratch):
  so = mutants.build(mutant, scratch)
  if any(f.startswith('live:') for f in mutant.killed_by):
    from oracle import ref_live
    if ref_live.reference_path() is None:
      pytest.skip('judged by a level stepped next to the live reference, which is neither under /root/reference nor built under oracle/_ref')
  if mutant.equivalent:  # no fixture CAN tell it apart (the reason is the mutant's `equivalent`); it still has to compile
    assert not mutant.killed_by
    return
  with mutants.loaded(so):
    for f in mutant.killed_by:
      assert not mutants.fixture_passes(f), '%s (%s) passes %s' % (mutant.name, mutant.cite, f)
    for f in mutant.survives:
      assert mutants.fixture_passes(f), '%s: %s was expected not to tell it apart' % (mutant.name, f)
  assert mutants.fixture_passes(mutant.killed_by[0])  # (the real library is back)


def test_every_mutant_changes_exactly_one_place():
  names = [m.name for m in mutants.MUTANTS]
  assert len(set(names)) == len(names)
  for m in mutants.MUTANTS:
    texts = mutants.mutated_sources(m)  # raises if the anchor is not unique in the oracle's source
    assert m.new == '' or m.new in texts[m.source]
