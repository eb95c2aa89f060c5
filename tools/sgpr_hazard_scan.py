#!/usr/bin/env python3
"""Scans AMDGPU assembly (hipcc -save-temps) for the hazard the compiler cannot see through inline asm: a VALU
instruction that writes an SGPR -- v_readlane_b32 / v_readfirstlane_b32, a v_cmp with an SGPR-pair destination, the
carry-out of v_add_co / v_sub_co / v_mad_u64_u32 and friends, v_div_scale -- followed within 5 wait states by a
VMEM instruction (global / buffer / scratch loads, stores, atomics, LDS-DMA) that uses that SGPR as its scalar base
or offset.  The ISA wants 5 wait states there; the compiler inserts them for its own VMEM instructions, but an
`asm volatile("global_store_dword %0, %1, %2" :: "v", "v", "s"(base))` is opaque to its hazard recogniser.

Control flow is followed: the pending VALU writes at a branch travel to its target label (and fall through a
conditional branch), merged at every label with the youngest age per register, iterated to a fixed point -- a write at
the end of a predecessor block followed by an unguarded store at the head of a branch target is a hit.
Usage: sgpr_hazard_scan.py file.s | file.hsaco [...]; exit status 1 if anything was found.  (A code object is
disassembled first: what the run-time builds of pcx_generic_step are checked with, tests/test_generic_specialised.py.)"""
import re
import sys

READLANE = re.compile(r'^\s*(v_readlane_b32|v_readfirstlane_b32)\s+(s\d+)\b')
# VALU instructions whose FIRST operand is an SGPR pair: compares in their VOP3 form
VCMP_SDST = re.compile(r'^\s*v_cmpx?_\w+\s+s\[(\d+):(\d+)\]')
# ... and those that write an SGPR pair as their SECOND operand (carry / scale outputs)
CARRY_SDST = re.compile(r'^\s*(v_add_co_\w+|v_sub_co_\w+|v_subrev_co_\w+|v_addc_co_\w+|v_subb_co_\w+|v_subbrev_co_\w+|'
                        r'v_mad_u64_u32|v_mad_i64_i32|v_div_scale_\w+)\s+v(?:\[\d+:\d+\]|\d+)\s*,\s*s\[(\d+):(\d+)\]')
VMEM = re.compile(r'^\s*(global_|buffer_|scratch_|tbuffer_)\S*\s+(.*)')
SREG = re.compile(r's\[(\d+):(\d+)\]|\bs(\d+)\b')
NOP = re.compile(r'^\s*s_nop\s+(\d+)')
SALU_DST = re.compile(r"^\s*(s_(?!nop|waitcnt|cbranch|branch|barrier|endpgm|sleep|setprio|sethalt|cmp|bitcmp|load|buffer_load|atomic|"
                      r"store|dcache|sendmsg|trap|icache|setreg|set_gpr)\w+)\s+(?:s\[(\d+):(\d+)\]|s(\d+)\b)")
BRANCH = re.compile(r'^\s*(s_branch|s_cbranch_\w+)\s+(\.?\w+)')
LABEL = re.compile(r'^(\.?\w+):')
WINDOW = 5


def merge(a, b):
  """Pending writes {sgpr: age}: the youngest age of each register."""
  out = dict(a)
  for reg, age in b.items():
    if reg not in out or age < out[reg]:
      out[reg] = age
  return out


def scan_kernel(path, kernel, lines):
  """lines: [(line number, text)] of one kernel.  Returns the hits as a sorted list of strings."""
  incoming = {}  # label -> pending writes arriving over branches
  hits = {}
  for _ in range(8):  # to a fixed point (the window is 5 wait states: a handful of rounds at most)
    changed = False
    hits = {}
    recent = {}
    reachable = True  # (after an unconditional branch nothing falls through)
    for n, line in lines:
      lab = LABEL.match(line)
      if lab and not line.startswith('_Z'):
        arriving = incoming.get(lab.group(1), {})
        recent = merge(recent, arriving) if reachable else dict(arriving)
        reachable = True
        continue
      t = line.strip()
      if not t or t.startswith(';') or t.startswith('.'):
        continue
      m = VMEM.match(line)
      if m:
        used = set()
        for a, b, c in SREG.findall(m.group(2)):
          if a:
            used.update(range(int(a), int(b) + 1))
          else:
            used.add(int(c))
        for reg, age in recent.items():
          if reg in used and age < WINDOW:
            hits[(n, reg)] = '%s:%d: %s uses s%d written by a VALU %d wait state(s) earlier  [%s]' % (
                path, n, t.split()[0], reg, age, kernel[:90])
      sal = SALU_DST.match(line)
      if sal:  # an SALU write replaces the VALU-written value: what follows reads the SALU's result (no hazard)
        a, b, c = sal.group(2), sal.group(3), sal.group(4)
        gone = set(range(int(a), int(b) + 1)) if a else {int(c)}
        recent = {r: age for r, age in recent.items() if r not in gone}
      nop = NOP.match(line)
      step = 1 + int(nop.group(1)) if nop else 1
      recent = {r: age + step for r, age in recent.items() if age + step <= WINDOW}
      w = READLANE.match(line)
      if w:
        recent[int(w.group(2)[1:])] = 0
      for rx in (VCMP_SDST, CARRY_SDST):
        w = rx.match(line)
        if w:
          lo, hi = int(w.groups()[-2]), int(w.groups()[-1])
          for r in range(lo, hi + 1):
            recent[r] = 0
      br = BRANCH.match(line)
      if br:
        old = incoming.get(br.group(2), {})
        new = merge(old, recent)
        if new != old:
          incoming[br.group(2)] = new
          changed = True
        if br.group(1) == 's_branch':
          reachable = False
          recent = {}
    if not changed:
      break
  return [hits[k] for k in sorted(hits)]


OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
DIS_LABEL = re.compile(r'^[0-9a-f]+ <(L\d+)>:')
DIS_SYMBOL = re.compile(r'^[0-9a-f]+ <(\w+)>:')


def disassemble(path):
  """A code object (the run-time builds of pcx_generic_step have no .s file) as lines this scanner reads: llvm-objdump
  with symbolised branch targets, `<L5>:` as `L5:`, kernel symbols as `_Z...:` headers, encodings stripped."""
  import subprocess
  out = subprocess.check_output([OBJDUMP, '-d', '--symbolize-operands', '--no-show-raw-insn', path]).decode()
  lines = []
  for line in out.splitlines():
    line = line.split('//')[0].rstrip()
    m = DIS_LABEL.match(line)
    if m:
      lines.append(m.group(1) + ':')
      continue
    m = DIS_SYMBOL.match(line)
    if m:
      lines.append('_Z' + m.group(1) + ':')
      continue
    lines.append(line)
  return lines


def main(argv):
  bad = 0
  for path in argv:
    text = disassemble(path) if path.endswith(('.hsaco', '.co', '.o')) else open(path).read().splitlines()
    kernel, body = '?', []
    out = []
    for n, line in enumerate(text, 1):
      if line.endswith(':') and line.startswith('_Z'):
        out += scan_kernel(path, kernel, body)
        kernel, body = line[:-1], []
        continue
      body.append((n, line))
    out += scan_kernel(path, kernel, body)
    for h in out:
      print(h)
    bad += len(out)
  print('%d hazard(s)' % bad)
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main(sys.argv[1:]))
