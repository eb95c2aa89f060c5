#!/usr/bin/env python3
"""Scans AMDGPU assembly (hipcc --offload-device-only -S) for the hazard the compiler cannot
see through inline asm: a VALU instruction that writes an SGPR (v_readlane_b32 / v_readfirstlane_b32 /
v_cmp into an SGPR pair) followed within 5 wait states by a VMEM instruction that uses that
SGPR as its scalar base.  The ISA wants 5 wait states there; the compiler inserts them for its own
VMEM instructions but an `asm volatile("global_store_dword %0, %1, %2" :: "v", "v", "s"(base))`
is opaque to its hazard recogniser.  Usage: sgpr_hazard_scan.py file.s [...]"""
import re
import sys

VALU_SGPR = re.compile(r'^\s*(v_readlane_b32|v_readfirstlane_b32)\s+(s\d+)')
VMEM = re.compile(r'^\s*(global_store|global_load|buffer_|scratch_)\S*\s+(.*)')
SREG = re.compile(r's\[(\d+):(\d+)\]|\bs(\d+)\b')
NOP = re.compile(r'^\s*s_nop\s+(\d+)')
SALU_DST = re.compile(r"^\s*(s_(?!nop|waitcnt|cbranch|branch|barrier|endpgm|sleep|setprio|sethalt|cmp|bitcmp)\w+)\s+(?:s\[(\d+):(\d+)\]|s(\d+)\b)")

bad = 0
for path in sys.argv[1:]:
  lines = open(path).read().splitlines()
  kernel = '?'
  recent = []  # (sgpr index, wait states since)
  for n, line in enumerate(lines, 1):
    if line.endswith(':') and line.startswith('_Z'):
      kernel = line[:-1]
      recent = []
      continue
    t = line.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
      continue
    m = VMEM.match(line)
    if m:
      used = set()
      for a, b, c in SREG.findall(m.group(2)):
        if a:
          used.update(range(int(a), int(b) + 1))
        else:
          used.add(int(c))
      for reg, age in recent:
        if reg in used and age < 5:
          bad += 1
          print('%s:%d: %s uses s%d written by a VALU %d wait state(s) earlier  [%s]' % (path, n, t.split()[0], reg, age, kernel[:90]))
    sal = SALU_DST.match(line)
    if sal:  # an SALU write replaces the VALU-written value: what follows reads the SALU's result (no hazard)
      a, b, c = sal.group(2), sal.group(3), sal.group(4)
      gone = set(range(int(a), int(b) + 1)) if a else {int(c)}
      recent = [(r, age) for r, age in recent if r not in gone]
    nop = NOP.match(line)
    step = 1 + int(nop.group(1)) if nop else 1
    recent = [(r, a + step) for r, a in recent if a + step < 6]
    w = VALU_SGPR.match(line)
    if w:
      recent.append((int(w.group(2)[1:]), 0))
print('%d hazard(s)' % bad)
sys.exit(1 if bad else 0)
