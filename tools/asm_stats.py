#!/usr/bin/env python3
"""Static instruction statistics of one kernel in a -save-temps gfx950 assembly file.

  python tools/asm_stats.py <file.s> <substring of the mangled kernel name> [--dump out.s]
"""
import collections
import re
import sys


def kernels(path):
  name, body, out = None, [], {}
  for line in open(path):
    m = re.match(r'^(_Z\S+):\s*(;.*)?$', line)
    if m and not line.startswith('.L'):
      name, body = m.group(1), []
      out[name] = body
      continue
    if name is not None:
      body.append(line)
      if '.end_amdhsa_kernel' in line:
        name = None
  return out


def stats(body):
  c = collections.Counter()
  for line in body:
    t = line.strip()
    if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
      continue
    op = t.split()[0]
    c['total'] += 1
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'):
      c['lane moves'] += 1
    if op.startswith('v_'):
      c['VALU'] += 1
    elif op.startswith('s_waitcnt'):
      c['s_waitcnt'] += 1
    elif op.startswith('s_cbranch') or op.startswith('s_branch'):
      c['branches'] += 1
    elif op.startswith('s_load') or op.startswith('s_buffer_load'):
      c['SMEM'] += 1
    elif op.startswith('s_'):
      c['SALU'] += 1
    elif op.startswith('ds_'):
      c['LDS'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
      c['VMEM'] += 1
    if op in ('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_hi_i32', 'v_mad_u64_u32', 'v_mad_i64_i32'):
      c['quarter-rate mul'] += 1
  for line in body:
    m = re.search(r'\.(vgpr_count|sgpr_count|amdhsa_next_free_vgpr|amdhsa_next_free_sgpr|amdhsa_group_segment_fixed_size|amdhsa_private_segment_fixed_size)\S*\s+(\S+)', line)
    if m:
      c[m.group(1)] = m.group(2)
  return c


def main():
  ks = kernels(sys.argv[1])
  want = sys.argv[2]
  for name, body in ks.items():
    if want in name:
      print(name[:150])
      print('  ' + ', '.join('%s %s' % kv for kv in stats(body).items()))
      if '--dump' in sys.argv:
        open(sys.argv[sys.argv.index('--dump') + 1], 'w').writelines(body)


if __name__ == '__main__':
  main()
