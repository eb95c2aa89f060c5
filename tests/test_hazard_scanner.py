"""tools/sgpr_hazard_scan.py, the build-time check behind the inline-asm plane stores (pcx_internal.h
saddr_store_dword; pycolab_amd/csrc/Makefile runs it on every kernel file's assembly): it must flag a scalar base
reloaded by a VALU instruction right before a VMEM instruction that uses it, and accept the forms the kernels rely
on -- enough wait states in between, an s_nop, or the guarded store that copies the base with s_mov_b64 first."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, 'tools', 'sgpr_hazard_scan.py')

HEADER = '_ZN3pcx2sm4testEv:\n'
FILLER = '\tv_add_u32_e32 v1, v2, v3\n'


def scan(tmp_path, body):
  path = os.path.join(str(tmp_path), 'k.s')
  with open(path, 'w') as f:
    f.write(HEADER + body + '\ts_endpgm\n')
  p = subprocess.run([sys.executable, SCAN, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
  return p.returncode, p.stdout


def test_flags_a_base_reloaded_right_before_the_store(tmp_path):
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n\tv_readlane_b32 s31, v158, 26\n'
                           '\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 1 and '2 hazard(s)' in out and 's30' in out and 's31' in out
  rc, out = scan(tmp_path, '\tv_readfirstlane_b32 s4, v9\n' + FILLER * 3 + '\tglobal_load_dword v5, v6, s[4:5]\n')
  assert rc == 1 and '1 hazard(s)' in out


def test_accepts_enough_wait_states(tmp_path):
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n' + FILLER * 5 + '\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 0, out
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n\ts_nop 4\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 0, out
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n\ts_nop 2\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 1, out  # 1 + 3 wait states: still short of five


def test_accepts_the_guarded_store_and_other_registers(tmp_path):
  # saddr_store_dword<true>: the base is copied by an SALU instruction inside the asm block; the store reads the copy
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n\tv_readlane_b32 s31, v158, 26\n'
                           '\ts_mov_b64 s[40:41], s[30:31]\n\tglobal_store_dword v21, v27, s[40:41]\n')
  assert rc == 0, out
  # an SALU write to the same register supersedes the VALU-written value
  rc, out = scan(tmp_path, '\tv_readlane_b32 s30, v158, 25\n\ts_mov_b32 s30, s12\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 0, out
  rc, out = scan(tmp_path, '\tv_readlane_b32 s10, v158, 25\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 0, out


def test_every_kernel_file_of_the_build_is_scanned():
  mk = open(os.path.join(ROOT, 'pycolab_amd', 'csrc', 'Makefile')).read()
  assert 'sgpr_hazard_scan.py' in mk and '-save-temps' in mk


def test_follows_branches_to_their_targets(tmp_path):
  """A VALU write of the base at the end of a predecessor block, the unguarded store at the head of the branch target
  (the linear scan of round 3 lost the write at the label in between)."""
  body = ('\tv_readlane_b32 s30, v158, 25\n\tv_readlane_b32 s31, v158, 26\n\ts_cbranch_scc1 .LBB0_7\n'
          + FILLER * 8 +
          '\ts_branch .LBB0_9\n.LBB0_7:\n\tglobal_store_dword v21, v27, s[30:31]\n.LBB0_9:\n')
  rc, out = scan(tmp_path, body)
  assert rc == 1 and 's30' in out and 's31' in out, out
  # ... far enough from the branch, the same store is fine; and nothing falls through an unconditional branch
  body = ('\tv_readlane_b32 s30, v158, 25\n\ts_cbranch_scc1 .LBB0_7\n' + FILLER * 8 + '\ts_branch .LBB0_9\n'
          '.LBB0_7:\n' + FILLER * 4 + '\tglobal_store_dword v21, v27, s[30:31]\n.LBB0_9:\n')
  rc, out = scan(tmp_path, body)
  assert rc == 0, out
  body = ('\tv_readlane_b32 s30, v158, 25\n\ts_branch .LBB0_9\n.LBB0_7:\n\tglobal_store_dword v21, v27, s[30:31]\n.LBB0_9:\n')
  rc, out = scan(tmp_path, body)
  assert rc == 0, out
  # a loop: the write at the bottom of the body reaches the store at its top over the back edge
  body = ('.LBB0_3:\n\tglobal_store_dword v21, v27, s[30:31]\n' + FILLER * 9 +
          '\tv_readlane_b32 s31, v158, 26\n\ts_cbranch_scc1 .LBB0_3\n')
  rc, out = scan(tmp_path, body)
  assert rc == 1 and 's31' in out, out


def test_other_valu_writes_of_sgprs_and_other_vmem_instructions(tmp_path):
  """Compares with an SGPR-pair destination, carry-outs, and the atomics / LDS-DMA of the persistent shapes."""
  rc, out = scan(tmp_path, '\tv_cmp_lt_u32_e64 s[30:31], v1, v2\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 1 and '2 hazard(s)' in out, out
  rc, out = scan(tmp_path, '\tv_mad_u64_u32 v[8:9], s[2:3], v6, s4, v[8:9]\n\tglobal_load_lds_dword v3, s[2:3]\n')
  assert rc == 1, out
  rc, out = scan(tmp_path, '\tv_readfirstlane_b32 s8, v4\n\tglobal_atomic_add v58, v5, v44, s[8:9] sc0\n')
  assert rc == 1, out
  rc, out = scan(tmp_path, '\tv_add_co_u32_e64 v1, s[6:7], v2, v3\n\ts_mov_b64 s[10:11], s[6:7]\n\tglobal_atomic_add v58, v5, v44, s[10:11] sc0\n')
  assert rc == 0, out
  # a scalar load is no SALU write of its destination for this purpose, but it is not a VALU write either
  rc, out = scan(tmp_path, '\ts_load_dwordx2 s[30:31], s[0:1], 0x10\n\ts_waitcnt lgkmcnt(0)\n\tglobal_store_dword v21, v27, s[30:31]\n')
  assert rc == 0, out
