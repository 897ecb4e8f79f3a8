"""Instruction mix per kernel from the device assembly (hipcc --cuda-device-only -S): VALU count, packed-fp32 share, MFMA, LDS,
VGPR / occupancy lines.  Usage: python tools/isa_mix.py dev.s [name filter ...]"""
import collections
import re
import sys

cur = None
cnt = collections.defaultdict(collections.Counter)
meta = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r'^\s+(v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)\s', line)
    if m and cur:
        cnt[cur][m.group(1)] += 1
        continue
    m = re.match(r'^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)', line)
    if m and cur:
        meta[cur][m.group(1)] = int(m.group(2))
flt = sys.argv[2:]
for k, c in cnt.items():
    if flt and not any(s in k for s in flt):
        continue
    valu = sum(v for i, v in c.items() if i.startswith('v_') and 'mfma' not in i)
    pk = sum(v for i, v in c.items() if i.startswith('v_pk_'))
    fp = sum(v for i, v in c.items() if re.match(r'v_(fma|fmac|mul|add|sub|fmamk|fmaak|mac|max|min)_f32', i))
    print('%-90s valu %5d  pk %4d  scalar-fp %4d  cndmask %4d  mov %4d  mfma %3d  ds %4d  glb %4d  trans %3d | %s' % (
        k[:90], valu, pk, fp, c['v_cndmask_b32'], c['v_mov_b32'] + c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32'],
        sum(v for i, v in c.items() if 'mfma' in i), sum(v for i, v in c.items() if i.startswith('ds_')),
        sum(v for i, v in c.items() if i.startswith(('global_', 'buffer_'))),
        sum(v for i, v in c.items() if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', i)), meta.get(k)))
