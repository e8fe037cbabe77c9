#!/usr/bin/env python3
"""Condensed instruction-class sequence of one kernel from hipcc -S output: isa_seq.py file.s mangled_name_substring"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*%s\S*):' % re.escape(key), s, re.M)
name = m.group(1)
i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
seq = []
for l in s[i:j].split('\n'):
    l = l.strip()
    m = re.match(r'(s_waitcnt[^;]*|s_barrier|v_mfma\w+|buffer_load\w+|buffer_store\w+|global_\w+|ds_read\w+|ds_write\w+|s_cbranch\w+|s_branch|\.LBB\S+:|v_\w+|s_\w+)', l)
    if not m: continue
    tok = m.group(1).strip()
    if tok.startswith('v_mfma'): tok = 'MFMA'
    elif tok.startswith('v_'): tok = 'v'
    elif tok.startswith('s_') and not tok.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch')): tok = 's'
    if seq and seq[-1][0] == tok: seq[-1][1] += 1
    else: seq.append([tok, 1])
print(name)
print(' '.join(f"{t}x{c}" if c > 1 else t for t, c in seq))
