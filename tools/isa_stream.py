"""Compress a kernel's assembly into a per-basic-block instruction-class stream:  python tools/isa_stream.py file.s <mangled-name-prefix> [min_block_len]
M = MFMA, v = VALU, e = v_exp, c = v_cvt, L = ds_read, W = ds_write, G = global/buffer load/store, s = SALU, w = s_waitcnt, B = s_barrier, b = branch, n = s_nop"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if l.startswith(name))
out, cur, label = [], [], 'entry'
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_exp'): return 'e'
    if op.startswith('v_cvt') or op.startswith('v_fma_mix'): return 'c'
    if op.startswith('v_accvgpr'): return 'a'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'L'
    if op.startswith('ds_'): return 'W'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'G'
    if op == 's_waitcnt': return 'w'
    if op == 's_barrier': return 'B'
    if op == 's_nop': return 'n'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'b'
    if op.startswith('s_'): return 's'
    return '?'
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end') or l.strip().startswith('.end_amdhsa_kernel'): break
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        out.append((label, ''.join(cur))); cur = []; label = m.group(1); continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur.append(cls(t.split()[0]))
out.append((label, ''.join(cur)))
for lab, sq in out:
    if len(sq) >= minlen:
        print(lab, len(sq), 'M=%d v=%d e=%d c=%d a=%d L=%d G=%d' % tuple(sq.count(k) for k in 'MvecaLG'))
        for i in range(0, len(sq), 160): print('   ', sq[i:i + 160])
