"""Instruction-class counts per kernel (and per loop body) from a hipcc -S listing: python tools/asm_mix.py file.s name-substring ..."""
import re, sys, collections


def classify(t):
    if t.startswith('v_mfma'): return 'mfma'
    if t.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): return 'trans'
    if t.startswith('v_accvgpr'): return 'acc'
    if t.startswith('v_'): return 'valu'
    if t.startswith('ds_'): return 'ds'
    if t.startswith(('global_', 'buffer_', 'scratch_')): return t.split('_')[0]
    if t.startswith('s_waitcnt'): return 'wait'
    if t.startswith('s_barrier'): return 'barrier'
    if t.startswith('s_'): return 'salu'
    return t


def main():
    lines = open(sys.argv[1]).read().split('\n')
    for name in sys.argv[2:]:
        start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l))
        end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
        total = collections.Counter()
        blocks, cur, label = [], collections.Counter(), 'entry'
        for l in lines[start + 1:end]:
            t = l.strip().split(' ')[0].split('\t')[0]
            if not t or t.startswith(('.', ';')):
                if re.match(r'^\.LBB\S+:', l.strip()):
                    blocks.append((label, cur)); cur, label = collections.Counter(), l.strip().split(':')[0]
                continue
            c = classify(t)
            total[c] += 1; cur[c] += 1
        blocks.append((label, cur))
        print(name, dict(total))
        for lab, c in blocks:
            if sum(c.values()) >= 150:
                print('   ', lab, dict(c))


main()
