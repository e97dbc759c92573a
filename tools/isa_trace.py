"""Condensed instruction-class trace of a kernel's K loop from a gfx950 .s file (no GPU needed).

    python tools/isa_trace.py file.s SUBSTRING [--full]

Finds the kernel whose (mangled) name contains SUBSTRING, takes the code between its first and last s_barrier, and prints
one token per instruction: M = MFMA, r = ds_read, w = ds_write, D = LDS-DMA (buffer_load ... lds), g = other VMEM,
v = VALU, s = SALU, W(...) = s_waitcnt with its counters, B = s_barrier, '|' branch/label.  With --full the raw lines.
Also prints the kernel's register / LDS metadata.
"""
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    full = '--full' in sys.argv
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sub in l.split(':')[0]]
    if not starts:
        sys.exit('no kernel matches')
    for st in starts:
        name = lines[st].split(':')[0]
        body = []
        for l in lines[st + 1:]:
            t = l.strip()
            if t.startswith('s_endpgm'):
                break
            if not t or t.startswith(';') or t.startswith('.'):
                if re.match(r'^\.LBB', l):
                    body.append(l.strip())
                continue
            body.append(t)
        bars = [i for i, t in enumerate(body) if t.startswith('s_barrier')]
        print('==', name, 'instructions', len(body), 'barriers', len(bars))
        if len(bars) < 2:
            continue
        seg = body[bars[0]:bars[-1] + 1]
        if full:
            print('\n'.join(seg))
            continue
        out = []
        cnt = dict(M=0, r=0, D=0, v=0, s=0, g=0, w=0)
        for t in seg:
            o = t.split()[0]
            if o.startswith('.LBB'):
                out.append('\n' + o + ' ')
            elif o.startswith('v_mfma'):
                out.append('M'); cnt['M'] += 1
            elif o.startswith('ds_read'):
                out.append('r'); cnt['r'] += 1
            elif o.startswith('ds_write'):
                out.append('w'); cnt['w'] += 1
            elif o.startswith('buffer_load') and ' lds' in t:
                out.append('D'); cnt['D'] += 1
            elif o.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
                out.append('g'); cnt['g'] += 1
            elif o == 's_waitcnt':
                out.append(' W(' + ','.join(re.findall(r'(?:vmcnt|lgkmcnt)\(\d+\)', t)).replace('vmcnt', 'v').replace('lgkmcnt', 'l') + ') ')
            elif o == 's_barrier':
                out.append(' B\n')
            elif o.startswith(('s_cbranch', 's_branch')):
                out.append('|' + t.split()[-1] + ' ')
            elif o.startswith('s_'):
                out.append('s'); cnt['s'] += 1
            elif o.startswith('v_'):
                out.append('v'); cnt['v'] += 1
            else:
                out.append('?')
        print(''.join(out))
        print('counts', cnt)
    # metadata
    txt = '\n'.join(lines)
    for m in re.finditer(r'\.name:\s+(\S+)', txt):
        pass


if __name__ == '__main__':
    main()
