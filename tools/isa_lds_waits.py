"""Static census of EXPOSED LDS round trips in the K loops of the MFMA kernels (no GPU needed).

    python tools/isa_lds_waits.py [igemm wgrad_s1p wgrad_k4 wgrad attn_mfma mx8]

Compiles each csrc/<name>.hip to gfx950 assembly (`hipcc -S --cuda-device-only`) and, per kernel, looks at the code between
its first and last `s_barrier` (the K loop, for the unrolled kernels plus a little of the epilogue): counts the MFMAs, the
`ds_read`s and the "exposed" waits = `s_waitcnt ... lgkmcnt(0)` reached with at least one `ds_read` outstanding and NO MFMA
issued since that read -- the wave then sits through a full LDS round trip (>= ~128 cycles) with its matrix pipe idle, and
only the other wave(s) of the SIMD can fill the gap.  Round-2 finding: the tap-staged kernel has 9-14 such waits per K-step of
32 MFMAs (512 MFMA cycles), the patch kernels 8-23 per 100 MFMAs when the backward-epilogue prefetch holds ~96 VGPRs, 4-7
without it: the register allocator at 249/256 VGPRs issues fragment reads just in time.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'audio-depth-estimation_amd', 'csrc')
EXTRA = {'attn_mfma': ['-mllvm', '-amdgpu-mfma-vgpr-form', '-fno-honor-nans']}


def kernels(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            t = line.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                out[cur].append(t)
            if t.startswith('s_endpgm'):
                cur = None
    return out


def census(lines):
    ops = [ln.split()[0] for ln in lines]
    bars = [i for i, o in enumerate(ops) if o == 's_barrier']
    if len(bars) < 2:
        return None
    mfma = reads = exposed = pending = since = 0
    for ln in lines[bars[0]:bars[-1]]:
        o = ln.split()[0]
        if o.startswith('ds_read'):
            reads += 1
            pending += 1
            since = 0
        elif o.startswith('v_mfma'):
            mfma += 1
            since += 1
        elif o == 's_waitcnt' and 'lgkmcnt(0)' in ln:
            if pending and since == 0:
                exposed += 1
            pending = 0
    return mfma, reads, exposed


def main():
    names = sys.argv[1:] or ['igemm', 'wgrad_s1p', 'wgrad_k4', 'wgrad', 'attn_mfma', 'mx8']
    filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    tmp = tempfile.mkdtemp()
    for n in names:
        s = os.path.join(tmp, n + '.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=fast', *EXTRA.get(n, []),
                        '--cuda-device-only', '-S', '-o', s, os.path.join(CSRC, n + '.hip'), '-I', CSRC,
                        '-I', os.path.join(ROOT, 'include')], check=True, stderr=subprocess.DEVNULL)
        print(f'# {n}.hip')
        for k, lines in kernels(s).items():
            r = census(lines)
            if r and r[0] >= 16:
                name = subprocess.run([filt, k], capture_output=True, text=True).stdout.strip() if filt else k
                name = name.replace('(anonymous namespace)::', '')
                print(f'{r[2]:3d} exposed LDS waits / {r[0]:4d} MFMA ({100 * r[2] / r[0]:5.1f} per 100)   ds_reads {r[1]:4d}   {name[:110]}')


if __name__ == '__main__':
    main()
