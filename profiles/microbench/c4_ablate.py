"""Writes scratch/abl/libc4_<mask>.so: the tree's convt_whole.hip with parts of ctw4_kernel's loop compiled out (-DC4_ABL=<mask>, wrong
results, timing only) linked against the tree's other objects.  Run on the build box (CPU), then
`gpurun -- python profiles/microbench/mb_c4_ablate.py 0 1 2 ...`.  Mask bits: 1 no epilogue pieces in the loop, 2 no barrier, 4 no weight
LDS-DMA in the loop, 8 no fragment reads in the loop, 16 no MFMAs.  Extra -D flags after `--` (e.g. -DC4_READS_EARLY=1)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(R, 'anytime-3d-reconstruction_amd')
out = os.path.join(R, 'scratch', 'abl'); os.makedirs(out, exist_ok=True)
args = sys.argv[1:]
extra = []
if '--' in args:
    extra = args[args.index('--') + 1:]
    args = args[:args.index('--')]
objs = [os.path.join(PKG, 'lib', f) for f in sorted(os.listdir(os.path.join(PKG, 'lib'))) if f.endswith('.o') and f != 'convt_whole.o']
for m in (args or ['0', '1', '2', '3', '8', '16']):
    tag = m + ''.join('_' + e.lstrip('-D').replace('=', '') for e in extra)
    o = os.path.join(out, 'c4_%s.o' % tag)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-w', '-fno-slp-vectorize',
                           '-DVV_TEST_HOOKS', '-DC4_ABL=%s' % m] + extra + ['-c', os.path.join(PKG, 'csrc', 'convt_whole.hip'), '-o', o])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libc4_%s.so' % tag), o] + objs)
    print('built', tag)
