"""Builds lib/libvoxvae.so from csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libvoxvae.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function', '-Wno-shift-op-parentheses']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest(paths):
    import hashlib
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps):
    """Staleness by CONTENT: `<target>.srchash` holds the sha256 of the sources the target was built from (mtimes do not
    survive a checkout or the snapshot that carries the tree to the GPU box)."""
    stamp = target + '.srchash'
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps) + ' ' + ' '.join(FLAGS)


def _mark(target, deps):
    with open(target + '.srchash', 'w') as f:
        f.write(_digest(deps) + ' ' + ' '.join(FLAGS))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(PKG), 'include', 'voxvae.h'))
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append(([HIPCC] + FLAGS + ['-c', src, '-o', obj], obj, [src] + hdrs))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    def compile_one(job):
        cmd, obj, deps = job
        run(cmd)
        _mark(obj, deps)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    allsrc = sources() + hdrs
    if jobs or force or _stale(LIB, allsrc):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
        _mark(LIB, allsrc)
    return LIB


def is_current():
    """True when lib/libvoxvae.so was built from exactly the sources in the tree (content hash)."""
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(PKG), 'include', 'voxvae.h'))
    return not _stale(LIB, sources() + hdrs)


if __name__ == '__main__':
    import sys
    print(build(force='-f' in sys.argv, verbose=True))
