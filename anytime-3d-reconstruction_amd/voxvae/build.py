"""Builds the HIP libraries from csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU):

    lib/libvoxvae.so         the RELEASE library: no environment variable is read anywhere (`vv_hook` is a null pointer, common.h)
    lib/libvoxvae_hooks.so   the same sources with -DVV_TEST_HOOKS: the kernel-form overrides (VV_CTW_SHAPE, VV_FINAL_BCE, ...) that the
                             A/B tests and microbenchmarks use; only the sources that contain a hook are compiled a second time

and runs the ISA lint (voxvae/isa_lint.py) on everything it compiled: hand-counted `s_waitcnt vmcnt(N)`, in-flight asm outputs,
scratch-free loops and the register budgets the launchers assume are properties of the GENERATED code, so a compiler or source change
that breaks one fails the build.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libvoxvae.so')
LIB_HOOKS = os.path.join(LIBDIR, 'libvoxvae_hooks.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function', '-Wno-shift-op-parentheses']
HOOK_FLAG = '-DVV_TEST_HOOKS'
# per-source code-generation flags.  convt_whole.hip: the four-wave kernel places its epilogue piece by piece between MFMAs; the SLP
# vectoriser pairs up operations of different pieces into v_pk_* instructions (an anti-lever beside MFMAs, MI355X_MICROARCH.md) and piles the work
# up in one gap
EXTRA_FLAGS = {'convt_whole.hip': ['-fno-slp-vectorize']}


def flags_for(src, hooks=False):
    return FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ([HOOK_FLAG] if hooks else [])


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(PKG), 'include', 'voxvae.h'))
    return hdrs


def _digest(paths):
    import hashlib
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps, flags=FLAGS):
    """Staleness by CONTENT: `<target>.srchash` holds the sha256 of the sources the target was built from (mtimes do not
    survive a checkout or the snapshot that carries the tree to the GPU box)."""
    stamp = target + '.srchash'
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps) + ' ' + ' '.join(flags)


def _mark(target, deps, flags=FLAGS):
    with open(target + '.srchash', 'w') as f:
        f.write(_digest(deps) + ' ' + ' '.join(flags))


def _uses_hooks(src):
    with open(src) as f:
        text = f.read()
    return 'vv_hook(' in text or 'vv_chunk_samples(' in text       # (vv_chunk_samples reads VV_CHUNK_SAMPLES, common.h)


def build(force=False, verbose=False, lint=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(os.path.join(LIBDIR, 'hooks'), exist_ok=True)
    hdrs = _headers()
    objs, hobjs, jobs = [], [], []
    hflags = FLAGS + [HOOK_FLAG]
    for src in sources():
        base = os.path.basename(src)[:-4]
        obj = os.path.join(LIBDIR, base + '.o')
        objs.append(obj)
        fl = flags_for(src)
        if force or _stale(obj, [src] + hdrs, fl):
            jobs.append(([HIPCC] + fl + ['-c', src, '-o', obj], obj, [src] + hdrs, fl))
        if _uses_hooks(src):
            hobj = os.path.join(LIBDIR, 'hooks', base + '.o')
            hobjs.append(hobj)
            hfl = flags_for(src, True)
            if force or _stale(hobj, [src] + hdrs, hfl):
                jobs.append(([HIPCC] + hfl + ['-c', src, '-o', hobj], hobj, [src] + hdrs, hfl))
        else:
            hobjs.append(obj)

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    def compile_one(job):
        cmd, obj, deps, flags = job
        run(cmd)
        _mark(obj, deps, flags)

    jobs.sort(key=lambda j: -os.path.getsize(j[2][0]))       # the implicit GEMM (a minute per variant) first
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(compile_one, jobs))
    allsrc = sources() + hdrs
    if jobs or force or _stale(LIB, allsrc):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
        _mark(LIB, allsrc)
    if jobs or force or _stale(LIB_HOOKS, allsrc, hflags):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_HOOKS] + hobjs)
        _mark(LIB_HOOKS, allsrc, hflags)
    if lint:
        lint_build(force=force, verbose=verbose)
    return LIB


def lint_build(force=False, verbose=False):
    """The ISA lint over the kernel sources, once per content hash of csrc/ (lib/isa/lint.ok)."""
    from . import isa_lint
    allsrc = sources() + _headers() + [os.path.abspath(isa_lint.__file__)]
    ok = os.path.join(LIBDIR, 'isa', 'lint.ok')
    if not force and not _stale(ok, allsrc):
        return
    if verbose:
        print('ISA lint (hand-counted waits, in-flight asm outputs, scratch, register budgets) ...', flush=True)
    n = isa_lint.check_all()
    with open(ok, 'w') as f:
        f.write('%d sources clean\n' % n)
    _mark(ok, allsrc)


def is_current(hooks=False):
    """True when the library was built from exactly the sources in the tree (content hash)."""
    allsrc = sources() + _headers()
    return not (_stale(LIB_HOOKS, allsrc, FLAGS + [HOOK_FLAG]) if hooks else _stale(LIB, allsrc))


if __name__ == '__main__':
    import sys
    print(build(force='-f' in sys.argv, verbose=True))
