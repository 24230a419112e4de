"""Static check of the hand-counted waits in the gfx950 kernels; part of the BUILD (voxvae/build.py runs `check_all` after it compiled
anything, so a compiler or source change that breaks a hand-counted `vmcnt` fails the build, not a later GPU test).

Several kernels issue loads from inline asm and wait for them with hand-counted `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` so that
a look-ahead load can stay in flight across barriers and MFMA blocks (a compiler-placed wait would be vmcnt(0)).  To the
compiler an asm output is an ordinary, READY value: nothing stops the register allocator from copying it, or from handing its
registers to something else, while the load is still in flight -- round 2 lost a GPU to exactly that in a variant of the last
layer's kernel.  And a counted wait is only right while the number of vector-memory operations between a
load and its wait is what the author counted (FEWER operations than counted = the wait no longer covers the load; more
operations only make it stricter).  Both conditions are properties of the generated code, so they are checked on the
generated code:

  `hipcc --offload-device-only -S` -> per kernel a control-flow graph of instructions -> from every asm-issued load a forward
  walk along ALL paths, counting the same-counter operations issued since (k), until a wait with N <= k resolves it
  (the counters retire in order), and reporting
    * every instruction that names a register of the load's destination before that point ("touched in flight"),
    * (a scalar-memory load inside the window of an LDS read makes only lgkmcnt(0) count from there on: lgkmcnt is shared
      with SMEM, which returns out of order),
    * how many asm waits of that counter a load passes before it is resolved (`waits`): the ring depth the author intended.
      A count that drifts (an epilogue that now issues 7 stores where vmcnt(9) assumed 8) shows up as a load that survives
      one wait more than the number pinned in PINNED_WAITS below.

Compiler-issued loads need no check: the compiler counts only what it knows, asm-issued operations in between make its own
waits stricter, never weaker.
"""
import hashlib
import os
import re
import subprocess

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
ISA_DIR = os.path.join(PKG, 'lib', 'isa')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-w']   # = voxvae/build.py's code generation flags

VM_OP = re.compile(r'^(buffer_|global_|flat_|scratch_|image_|tbuffer_)')
REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def compile_isa(src, force=False):
    """csrc/<name>.hip -> lib/isa/<name>.s (cached by content hash of the source and the csrc headers)."""
    os.makedirs(ISA_DIR, exist_ok=True)
    name = os.path.basename(src)[:-4]
    out = os.path.join(ISA_DIR, name + '.s')
    h = hashlib.sha256()
    deps = [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(ROOT, 'include', 'voxvae.h')]
    for p in deps:
        with open(p, 'rb') as f:
            h.update(f.read())
    from . import build as _build
    flags = FLAGS + _build.EXTRA_FLAGS.get(os.path.basename(src), [])
    stamp = h.hexdigest() + ' ' + ' '.join(flags)
    sp = out + '.srchash'
    if not force and os.path.exists(out) and os.path.exists(sp) and open(sp).read() == stamp:
        return out
    subprocess.check_call([HIPCC] + flags + ['--offload-device-only', '-S', src, '-o', out], stderr=subprocess.DEVNULL)
    with open(sp, 'w') as f:
        f.write(stamp)
    return out


class Ins(object):
    __slots__ = ('line', 'op', 'args', 'asm', 'text')

    def __init__(self, line, op, args, asm, text):
        self.line, self.op, self.args, self.asm, self.text = line, op, args, asm, text


def parse(path):
    """-> {kernel name: dict(ins=[Ins], labels={label: index}, meta={...})}.  Only functions that are kernels (have an
    .amdhsa_kernel record) are kept."""
    funcs, cur, name, in_asm = {}, None, None, False
    meta = {}
    lines = open(path).read().split('\n')
    for no, raw in enumerate(lines, 1):
        s = raw.split(';;#')[0] if ';;#' not in raw else raw
        if ';;#ASMSTART' in raw:
            in_asm = True
            continue
        if ';;#ASMEND' in raw:
            in_asm = False
            continue
        t = raw.split(';')[0].strip() if not in_asm else raw.split(';')[0].strip()
        m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', raw)
        if m and not raw.startswith('.') and not raw.startswith('\t'):
            name = m.group(1)
            cur = {'ins': [], 'labels': {}}
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if raw.startswith('.Lfunc_end'):
            cur = None
            continue
        m = re.match(r'^(\.LBB\w+):', raw)
        if m:
            cur['labels'][m.group(1)] = len(cur['ins'])
            continue
        if not t or t.startswith('.') or t.startswith('//'):
            continue
        parts = t.split(None, 1)
        cur['ins'].append(Ins(no, parts[0], parts[1] if len(parts) > 1 else '', in_asm, t))
    text = '\n'.join(lines)
    for m in re.finditer(r'\.amdhsa_kernel (\S+)', text):
        meta[m.group(1)] = {}
    # metadata: spills / scratch per kernel (the YAML note at the end of the file)
    for blk in re.split(r'\n  - \.agpr_count:', text)[1:]:
        nm = re.search(r'\.name:\s+(\S+)', blk)
        if not nm:
            continue
        d = meta.setdefault(nm.group(1), {})
        for key in ('vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'vgpr_count'):
            mm = re.search(r'\.%s:\s+(\d+)' % key, blk)
            if mm:
                d[key] = int(mm.group(1))
    return {k: dict(v, meta=meta.get(k, {})) for k, v in funcs.items() if k in meta}


def regs_of(args):
    out = set()
    for m in REG.finditer(args):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def wait_counts(ins):
    """s_waitcnt -> {'vm': N or None, 'lgkm': N or None}."""
    if ins.op != 's_waitcnt':
        return None
    a = ins.args
    if re.fullmatch(r'\s*0\s*', a):
        return {'vm': 0, 'lgkm': 0}
    vm = re.search(r'vmcnt\((\d+)\)', a)
    lg = re.search(r'lgkmcnt\((\d+)\)', a)
    return {'vm': int(vm.group(1)) if vm else None, 'lgkm': int(lg.group(1)) if lg else None}


def asm_loads(fn):
    """[(index, counter, destination registers or None for an LDS-DMA)] for every load issued from inline asm."""
    out = []
    for i, ins in enumerate(fn['ins']):
        if not ins.asm:
            continue
        if ins.op.startswith('ds_read') or ins.op.startswith('ds_load'):
            out.append((i, 'lgkm', regs_of(ins.args.split(',')[0])))
        elif VM_OP.match(ins.op) and 'load' in ins.op:
            if re.search(r'\blds\b', ins.args):
                out.append((i, 'vm', None))
            else:
                out.append((i, 'vm', regs_of(ins.args.split(',')[0])))
    return out


def successors(fn, i):
    ins = fn['ins'][i]
    if ins.op == 's_endpgm':
        return []
    if ins.op == 's_branch':
        return [fn['labels'][ins.args.strip()]]
    if ins.op.startswith('s_cbranch'):
        return [fn['labels'][ins.args.strip()], i + 1]
    if ins.op in ('s_setpc_b64', 's_swappc_b64'):
        raise ValueError('indirect control flow at line %d' % ins.line)
    return [i + 1] if i + 1 < len(fn['ins']) else []


def walk(fn, start, counter, dest, kcap=12, wcap=6):
    """Forward from the asm load at `start` along all paths.  State per path: k = same-counter operations issued since the
    load (the load is complete at a wait with N <= k: the counters retire in order), w = asm waits of this counter passed
    while it was pending.  A scalar-memory load inside the window of an LDS read freezes k (lgkmcnt is shared with SMEM,
    which returns out of order: only lgkmcnt(0) resolves from then on).  States are explored smallest k first, so an
    instruction is expanded at most once per distinct w.
    -> dict(waits = max asm waits passed before resolution, touched = [Ins naming a destination register while in flight],
            notes = [str])."""
    import heapq
    ins_list = fn['ins']
    res = {'waits': 0, 'touched': [], 'notes': []}
    FROZEN = -1
    done = {}                      # instruction index -> (smallest k expanded, largest w expanded)
    heap = [(0, 0, s) for s in successors(fn, start)]
    heapq.heapify(heap)
    seen_touch = set()
    while heap:
        k, negw, i = heapq.heappop(heap)
        w = -negw
        while True:
            if i == start:                         # issued again (loop back-edge) without ever having been read: a dead look-ahead
                break
            prev = done.get(i)
            if prev is not None and prev[0] <= k and prev[1] >= w:
                break
            done[i] = (k if prev is None else min(k, prev[0]), w if prev is None else max(w, prev[1]))
            ins = ins_list[i]
            wc = wait_counts(ins)
            if wc is not None:
                n = wc['vm'] if counter == 'vm' else wc['lgkm']
                if n is not None:
                    if n == 0 or (k != FROZEN and n <= k):
                        res['waits'] = max(res['waits'], w + (1 if ins.asm else 0))
                        break                      # resolved on this path
                    if ins.asm:
                        w = min(w + 1, wcap)
            elif dest is not None and (regs_of(ins.args) & dest):
                if i not in seen_touch:
                    seen_touch.add(i)
                    res['touched'].append(ins)
                break                              # reported; whatever follows on this path is wrong anyway
            if k != FROZEN:
                if counter == 'vm' and VM_OP.match(ins.op):
                    k = min(k + 1, kcap)
                elif counter == 'lgkm':
                    if ins.op.startswith('ds_'):
                        k = min(k + 1, kcap)
                    elif ins.op.startswith(('s_load', 's_buffer_load', 's_scratch_load')):
                        k = FROZEN
            nxt = successors(fn, i)
            if not nxt:
                break                              # s_endpgm: the hardware drains the counters before the wave's LDS is released
            for s in nxt[1:]:
                heapq.heappush(heap, (k, -w, s))
            i = nxt[0]
    return res


def lint_file(src):
    """-> {kernel: dict(loads=n, dma=n, violations=[str], max_waits_vm, max_waits_lgkm, meta)}"""
    fns = parse(compile_isa(src))
    out = {}
    for name, fn in fns.items():
        loads = asm_loads(fn)
        if not loads:
            continue
        r = {'loads': 0, 'dma': 0, 'violations': [], 'max_waits_vm': 0, 'max_waits_lgkm': 0, 'meta': fn['meta']}
        for idx, counter, dest in loads:
            w = walk(fn, idx, counter, dest)
            li = fn['ins'][idx]
            if dest is None:
                r['dma'] += 1
            else:
                r['loads'] += 1
            key = 'max_waits_vm' if counter == 'vm' else 'max_waits_lgkm'
            r[key] = max(r[key], w['waits'])
            for t in w['touched']:
                r['violations'].append('line %d `%s`: destination touched in flight at line %d `%s`' % (li.line, li.text, t.line, t.text))
        out[name] = r
    return out


# kernel family -> (max asm vmcnt waits, max asm lgkmcnt waits) a look-ahead load passes before the wait that covers it: the ring
# depth of each kernel as built today (GPU-tested).  None = not statically bounded (the wait count is selected at run time from
# several `s_waitcnt`s; the walker cannot tell which paths are feasible).
PINNED_WAITS = {
    'conv_direct_kernel': (2, 2), 'conv_direct16_kernel': (2, 2), 'conv_direct16h_kernel': (2, 1), 'conv_direct_fp8_kernel': (2, 2),
    'convT_direct_kernel': (1, 2), 'convT_direct_fp8_kernel': (1, 2),
    'ctw_kernel': (4, 2), 'ctw16_kernel': (4, 2), 'ctw4_kernel': (4, 2),
    'final_bce_sweep_kernel': (3, 0), 'final_bce_sweepw_kernel': (3, 0), 'final_bce_sweep_fp8_kernel': (3, 0), 'final_bce_mfma_kernel': (0, 0),
    'igemm_kernel': (None, 0), 'pg_kernel': (None, 2), 'lt_e5_kernel': (1, 0),
    'sd_kernel': (2, 2), 'wgrad_bf16_kernel': (1, 1), 'wgrad_phase_kernel': (1, 2),
}
# kernels whose counted waits were written for a register-resident loop: no scratch, no spills
NO_SCRATCH = ('conv_direct_kernel', 'conv_direct16_kernel', 'conv_direct16h_kernel', 'conv_direct_fp8_kernel', 'convT_direct_fp8_kernel', 'ctw_kernel', 'ctw16_kernel',
              'final_bce_sweep_kernel', 'final_bce_sweepw_kernel', 'final_bce_sweep_fp8_kernel', 'sd_kernel', 'pg_kernel', 'wgrad_phase_kernel', 'igemm_kernel')
# kernel family -> (source, VGPR budget): launchers that deal the work items for a fixed number of workgroups per CU.  The first layer's
# plane form was written for four per CU (128 VGPRs); removing its timing ablations let the register allocator drift to 134 = three per
# CU under a launcher that still dealt for four (a 1.33-round grid, ~20 % of the kernel) and nothing noticed for a round.
RESIDENT_BUDGET = {'first_conv_chain_kernel': ('first_last.hip', 128), 'final_bce_sweep_kernel': ('first_last.hip', 168),
                   'final_bce_sweepw_kernel': ('first_last.hip', 128)}


def family(mangled):
    m = re.match(r'^_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?_kernel)', mangled)
    return m.group(1) if m else mangled


def check_file(src):
    """-> list of problem strings for one csrc/*.hip (empty = clean): in-flight asm outputs touched, ring depths past their pins,
    scratch where a counted-wait loop was written for registers, register budgets of the fixed-residency launchers."""
    problems = []
    for kernel, r in lint_file(src).items():
        fam = family(kernel)
        for v in r['violations'][:4]:
            problems.append('%s: %s' % (kernel, v))
        if fam not in PINNED_WAITS:
            problems.append('new kernel with asm loads: add %s to PINNED_WAITS (vm %d, lgkm %d)' % (fam, r['max_waits_vm'], r['max_waits_lgkm']))
            continue
        vm, lg = PINNED_WAITS[fam]
        if vm is not None and r['max_waits_vm'] > vm:
            problems.append('%s: a look-ahead load now survives %d asm vmcnt waits (pinned %d)' % (kernel, r['max_waits_vm'], vm))
        if lg is not None and r['max_waits_lgkm'] > lg:
            problems.append('%s: an LDS read now survives %d asm lgkmcnt waits (pinned %d)' % (kernel, r['max_waits_lgkm'], lg))
        if fam in NO_SCRATCH and (r['meta'].get('private_segment_fixed_size', 0) or r['meta'].get('vgpr_spill_count', 0)):
            problems.append('%s: scratch / spills in a loop written for registers: %s' % (kernel, r['meta']))
    base = os.path.basename(src)
    for fam, (fsrc, budget) in RESIDENT_BUDGET.items():
        if fsrc != base:
            continue
        seen = False
        for kernel, fn in parse(compile_isa(src)).items():
            if family(kernel) != fam:
                continue
            seen = True
            m = fn['meta']
            if m.get('vgpr_count', 0) > budget or m.get('private_segment_fixed_size', 0) or m.get('vgpr_spill_count', 0):
                problems.append('%s: %s against a budget of %d VGPRs, no scratch (its launcher deals work for a fixed residency)' % (kernel, m, budget))
        if not seen:
            problems.append('%s: no kernel of family %s found (RESIDENT_BUDGET)' % (base, fam))
    return problems


def check_all(sources=None, workers=4):
    """Lint every csrc/*.hip (ISA text cached by content hash under lib/isa/); raises RuntimeError listing every problem."""
    from concurrent.futures import ThreadPoolExecutor
    files = sources or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(compile_isa, files))           # the compiles run in parallel (subprocesses); the walks are quick
    problems = [p for f in files for p in check_file(f)]
    if problems:
        raise RuntimeError('ISA lint failed (voxvae/isa_lint.py):\n  ' + '\n  '.join(problems))
    return len(files)


if __name__ == '__main__':
    import sys
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    for f in files:
        for k, r in lint_file(f).items():
            print('%-28s %-70.70s loads %3d dma %3d waits vm %d lgkm %d spills %s scratch %s  %s' % (
                os.path.basename(f), k, r['loads'], r['dma'], r['max_waits_vm'], r['max_waits_lgkm'], r['meta'].get('vgpr_spill_count'),
                r['meta'].get('private_segment_fixed_size'), 'OK' if not r['violations'] else '%d VIOLATIONS' % len(r['violations'])))
            for v in r['violations'][:6]:
                print('      ', v)
