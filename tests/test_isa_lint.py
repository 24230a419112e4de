"""CPU test: the hand-counted waits of the gfx950 kernels hold in the GENERATED code (tests/isa_lint.py).

Every csrc/*.hip is compiled to ISA text (cached under lib/isa/ by content hash) and every load issued from inline asm is
followed along all paths of the kernel's control-flow graph:

  * no instruction may name a destination register of such a load before a wait has covered it (the register allocator
    copying / re-using an in-flight asm output: the cause of round 2's GPU memory fault);
  * the number of asm waits a load survives is pinned per kernel: a count that drifts (fewer vector-memory operations between
    a load and its `vmcnt(N)` than the author counted) shows up as one more;
  * the kernels with counted waits keep their accumulators in registers (no scratch) where that is the case today.

The walker itself is tested on hand-written ISA snippets that reproduce the three failure shapes.
"""
import os
import re

import pytest

import isa_lint as il

SOURCES = sorted(f for f in os.listdir(il.CSRC) if f.endswith('.hip'))

PINNED_WAITS, NO_SCRATCH, RESIDENT_BUDGET, _family = il.PINNED_WAITS, il.NO_SCRATCH, il.RESIDENT_BUDGET, il.family   # the pins live with the lint (the build enforces them)


@pytest.mark.parametrize('src', SOURCES)
def test_no_in_flight_asm_output_is_touched_and_ring_depths_hold(src):
    res = il.lint_file(os.path.join(il.CSRC, src))
    for kernel, r in res.items():
        fam = _family(kernel)
        assert not r['violations'], '%s: %s' % (kernel, r['violations'][:4])
        assert fam in PINNED_WAITS, 'new kernel with asm loads: add %s to PINNED_WAITS (vm %d, lgkm %d)' % (fam, r['max_waits_vm'], r['max_waits_lgkm'])
        vm, lg = PINNED_WAITS[fam]
        if vm is not None:
            assert r['max_waits_vm'] <= vm, '%s: a look-ahead load now survives %d asm vmcnt waits (pinned %d)' % (kernel, r['max_waits_vm'], vm)
        if lg is not None:
            assert r['max_waits_lgkm'] <= lg, '%s: an LDS read now survives %d asm lgkmcnt waits (pinned %d)' % (kernel, r['max_waits_lgkm'], lg)
        if fam in NO_SCRATCH:
            assert r['meta'].get('private_segment_fixed_size', 0) == 0 and r['meta'].get('vgpr_spill_count', 0) == 0, (kernel, r['meta'])


def test_kernels_dealt_for_a_fixed_residency_keep_their_register_budget():
    seen = set()
    for fam, (src, budget) in RESIDENT_BUDGET.items():
        for kernel, fn in il.parse(il.compile_isa(os.path.join(il.CSRC, src))).items():
            if _family(kernel) != fam:
                continue
            seen.add(fam)
            m = fn['meta']
            assert m.get('vgpr_count', 0) <= budget, (kernel, m)
            assert m.get('private_segment_fixed_size', 0) == 0 and m.get('vgpr_spill_count', 0) == 0, (kernel, m)
    assert seen == set(RESIDENT_BUDGET), seen


# ------------------------------------------------------------------------------------------------ the walker on known shapes
_HEAD = '''\t.text
_Z1kv:
'''
_TAIL = '''\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel _Z1kv
\t.end_amdhsa_kernel
'''


def _lint_text(tmp_path, body):
    p = tmp_path / 'k.s'
    p.write_text(_HEAD + body + _TAIL)
    fn = il.parse(str(p))['_Z1kv']
    out = []
    for idx, counter, dest in il.asm_loads(fn):
        out.append((fn['ins'][idx].text, il.walk(fn, idx, counter, dest)))
    return out


def _asm(*lines):
    return '\t;;#ASMSTART\n' + ''.join('\t%s\n' % ln for ln in lines) + '\t;;#ASMEND\n'


def test_walker_flags_a_dead_look_ahead_load_whose_registers_are_reused(tmp_path):
    """Round 2's variant (i) of the last layer (commit 53fa832): the look-ahead target load of the LAST step is dead, the
    compiler frees its registers at once, and the load lands on the logit accumulators of the last plane."""
    body = (_asm('global_load_dwordx2 v[2:3], v[2:3], off') + _asm('s_waitcnt vmcnt(5)') + '\tv_mov_b32_e32 v2, 0\n'
            '\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read2_b32 v[2:3], v2 offset0:111 offset1:180\n')
    (_, w), = _lint_text(tmp_path, body)
    assert [t.text for t in w['touched']] == ['v_mov_b32_e32 v2, 0']


def test_walker_flags_a_copy_of_an_in_flight_register_and_accepts_the_counted_wait(tmp_path):
    ok = (_asm('global_load_dwordx2 v[62:63], v[2:3], off')
          + ''.join(_asm('buffer_load_dwordx4 v2, s[16:19], s60 offen lds') for _ in range(4))
          + _asm('s_waitcnt vmcnt(5)') + '\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n'
          + _asm('s_waitcnt vmcnt(4)') + '\tv_add_f32_e32 v1, v62, v63\n')
    res = _lint_text(tmp_path, ok)
    load = [w for t, w in res if t.startswith('global_load')][0]
    assert not load['touched'] and load['waits'] == 2          # passes vmcnt(5) (k = 4 < 5), resolved by vmcnt(4)
    bad = ok.replace('\tv_mfma_f32_32x32x16_bf16', '\tv_mov_b32_e32 v70, v62\n\tv_mfma_f32_32x32x16_bf16')
    load = [w for t, w in _lint_text(tmp_path, bad) if t.startswith('global_load')][0]
    assert [t.text for t in load['touched']] == ['v_mov_b32_e32 v70, v62']


def test_walker_sees_a_counted_wait_that_no_longer_covers_its_load(tmp_path):
    """vmcnt(9) written for 8 stores + 1 younger DMA: with 7 stores the older DMA is still among the newest 9."""
    def body(nstores):
        return (_asm('buffer_load_dwordx4 v1, s[0:3], s4 offen lds') + _asm('buffer_load_dwordx4 v1, s[0:3], s5 offen lds')
                + ''.join('\tglobal_store_dwordx4 v[4:5], v[8:11], off\n' for _ in range(nstores))
                + _asm('s_waitcnt vmcnt(9)') + '\ts_barrier\n' + _asm('s_waitcnt vmcnt(0)'))
    first = lambda res: res[0][1]['waits']
    assert first(_lint_text(tmp_path, body(8))) == 1            # covered by the counted wait
    assert first(_lint_text(tmp_path, body(7))) == 2            # survives it: only the closing vmcnt(0) lands it


def test_walker_follows_loop_back_edges_and_scalar_loads_freeze_the_lgkm_count(tmp_path):
    loop = ('.LBB0_1:\n' + _asm('s_waitcnt lgkmcnt(1)') + '\tv_add_f32_e32 v9, v4, v5\n' + _asm('ds_read_b128 v[4:7], v20')
            + _asm('ds_read_b128 v[10:13], v20 offset:16') + '\ts_cbranch_scc1 .LBB0_1\n' + _asm('s_waitcnt lgkmcnt(0)'))
    res = _lint_text(tmp_path, loop)
    assert all(not w['touched'] for _, w in res)                # v[4:7] is waited (all but the newest 1) before it is read
    smem = loop.replace('\ts_cbranch_scc1', '\ts_load_dwordx2 s[0:1], s[2:3], 0x0\n\ts_cbranch_scc1')
    res = _lint_text(tmp_path, smem)                            # SMEM returns out of order: lgkmcnt(1) proves nothing any more
    assert any(w['touched'] for _, w in res)
