"""ISA audit of the ring GEMM: the counted `s_waitcnt vmcnt(N)` at a tile's start allow for EXACTLY the output stores its predecessor left in flight
(NST = 16 per wave and tile, 8 with the SwiGLU epilogue; masked lanes store to a dump slot so the count never depends on the tile).  A change that makes hipcc emit
one store more or fewer would turn those waits into under-waits (stale LDS reads) without failing to compile.  This test cross-compiles one instantiation per
epilogue for gfx950 (no GPU needed) and counts the stores, spills and MFMAs in the ISA."""
import os, re, shutil, subprocess, tempfile
import pytest
from conftest import ROOT

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')


@pytest.mark.parametrize('epi,stores', [(0, 16), (1, 16), (2, 16), (3, 16), (4, 8)], ids=['none', 'gelu_tanh', 'gelu_erf', 'resid', 'swiglu'])
def test_ring_kernel_store_count_and_no_spills(epi, stores):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'p.s')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'mmduet_amd', 'csrc'), f'-DPROBE_EPI={epi}',
                            '--cuda-device-only', '-S', '-o', out, os.path.join(ROOT, 'tools', 'probes', 'ring_probe.hip')], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        s = open(out).read()
    # 32 fp32 slab stores of the split-K exit (8 rows x 4 tiles) + the epilogue's NST
    assert len(re.findall(r'\bglobal_store_dwordx4\b', s)) == 32 + stores
    assert not re.search(r'\bscratch_(load|store)', s), 'the ring kernel spills'
    m = re.search(r'\.vgpr_spill_count:\s*(\d+)', s)
    assert m and int(m.group(1)) == 0
    # 6 step bodies (pend pair, steady pair, tail pair) of 32 MFMAs
    assert len(re.findall(r'\bv_mfma_f32_16x16x32_bf16\b', s)) == 192


@pytest.mark.parametrize('ns,nb', [(3, 3), (4, 2)])
@pytest.mark.parametrize('epi,stores', [(0, 16), (3, 16), (4, 8)], ids=['none', 'resid', 'swiglu'])
def test_ringw_kernel_store_count_loads_and_no_spills(epi, stores, ns, nb):
    """gemm_ringw_kernel (W fragments by inline-asm global loads, X-only LDS ring): the hand-counted waits assume exactly 4 W loads + 2 X DMAs per step and wave and NST
    output stores per tile; no spill (a scratch reload would be waited for with vmcnt(0) and drain the ring), no compiler-inserted vmcnt wait inside the step bodies."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'p.s')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'mmduet_amd', 'csrc'), f'-DPROBE_EPI={epi}', f'-DPROBE_NS={ns}',
                            f'-DPROBE_NB={nb}', '--cuda-device-only', '-S', '-o', out, os.path.join(ROOT, 'tools', 'probes', 'ringw_probe.hip')], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        s = open(out).read()
    assert len(re.findall(r'\bglobal_store_dwordx4\b', s)) == 32 + (stores if epi == 3 else 2 * stores)          # (both store forms are compiled: lane-adjacent and the direct A/B form; the residual epilogue has the adjacent form only)
    assert not re.search(r'\bscratch_(load|store)', s), 'the ringw kernel spills'
    m = re.search(r'\.vgpr_spill_count:\s*(\d+)', s)
    assert m and int(m.group(1)) == 0
    per = {(3, 3): 3, (4, 2): 4}[(ns, nb)]
    n_mfma = len(re.findall(r'\bv_mfma_f32_16x16x32_bf16\b', s))
    assert n_mfma == 3 * per * 32, n_mfma                  # pend period + steady period + tail period, 32 MFMAs per step
    # every step body: 4 W loads per step (prologue: 4 x LA more per prologue site), and every vmcnt wait is one of the hand-written counted forms
    la, issue = nb - 1, 6
    vm_step = (la - 1) * issue if ns == la + 1 else 2 + (la - 1) * issue
    allowed = {0, 1, 2, vm_step, issue, vm_step + stores, 4 * la + 2 * (ns - 1), 4 * la + 2 * (ns - 1) + stores, 4 * la + 2 * ns}
    waits = {int(x) for x in re.findall(r's_waitcnt vmcnt\((\d+)\)', s)}
    assert waits <= allowed, (waits, allowed)


def _regs(tok):
    """v[a:b] / vN -> set of VGPR numbers"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def _mentions(line):
    out = set()
    for tok in re.findall(r'v\[\d+:\d+\]|\bv\d+\b', line):
        out |= _regs(tok)
    return out


VMEM = re.compile(r'^\s*(global_load|global_store|buffer_load|buffer_store|global_atomic|scratch_)')


@pytest.mark.parametrize('epi', [0, 1, 3], ids=['none', 'gelu_tanh', 'resid'])
def test_ringw_asm_load_destinations_are_untouched_until_their_counted_wait(epi):
    """The W fragments, the bias quads and the residual pieces of gemm_ringw_kernel arrive by inline-asm loads hipcc knows nothing about: a register copy (phi, spill,
    re-allocation) placed between such a load and the counted `s_waitcnt vmcnt(N)` that covers it would copy a register the load has not written yet -- silently, and
    only when the memory system is slow.  Walk the ISA: after every asm load, no instruction may mention its destination registers until a vmcnt wait whose count is
    at most the number of vector-memory operations issued since (vmcnt retires in issue order).  Straight-line walk in text order; the steady K loop is also walked
    around its back edge."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'p.s')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'mmduet_amd', 'csrc'), f'-DPROBE_EPI={epi}', '-DPROBE_NS=3',
                            '-DPROBE_NB=3', '--cuda-device-only', '-S', '-o', out, os.path.join(ROOT, 'tools', 'probes', 'ringw_probe.hip')], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in open(out).read().split('\n')]
    body = [l.split(';')[0].strip() if not l.strip().startswith(';;#') else l.strip() for l in lines]
    body = [l for l in body if l and (not l.startswith('.') or re.match(r'^\.LBB\d+_\d+:', l))]
    # the steady loop: the block with 96 MFMAs that branches back to its own label
    where = {l[:-1]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:$', l.split()[0] if l else '')}
    loop = None
    for i, l in enumerate(body):
        m = re.match(r'^s_cbranch\w+ (\.LBB\d+_\d+)', l)
        if m and m.group(1) in where and where[m.group(1)] < i:
            region = [x for x in body[where[m.group(1)] + 1:i] if not re.match(r'^\.LBB', x)]
            if sum('v_mfma' in x for x in region) == 96:
                loop = region
    assert loop is not None, 'steady loop not found'

    def check(seq, wrap, form):
        n_checked = 0
        for p, l in enumerate(seq):
            m = re.match(r'^(global_load_dwordx[24]) (v\[\d+:\d+\]), (v\[\d+:\d+\]|v\d+), (off|s\[\d+:\d+\])', l)
            if not m or (p > 0 and 'ASMSTART' not in seq[p - 1]) or (m.group(4) == 'off') != (form == 'off'):
                continue
            dest = _regs(m.group(2))
            issued, covered = 0, False
            idxs = list(range(p + 1, len(seq))) + (list(range(0, p)) if wrap else [])
            for q in idxs:
                x = seq[q]
                w = re.search(r's_waitcnt vmcnt\((\d+)\)', x)
                if w and int(w.group(1)) <= issued:
                    covered = True
                    break
                if VMEM.match(x):
                    issued += 1
                    if _mentions(x.split(',', 1)[0]) & dest and x.startswith('global_load'):
                        raise AssertionError(f'destination of `{l}` re-loaded before its wait: `{x}`')
                    continue
                if x.startswith('s_') or x.startswith(';') or 'ASM' in x:
                    continue
                assert not (_mentions(x) & dest), f'`{x}` touches the destination of the in-flight `{l}`'
            if covered:
                n_checked += 1
        return n_checked

    body_keep = [l for l in body]
    assert check(loop, True, 'saddr') == 12                 # steady loop: 3 steps x 4 W fragments (uniform base + lane offset form)
    # epilogue operands (`off` form: bias quads, residual pieces) sit in straight-line code
    assert check([l for l in body_keep if not re.match(r'^\.LBB', l)], False, 'off') >= 4 + (15 if epi == 3 else 0)          # (text order is not execution order: the last residual's wait may be laid out elsewhere)
