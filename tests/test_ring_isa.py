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
