"""Library-wide ISA audit: no kernel of libmmduet_hip spills.  A scratch reload is a VMEM load -- hipcc waits for it with `vmcnt(0)`, which also drains whatever LDS-DMA ring the kernel
keeps in flight, and a spilling kernel does not fail any parity test.  Cross-compiles every .hip file for gfx950 with the Makefile's flags (no GPU needed; gemm.hip takes about a minute)
and reads the assembly: no `scratch_` instruction anywhere, and a non-empty private segment only where it is known and harmless (the decode attention parks ten SGPRs in VGPR lanes
around its tile loop; the slot reserved for that VGPR is never touched)."""
import os, re, shutil, subprocess, tempfile
import pytest
from conftest import ROOT

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
CSRC = os.path.join(ROOT, 'mmduet_amd', 'csrc')
KNOWN_PRIVATE = {'_Z18attn_gqa128_kernelILi1ELi4ELi4EEv5AttnP'}


@pytest.mark.parametrize('src', ['attn.hip', 'ops.hip', 'gemm.hip', 'model.hip', 'comm.hip'])
def test_no_kernel_spills(src):
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-Wno-unused-value', '-I' + os.path.join(ROOT, 'include')]
    if src == 'attn.hip':
        flags.append('-fno-honor-nans')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        r = subprocess.run([HIPCC, *flags, '--cuda-device-only', '-S', '-o', out, os.path.join(CSRC, src)], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        s = open(out).read()
    spills = [l.strip() for l in s.split('\n') if re.match(r'\s*scratch_(load|store)', l)]
    assert not spills, f'{src}: {len(spills)} scratch instructions, e.g. {spills[:3]}'
    for name, size in re.findall(r'\.set (\w+)\.private_seg_size, (\d+)', s):
        assert int(size) == 0 or name in KNOWN_PRIVATE, f'{src}: {name} reserves {size} bytes of private memory'
