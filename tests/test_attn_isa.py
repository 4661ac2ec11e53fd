"""ISA audit of the ring forms of the LLM attention kernel (attn_gqa128_kernel<2, 4, 8> chunks, <1, 4> decode): their K / V tiles arrive by LDS-DMA three tiles deep and are
waited for with COUNTED `s_waitcnt vmcnt(N)`.  Two silent compiler-side regressions would undo that without failing a parity test: a register spill inside the tile loop
(scratch reloads are VMEM: hipcc puts `vmcnt(0)` behind each one and drains the ring) and a compiler-inserted `vmcnt(0)` in the loop (an ordinary global load or a second
LDS object next to the DMAs).  Cross-compiles attn.hip for gfx950 with the Makefile's flags (no GPU needed) and reads the ISA."""
import os, re, shutil, subprocess, tempfile
import pytest
from conftest import ROOT

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')


@pytest.fixture(scope='module')
def attn_isa():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'attn.s')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-honor-nans', '-Wno-unused-result', '-Wno-unused-value',
                            '--cuda-device-only', '-S', '-o', out, os.path.join(ROOT, 'mmduet_amd', 'csrc', 'attn.hip')], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(out).read()


def _kernel(s, mangled):
    m = re.search(r'^' + re.escape(mangled) + r':', s, re.M)
    assert m, f'{mangled} not in the ISA'
    body = s[m.start():]
    end = body.index('s_endpgm')
    meta = body[end:end + 8000]
    return body[:end].split('\n'), meta


def _loop(lines):
    inl = [i for i, l in enumerate(lines) if re.search(r'in Loop: Header=BB\d+_\d+ Depth=1', l)]
    assert inl
    return lines[min(inl) - 1:max(inl) + 2]


def test_chunk_ring_kernel_has_no_spills_and_only_counted_waits(attn_isa):
    lines, meta = _kernel(attn_isa, '_Z18attn_gqa128_kernelILi2ELi4ELi8EEv5AttnP')
    assert not any('scratch_' in l for l in lines), 'the chunk attention spills'
    assert re.search(r'; ScratchSize: 0\b', meta)
    vg = int(re.search(r'; NumVgprs: (\d+)', meta).group(1))
    assert vg <= 256                                            # 8 waves per CU
    loop = _loop(lines)
    assert sum('s_barrier' in l for l in loop) == 1             # one barrier per tile
    assert sum('v_mfma_f32_16x16x32_bf16' in l for l in loop) == 64
    assert sum('global_load_lds_dwordx4' in l for l in loop) == 8          # this wave's 4 pieces, once for either group's issue point
    # every vmcnt wait in the loop is one of seg_barrier's hand-written ones (inside an inline-asm block: vmcnt(4) with a tile in flight, vmcnt(0) at the end of the
    # keys); hipcc's own would sit outside ASMSTART / ASMEND (it once put `vmcnt(0) lgkmcnt(0)` in front of the score MFMAs for the q fragments' loads: every tile)
    waits = [(i, l.strip()) for i, l in enumerate(loop) if 'vmcnt' in l]
    assert len(waits) >= 2
    for i, w in waits:
        assert w in ('s_waitcnt vmcnt(0)', 's_waitcnt vmcnt(4)'), w
        assert 'ASMSTART' in loop[i - 1], (w, loop[i - 1])
    assert not any(re.match(r'\s*global_load_dword', l) for l in loop), 'an ordinary global load inside the ring loop'


def test_decode_ring_kernel_loaders_use_counted_waits(attn_isa):
    lines, meta = _kernel(attn_isa, '_Z18attn_gqa128_kernelILi1ELi4ELi4EEv5AttnP')
    assert not any('scratch_' in l for l in lines), 'the decode attention spills'
    text = '\n'.join(lines)
    for n in (22, 20, 11, 10):                                  # loaders: 11 / 11 / 10 DMA pieces per tile, one or two younger tiles in flight
        assert re.search(rf's_waitcnt vmcnt\({n}\)', text), n
    assert len(re.findall(r'global_load_lds_dwordx4 .* nt', text)) >= 11          # K / V stream with the nt policy


def test_two_slot_forms_unchanged(attn_isa):
    for name in ('_Z18attn_gqa128_kernelILi2ELi2ELi4EEv5AttnP', '_Z18attn_gqa128_kernelILi1ELi2ELi4EEv5AttnP'):
        lines, meta = _kernel(attn_isa, name)
        assert not any('scratch_' in l for l in lines), name


@pytest.mark.parametrize('mangled', ['_Z20attn_d72_ring_kernelILb1EEv5AttnP', '_Z20attn_d72_ring_kernelILb0EEv5AttnP'])
def test_vit_ring_kernel_fits_four_blocks_and_keeps_its_dma_in_flight(attn_isa, mangled):
    """attn_d72_ring_kernel (fp16 / bf16 tower): <= 128 registers and no scratch are what put four blocks on a CU (at 130 it is three; a spill reload is VMEM and drains the
    DMA of the next tile).  vmcnt waits of the tile loop: the hand-written one in front of the barrier, and hipcc's own in front of the FIRST transposing V read of a tile
    (the builtin carries no memory operand; once drained the second half tile needs none) -- a third one would mean an ordinary global load or a spill crept in."""
    lines, meta = _kernel(attn_isa, mangled)
    assert not any('scratch_' in l for l in lines), 'the ViT ring attention spills'
    assert re.search(r'; ScratchSize: 0\b', meta)
    assert int(re.search(r'; NumVgprs: (\d+)', meta).group(1)) <= 128
    hdr = [i for i, l in enumerate(lines) if 'This Inner Loop Header: Depth=1' in l]
    inl = [i for i, l in enumerate(lines) if re.search(r'in Loop: Header=BB\d+_\d+ Depth=1', l)]
    assert len(hdr) == 1 and inl and max(inl) > hdr[0]
    end = next(i for i in range(max(inl) + 1, len(lines)) if re.match(r'\.LBB\d+_\d+:', lines[i]))          # the label behind the loop's last block
    loop = lines[hdr[0]:end]
    assert sum('s_barrier' in l for l in loop) == 1
    assert sum('global_load_lds_dwordx4' in l for l in loop) == 5          # this wave's five 1 KB blocks of the next tile
    assert sum('v_mfma_f32_16x16x32' in l for l in loop) == 36 and sum('v_mfma_f32_16x16x16' in l for l in loop) == 8
    assert sum('ds_read_b64_tr_b16' in l for l in loop) == 20
    waits = [(i, l.strip()) for i, l in enumerate(loop) if 'vmcnt' in l]
    assert len(waits) == 2 and 'ASMSTART' in loop[waits[0][0] - 1], waits
    first_tr = min(i for i, l in enumerate(loop) if 'ds_read_b64_tr_b16' in l)
    assert waits[0][0] < first_tr and waits[1][0] < first_tr                 # both in front of the tile's first V read: nothing waits on the NEXT tile's DMA behind it
    assert not any(re.match(r'\s*global_load_dword', l) for l in loop), 'an ordinary global load inside the ring loop'
    # the ten transposing V reads of a half tile are issued as one group (sched_barrier fences in the source; not a correctness condition, see below)
    tr = [i for i, l in enumerate(loop) if 'ds_read_b64_tr_b16' in l]
    for grp in (tr[:10], tr[10:]):
        assert not any('v_mfma' in l for l in loop[grp[0]:grp[-1] + 1]), 'an MFMA between the transposing V reads of one half tile'
    # THE correctness condition found in round 5: the 16-deep MFMA that finishes a score tile reads a 32-deep MFMA's result as SrcC; issued directly behind its producer the
    # first row tile of every wave came back wrong and different from run to run.  At least one other MFMA has to sit between producer and consumer (D72_MFMA_PIN in the source).
    mf = [(i, l.strip()) for i, l in enumerate(loop) if 'v_mfma' in l]
    def _regs(tok):
        m = re.match(r'v\[(\d+):(\d+)\]', tok.strip())
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
    checked = 0
    for k, (i, l) in enumerate(mf):
        if 'v_mfma_f32_16x16x16' not in l: continue
        ops_ = [x.strip() for x in l.split(None, 1)[1].split(',')]
        srcc = _regs(ops_[3])
        prev = mf[k - 1][1]
        pdst = _regs(prev.split(None, 1)[1].split(',')[0])
        assert not (srcc & pdst and 'v_mfma_f32_16x16x32' in prev), f'16-deep MFMA directly behind the 32-deep MFMA that produces its SrcC: {prev} -> {l}'
        # round 6 (D72_TAIL_SEPARATE): the tail accumulates onto ZERO in registers of its own -- no 16-deep MFMA reads any MFMA's result, whatever the instruction order
        assert ops_[3].strip() == '0', f'the 16-deep tail MFMA takes a register as SrcC: {l}'
        checked += 1
    assert checked == 8


def test_w1_kernel_has_no_spills_vgpr_form_mfmas_and_only_counted_waits(attn_isa):
    """attn_gqa128_w1_kernel (attn_w1.h): <= 256 registers and NO AGPRs -- beyond 256 hipcc switches every MFMA to the AGPR form and pays an accvgpr move per score;
    no scratch (a reload is VMEM: hipcc waits vmcnt(0) behind it and drains the K / V ring); every vmcnt wait of the kernel is one of the hand-written counted ones."""
    name = '_Z21attn_gqa128_w1_kernelILi2ELi8EEv5AttnP'          # (several s_endpgm: the 2 / 1 / 0 row-tile programs end separately -- cut at the function's end label)
    body = attn_isa[re.search(r'^' + name + r':', attn_isa, re.M).start():]
    end = body.index('.Lfunc_end')
    lines, meta = body[:end].split('\n'), body[end:end + 8000]
    assert not any('scratch_' in l for l in lines), 'the w1 attention spills'
    assert re.search(r'; ScratchSize: 0\b', meta)
    assert int(re.search(r'; NumVgprs: (\d+)', meta).group(1)) <= 256 and int(re.search(r'; NumAgprs: (\d+)', meta).group(1)) == 0
    assert not any('v_accvgpr' in l for l in lines)
    text = [l.strip() for l in lines]
    first_dma = min(i for i, l in enumerate(text) if l.startswith('global_load_lds_dwordx4'))
    own = []
    for i, l in enumerate(text):
        if i > first_dma and 'vmcnt' in l and not l.startswith(';'):
            # behind the first DMA: only the hand-written counted waits (inside ASMSTART / ASMEND) -- and the compiler's own wait for the q fragments' ordinary loads, which are
            # OLDER than the first three tiles' DMAs (vmcnt(N >= 12): the DMAs stay in flight) -- and the epilogue's
            w = re.fullmatch(r's_waitcnt vmcnt\((\d+)\)', l)
            if w and 'ASMSTART' in text[i - 1]:
                assert int(w.group(1)) in (0, 4, 8), l
            elif not any(x.startswith('global_store') or x.startswith('buffer_store') for x in text[first_dma:i]):
                own.append(i)
                assert any(x.startswith('global_load_lds_dwordx4') for x in text[i:i + 40]), 'a compiler-inserted vmcnt wait that is not the q fragments\' (tiles 1, 2 are staged right behind that one)'
    assert len(own) <= 2                                        # the 2- and 1-row-tile programs' q loads
    loops = [i for i, l in enumerate(lines) if 'This Inner Loop Header: Depth=1' in l]
    assert len(loops) >= 3                                      # the 2 / 1 / 0 row-tile programs
    assert sum('s_barrier' in l for l in lines) >= 6            # ... each with its prologue barrier and one per tile


def test_chunk_kernel_contiguous_form_has_no_spills_and_only_counted_waits(attn_isa):
    """attn_gqa128_chunk_kernel (attn_chunk.h): the wave program of the <2, 4, 8> form inside a segment loop -- same audit: no scratch, <= 256 registers, one barrier per tile in the
    tile loop plus the one between segments, every vmcnt wait of the tile loop hand-written."""
    name = '_Z24attn_gqa128_chunk_kernel5AttnP8ChunkTab'
    body = attn_isa[re.search(r'^' + name + r':', attn_isa, re.M).start():]
    end = body.index('.Lfunc_end')
    lines, meta = body[:end].split('\n'), body[end:end + 8000]
    assert not any('scratch_' in l for l in lines), 'the chunk attention spills'
    assert re.search(r'; ScratchSize: 0\b', meta) and int(re.search(r'; NumVgprs: (\d+)', meta).group(1)) <= 256
    inner = [i for i, l in enumerate(lines) if re.search(r'in Loop: Header=BB\d+_\d+ Depth=2', l)]
    assert inner, 'tile loop (depth 2: inside the segment loop) not found'
    loop = lines[min(inner) - 1:max(inner) + 2]
    assert sum('v_mfma_f32_16x16x32_bf16' in l for l in loop) == 64
    for i, l in enumerate(loop):
        if 'vmcnt' in l and not l.strip().startswith(';'):
            assert l.strip() in ('s_waitcnt vmcnt(0)', 's_waitcnt vmcnt(4)') and 'ASMSTART' in loop[i - 1], l
    assert not any(re.match(r'\s*global_load_dword', l) for l in loop), 'an ordinary global load inside the ring loop'


@pytest.mark.parametrize('mangled', ['_Z24attn_gqa128_chunk_kernel5AttnP8ChunkTab', '_Z18attn_gqa128_kernelILi2ELi4ELi8EEv5AttnP', '_Z18attn_gqa128_kernelILi1ELi4ELi4EEv5AttnP',
                                     '_Z18attn_gqa128_kernelILi1ELi2ELi4EEv5AttnP', '_Z18attn_gqa128_kernelILi2ELi2ELi4EEv5AttnP', '_Z21attn_gqa128_w1_kernelILi2ELi8EEv5AttnP',
                                     '_Z20attn_d72_ring_kernelILb1EEv5AttnP', '_Z20attn_d72_ring_kernelILb0EEv5AttnP'])
def test_ring_attention_dmas_take_the_scalar_base_form_and_no_agpr_moves(attn_isa, mangled):
    """Round 5: inside the chunk and decode loops hipcc had re-associated (scalar base + tile offset + lane offset) into 64-bit per-lane addresses -- two v_lshl_add_u64 + two moves
    per LDS-DMA of loops that are bound by the vector issue port; the bases are opaque scalar pairs now and every DMA is `global_load_lds_dwordx4 v_off32, s[base:base+1]`.  And the
    decode form, compiled for one wave per SIMD, kept its accumulators in AGPRs (a v_accvgpr move per score read back): all attention forms are bounded to <= 256 registers."""
    body = attn_isa[re.search(r'^' + mangled + r':', attn_isa, re.M).start():]
    lines = body[:body.index('.Lfunc_end')].split('\n')
    dma = [l.strip() for l in lines if 'global_load_lds_dwordx4' in l]
    form = r'global_load_lds_dwordx4 v\d+, (s\[|vcc)'          # (the scalar pair may be vcc)
    assert dma and all(re.match(form, l) for l in dma), [l for l in dma if not re.match(form, l)][:3]
    assert not any('v_accvgpr' in l for l in lines)
