#!/usr/bin/env python
"""Effective clock and MFMA-busy share of cycles per kernel, from ONE rocprofv3 pass `--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [...]` (rocpd database):

    python tools/pmc_clock.py db [kernel-substring] > profiles/rNN_pmc_clock.json

Per dispatch the pass holds the counters AND the kernel's duration, so cycles and wall belong to the same execution:
    effective_clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / duration          (the counter is summed over the XCDs)
    mfma_busy_frac_of_cycles = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
(MI355X guide, "DVFS give-back": a kernel's FLOP fraction of the 2.4 GHz peak mixes the schedule's share of cycles with the clock the chip granted; these two split it.)
Kernel classes as bench.py's `mmd_prof_*` (tools/pmc_traffic.py)."""
import json, re, sqlite3, sys, collections

XCDS, SIMDS = 8, 1024
CLASSES = {'gemm_tile': ('gemm_ringx_kernel', 'gemm_ring256_kernel', 'gemm_big_kernel', 'gemm_tile_kernel'), 'gemm_skinny': ('gemm_gemv16_kernel', 'gemm_skinny_kernel', 'gemm_stream_kernel'),
           'attn_llm': ('attn_gqa128_kernel', 'attn_gqa128_chunk_kernel', 'attn_gqa128_w1_kernel'), 'attn_vit': ('attn_d72_ring_kernel', 'attn_rowmajor_kernel')}


def load(path, sub=''):
    db = sqlite3.connect(path); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
    cn = 'counter_name' if 'counter_name' in cols else [c for c in cols if 'counter' in c and 'name' in c][0]
    vn = 'value' if 'value' in cols else [c for c in cols if 'value' in c][0]
    did = 'dispatch_id' if 'dispatch_id' in cols else ('id' if 'id' in cols else None)
    disp = collections.OrderedDict()          # dispatch -> dict(name, counters, dur)
    if 'start' in cols and 'end' in cols and did:
        for d, k, c, v, s, e in cur.execute(f"select {did}, {kn}, {cn}, {vn}, start, end from counters_collection"):
            r = disp.setdefault(d, dict(name=k, c=collections.defaultdict(float), dur=e - s)); r['c'][c] += v
    else:
        kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        kd = 'dispatch_id' if 'dispatch_id' in kcols else 'id'
        durs = {d: (e - s) for d, s, e in cur.execute(f"select {kd}, start, end from kernels")}
        for d, k, c, v in cur.execute(f"select {did}, {kn}, {cn}, {vn} from counters_collection"):
            r = disp.setdefault(d, dict(name=k, c=collections.defaultdict(float), dur=durs.get(d))); r['c'][c] += v
    return [r for r in disp.values() if sub in r['name'] and r['dur']]


def summarise(rows):
    g = sum(r['c'].get('GRBM_GUI_ACTIVE', 0.0) for r in rows); m = sum(r['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for r in rows); t = sum(r['dur'] for r in rows)
    out = dict(dispatches=len(rows), avg_us=round(t / len(rows) / 1e3, 2))
    if g > 0:
        cyc = g / XCDS
        out.update(effective_clock_ghz=round(cyc / t, 3), mfma_busy_frac_of_cycles=round(m / SIMDS / cyc, 4))
        # GRBM_GUI_ACTIVE is sampled around the dispatch, a few microseconds wider than the kernel runs: for kernels of tens of microseconds the quotient reads above the chip's
        # 2.4 GHz (the window, not the clock); it is a clock only where the kernel is long against that margin
        out['clock_reliable'] = bool(t / len(rows) >= 100e3)
        w, wa = sum(r['c'].get('SQ_WAVE_CYCLES', 0.0) for r in rows), sum(r['c'].get('SQ_WAIT_ANY', 0.0) for r in rows)
        if w > 0:
            out['wait_any_frac_of_wave_cycles'] = round(wa / w, 3)
    return out


def main(path, sub=''):
    rows = load(path, sub)
    short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '')[:70]
    res = {'_note': 'tools/pmc_clock.py: one rocprofv3 --kernel-trace --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...); clock = GRBM_GUI_ACTIVE / 8 XCDs / duration of the same '
                    'dispatches, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / those cycles; peak figures (2.5 PF) assume 2.4 GHz'}
    for cls, pre in CLASSES.items():
        sel = [r for r in rows if short(r['name']).startswith(pre)]
        if sel:
            res[cls] = summarise(sel)
    by = collections.defaultdict(list)
    for r in rows:
        by[short(r['name'])].append(r)
    res['kernels'] = {k: summarise(v) for k, v in sorted(by.items(), key=lambda kv: -sum(r['dur'] for r in kv[1]))[:40]}
    json.dump(res, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
