"""The reference's OWN driver classes, unchanged, over the product's Python surface (build container only).

`/root/reference/test/inference.py::LiveInferForBenchmark` (constructor included) and `/root/reference/demo/liveinfer.py::LiveInferForDemo` are imported
with `models` bound to the two lines INTEGRATION.md section 1 prescribes; mmduet_amd runs as shipped except that libmmduet_hip.so is replaced by an
oracle-backed stand-in for its C entry points (tests/cabi_oracle_shim.py -- there is no GPU in the build container).  Their results must equal what
the reference's own model produced for the same streams (tests/golden/cfgA_streams.json, recorded by tests/golden/make_golden.py): per-frame scores,
response token ids and text, response times, the final KV length -- including `remove_assistant_turns`, which works only if a handle held before
generation still denotes the pre-generation context (the product's (arena, length) handles), and the repetition-penalty list.
Self-skips where /root/reference does not exist (the GPU box)."""
import json, os, subprocess, sys
import pytest
from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference checkout (build container only)')


@pytest.fixture(scope='module')
def run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ref_conformance_runner.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('CONFORMANCE_JSON ')][-1]
    return json.loads(line[len('CONFORMANCE_JSON '):]), json.load(open(os.path.join(GOLDEN, 'cfgA_streams.json')))


def _close(a, b, tol=2e-4):
    return abs(a - b) <= tol


def test_reference_benchmark_driver_runs_unchanged_on_the_product_surface(run):
    got, meta = run
    assert set(got['benchmark']) == set(meta['cases'])
    for name, want in meta['cases'].items():
        g = got['benchmark'][name]
        assert g['handle_type'] == 'KVCacheHandle'
        assert len(g['debug_data']) == want['T'] == len(want['debug_data']), name
        for a, b in zip(g['debug_data'], want['debug_data']):
            assert a['time'] == b['time'] and _close(a['informative_score'], b['informative_score']) and _close(a['relevance_score'], b['relevance_score']), (name, a, b)
        assert g['generated'] == want['generated'], name                               # response token ids
        assert g['responses'] == want['responses'], name                               # roles, times, decoded text
        assert g['final_kv_len'] == want['final_kv_len'], (name, g['final_kv_len'], want['final_kv_len'])
        assert g['penalty_ids'] == want['penalty_ids'], name
    # the cases cover both keep / remove modes and at least one response each way
    assert any(c['opts'].get('remove_assistant_turns') and c['n_responses'] > 0 for c in meta['cases'].values())
    assert any(not c['opts'].get('remove_assistant_turns') and c['n_responses'] > 0 for c in meta['cases'].values())


def test_reference_demo_driver_input_one_frame(run):
    """LiveInferForDemo.input_one_frame / encode_given_query (demo/liveinfer.py:60-105): frame by frame, the same scores, responses and context."""
    got, meta = run
    for name, want in meta['cases'].items():
        g = got['demo'][name]
        assert len(g['rows']) == want['T'], name
        for i, (row, b) in enumerate(zip(g['rows'], want['debug_data'])):
            assert row['frame_idx'] == i + 1 and row['time'] == round(b['time'], 1)
            assert _close(row['informative_score'], b['informative_score']) and _close(row['relevance_score'], b['relevance_score']), (name, i)
        assert g['generated'] == want['generated'], name
        texts = [r['response'] for r in g['rows'] if r['response'] is not None]
        assert texts == [r['content'] for r in want['responses'] if r['role'] == 'assistant'], name
        assert g['final_kv_len'] == want['final_kv_len'], name


def test_the_product_layer_really_ran(run):
    """The shim counts the C entry points the product's Python layer called: arenas created, steps issued, O(1) truncates (continuing from an older handle)."""
    calls = run[0]['calls']
    for k in ('mmd_create', 'mmd_load_tensor', 'mmd_preprocess_frames', 'mmd_vit_encode', 'mmd_embed_tokens', 'mmd_llm_step', 'mmd_video_heads', 'mmd_lm_head', 'mmd_stream_create', 'mmd_kv_truncate'):
        assert calls.get(k, 0) > 0, (k, calls)
