import json
import torch
import math, os
import pytest
from mmduet_amd.results import (result_record, smooth_pred_list, normalize_pred_list, save_frame_features, load_frame_features,
                                is_time_in_span, keep_longest_true_span, calculate_iou, calculate_iou_span, qvh_to_charades_format,
                                grounding_sweep, qvh_saliency_scores, GROUNDING_THRESHOLDS)

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_feed.json')))


def _same(a, b, tol=1e-12):
    return (math.isnan(a) and math.isnan(b)) or abs(a - b) <= tol


def test_result_record_live_and_legacy_keys():
    dbg = [{'time': 0.0, 'informative_score': 0.12345, 'relevance_score': 0.9}, {'time': 1.0, 'informative_score': 0.5, 'relevance_score': 0.25}]
    rec = result_record('q1', [{'time': 0.0, 'content': 'hi', 'role': 'user'}], 2.0, dbg)
    assert rec['debug_data'][0]['informative_score'] == 0.123 and rec['debug_data'][0]['video_time'] == 0.0
    assert rec['debug_data'][1]['relevance_score_pair'] == [0.75, 0.25]          # the evaluator reads ['relevance_score'][1]-style pairs
    json.dumps(rec)
    assert 'video_time' not in result_record('q', [], 1.0, dbg, legacy_keys=False)['debug_data'][0]
    ev = result_record('q', [], 1.0, dbg, evaluator_format=True)['debug_data']     # the shape test/evaluate.py:319-325 reads
    assert ev[1]['relevance_score'] == [0.75, 0.25] and ev[1]['video_time'] == 1.0 and 'time' not in ev[1]


def test_score_postprocessing_matches_reference_formulas():
    p = [0.0, 1.0, 0.0, 1.0, 4.0]
    w = 1   # test/evaluate.py:166-167
    ref = [sum(p[max(0, i - w):i + w + 1]) / len(p[max(0, i - w):i + w + 1]) for i in range(len(p))]
    assert smooth_pred_list(p, w) == ref
    assert normalize_pred_list(p) == [0.0, 0.25, 0.0, 0.25, 1.0]


def test_feature_file_roundtrip(tmp_path):
    x = torch.randn(3 * 4, 8)
    save_frame_features(tmp_path / 'v.pt', x)
    y = load_frame_features(tmp_path / 'v.pt', 4, device='cpu')
    assert y.shape == (3, 4, 8) and torch.allclose(y.float(), x.reshape(3, 4, 8), atol=2e-2)
    # the reference's layout [T, tokens, C] (data/utils.py:114-117), bf16 and fp32, .pt and .npy
    save_frame_features(tmp_path / 'w.pt', x, tokens_per_frame=4, to_bf16=False)
    z = load_frame_features(tmp_path / 'w.pt')
    assert z.dtype == torch.float32 and torch.equal(z, x.reshape(3, 4, 8))
    import numpy as np
    np.save(tmp_path / 'u.npy', x.reshape(3, 4, 8).numpy())
    assert torch.equal(load_frame_features(str(tmp_path / 'u.npy')), x.reshape(3, 4, 8))
    with pytest.raises(ValueError):
        load_frame_features(tmp_path / 'v.pt')                 # flat file without tokens_per_frame


@pytest.mark.parametrize('ci', range(len(GOLD['cases'])))
def test_evaluator_feed_matches_reference_outputs(ci):
    """Golden: the reference's own helpers run on seeded streams (tests/golden/make_eval_golden.py)."""
    c = GOLD['cases'][ci]
    assert [is_time_in_span(t, c['spans']) for t in c['times']] == c['gold']
    live = [{'time': t, 'informative_score': 0.0, 'relevance_score': s} for t, s in zip(c['times'], c['scores'])]
    for fmt in ({'legacy_keys': False}, {'legacy_keys': True}, {'evaluator_format': True}):
        dbg = result_record('q', [], c['times'][-1], live, ndigits=6, **fmt)['debug_data']
        for w, g in c['windows'].items():
            w = int(w)
            sm = smooth_pred_list(c['scores'], w)
            assert all(_same(a, b) for a, b in zip(sm, g['smooth']))
            assert all(_same(a, b) for a, b in zip(normalize_pred_list(sm), g['normalized']))
            sweep = grounding_sweep(dbg, c['spans'], w)
            assert len(sweep) == len(GROUNDING_THRESHOLDS) == 21
            for th, iou in sweep.items():
                assert _same(iou, g['iou'][f'{th:.2f}'])
            nm = normalize_pred_list(sm)
            for th, iou in g['iou_longest_span'].items():
                assert _same(calculate_iou(nm, c['gold'], float(th), pred_get_largest_span=True), iou)
            sal = qvh_saliency_scores(dbg, w)
            assert len(sal) == len(g['saliency']) and all(_same(a, b, 1e-9) for a, b in zip(sal, g['saliency']))


def test_span_helpers_match_reference_outputs():
    for e in GOLD['longest_span']:
        mask, n = keep_longest_true_span(e['in'])
        assert [mask, n] == e['out']
    for e in GOLD['span_iou']:
        assert _same(calculate_iou_span(e['pred'], e['gold']), e['iou'])
    for e in GOLD['qvh_to_charades']:
        assert qvh_to_charades_format(json.loads(json.dumps(e['in'])))['timestamps'] == e['timestamps']
