import json
import torch
from mmduet_amd.results import result_record, smooth_pred_list, normalize_pred_list, save_frame_features, load_frame_features


def test_result_record_live_and_legacy_keys():
    dbg = [{'time': 0.0, 'informative_score': 0.12345, 'relevance_score': 0.9}, {'time': 1.0, 'informative_score': 0.5, 'relevance_score': 0.25}]
    rec = result_record('q1', [{'time': 0.0, 'content': 'hi', 'role': 'user'}], 2.0, dbg)
    assert rec['debug_data'][0]['informative_score'] == 0.123 and rec['debug_data'][0]['video_time'] == 0.0
    assert rec['debug_data'][1]['relevance_score_pair'] == [0.75, 0.25]          # the evaluator reads ['relevance_score'][1]-style pairs
    json.dumps(rec)
    assert 'video_time' not in result_record('q', [], 1.0, dbg, legacy_keys=False)['debug_data'][0]


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
