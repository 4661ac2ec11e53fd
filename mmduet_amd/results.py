"""Output records of the stream driver and the pre-extracted-feature file format (host side).

* `result_record` builds the JSONL line `test/inference.py:357-359` writes (question_id, model_response_list,
  video_duration, debug_data rounded to 3 decimals).  The reference's evaluator (`test/evaluate.py:319-325,379-382`) still
  reads the keys of the deprecated loop (`video_time`, `relevance_score` as a [p0, p1] list); `legacy_keys=True` adds them next to
  the live keys so `test/evaluate.py --func grounding / qvh_highlight` runs on this build's output unchanged.
* `smooth_pred_list` / `normalize_pred_list` restate the two score post-processing helpers of `test/evaluate.py:166-173`
  (window mean, min-max normalisation) for callers that want grounding curves without importing the evaluator.
* `save_frame_features` / `load_frame_features`: the per-video `.pt` of `[T, tokens, C]` that `data/utils.py:99-117`
  (`distributed_encode`) writes, so Phase A (vision) can be cached and Phase B fed from disk.
"""
import json
import torch


def round_numbers(data, n):
    if isinstance(data, list):
        return [round_numbers(d, n) for d in data]
    if isinstance(data, dict):
        return {k: round_numbers(v, n) for k, v in data.items()}
    if isinstance(data, float):
        return round(data, n)
    return data


def result_record(question_id, model_response_list, video_duration, debug_data_list, legacy_keys=True, ndigits=3):
    debug = []
    for d in debug_data_list:
        e = dict(d)
        if legacy_keys:
            e['video_time'] = d['time']
            inf, rel = d['informative_score'], d['relevance_score']
            e['informative_score_pair'] = [1.0 - inf, inf]
            e['relevance_score_pair'] = [1.0 - rel, rel]
        debug.append(e)
    return {'question_id': question_id, 'model_response_list': model_response_list, 'video_duration': video_duration,
            'debug_data': round_numbers(debug, ndigits)}


def write_jsonl(path, records):
    with open(path, 'w') as f:
        for r in records:
            f.write(json.dumps(r) + '\n')


def smooth_pred_list(pred_list, window_size):
    """test/evaluate.py:166-167: centred window mean (window shrinks at the borders)."""
    n = len(pred_list)
    return [sum(pred_list[max(0, i - window_size):min(n, i + window_size + 1)]) / len(pred_list[max(0, i - window_size):min(n, i + window_size + 1)])
            for i in range(n)]


def normalize_pred_list(pred_list):
    """test/evaluate.py:170-173: min-max normalisation."""
    lo, hi = min(pred_list), max(pred_list)
    return [(p - lo) / (hi - lo) if hi > lo else 0.0 for p in pred_list]


def save_frame_features(path, frame_embeds, to_bf16=True):
    """frame_embeds: [T*tokens, C] or [T, tokens, C] device tensor from `model.visual_embed`."""
    t = frame_embeds.detach().to('cpu')
    torch.save(t.to(torch.bfloat16) if to_bf16 else t, path)


def load_frame_features(path, tokens_per_frame, device='cuda', dtype=torch.bfloat16):
    t = torch.load(path, map_location='cpu')
    return t.reshape(-1, tokens_per_frame, t.shape[-1]).to(device=device, dtype=dtype)
