"""Output records of the stream driver and the pre-extracted-feature file format (host side).

* `result_record` builds the JSONL line `test/inference.py:357-359` writes (question_id, model_response_list,
  video_duration, debug_data rounded to 3 decimals).  The reference's evaluator (`test/evaluate.py:319-325,379-382`) still
  reads the keys of the deprecated loop (`video_time`, `relevance_score` as a [p0, p1] list); `legacy_keys=True` adds them next to
  the live keys so `test/evaluate.py --func grounding / qvh_highlight` runs on this build's output unchanged.
* `smooth_pred_list` / `normalize_pred_list` restate the two score post-processing helpers of `test/evaluate.py:166-173`
  (window mean, min-max normalisation) for callers that want grounding curves without importing the evaluator.
* `grounding_sweep` / `qvh_saliency_scores` are the evaluator's per-stream feeds (`test/evaluate.py:373-392` threshold sweep
  + IoU, `:319-330` two-second clip sums) with their helpers `is_time_in_span` (:102-106), `keep_longest_true_span`
  (:109-126), `calculate_iou` (:129-137), `calculate_iou_span` (:140-145), `qvh_to_charades_format` (:148-163).  Pinned by
  tests/golden/eval_feed.json (outputs of the reference's functions, tests/golden/make_eval_golden.py).
* `save_frame_features` / `load_frame_features`: the per-video `.pt` of `[T, tokens, C]` that `data/utils.py:99-117`
  (`distributed_encode`) writes, so Phase A (vision) can be cached and Phase B fed from disk.
"""
import json
import numpy as np
import torch


def round_numbers(data, n):
    if isinstance(data, list):
        return [round_numbers(d, n) for d in data]
    if isinstance(data, dict):
        return {k: round_numbers(v, n) for k, v in data.items()}
    if isinstance(data, float):
        return round(data, n)
    return data


def result_record(question_id, model_response_list, video_duration, debug_data_list, legacy_keys=True, ndigits=3, evaluator_format=False):
    """One output line of test/inference.py:357-359.

    The live loop writes `time` and float scores (test/inference.py:243-246,286) while the reference's evaluator still
    reads the deprecated loop's shape (`video_time`, `relevance_score` = [p0, p1]; test/inference.py:136,
    test/evaluate.py:319-325,379-382) and cannot consume the live shape.  `legacy_keys` ADDS `video_time` and
    `*_score_pair` next to the live keys (a superset, safe for live consumers); `evaluator_format=True` instead writes the
    deprecated shape itself, so `test/evaluate.py --func grounding|qvh_highlight` runs on the file unchanged."""
    debug = []
    for d in debug_data_list:
        e = dict(d)
        inf, rel = d['informative_score'], d['relevance_score']
        if evaluator_format:
            e = {k: v for k, v in d.items() if k not in ('time', 'informative_score', 'relevance_score')}
            e.update(video_time=d['time'], informative_score=[1.0 - inf, inf], relevance_score=[1.0 - rel, rel])
        elif legacy_keys:
            e['video_time'] = d['time']
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
    """test/evaluate.py:170-173: min-max normalisation.  A constant list gives NaN for every entry, as the evaluator's
    np.float64 arithmetic does (0/0 -> nan with a warning, no exception): every threshold test on it is then False."""
    lo, hi = min(pred_list), max(pred_list)
    span = hi - lo
    return [(p - lo) / span if span != 0 else float('nan') for p in pred_list]


def is_time_in_span(time, spans):
    """test/evaluate.py:102-106: closed-interval membership in any [start, end] span."""
    return any(s[0] <= time <= s[1] for s in spans)


def keep_longest_true_span(boolean_list):
    """test/evaluate.py:109-126: keep only the first longest run of True.  Returns (mask, run length)."""
    best_len, best_at, run = 0, -1, 0
    for i, v in enumerate(boolean_list):
        run = run + 1 if v else 0
        if run > best_len:
            best_len, best_at = run, i - run + 1
    mask = [False] * len(boolean_list)
    if best_at >= 0:
        mask[best_at:best_at + best_len] = [True] * best_len
    return mask, best_len


def calculate_iou(pred_scores, gold_scores, threshold, pred_get_largest_span=False):
    """test/evaluate.py:129-137: frame-level IoU of (score >= threshold) against the gold mask."""
    assert len(pred_scores) == len(gold_scores)
    pred = [p >= threshold for p in pred_scores]
    if pred_get_largest_span:
        pred, _ = keep_longest_true_span(pred)
    inter = sum(1 for p, g in zip(pred, gold_scores) if p and g)
    union = sum(1 for p, g in zip(pred, gold_scores) if p or g)
    return inter / union if union else 0


def calculate_iou_span(pred_span, gold_span):
    """test/evaluate.py:140-145: IoU of two inclusive [start, end] spans (the +1 is the reference's)."""
    inter = max(0, min(pred_span[1], gold_span[1]) - max(pred_span[0], gold_span[0]) + 1)
    union = max(pred_span[1], gold_span[1]) - min(pred_span[0], gold_span[0]) + 1
    return inter / union if union else 0


def qvh_to_charades_format(example):
    """test/evaluate.py:148-163: QVHighlights saliency annotation -> [[start_sec, end_sec], ...] on `example['timestamps']`.
    The quirks are the reference's: a run of clips whose best saliency is >= 4 is closed by the first clip below 4 as the
    zero-length span [2*clip, 2*clip] of THAT clip; only a run still open at the end keeps its start."""
    spans, open_at, clip = [], None, None
    for sal, clip in zip(example['answer']['saliency_scores'], example['answer']['relevant_clip_ids']):
        if max(sal) >= 4:
            if open_at is None:
                open_at = clip
        elif open_at is not None:
            spans.append([clip * 2, clip * 2])
            open_at = None
    if open_at is not None:
        spans.append([open_at * 2, clip * 2 + 2])
    example['timestamps'] = spans
    return example


GROUNDING_THRESHOLDS = tuple(np.arange(0.30, 0.71, 0.02))          # test/evaluate.py:373


def _relevance_series(debug_data):
    """(times, p1 scores) of one record in any of the three shapes result_record writes (missing score -> 0, :381-384)."""
    times, scores = [], []
    for e in debug_data:
        times.append(e['video_time'] if 'video_time' in e else e['time'])
        r = e.get('relevance_score', 0)
        scores.append(r[1] if isinstance(r, (list, tuple)) else r)
    return times, scores


def grounding_sweep(debug_data, gold_timestamps, smooth_window_size, thresholds=GROUNDING_THRESHOLDS):
    """Per-stream body of `--func grounding` (test/evaluate.py:376-392): smooth, min-max normalise, and score the thresholded
    mask against the gold spans at every threshold.  Returns {threshold: IoU}."""
    times, scores = _relevance_series(debug_data)
    pred = normalize_pred_list(smooth_pred_list(scores, smooth_window_size))
    gold = [is_time_in_span(t, gold_timestamps) for t in times]
    return {th: calculate_iou(pred, gold, th) for th in thresholds}


def qvh_saliency_scores(debug_data, smooth_window_size):
    """Per-stream body of `--func qvh_highlight` (test/evaluate.py:319-330): smoothed relevance summed over 2-second clips."""
    times, scores = _relevance_series(debug_data)
    per_clip = int(2 / (times[1] - times[0]))
    sm = smooth_pred_list(scores, smooth_window_size)
    return [sum(sm[i:i + per_clip]) for i in range(0, len(sm), per_clip)]


from .features import save_frame_features, load_frame_features      # moved to mmduet_amd/features.py (kept importable from here)  # noqa: E402,F401
