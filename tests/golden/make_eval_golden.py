#!/usr/bin/env python
"""Generate tests/golden/eval_feed.json: outputs of the reference's score post-processing helpers on seeded inputs.

Runs HERE only (needs /root/reference).  test/evaluate.py cannot be imported as a module (relative imports of the
qvh / dvc evaluators and heavyweight top-level imports), so the pure helper functions are pulled out of its AST and
executed in a namespace that holds numpy -- their code is run, never stored: only inputs and outputs are written.
"""
import ast, json, os, random
import numpy as np

REF = '/root/reference/test/evaluate.py'
WANT = {'is_time_in_span', 'keep_longest_true_span', 'calculate_iou', 'calculate_iou_span', 'qvh_to_charades_format',
        'smooth_pred_list', 'normalize_pred_list'}


def reference_helpers():
    tree = ast.parse(open(REF).read())
    ns = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANT:
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, 'exec'), ns)
    assert WANT <= set(ns), WANT - set(ns)
    return ns


def main():
    ref = reference_helpers()
    rng = random.Random(1234)
    cases = []
    for ci in range(6):
        T = rng.choice([7, 20, 33, 64, 120])
        fps = rng.choice([0.5, 1.0, 2.0])
        times = [i / fps for i in range(T)]
        scores = [round(rng.random(), 3) for _ in range(T)]
        spans = sorted([sorted([round(rng.uniform(0, times[-1]), 1), round(rng.uniform(0, times[-1]), 1)]) for _ in range(rng.choice([1, 2, 3]))])
        gold = [ref['is_time_in_span'](t, spans) for t in times]
        per_window = {}
        for w in (0, 1, 4, 14):
            sm_np = ref['smooth_pred_list'](scores, w)            # np.float64 items, as the evaluator passes them on
            with np.errstate(all='ignore'):
                nm = [float(x) for x in ref['normalize_pred_list'](sm_np)]      # constant list -> NaN (0/0 on np.float64), not an exception
            sm = [float(x) for x in sm_np]
            ious = {f'{th:.2f}': ref['calculate_iou'](nm, gold, th) for th in np.arange(0.30, 0.71, 0.02)}
            ious_span = {f'{th:.2f}': ref['calculate_iou'](nm, gold, th, pred_get_largest_span=True) for th in (0.3, 0.5, 0.7)}
            two = int(2 / (times[1] - times[0]))
            sal = [float(sum(sm[i:i + two])) for i in range(0, len(sm), two)]
            per_window[str(w)] = {'smooth': sm, 'normalized': nm, 'iou': ious, 'iou_longest_span': ious_span, 'saliency': sal}
        cases.append({'times': times, 'scores': scores, 'spans': spans, 'gold': gold, 'windows': per_window})
    bools = [[rng.random() < 0.5 for _ in range(rng.choice([1, 5, 17]))] for _ in range(8)] + [[False] * 4, [True] * 3, []]
    longest = [{'in': b, 'out': list(ref['keep_longest_true_span'](list(b)))} for b in bools]
    span_iou = []
    for _ in range(10):
        a = sorted([rng.uniform(0, 50), rng.uniform(0, 50)]); b = sorted([rng.uniform(0, 50), rng.uniform(0, 50)])
        span_iou.append({'pred': a, 'gold': b, 'iou': ref['calculate_iou_span'](a, b)})
    qvh = []
    for _ in range(6):
        n = rng.choice([1, 3, 8, 15])
        ids = sorted(rng.sample(range(0, 75), n))
        sal = [[rng.randint(0, 4) for _ in range(3)] for _ in range(n)]
        ex = {'answer': {'saliency_scores': sal, 'relevant_clip_ids': ids}}
        out = ref['qvh_to_charades_format'](json.loads(json.dumps(ex)))
        qvh.append({'in': ex, 'timestamps': out['timestamps']})
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'eval_feed.json')
    json.dump({'cases': cases, 'longest_span': longest, 'span_iou': span_iou, 'qvh_to_charades': qvh}, open(out_path, 'w'))
    print('wrote', out_path, os.path.getsize(out_path), 'bytes')


if __name__ == '__main__':
    main()
