"""Host logic of the product stream driver (mmduet_amd/inference.py), exercised on CPU by injecting the oracle model:
must reproduce what the REFERENCE driver + reference model produced (tests/golden/cfgA_streams.json)."""
import pytest
import torch
from helpers import oracle_model, stream_cases, run_stream_case, make_args, tokenizer_for
from mmduet_amd.inference import LiveInferForBenchmark

META = stream_cases()


def _check(d, case, score_tol=2e-5):
    ref = case['debug_data']
    assert len(d.debug_data_list) == len(ref) == case['T']
    for got, exp in zip(d.debug_data_list, ref):
        assert got['time'] == pytest.approx(exp['time'], abs=1e-9)
        assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=score_tol)
        assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=score_tol)
    assert d.response_token_ids == case['generated']
    exp_resp = case['responses']
    assert [(r['role'], r['time']) for r in d.responses] == [(r['role'], pytest.approx(r['time'])) for r in exp_resp]
    assert [r['content'] for r in d.responses] == [r['content'] for r in exp_resp]
    assert len(d.past_key_values) == case['final_kv_len']
    assert [int(x) for x in d.generated_token_ids] == case['penalty_ids']


@pytest.mark.parametrize('name', list(META['cases']))
@pytest.mark.parametrize('k', [1, 3])
def test_stream_matches_reference_driver(name, k):
    model, _, _ = oracle_model('A')
    case = META['cases'][name]
    d = run_stream_case(LiveInferForBenchmark, model, name, case, META, frames_per_forward=k)
    _check(d, case)
    if k > 1 and case['n_responses'] == 0:
        assert d.forward_calls < case['T'] + len(case['conversation']) + 1      # chunking really happened


def test_exactly_one_threshold_required():
    model, _, _ = oracle_model('A')
    tok = tokenizer_for(model.config)
    with pytest.raises(ValueError):
        LiveInferForBenchmark(make_args(), model=model, tokenizer=tok)
    with pytest.raises(ValueError):
        LiveInferForBenchmark(make_args(stream_end_prob_threshold=0.5, stream_end_score_sum_threshold=2.0), model=model, tokenizer=tok)
    with pytest.raises(AssertionError):
        LiveInferForBenchmark(make_args(stream_end_prob_threshold=0.5, bf16=True, fp16=True), model=model, tokenizer=tok)


def test_demo_driver_single_frame_steps():
    """LiveInferForDemo.input_one_frame == the benchmark loop, one frame at a time (demo/liveinfer.py:69-105)."""
    from mmduet_amd.liveinfer import LiveInferForDemo
    from helpers import stream_frames
    name = 'prob_keep_pen'
    case = META['cases'][name]
    model, _, _ = oracle_model('A')
    opts = case['opts']
    args = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12,
                     stream_end_prob_threshold=opts['stream_end_prob_threshold'], score_heads=opts['score_heads'],
                     repetition_penalty=opts['repetition_penalty'])
    tok = tokenizer_for(model.config)
    model.config.eos_token_id = META['eos_token_id']          # the fixture's synthetic "eos" (after the tokenizer set its own)
    d = LiveInferForDemo(args, model=model, tokenizer=tok)
    d.input_video_stream(stream_frames(name))
    d.encode_given_query(case['conversation'][0]['content'])      # the fixture's query is due at t=0
    outs = []
    while d.frame_embeds_queue:
        outs.append(d.input_one_frame())
    assert [o['informative_score'] for o in outs] == pytest.approx([x['informative_score'] for x in case['debug_data']], abs=2e-5)
    assert sum(o['response'] is not None for o in outs) == case['n_responses']
    assert len(d.past_key_values) == case['final_kv_len']


def test_chunk_size_respects_model_step_capacity():
    model, _, _ = oracle_model('A')
    model.max_step_tokens = 128 + 3 * 4            # room for 3 frames of 4 tokens
    tok = tokenizer_for(model.config)
    d = LiveInferForBenchmark(make_args(stream_end_prob_threshold=2.0, frames_per_forward=50), model=model, tokenizer=tok)
    from helpers import stream_frames
    d.input_video_stream(stream_frames('grounding_q0'))
    assert d._chunk_size() == 3
    d.inference()
    assert len(d.debug_data_list) == 10 and d.forward_calls == 4


# ---- several streams in shared forwards: the scheduler's host logic, on CPU around the oracle ------------------------------
class _OracleWithMultiStep:
    """The oracle model plus a `multi_step` that simply loops over the segments (what mmd_frame_step_multi does in one forward)."""

    def __init__(self, model):
        self.__dict__['_m'] = model
        self.__dict__['max_step_tokens'] = 4096
        self.__dict__['device'] = torch.device('cpu')
        self.__dict__['multi_calls'] = []

    def __getattr__(self, name):
        return getattr(self._m, name)

    def __call__(self, *a, **k):
        return self._m(*a, **k)

    def multi_step(self, segments, want_logits=True):
        self.multi_calls.append([s['x'].reshape(-1, s['x'].shape[-1]).shape[0] for s in segments])
        out = []
        for s in segments:
            x = s['x'].reshape(1, -1, s['x'].shape[-1])
            r = self._m(inputs_embeds=x, past_key_values=s['cache'], use_cache=True, return_dict=True)
            rows = list(s.get('head_rows', ()))
            heads = torch.cat([r.informative_logits[0, rows], r.relevance_logits[0, rows]], dim=-1).float() if rows else None
            want = s.get('hidden', 'none')
            hid = r.hidden_states[0] if want == 'all' else (r.hidden_states[0, -1:] if want == 'last' else None)
            out.append(dict(heads=heads, hidden=hid, logits=r.logits[0, -1:].float() if want == 'last' else None, cache=r.past_key_values))
        return out


class _OracleSampler:
    def begin(self, eos, penalty, prev, max_new):
        self.eos, self.penalty, self.prev, self.tok = eos, penalty, list(prev or []), None


class _OracleWithRounds(_OracleWithMultiStep):
    """... plus `round_multi` / `new_sampler` with the semantics of mmd_round_multi / mmd_sampler (sampling inside the round, feed rows gathered from the sampler's last
    token): the scheduler then keeps a talking slot parked for the whole response."""

    def new_sampler(self):
        return _OracleSampler()

    def round_multi(self, segments):
        self.multi_calls.append([1 if s.get('feed') else s['x'].reshape(-1, s['x'].shape[-1]).shape[0] for s in segments])
        out = []
        for s in segments:
            sp = s.get('sampler')
            x = self._m.get_input_embeddings()(torch.tensor([[sp.tok]])) if s.get('feed') else s['x']
            x = x.reshape(1, -1, x.shape[-1])
            r = self._m(inputs_embeds=x, past_key_values=s['cache'], use_cache=True, return_dict=True)
            rows = list(s.get('head_rows', ()))
            heads = torch.cat([r.informative_logits[0, rows], r.relevance_logits[0, rows]], dim=-1).float() if rows else None
            tok = None
            if s.get('sample'):
                scores = r.logits[0, -1].float().clone()
                if sp.penalty is not None and sp.prev:
                    idx = torch.as_tensor(sp.prev, dtype=torch.long)
                    picked = scores[idx]
                    scores[idx] = torch.where(picked < 0, picked * sp.penalty, picked / sp.penalty)
                tok = int(scores.argmax(-1))
                sp.tok = tok
                if sp.penalty is not None and tok != sp.eos:
                    sp.prev.append(tok)
            out.append(dict(heads=heads, token=tok, cache=r.past_key_values))
        return out


@pytest.mark.parametrize('native_rounds', [False, True], ids=['python_decode', 'round_multi'])
@pytest.mark.parametrize('n_slots,k', [(1, 1), (3, 1), (6, 3), (4, 2)])
def test_multistream_scheduler_matches_reference_driver(n_slots, k, native_rounds):
    from mmduet_amd.multistream import MultiStreamInfer
    from helpers import stream_frames
    base, _, _ = oracle_model('A')
    model = (_OracleWithRounds if native_rounds else _OracleWithMultiStep)(base)
    tok = tokenizer_for(base.config)
    base.config.eos_token_id = META['eos_token_id']
    names = list(META['cases'])
    videos = []
    for n in names:
        case, opts = META['cases'][n], META['cases'][n]['opts']
        a = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12,
                      stream_end_prob_threshold=opts.get('stream_end_prob_threshold'), stream_end_score_sum_threshold=opts.get('stream_end_score_sum_threshold'),
                      score_heads=opts.get('score_heads', 'informative_score'), remove_assistant_turns=opts.get('remove_assistant_turns', False),
                      repetition_penalty=opts.get('repetition_penalty'), running_list_length=opts.get('running_list_length', 20), frames_per_forward=k)
        videos.append(dict(frames=stream_frames(n), conversation=case['conversation'], args=a))
    ms = MultiStreamInfer(videos[0]['args'], model=model, tokenizer=tok, n_slots=n_slots)
    results = ms.run(videos)
    for n, res in zip(names, results):
        case = META['cases'][n]
        assert len(res['debug_data']) == case['T']
        for got, exp in zip(res['debug_data'], case['debug_data']):
            assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=2e-5) and got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=2e-5)
        assert res['response_token_ids'] == case['generated'] and res['final_kv_len'] == case['final_kv_len'] and res['generated_token_ids'] == case['penalty_ids']
        assert [(r['role'], r['content']) for r in res['responses']] == [(r['role'], r['content']) for r in case['responses']]
    widths = [len(c) for c in model.multi_calls]
    assert max(widths) == min(n_slots, len(names))                       # the forwards really carried one segment per live slot
    if n_slots > 1:
        assert any(1 in c and max(c) > 1 for c in model.multi_calls)     # a generating stream (1 row) rode with other streams' rows
    if native_rounds:          # (a round that carries a query turn -- all hidden rows wanted -- next to a talking stream is two model calls)
        assert ms.rounds <= len(model.multi_calls) <= ms.rounds + len(names)
    else:
        assert ms.rounds == len(model.multi_calls)


def test_multistream_scheduler_propagates_errors():
    from mmduet_amd.multistream import MultiStreamInfer
    from helpers import stream_frames
    base, _, _ = oracle_model('A')
    model = _OracleWithMultiStep(base)
    def boom(segments, want_logits=True):
        raise RuntimeError('device fault')
    model.__dict__['multi_step'] = boom
    case = META['cases']['grounding_q0']
    a = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], stream_end_prob_threshold=2.0)
    ms = MultiStreamInfer(a, model=model, tokenizer=tokenizer_for(base.config), n_slots=2)
    with pytest.raises(RuntimeError, match='device fault'):
        ms.run([dict(frames=stream_frames('grounding_q0'), conversation=case['conversation'])] * 2)


@pytest.mark.parametrize('fps,k,q_times', [(3.0, 4, (10.0, 17.0)), (10.0, 5, (1.3, 2.0)), (3.0, 7, (0.0, 3.0 + 1e-9)), (7.0, 3, (1.0, 2.0, 3.0))])
def test_mid_stream_query_lands_on_the_same_frame_for_every_chunk_size(fps, k, q_times):
    """A user query due at time t is encoded before the first frame whose ACCUMULATED clock (video_time += 1/fps,
    test/inference.py:281,311) reaches t.  j/fps and j additions of 1/fps differ in the last ulp for fps 3, 7, 10: the chunked
    schedule must replay the accumulation, or the query slips by one frame against the one-frame-per-forward schedule."""
    from helpers import stream_frames
    model, _, _ = oracle_model('A')
    tok = tokenizer_for(model.config)
    frames = torch.cat([stream_frames('grounding_q0')] * 6)[:56]           # 56 frames
    conv = [{'role': 'user', 'content': f'what happens at {t}?', 'time': t} for t in q_times]

    def run(kk):
        d = LiveInferForBenchmark(make_args(frame_fps=fps, system_prompt=META['system_prompt'], stream_end_prob_threshold=2.0, frames_per_forward=kk),
                                  model=model, tokenizer=tok)
        query_frames = []
        enc = d._encode_query
        d._encode_query = lambda: (query_frames.append(d.frame_idx), enc())[1]
        d.input_video_stream(frames)
        d.input_query_stream(conv)
        d.inference()
        return query_frames, [x['informative_score'] for x in d.debug_data_list], len(d.past_key_values)

    q1, s1, n1 = run(1)
    qk, sk, nk = run(k)
    assert qk == q1 and nk == n1
    assert sk == pytest.approx(s1, abs=2e-5)


@pytest.mark.parametrize('level', ['embed', 'tower'])
def test_feature_file_stream_equals_frame_stream(tmp_path, level):
    """f4: Phase A from a feature file (mmduet_amd/features.py; layout of data/utils.py:99-117) queues exactly what input_video_stream would have."""
    from helpers import stream_frames
    from mmduet_amd.features import extract_features, save_frame_features, load_frame_features
    name = 'sum_remove'
    case, opts = META['cases'][name], META['cases'][name]['opts']
    model, _, _ = oracle_model('A')
    d = run_stream_case(LiveInferForBenchmark, model, name, case, META)
    feats = extract_features(model, stream_frames(name), level)
    assert feats.shape[0] == case['T'] and feats.ndim == 3
    save_frame_features(tmp_path / 'clip.pt', feats, to_bf16=False)
    args = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12, stream_end_score_sum_threshold=opts.get('stream_end_score_sum_threshold'),
                     stream_end_prob_threshold=opts.get('stream_end_prob_threshold'), score_heads=opts.get('score_heads', 'informative_score'),
                     remove_assistant_turns=opts.get('remove_assistant_turns', False), repetition_penalty=opts.get('repetition_penalty'),
                     running_list_length=opts.get('running_list_length', 20))
    tok = tokenizer_for(model.config)
    model.config.eos_token_id = META['eos_token_id']
    d2 = LiveInferForBenchmark(args, model=model, tokenizer=tok)
    d2.input_feature_stream(str(tmp_path / 'clip.pt'))
    d2.input_query_stream(case['conversation'])
    d2.responses = d2.inference()
    assert d2.debug_data_list == d.debug_data_list and d2.response_token_ids == d.response_token_ids
    assert len(d2.past_key_values) == len(d.past_key_values)
    with pytest.raises(ValueError):
        d2.input_feature_stream(torch.zeros(3, 5, 7))
