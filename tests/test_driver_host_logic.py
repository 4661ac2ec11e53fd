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
