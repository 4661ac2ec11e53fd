"""Several streams per GPU in shared forwards (mmduet_amd/multistream.py, mmd_frame_step_multi): every stream must come out as if
it had run alone -- checked against the reference-generated whole-stream fixtures, all six cases running CONCURRENTLY."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from helpers import hip_model, stream_cases, stream_frames, make_args, tokenizer_for
from mmduet_amd.multistream import MultiStreamInfer

META = stream_cases()


def _args_for(case, k):
    opts = case['opts']
    return make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12,
                     stream_end_prob_threshold=opts.get('stream_end_prob_threshold'),
                     stream_end_score_sum_threshold=opts.get('stream_end_score_sum_threshold'),
                     score_heads=opts.get('score_heads', 'informative_score'),
                     remove_assistant_turns=opts.get('remove_assistant_turns', False),
                     repetition_penalty=opts.get('repetition_penalty'),
                     running_list_length=opts.get('running_list_length', 20), frames_per_forward=k)


@pytest.fixture(scope='module')
def model_f32():
    m = hip_model('A', torch.float32)[0]
    m.tok = tokenizer_for(m.config)                      # (the tokenizer builder writes its own eos into the config ...)
    m.config.eos_token_id = META['eos_token_id']         # ... the fixture's synthetic "eos" goes in afterwards (make_golden.py)
    return m


@pytest.mark.parametrize('python_decode', [False, True], ids=['native_rounds', 'python_decode'])
@pytest.mark.parametrize('n_slots', [2, 6])
@pytest.mark.parametrize('k', [1, 4])
def test_concurrent_streams_match_reference_fixtures(model_f32, n_slots, k, python_decode):
    """(native_rounds: responses decode inside mmd_round_multi -- sampling, repetition penalty and the next token's embedding on the device, a talking slot parked for
    the whole response; python_decode: the per-token host loop over mmd_frame_step_multi, the cross-check.)"""
    names = list(META['cases'])
    videos = [dict(frames=stream_frames(n), conversation=META['cases'][n]['conversation'], args=_args_for(META['cases'][n], k)) for n in names]
    ms = MultiStreamInfer(videos[0]['args'], model=model_f32, tokenizer=model_f32.tok, n_slots=n_slots)
    ms.python_decode = python_decode
    results = ms.run(videos)
    for name, res in zip(names, results):
        case = META['cases'][name]
        assert len(res['debug_data']) == case['T'], name
        for got, exp in zip(res['debug_data'], case['debug_data']):
            assert got['time'] == pytest.approx(exp['time'])
            assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=2e-4), name
            assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=2e-4), name
        assert res['response_token_ids'] == case['generated'], name
        assert [(r['role'], r['time'], r['content']) for r in res['responses']] == [(r['role'], pytest.approx(r['time']), r['content']) for r in case['responses']]
        assert res['final_kv_len'] == case['final_kv_len'], name
        assert res['generated_token_ids'] == case['penalty_ids'], name
    # the forwards really were shared: fewer rounds than the streams would have needed one by one
    assert ms.rounds < sum(r['forward_calls'] + sum(len(g) for g in r['response_token_ids']) for r in results)


def test_multi_step_equals_separate_calls(model_f32):
    m = model_f32
    H = m.config.hidden_size
    g = torch.Generator().manual_seed(5)
    xa, xb, xc = (torch.randn(n, H, generator=g).cuda() * 0.3 for n in (7, 1, 12))
    pa, pb = (torch.randn(n, H, generator=g).cuda() * 0.3 for n in (20, 33))
    # separate: stream A continues a 20-token context, stream B a 33-token one, stream C starts empty
    ca = m(inputs_embeds=pa[None]).past_key_values; cb = m(inputs_embeds=pb[None]).past_key_values
    ra = m(inputs_embeds=xa[None], past_key_values=ca); rb = m(inputs_embeds=xb[None], past_key_values=cb); rc = m(inputs_embeds=xc[None])
    want_heads = [m.video_heads(r.hidden_states[0]) for r in (ra, rb, rc)]
    want_logits = [r.logits[0, -1] for r in (ra, rb, rc)]
    # merged
    ca2 = m(inputs_embeds=pa[None]).past_key_values; cb2 = m(inputs_embeds=pb[None]).past_key_values
    out = m.multi_step([dict(x=xa, cache=ca2, head_rows=[0, 6], hidden='last'), dict(x=xb, cache=cb2, head_rows=[0], hidden='last'),
                        dict(x=xc, cache=None, head_rows=[11], hidden='all')])
    assert [len(o['cache']) for o in out] == [27, 34, 12]
    torch.testing.assert_close(out[0]['heads'], want_heads[0][[0, 6]].cpu(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(out[1]['heads'], want_heads[1][[0]].cpu(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(out[2]['heads'], want_heads[2][[11]].cpu(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(out[0]['logits'][0], want_logits[0], atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(out[1]['logits'][0], want_logits[1], atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(out[2]['hidden'], rc.hidden_states[0], atol=2e-5, rtol=1e-5)
    # and the arenas continue correctly afterwards
    nxt = torch.randn(3, H, generator=g).cuda() * 0.3
    r1 = m(inputs_embeds=nxt[None], past_key_values=ra.past_key_values); r2 = m(inputs_embeds=nxt[None], past_key_values=out[0]['cache'])
    torch.testing.assert_close(r2.hidden_states, r1.hidden_states, atol=2e-5, rtol=1e-5)
    with pytest.raises(ValueError):
        m.multi_step([dict(x=xa, cache=out[1]['cache']), dict(x=xb, cache=out[1]['cache'])])      # one arena twice


def test_round_multi_rejects_misuse_and_matches_multi_step(model_f32):
    """mmd_round_multi: the plain part (rows + heads) equals mmd_frame_step_multi; a FEED segment must be one row with a sampler of this context, a sampler may sample once
    per round, an arena may appear once; the error leaves the arenas untouched."""
    from mmduet_amd._lib import MmduetError
    m = model_f32
    H = m.config.hidden_size
    g = torch.Generator().manual_seed(9)
    xa, xb = (torch.randn(n, H, generator=g).cuda() * 0.3 for n in (9, 5))
    ca = m(inputs_embeds=xa[None]).past_key_values; cb = m(inputs_embeds=xb[None]).past_key_values
    ya, yb = (torch.randn(n, H, generator=g).cuda() * 0.3 for n in (6, 3))
    ref = m.multi_step([dict(x=ya, cache=m.cache_prefix(ca, 9), head_rows=[2, 5]), dict(x=yb, cache=m.cache_prefix(cb, 5), head_rows=[2], hidden='last')])
    smp = m.new_sampler(); smp.begin(-1, None, None, 4)
    out = m.round_multi([dict(x=ya, cache=m.cache_prefix(ca, 9), head_rows=[2, 5]), dict(x=yb, cache=m.cache_prefix(cb, 5), head_rows=[2], sampler=smp, sample=True)])
    torch.testing.assert_close(out[0]['heads'], ref[0]['heads'], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(out[1]['heads'], ref[1]['heads'], atol=2e-5, rtol=1e-5)
    assert out[1]['token'] == int(ref[1]['logits'][0].argmax()) and out[0]['token'] is None
    assert [len(o['cache']) for o in out] == [15, 8]
    # the next round feeds the drawn token: equal to embedding it by hand
    nxt = m.round_multi([dict(x=None, cache=out[1]['cache'], sampler=smp, feed=True, sample=True)])
    by_hand = m(inputs_embeds=m.get_input_embeddings()(torch.tensor([[out[1]['token']]], device=m.device)).view(1, 1, H), past_key_values=m.cache_prefix(out[1]['cache'], 8))
    assert nxt[0]['token'] == int(by_hand.logits[0, -1].argmax()) and len(nxt[0]['cache']) == 9
    la, lb = len(out[0]['cache']), len(nxt[0]['cache'])
    with pytest.raises(MmduetError):          # feed without a sampler
        m.round_multi([dict(x=None, cache=out[0]['cache'], feed=True)])
    with pytest.raises(MmduetError):          # one sampler sampling twice in a round
        m.round_multi([dict(x=ya, cache=m.cache_prefix(out[0]['cache'], la), sampler=smp, sample=True), dict(x=yb, cache=m.cache_prefix(nxt[0]['cache'], lb), sampler=smp, sample=True)])
    with pytest.raises(ValueError):           # one arena twice
        m.round_multi([dict(x=ya, cache=out[0]['cache']), dict(x=yb, cache=out[0]['cache'])])
    assert out[0]['cache'].arena.length() == la and nxt[0]['cache'].arena.length() == lb
