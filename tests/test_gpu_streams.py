"""Whole-stream parity on the GPU: product driver + HIP model vs what the reference driver + reference model produced."""
import pytest
import torch

pytestmark = pytest.mark.gpu
import json
import numpy as np
from helpers import hip_model, stream_cases, run_stream_case, tokenizer_for
from mmduet_amd.inference import LiveInferForBenchmark

META = stream_cases()


@pytest.fixture(scope='module')
def model_f32():
    return hip_model('A', torch.float32)[0]


@pytest.mark.parametrize('name', list(META['cases']))
@pytest.mark.parametrize('k', [1, 4])
def test_stream_fp32_matches_reference(model_f32, name, k):
    case = META['cases'][name]
    d = run_stream_case(LiveInferForBenchmark, model_f32, name, case, META, frames_per_forward=k)
    assert len(d.debug_data_list) == case['T']
    for got, exp in zip(d.debug_data_list, case['debug_data']):
        assert got['time'] == pytest.approx(exp['time'])
        assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=2e-4)
        assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=2e-4)
    assert d.response_token_ids == case['generated']
    assert [(r['role'], r['time'], r['content']) for r in d.responses] == [(r['role'], pytest.approx(r['time']), r['content']) for r in case['responses']]
    assert len(d.past_key_values) == case['final_kv_len']
    assert [int(x) for x in d.generated_token_ids] == case['penalty_ids']
    if k > 1 and case['n_responses'] == 0:
        assert d.forward_calls <= -(-case['T'] // k) + len(case['conversation']) + 1


def test_stream_bf16_runs_and_stays_close():
    m = hip_model('A', torch.bfloat16)[0]
    case = META['cases']['grounding_q0']
    d = run_stream_case(LiveInferForBenchmark, m, 'grounding_q0', case, META, dtype=torch.bfloat16)
    for got, exp in zip(d.debug_data_list, case['debug_data']):
        assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=5e-2)
        assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=5e-2)
    assert len(d.past_key_values) == case['final_kv_len']


@pytest.mark.gpu
def test_cli_over_predecoded_and_raw_decoded_entries(tmp_path, monkeypatch):
    """`python -m mmduet_amd`: both entry kinds (letter-boxed frames / raw decoder output + reference sampling on the GPU), JSONL
    shape, evaluator_format, time instruction prefix."""
    import mmduet_amd.inference as inf
    import mmduet_amd.__main__ as cli
    from mmduet_amd.results import grounding_sweep
    model, cfgd, _ = hip_model('A')
    tok = tokenizer_for(model.config)
    monkeypatch.setattr(inf, 'build_model_and_tokenizer', lambda **kw: (model, tok))
    R = model.config.frame_resolution
    rng = np.random.default_rng(3)
    np.save(tmp_path / 'clip.npy', rng.integers(0, 256, (5, 3, R, R), dtype=np.uint8))
    np.save(tmp_path / 'raw.npy', rng.integers(0, 256, (40, 30, 50, 3), dtype=np.uint8))
    entries = [{'question_id': 'q0', 'frames': 'clip.npy', 'fps': 1.0, 'video_duration': 5.0, 'conversation': [{'role': 'user', 'content': 'what happens?', 'time': 0.0}]},
               {'question_id': 'q1', 'decoded': 'raw.npy', 'input_fps': 10.0, 'conversation': [{'role': 'user', 'content': 'describe', 'time': 0.0}]},
               {'question_id': 'q2', 'frames': 'missing.npy', 'conversation': [{'role': 'user', 'content': 'x', 'time': 0.0}]}]
    json.dump(entries, open(tmp_path / 'test.json', 'w'))
    out = tmp_path / 'out.jsonl'
    cli.main(['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', str(tmp_path), '--test_fname', str(tmp_path / 'test.json'),
              '--output_fname', str(out), '--frame_fps', '2', '--frame_resolution', str(R), '--max_num_frames', '6', '--time_instruction_format', 'vtimellm',
              '--stream_end_prob_threshold', '0.5', '--evaluator_format', 'true', '--max_new_tokens', '4'])
    recs = [json.loads(l) for l in open(out)]
    assert [r['question_id'] for r in recs] == ['q0', 'q1']                       # the unreadable entry is skipped (test/datasets.py:102-104)
    assert len(recs[0]['debug_data']) == 5 and len(recs[1]['debug_data']) == 6      # 4 s of 10-fps video at 2 fps = 8 frames, cut at max_num_frames
    assert recs[1]['video_duration'] == 4.0
    e = recs[1]['debug_data'][1]
    assert e['video_time'] == 0.5 and len(e['relevance_score']) == 2 and abs(sum(e['relevance_score']) - 1) < 2e-3
    assert recs[1]['model_response_list'][0]['content'].startswith('This is a video with 6 frames.\n')
    assert len(grounding_sweep(recs[1]['debug_data'], [[0.5, 1.5]], 1)) == 21
    # the same entries, two videos at a time in shared forwards: same records (fp32 tiny model: scores to 1e-3 after the 3-digit rounding)
    out2 = tmp_path / 'out2.jsonl'
    cli.main(['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', str(tmp_path), '--test_fname', str(tmp_path / 'test.json'),
              '--output_fname', str(out2), '--frame_fps', '2', '--frame_resolution', str(R), '--max_num_frames', '6', '--time_instruction_format', 'vtimellm',
              '--stream_end_prob_threshold', '0.5', '--evaluator_format', 'true', '--max_new_tokens', '4', '--streams_per_gpu', '2'])
    recs2 = [json.loads(l) for l in open(out2)]
    recs2.sort(key=lambda r: r['question_id'])                 # shared forwards write records as videos complete
    assert [r['question_id'] for r in recs2] == ['q0', 'q1']
    for a, b in zip(recs, recs2):
        assert a['model_response_list'] == b['model_response_list'] and len(a['debug_data']) == len(b['debug_data'])
        for x, y in zip(a['debug_data'], b['debug_data']):
            assert x['video_time'] == y['video_time'] and abs(x['relevance_score'][1] - y['relevance_score'][1]) <= 1.5e-3


def test_cli_prefetch_gives_identical_records(tmp_path, monkeypatch):
    """`python -m mmduet_amd --num_workers N` (mmduet_amd/prefetch.py; the reference: DataLoader(num_workers=4), test/inference.py:341): clips decoded by loader threads into pinned
    buffers while the GPU runs the current one, raw decoder output uploaded on a copy stream, letter-boxed frames per tower batch on the tower stream -- against the inline loader
    (--num_workers 0): the same records, byte for byte, for every entry kind (frames / raw decoder output / Motion-JPEG container / unreadable), one and two streams per GPU."""
    import mmduet_amd.inference as inf
    import mmduet_amd.__main__ as cli
    from mmduet_amd.video_decode import write_mjpeg_avi
    model, cfgd, _ = hip_model('A')
    tok = tokenizer_for(model.config)
    monkeypatch.setattr(inf, 'build_model_and_tokenizer', lambda **kw: (model, tok))
    R = model.config.frame_resolution
    rng = np.random.default_rng(7)
    entries = []
    for i in range(3):
        np.save(tmp_path / f'clip{i}.npy', rng.integers(0, 256, (4 + i, 3, R, R), dtype=np.uint8))
        entries.append({'question_id': f'f{i}', 'frames': f'clip{i}.npy', 'fps': 1.0, 'video_duration': 4.0 + i, 'conversation': [{'role': 'user', 'content': 'what happens?', 'time': 0.0}]})
    for i in range(2):
        np.save(tmp_path / f'raw{i}.npy', rng.integers(0, 256, (30 + 10 * i, 30, 50, 3), dtype=np.uint8))
        entries.append({'question_id': f'd{i}', 'decoded': f'raw{i}.npy', 'input_fps': 10.0, 'conversation': [{'role': 'user', 'content': 'describe', 'time': 0.0}]})
    for i in range(3):
        write_mjpeg_avi(tmp_path / f'v{i}.avi', rng.integers(0, 256, (24 + 6 * i, 36, 48, 3)).astype(np.uint8), 8.0)
        entries.append({'question_id': f'v{i}', 'video': f'v{i}.avi', 'conversation': [{'role': 'user', 'content': 'describe', 'time': 0.0}]})
    entries.insert(4, {'question_id': 'broken', 'frames': 'missing.npy', 'conversation': [{'role': 'user', 'content': 'x', 'time': 0.0}]})
    json.dump(entries, open(tmp_path / 'test.json', 'w'))

    def run(tag, *extra):
        out = tmp_path / f'{tag}.jsonl'
        cli.main(['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', str(tmp_path), '--test_fname', str(tmp_path / 'test.json'), '--output_fname', str(out),
                  '--frame_fps', '2', '--frame_resolution', str(R), '--max_num_frames', '6', '--time_instruction_format', 'vtimellm', '--stream_end_prob_threshold', '0.5',
                  '--max_new_tokens', '4', *extra])
        return [json.loads(l) for l in open(out)]
    inline, pre = run('inline', '--num_workers', '0'), run('pre', '--num_workers', '4')
    assert [r['question_id'] for r in inline] == [e['question_id'] for e in entries if e['question_id'] != 'broken']
    assert inline == pre
    two_inline, two_pre = run('two_inline', '--num_workers', '0', '--streams_per_gpu', '2'), run('two_pre', '--num_workers', '3', '--streams_per_gpu', '2')
    key = lambda r: r['question_id']
    assert sorted(two_inline, key=key) == sorted(two_pre, key=key) and len(two_pre) == len(inline)


def test_demo_driver_on_gpu_with_concurrent_query_thread(model_f32):
    """LiveInferForDemo on the HIP model: `input_one_frame` from the generator thread while `encode_given_query` arrives from a second
    thread (the Gradio handler, demo/app.py:84-85).  The reference shares past_key_values unlocked; here both go through one lock, so
    whatever the interleaving, the run equals SOME serial order: scores before the query equal the no-query stream, the context length
    accounts for every call exactly once, and nothing raises."""
    import threading, time
    from mmduet_amd.liveinfer import LiveInferForDemo
    from helpers import make_args, stream_frames
    name = 'prob_keep_pen'
    case, opts = META['cases'][name], META['cases'][name]['opts']
    tok = tokenizer_for(model_f32.config)
    model_f32.config.eos_token_id = META['eos_token_id']

    def driver():
        a = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12, stream_end_prob_threshold=opts['stream_end_prob_threshold'],
                      score_heads=opts['score_heads'], repetition_penalty=opts['repetition_penalty'])
        d = LiveInferForDemo(a, model=model_f32, tokenizer=tok)
        d.input_video_stream(stream_frames(name))
        return d

    # serial reference: query first (the fixture's query is due at t = 0), then every frame
    d0 = driver()
    d0.encode_given_query(case['conversation'][0]['content'])
    outs0 = []
    while d0.frame_embeds_queue:
        outs0.append(d0.input_one_frame())
    assert [o['informative_score'] for o in outs0] == pytest.approx([x['informative_score'] for x in case['debug_data']], abs=2e-4)
    assert len(d0.past_key_values) == case['final_kv_len']

    # two threads: frames step while a second user query lands somewhere in the middle
    d1 = driver()
    d1.encode_given_query(case['conversation'][0]['content'])
    errors, landed = [], []

    def handler():
        try:
            time.sleep(0.01)
            n_before = d1.frame_idx
            d1.encode_given_query('and what happens next?')
            landed.append(n_before)
        except Exception as e:          # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=handler)
    outs1 = []
    t.start()
    while d1.frame_embeds_queue:
        outs1.append(d1.input_one_frame())
    t.join()
    assert not errors and len(landed) == 1 and len(outs1) == case['T']
    # frames stepped before the second query saw exactly the serial run's context
    j = next((i for i, o in enumerate(outs1) if o['frame_idx'] > landed[0]), len(outs1))
    pre = min(j, next((i for i, o in enumerate(outs0) if o['response'] is not None), len(outs0)) + 1)
    assert [o['informative_score'] for o in outs1[:pre]] == pytest.approx([o['informative_score'] for o in outs0[:pre]], abs=1e-6)
    assert all(0.0 <= o['informative_score'] <= 1.0 for o in outs1)


@pytest.mark.parametrize('vmm', [True, False], ids=['virtual-memory', 'realloc'])
def test_live_arena_growth_across_reallocations_matches_presized_arena(vmm, monkeypatch):
    """A LIVE KV arena that starts at 512 tokens and grows past 100 k tokens must hold exactly the context of an arena that was sized for the whole
    stream up front: identical head logits (bit for bit -- same kernels, same data) at every probe and at the end.  Both growth mechanisms:
    the virtual-memory arena (pages mapped behind a fixed address range: no copy, never a second arena) and the fallback (>= 8 reallocations,
    each moving the K rows and the 64-token V^T blocks of every layer)."""
    from helpers import hip_model
    monkeypatch.setenv('MMDUET_KV_NO_VMM', '0' if vmm else '1')
    m_grow = hip_model('A', torch.bfloat16)[0]
    m_grow.kv_initial_tokens = 512
    m_big = hip_model('A', torch.bfloat16)[0]
    m_big.kv_initial_tokens = 110_000
    H = m_grow.config.hidden_size
    g = torch.Generator(device='cuda').manual_seed(0)
    from mmduet_amd._lib import lib
    cg = cb = None
    caps = set()
    total = 0
    for step in range(212):
        S = 509 if step % 2 == 0 else 448                   # not multiples of 64: appends straddle the V^T block boundary
        x = (torch.randn(1, S, H, generator=g, device='cuda') * 0.5).to(torch.bfloat16)
        og = m_grow(inputs_embeds=x, past_key_values=cg); cg = og.past_key_values
        ob = m_big(inputs_embeds=x, past_key_values=cb); cb = ob.past_key_values
        total += S
        caps.add(int(lib().mmd_kv_capacity(cg.arena.h)))
        if step % 20 == 0 or step == 211:
            assert torch.equal(og.informative_logits[0, -1], ob.informative_logits[0, -1]), (step, total)
            assert torch.equal(og.hidden_states[0, -1], ob.hidden_states[0, -1])
    assert total > 100_000 and len(cg) == len(cb) == total
    assert len(caps) >= (2 if vmm else 3) and max(caps) >= total and int(lib().mmd_kv_capacity(cb.arena.h)) >= 110_000       # the live arena really grew
    stride = int(lib().mmd_kv_stride(cg.arena.h))
    assert (stride >= (4 << 20)) if vmm else (stride == max(caps))      # virtual arena: the row stride is the reserved capacity, not what is backed
    # rollback into the grown arena and replay: same result as the first time
    mid = m_grow.cache_prefix(cg, total - 448)
    x = (torch.randn(1, 448, H, generator=torch.Generator(device='cuda').manual_seed(99), device='cuda') * 0.5).to(torch.bfloat16)
    a = m_grow(inputs_embeds=x, past_key_values=mid).informative_logits[0, -1].clone()
    b = m_big(inputs_embeds=x, past_key_values=m_big.cache_prefix(cb, total - 448)).informative_logits[0, -1]
    assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('level', ['embed', 'tower'])
def test_feature_file_stream_is_bit_identical_to_in_hbm_stream(tmp_path, dtype, level):
    """f4 end to end on the GPU: extract -> save ([T, tokens, C], data/utils.py:114-117 layout, model dtype) -> input_feature_stream -> Phase B.
    The file holds the very bits the in-HBM path hands on (tower output / pooled embeddings), so every score and token is IDENTICAL."""
    from helpers import make_args, stream_frames
    from mmduet_amd.features import extract_features, save_frame_features
    m = hip_model('A', dtype)[0]
    name = 'prob_keep_pen'
    case, opts = META['cases'][name], META['cases'][name]['opts']
    ref = run_stream_case(LiveInferForBenchmark, m, name, case, META, dtype=dtype)
    feats = extract_features(m, stream_frames(name), level)
    save_frame_features(tmp_path / 'clip.pt', feats, to_bf16=(dtype == torch.bfloat16))
    a = make_args(frame_fps=case['fps'], system_prompt=META['system_prompt'], max_new_tokens=12, stream_end_prob_threshold=opts['stream_end_prob_threshold'],
                  score_heads=opts['score_heads'], repetition_penalty=opts['repetition_penalty'], bf16=(dtype == torch.bfloat16))
    tok = tokenizer_for(m.config)
    m.config.eos_token_id = META['eos_token_id']
    d = LiveInferForBenchmark(a, model=m, tokenizer=tok)
    d.input_feature_stream(str(tmp_path / 'clip.pt'))
    d.input_query_stream(case['conversation'])
    d.inference()
    assert d.debug_data_list == ref.debug_data_list
    assert d.response_token_ids == ref.response_token_ids and len(d.past_key_values) == len(ref.past_key_values)


def test_cli_reads_feature_files(tmp_path, monkeypatch):
    """`python -m mmduet_amd --features_dir ...`: entries with "features" skip Phase A; records equal the frame-fed run of the same clips."""
    import mmduet_amd.inference as inf
    import mmduet_amd.__main__ as cli
    from helpers import stream_frames
    from mmduet_amd.features import extract_features, save_frame_features
    model, cfgd, _ = hip_model('A')
    tok = tokenizer_for(model.config)
    monkeypatch.setattr(inf, 'build_model_and_tokenizer', lambda **kw: (model, tok))
    R = model.config.frame_resolution
    names = ['grounding_q0', 'prob_keep']
    entries_f, entries_v = [], []
    for n in names:
        case = META['cases'][n]
        fr = stream_frames(n)
        np.save(tmp_path / f'{n}.npy', fr.numpy())
        save_frame_features(tmp_path / f'{n}.pt', extract_features(model, fr, 'embed'), to_bf16=False)
        base = dict(question_id=n, fps=case['fps'], video_duration=case['T'] / case['fps'], conversation=case['conversation'])
        entries_v.append(dict(base, frames=f'{n}.npy')); entries_f.append(dict(base, features=f'{n}.pt'))
    outs = []
    for tag, entries, extra in (('v', entries_v, []), ('f', entries_f, ['--features_dir', str(tmp_path)]), ('f2', entries_f, ['--features_dir', str(tmp_path), '--streams_per_gpu', '2'])):
        json.dump(entries, open(tmp_path / f'{tag}.json', 'w'))
        cli.main(['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', str(tmp_path), '--test_fname', str(tmp_path / f'{tag}.json'),
                  '--output_fname', str(tmp_path / f'{tag}.jsonl'), '--frame_fps', '1', '--frame_resolution', str(R), '--max_num_frames', '100',
                  '--stream_end_prob_threshold', '0.5', '--max_new_tokens', '4'] + extra)
        outs.append(sorted((json.loads(l) for l in open(tmp_path / f'{tag}.jsonl')), key=lambda r: r['question_id']))
    assert outs[0] == outs[1] and len(outs[0]) == 2
    for a, b in zip(outs[1], outs[2]):                         # shared forwards: same records up to the 3-digit score rounding
        assert a['model_response_list'] == b['model_response_list'] and len(a['debug_data']) == len(b['debug_data'])


@pytest.mark.parametrize('name', ['sum_remove', 'sum_remove_pen'])
def test_remove_turns_mode_keeps_the_chunk_tail_instead_of_replaying(model_f32, name):
    """remove_assistant_turns: a response in the middle of a multi-frame chunk does not stay in the context, so the frames behind it are restored from a KV stash
    (mmd_kv_stash / mmd_kv_unstash) instead of being recomputed.  Same records as the replaying schedule and as the reference driver; no frame is replayed."""
    case = META['cases'][name]
    runs = {}
    for reuse in (True, False):
        orig = LiveInferForBenchmark.__init__

        def patched(self, *a, _reuse=reuse, **k):
            orig(self, *a, **k)
            self.reuse_chunk_tail = _reuse
        LiveInferForBenchmark.__init__ = patched
        try:
            runs[reuse] = run_stream_case(LiveInferForBenchmark, model_f32, name, case, META, frames_per_forward=5)
        finally:
            LiveInferForBenchmark.__init__ = orig
    a, b = runs[True], runs[False]
    assert a.replayed_frames == 0 and b.replayed_frames > 0 and case['n_responses'] > 0
    assert a.response_token_ids == b.response_token_ids == case['generated']
    assert len(a.past_key_values) == len(b.past_key_values) == case['final_kv_len']
    for x, y, ref in zip(a.debug_data_list, b.debug_data_list, case['debug_data']):
        assert x['informative_score'] == pytest.approx(y['informative_score'], abs=1e-5) == pytest.approx(ref['informative_score'], abs=2e-4)
    assert a.forward_calls < b.forward_calls


@pytest.mark.parametrize('name', [n for n, c in META['cases'].items() if c['n_responses'] > 0][:2])
def test_tower_schedule_does_not_change_results(model_f32, name, monkeypatch):
    """Where the tower batches run (all up front, one batch ahead + bursts of capped-grid batches beside a response's decoding, same stream as the LLM)
    is scheduling only: scores, responses and the final context are bit-identical."""
    case = META['cases'][name]
    runs = []
    for sched in (dict(vit_burst_batches=0), dict(vit_burst_batches=3, vit_burst_blocks=128), dict(vit_burst_batches=1, vit_burst_blocks=32),
                  dict(vit_burst_batches=0, vit_lookahead_batches=0), dict(overlap_vision=False)):
        class D(LiveInferForBenchmark):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                for key, v in sched.items():
                    setattr(self, key, v)
        d = run_stream_case(D, model_f32, name, case, META, frames_per_forward=4)
        runs.append(([(x['informative_score'], x['relevance_score']) for x in d.debug_data_list], d.response_token_ids, len(d.past_key_values)))
    assert all(r == runs[0] for r in runs[1:])
    assert runs[0][1] == case['generated']
