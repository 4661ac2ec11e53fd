"""The multi-GPU seam end to end on CPU (the reference's seam: test/inference.py:335-338 `--start_idx/--end_idx`): the product CLI
(`python -m mmduet_amd`) under a torchrun-style world of 2 (gloo) -- `i % world` shards, `<output>.rank<r>` files, ONE score all-gather, rank 0's
`<output>.scores.json` -- against the same CLI run with one rank.  The native library is the oracle-backed C-ABI stand-in (tests/cabi_oracle_shim.py,
no GPU here), everything above it is the shipped Python; single-stream and `--streams_per_gpu 2` (shared forwards)."""
import json, os, socket, subprocess, sys
import numpy as np
import pytest
from conftest import ROOT, GOLDEN, load_golden_weights
from helpers import product_config, stream_cases, stream_frames

RUNNER = os.path.join(ROOT, 'tests', 'cli_shim_runner.py')


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    """A checkpoint directory of the tiny golden model + a 5-entry test file over the golden streams' frames (one entry unreadable: skipped like the reference does)."""
    from mmduet_amd.weights import save_checkpoint
    from mmduet_amd.tokenization_live import build_byte_level_tokenizer
    d = tmp_path_factory.mktemp('cli_world2')
    cfgd, w = load_golden_weights('A')
    ckpt = str(d / 'llava-tiny-cfgA')
    save_checkpoint(w, ckpt, product_config(cfgd))
    build_byte_level_tokenizer().save_pretrained(ckpt)
    meta = stream_cases()
    names = list(meta['cases'])[:4]
    entries = []
    for j, n in enumerate(names):
        np.save(d / f'{n}.npy', stream_frames(n).numpy())
        case = meta['cases'][n]
        entries.append(dict(question_id=f'q{j}', frames=f'{n}.npy', fps=case['fps'], video_duration=case['T'] / case['fps'],
                            conversation=[t for t in case['conversation'] if t['role'] == 'user']))
    entries.insert(2, dict(question_id='broken', frames='missing.npy', fps=1.0, conversation=[{'role': 'user', 'content': 'x', 'time': 0.0}]))
    (d / 'test.json').write_text(json.dumps(entries))
    flags = ['--llm_pretrained', ckpt, '--input_dir', str(d), '--test_fname', str(d / 'test.json'), '--bf16', 'false', '--overlap_vision', 'false',
             '--frame_num_tokens', str(cfgd['frame_num_tokens']), '--video_pooling_stride', str(cfgd['video_pooling_stride']), '--frame_resolution', str(cfgd['frame_resolution']),
             '--stream_end_prob_threshold', '0.5', '--max_new_tokens', '6', '--system_prompt', meta['system_prompt'], '--frames_per_forward', '3']
    return d, flags, [e['question_id'] for e in entries]


def _run(flags, out, world, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
        if world > 1:
            env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env['OMP_NUM_THREADS'] = '2'
        procs.append(subprocess.Popen([sys.executable, RUNNER] + flags + ['--output_fname', out] + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    logs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        logs.append(o)
    return logs


def _records(path):
    return {json.loads(l)['question_id']: json.loads(l) for l in open(path) if l.strip()}


@pytest.mark.parametrize('streams_per_gpu', [1, 2])
def test_cli_two_ranks_equal_one_rank(dataset, streams_per_gpu):
    d, flags, qids = dataset
    extra = ['--streams_per_gpu', str(streams_per_gpu)]
    one = str(d / f'one_s{streams_per_gpu}.jsonl')
    two = str(d / f'two_s{streams_per_gpu}.jsonl')
    _run(flags, one, 1, extra)
    logs = _run(flags, two, 2, extra)
    ref = _records(one)
    assert set(ref) == {q for q in qids if q != 'broken'}                      # the unreadable clip was skipped, the rest ran
    assert not os.path.exists(one + '.scores.json')                            # one rank: no collective, no merged file
    # i % world sharding: rank r wrote exactly its entries into <output>.rank<r>
    shards = [_records(f'{two}.rank{r}') for r in range(2)]
    for r in range(2):
        assert set(shards[r]) == {q for i, q in enumerate(qids) if i % 2 == r and q != 'broken'}
    merged = {**shards[0], **shards[1]}
    assert merged == ref                                                       # records (scores, responses, times) equal the 1-rank run, bit for bit (same CPU arithmetic)
    assert any(r['model_response_list'] and any(t['role'] == 'assistant' for t in r['model_response_list']) for r in ref.values())
    # the gathered score block, written by rank 0 in dataset order, equals the per-frame scores of the 1-rank run
    scores = json.load(open(two + '.scores.json'))
    assert list(scores) == qids and scores['broken'] == []
    for q, rec in ref.items():
        want = [[x['informative_score'], x['relevance_score']] for x in rec['debug_data']]
        got = scores[q]
        assert len(got) == len(want) and all(abs(a - b) <= 5.1e-4 for g, w in zip(got, want) for a, b in zip(g, w)), q      # the record is rounded to 3 decimals
    # response mode: the generated token ids were gathered too, keyed like the scores; the texts of the records are their decoding
    resp = json.load(open(two + '.responses.json'))
    assert list(resp) == qids and resp['broken'] == []
    for q, rec in ref.items():
        n_assistant = sum(t['role'] == 'assistant' for t in rec['model_response_list'])
        assert len(resp[q]) == n_assistant and all(len(t) > 0 for t in resp[q]), q
    assert not os.path.exists(one + '.responses.json')
    # the product layer really ran on both ranks, through the driver's native entry points
    for log in logs:
        calls = dict(kv.split('=') for kv in [l for l in log.splitlines() if l.startswith('CLI_CALLS ')][-1].split()[1:])
        assert int(calls.get('mmd_vit_encode_frames', 0)) > 0
        if streams_per_gpu > 1:          # shared forwards: frame chunks, queries and decode rows (lm_head inside the call) all travel through mmd_frame_step_multi
            assert int(calls.get('mmd_frame_step_multi', 0)) > 0 and int(calls.get('mmd_frame_step', 0)) == 0
        else:
            assert int(calls.get('mmd_frame_step', 0)) > 0 and int(calls.get('mmd_greedy_generate', 0)) > 0
