"""world_size-2 gloo run of the data-parallel score gather (mmduet_amd/distributed.py) on CPU."""
import os, socket, subprocess, sys, json
from conftest import ROOT

WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MMD_ROOT"])
from mmduet_amd.distributed import init_distributed, shard_indices, shard_shape, gather_scores
import torch.distributed as dist
rank, world, local = init_distributed(backend="gloo")
calls = []
_ag = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda out, inp, *a, **k: (calls.append(tuple(inp.shape)), _ag(out, inp, *a, **k))[1]
lengths = [5, 9, 3, 7, 4]
mine = shard_indices(len(lengths), rank, world)
local_scores = [torch.full((lengths[i], 2), float(i)) + torch.arange(lengths[i])[:, None] * 0.01 for i in mine]
n_max, t_max = shard_shape(len(lengths), world, lengths)
scores, lens = gather_scores(local_scores, t_max=t_max, n_max=n_max)      # shape known from the dataset: ONE collective
assert calls == [(n_max, t_max + 1, 2)], calls
s2, l2 = gather_scores(local_scores)                                      # shape unknown: + one 16-byte shape exchange
assert len(calls) == 3 and torch.equal(l2, lens) and torch.equal(torch.nan_to_num(s2), torch.nan_to_num(scores))
out = {"rank": rank, "mine": mine, "shape": list(scores.shape), "lens": lens.tolist(),
       "first": [[float(scores[r, j, 0, 0]) for j in range(scores.shape[1])] for r in range(world)],
       "balanced": [shard_indices(len(lengths), r, world, lengths) for r in range(world)]}
print("RESULT " + json.dumps(out), flush=True)
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def test_gather_scores_world2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MMD_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith('RESULT ')][0][7:]))
    a, b = sorted(outs, key=lambda x: x['rank'])
    assert a['mine'] == [0, 2, 4] and b['mine'] == [1, 3]
    assert a['shape'] == b['shape'] == [2, 3, 9, 2]                 # [world, n_max, t_max, 2]
    assert a['lens'] == b['lens'] == [[5, 3, 4], [9, 7, 0]]
    assert a['first'][0] == [0.0, 2.0, 4.0] and a['first'][1][:2] == [1.0, 3.0]
    assert a['first'] == b['first'] or all(x == y or (x != x and y != y) for r in range(2) for x, y in zip(a['first'][r], b['first'][r]))
    bal = a['balanced']
    assert sorted(bal[0] + bal[1]) == [0, 1, 2, 3, 4] and abs(sum([5, 9, 3, 7, 4][i] for i in bal[0]) - 14) <= 2


FAIL_WORKER = r'''
import os, sys, json, types, torch
sys.path.insert(0, os.environ["MMD_ROOT"])
from mmduet_amd.distributed import init_distributed, shard_indices
from mmduet_amd.__main__ import _gather_and_write
import torch.distributed as dist
rank, world, local = init_distributed(backend="gloo")
data = [{"question_id": f"q{i}"} for i in range(5)]
mine = shard_indices(len(data), rank, world)
done = mine if rank == 0 else mine[:1]          # rank 1 "fails" after its first clip
scores = {i: [[0.25 + i, 0.5]] * (3 + i) for i in done}
ids = {i: [[7, 8, 9 + i]] for i in done}
args = types.SimpleNamespace(output_fname=os.environ["OUT"])
_gather_and_write(args, data, mine, world, rank, scores, ids, RuntimeError("device fault") if rank == 1 else None, torch.device("cpu"))
print("RESULT ok", flush=True)
dist.destroy_process_group()
'''


def test_failed_rank_joins_the_collectives_and_the_files_say_partial(tmp_path):
    """a rank that failed still takes part in the gathers (the other ranks must not hang) -- but the merged files then carry the suffix `.partial` and name the
    failed ranks, so its unprocessed clips (zero-length streams) cannot be mistaken for unreadable ones."""
    script = tmp_path / 'worker.py'
    script.write_text(FAIL_WORKER)
    port = _free_port()
    out = str(tmp_path / 'run.jsonl')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MMD_ROOT=ROOT, OUT=out)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0 and 'RESULT ok' in o, e[-2000:]
    assert not os.path.exists(out + '.scores.json')
    sc = json.load(open(out + '.scores.json.partial'))
    assert sc['__failed_ranks__'] == [1]
    assert len(sc['q0']) == 3 and len(sc['q2']) == 5 and len(sc['q4']) == 7          # rank 0's clips complete
    assert len(sc['q1']) == 4 and sc['q3'] == []                                       # rank 1: one clip done, one never reached
    ids = json.load(open(out + '.responses.json.partial'))
    assert ids['__failed_ranks__'] == [1] and ids['q0'] == [[7, 8, 9]] and ids['q3'] == []


def test_single_process_gather_is_local():
    import torch
    from mmduet_amd.distributed import gather_scores, shard_indices
    s, l = gather_scores([torch.ones(4, 2), torch.zeros(2, 2)])
    assert s.shape == (1, 2, 4, 2) and l.tolist() == [[4, 2]]
    assert shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert torch.isnan(s[0, 1, 2:]).all() and s[0, 0].eq(1).all() and s[0, 1, :2].eq(0).all()
