"""True layer shapes (LLaVA-OV-Qwen2-7B widths, SigLIP-so400m widths; reduced layer count and vocab so the CPU oracle
finishes in seconds): bf16 HIP path vs the oracle executed in bf16, plus size-independent properties of the schedule."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from oracle import duet_oracle as O

TOL = 6e-2        # bf16: accumulation-order differences on O(1) logits (the fp32 path is held to 3e-4 in test_gpu_model.py)


@pytest.fixture(scope='module')
def true_shape():
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    ocfg = O.OracleConfig(vocab_size=2048, num_hidden_layers=2, vit_layers=1)          # every other dimension is the 7B / so400m default
    w = O.random_weights(ocfg, seed=3, dtype=torch.bfloat16, scale='unit')
    pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=2048, num_hidden_layers=2, vit_num_hidden_layers=2, vit_layers_removed=1,
                                        frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>')
    m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=4, max_step_tokens=512, kv_initial_tokens=1024)
    m.load_state_dict(w)
    return m, O.OracleModel(ocfg, w), ocfg


def maxerr(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def test_vit_projector_pool_true_shape(true_shape):
    m, om, cfg = true_shape
    g = torch.Generator().manual_seed(0)
    px = torch.randn(2, 3, 384, 384, generator=g).to(torch.bfloat16)
    ve = m.visual_embed(px.cuda())
    ref = om.visual_embed(px)
    assert ve.shape == (2 * 49, 3584)
    scale = ref.float().abs().max().item()
    assert maxerr(ve, ref) < TOL * max(1.0, scale), (maxerr(ve, ref), scale)


def test_llm_steps_true_shape(true_shape):
    m, om, cfg = true_shape
    g = torch.Generator().manual_seed(1)
    steps = [torch.randn(1, 59, 3584, generator=g), torch.randn(1, 49, 3584, generator=g), torch.randn(1, 1, 3584, generator=g),
             torch.randn(1, 98, 3584, generator=g)]
    cache = ocache = None
    for x in steps:
        x = (x * 0.5).to(torch.bfloat16)
        out = m(inputs_embeds=x.cuda(), past_key_values=cache); cache = out.past_key_values
        ref = om(inputs_embeds=x, past_key_values=ocache); ocache = ref.past_key_values
        assert maxerr(out.informative_logits[0, -1], ref.informative_logits[0, -1]) < TOL
        assert maxerr(out.relevance_logits[0, -1], ref.relevance_logits[0, -1]) < TOL
        assert maxerr(out.logits[0, -1], ref.logits[0, -1]) < 2 * TOL
    assert len(cache) == 59 + 49 + 1 + 98


def test_chunked_forward_equals_per_frame_forward(true_shape):
    """Causality: feeding k frames in one forward gives each frame's head logits as if fed one by one (up to bf16
    accumulation order: the per-frame step runs the split-K skinny GEMM, the chunk the big-tile GEMM)."""
    m, om, cfg = true_shape
    g = torch.Generator().manual_seed(2)
    frames = [(torch.randn(49, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda() for _ in range(6)]
    prompt = (torch.randn(1, 20, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda()
    base = m(inputs_embeds=prompt).past_key_values
    per, cache = [], base
    for f in frames:
        sc, cache = m.frame_step(f[None], cache, [48]); per.append(sc[0])
    n_end = len(cache)
    sc_chunk, cache2 = m.frame_step(torch.cat(frames)[None], m.cache_prefix(base, len(base)), [49 * (j + 1) - 1 for j in range(6)])
    assert len(cache2) == n_end
    for j in range(6):
        assert maxerr(sc_chunk[j], per[j]) < TOL
    # truncate to the end of frame 3 and replay frames 4..5: same scores again (O(1) rollback of speculative chunks)
    mid = m.cache_prefix(cache2, len(base) + 49 * 3)
    sc_replay, cache3 = m.frame_step(torch.cat(frames[3:])[None], mid, [48, 97, 146])
    assert len(cache3) == n_end
    for j in range(3):
        assert maxerr(sc_replay[j], sc_chunk[3 + j]) < TOL
    # determinism: the same call twice is bit-identical
    sc_a, _ = m.frame_step(frames[0][None], m.cache_prefix(base, len(base)), [48])
    sc_b, _ = m.frame_step(frames[0][None], m.cache_prefix(base, len(base)), [48])
    assert torch.equal(sc_a, sc_b)


@pytest.mark.parametrize('penalty', [None, 1.3])
def test_graph_replayed_decode_equals_call_loop(penalty, monkeypatch):
    """mmd_greedy_generate replays a captured hipGraph per token (position / arena / penalty list in device state);
    it must produce the tokens of the plain call-by-call loop, leave the same KV length, and be reusable across
    streams (different arenas) and contexts lengths."""
    from mmduet_amd.modeling_live import fast_greedy_generate, VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    monkeypatch.setenv('MMDUET_GRAPH', '1')            # read when the native context is created
    ocfg = O.OracleConfig(vocab_size=2048, num_hidden_layers=2, vit_layers=1)
    w = O.random_weights(ocfg, seed=3, dtype=torch.bfloat16, scale='unit')
    pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=2048, num_hidden_layers=2, vit_num_hidden_layers=2, vit_layers_removed=1,
                                        frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>')
    m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=1, max_step_tokens=1024, kv_initial_tokens=1024)
    m.load_state_dict(w)
    g = torch.Generator().manual_seed(7)
    for ctx_len in (30, 700):
        ctx = (torch.randn(1, ctx_len, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda()
        prompt = (torch.randn(1, 13, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda()
        res = []
        for loop in (False, True):
            base = m(inputs_embeds=ctx).past_key_values                   # a fresh arena each time
            out = torch.zeros(1, 12, dtype=torch.long, device='cuda'); seen = [5, 9]
            m.python_generate_loop = loop
            try:
                ids, cache, seen = fast_greedy_generate(model=m, inputs_embeds=prompt, past_key_values=base, eos_token_id=-1,
                                                        inplace_output_ids=out, repetition_penalty=penalty, generated_token_ids=seen)
            finally:
                m.python_generate_loop = False
            nxt = m(inputs_embeds=prompt[:, :3], past_key_values=cache)   # the context left behind is usable and identical
            res.append((ids[0].tolist(), list(seen), len(cache), nxt.informative_logits[0, -1].tolist()))
        assert res[0][0] == res[1][0], (res[0][0], res[1][0])
        assert res[0][1] == res[1][1] and res[0][2] == res[1][2] == ctx_len + 13 + 11
        assert res[0][3] == pytest.approx(res[1][3], abs=2e-2)


def test_fused_and_unfused_schedules_agree(true_shape, monkeypatch):
    """The fused slab consumers (reduce+RoPE+append, reduce+residual+RMSNorm) keep the unfused rounding points."""
    import subprocess, sys, os, json
    code = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MMD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MMD_ROOT"], "tests"))
from oracle import duet_oracle as O
from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
ocfg = O.OracleConfig(vocab_size=2048, num_hidden_layers=2, vit_layers=1)
w = {k: v for k, v in O.random_weights(ocfg, seed=3, dtype=torch.bfloat16, scale="unit").items()}
pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=2048, num_hidden_layers=2, vit_num_hidden_layers=2, vit_layers_removed=1, frame_num_tokens=49, frame_resolution=384)
m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=1, max_step_tokens=int(os.environ.get('MMD_MAX_STEP', 128)), kv_initial_tokens=int(os.environ.get('MMD_KV_TOKENS', 512)))
m.load_state_dict(w)
g = torch.Generator().manual_seed(5)
c = None; res = []
for S in [int(v) for v in os.environ['MMD_SIZES'].split(',')]:
    x = (torch.randn(1, S, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda()
    o = m(inputs_embeds=x, past_key_values=c); c = o.past_key_values
    res.append(o.informative_logits[0, -1].tolist() + o.logits[0, -1, :8].tolist())
print("RES " + json.dumps(res))
'''
    from conftest import ROOT
    def run(sizes='49,1,30,5,16,1,3', **kw):
        env = dict(os.environ, MMD_ROOT=ROOT, MMD_SIZES=sizes, **kw)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('RES ')][0][4:])
    # one K slab per decode GEMV: same reduction order, same rounding points -> bit-identical
    assert run('49,1,30', MMDUET_NO_FUSE='0', MMDUET_GEMV_KSPLIT_SHORT='1', MMDUET_NO_CHAIN='1') == run('49,1,30', MMDUET_NO_FUSE='1', MMDUET_GEMV_KSPLIT_SHORT='1')
    # the shipped schedule sums the decode qkv / o products in two fp32 K slabs and (rows <= 16) takes the RMSNorm statistics from per-n-tile
    # partial sums (GemvChain): different fp32 summation orders, bf16-rounding-level apart -- and deterministic run to run
    a, b, a2 = run(MMDUET_NO_FUSE='0'), run(MMDUET_NO_FUSE='1'), run(MMDUET_NO_FUSE='0')
    assert a == a2
    for ra, rb in zip(a, b):
        assert ra == pytest.approx(rb, abs=2e-2, rel=2e-2)
    c = run(MMDUET_NO_FUSE='0', MMDUET_NO_CHAIN='1')
    for ra, rc in zip(a, c):
        assert ra == pytest.approx(rc, abs=2e-2, rel=2e-2)
    # q / k / v prepared inside the decode attention kernel vs by slab_rope_append: same slabs, same rounding points, same (cos, sin) -> same bits
    assert a == run(MMDUET_NO_FUSE='0', MMDUET_NO_ROPE_FUSE='1')
    # ... also when the new positions straddle a 64-key tile / split boundary (62,63 | 64), start a stream (n = 0) or sit on the last key of a split
    edge = '3,4,55,3,2,60,4,1,1'
    assert run(edge, MMDUET_NO_FUSE='0') == run(edge, MMDUET_NO_FUSE='0', MMDUET_NO_ROPE_FUSE='1')
    # ... and at the benchmark's context length: 15 k keys in 64 splits, the new position in the last one
    long_ctx = ','.join(['120'] * 125 + ['3', '1', '4', '2', '1'])
    got = run(long_ctx, MMDUET_NO_FUSE='0', MMD_KV_TOKENS='16384')
    assert got[-5:] == run(long_ctx, MMDUET_NO_FUSE='0', MMDUET_NO_ROPE_FUSE='1', MMD_KV_TOKENS='16384')[-5:]
    # the decode attention's four-slot loader / compute ring against its two-slot form: same tiles, same per-tile arithmetic, same split ranges -> same bits
    # (edge: 1-5 tiles per split, empty splits, positions on tile boundaries; long: 4 tiles per split of 15 k keys, the ring refilled once)
    assert run(edge, MMDUET_NO_FUSE='0') == run(edge, MMDUET_NO_FUSE='0', MMDUET_ATTN_DECODE_RING='0')
    assert got[-5:] == run(long_ctx, MMDUET_NO_FUSE='0', MMDUET_ATTN_DECODE_RING='0', MMD_KV_TOKENS='16384')[-5:]
    mid = ','.join(['100'] * 4 + ['1', '2', '1'] + ['128'] * 3 + ['1', '1', '2'] + ['128'] * 40 + ['1', '2'])          # contexts 400 / 790 / 5.9 k: 1 / 1 / 2 tiles per split
    assert run(mid, MMDUET_NO_FUSE='0', MMD_KV_TOKENS='8192') == run(mid, MMDUET_NO_FUSE='0', MMDUET_ATTN_DECODE_RING='0', MMD_KV_TOKENS='8192')
    # a chunk's split-K down_proj folded by ONE reduce + residual + RMSNorm pass (M >= 512, K = 18944) against splitk_reduce followed by the RMSNorm launch: same
    # slabs, same slab order, same rounding points of the residual stream; the row's sum of squares is accumulated in another order (256 threads x 4 columns vs
    # one wave x 8 columns), so the normalised activations agree to bf16 rounding, not to the bit.  Two chunks: the second consumes the first one's fused output
    chunk = '640,700,1'
    fa, fb = run(chunk, MMDUET_NO_FUSE='0', MMD_MAX_STEP='768', MMD_KV_TOKENS='2048'), run(chunk, MMDUET_NO_FUSE='0', MMDUET_NO_SLAB_NORM='1', MMD_MAX_STEP='768', MMD_KV_TOKENS='2048')
    assert fa == run(chunk, MMDUET_NO_FUSE='0', MMD_MAX_STEP='768', MMD_KV_TOKENS='2048')                  # deterministic
    for ra, rb in zip(fa, fb):
        assert ra == pytest.approx(rb, abs=2e-2, rel=2e-2)
    # the chunk's vectorised RoPE + KV append (per-step cos / sin table, LDS-transposed V) against the scalar kernel: same arithmetic -> same bits; the chunks start at
    # positions 0 / 640 / 1340 (partial 64-token arena blocks at both ends of the V transpose), then a decode row reads the arena back
    assert fa == run(chunk, MMDUET_NO_FUSE='0', MMDUET_NO_CHUNK_ROPE='1', MMD_MAX_STEP='768', MMD_KV_TOKENS='2048')
    odd = '70,333,129,1,64,2'
    assert run(odd, MMDUET_NO_FUSE='0', MMD_MAX_STEP='384', MMD_KV_TOKENS='1024') == run(odd, MMDUET_NO_FUSE='0', MMDUET_NO_CHUNK_ROPE='1', MMD_MAX_STEP='384', MMD_KV_TOKENS='1024')
    ref = run(long_ctx, MMDUET_NO_FUSE='1', MMD_KV_TOKENS='16384')          # unfused launch schedule throughout
    for ra, rb in zip(got[-5:], ref[-5:]):
        assert ra == pytest.approx(rb, abs=3e-2, rel=3e-2)


def test_chunk_last_layer_on_read_rows_only(true_shape):
    """A chunk's last decoder layer runs o_proj / MLP / final norm only on the rows something reads (frame-end head rows, the logit row) through the weight-streaming
    kernels; MMDUET_FULL_LAST_LAYER=1 keeps all rows on the tile GEMMs.  Same rows, same weights, other GEMM kernels -> bf16-rounding-level agreement on the chunk itself,
    and the KV arena it leaves behind is the same to the bit (a following small step reads it back)."""
    import subprocess, sys, os, json
    code = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MMD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MMD_ROOT"], "tests"))
from oracle import duet_oracle as O
from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
ocfg = O.OracleConfig(vocab_size=2048, num_hidden_layers=2, vit_layers=1)
w = {k: v for k, v in O.random_weights(ocfg, seed=3, dtype=torch.bfloat16, scale="unit").items()}
pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=2048, num_hidden_layers=2, vit_num_hidden_layers=2, vit_layers_removed=1, frame_num_tokens=49, frame_resolution=384)
m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=1, max_step_tokens=768, kv_initial_tokens=4096)
m.load_state_dict(w)
g = torch.Generator().manual_seed(11)
def rnd(S): return (torch.randn(S, 3584, generator=g) * 0.5).to(torch.bfloat16).cuda()
res = []
# single stream: 64 frame-end rows of a 640-row chunk, then one row twice + the first row of a 333-row chunk, then 65 rows (all rows computed), then a small step
c = None
for S, rows in ((640, list(range(9, 640, 10))), (333, [332, 332, 0]), (130, list(range(0, 130, 2))), (5, [4])):
    h, c = m.frame_step(rnd(S), c, rows)
    res.append(h.flatten().tolist())
# two streams in one forward: heads on one, last-row logits on the other; then each stream alone on a small step
a = m.multi_step([dict(x=rnd(300), cache=None, head_rows=[99, 199, 299], hidden='none'), dict(x=rnd(150), cache=None, head_rows=[], hidden='last')], want_logits=True)
res.append(a[0]['heads'].flatten().tolist() + a[1]['logits'][0, :16].float().cpu().tolist() + a[1]['hidden'].float().cpu().flatten()[:16].tolist())
for k in (0, 1):
    h, _ = m.frame_step(rnd(3), a[k]['cache'], [2]); res.append(h.flatten().tolist())
print("RES " + json.dumps(res))
'''
    from conftest import ROOT
    def run(**kw):
        env = dict(os.environ, MMD_ROOT=ROOT, **kw)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('RES ')][0][4:])
    a, b, a2 = run(), run(MMDUET_FULL_LAST_LAYER='1'), run()
    assert a == a2
    assert len(a[0]) == 64 * 4 and len(a[2]) == 65 * 4
    for ra, rb in zip(a, b):
        assert ra == pytest.approx(rb, abs=3e-2, rel=3e-2)
    assert a[2] == b[2]                         # 65 read rows: both runs take the all-rows path
    assert a[3] == b[3] and a[5] == b[5] and a[6] == b[6]        # later steps read only K / V, which do not depend on the last layer's tail


def test_multi_stream_step_true_shape(true_shape):
    """bf16 at the 7B widths: one forward over three arenas (a 98-row frame chunk, a 49-row frame, one decode row) against the oracle
    run stream by stream -- the merged forward takes the tile-GEMM path where the single-stream steps take the weight-streaming one."""
    m, om, cfg = true_shape
    g = torch.Generator().manual_seed(7)
    ctxs = [(torch.randn(1, n, 3584, generator=g) * 0.5).to(torch.bfloat16) for n in (130, 64, 200)]
    news = [(torch.randn(1, n, 3584, generator=g) * 0.5).to(torch.bfloat16) for n in (98, 49, 1)]
    caches, ocaches = [], []
    for c in ctxs:
        caches.append(m(inputs_embeds=c.cuda()).past_key_values)
        ocaches.append(om(inputs_embeds=c).past_key_values)
    out = m.multi_step([dict(x=news[0].cuda(), cache=caches[0], head_rows=[48, 97]), dict(x=news[1].cuda(), cache=caches[1], head_rows=[48]),
                        dict(x=news[2].cuda(), cache=caches[2], hidden='last')])
    refs = [om(inputs_embeds=x, past_key_values=oc) for x, oc in zip(news, ocaches)]
    for o, r, rows in ((out[0], refs[0], [48, 97]), (out[1], refs[1], [48])):
        want = torch.cat([r.informative_logits[0, rows], r.relevance_logits[0, rows]], dim=-1)
        assert maxerr(o['heads'], want) < TOL
    assert maxerr(out[2]['logits'][0], refs[2].logits[0, -1]) < 2 * TOL
    assert [len(o['cache']) for o in out] == [228, 113, 201]
    # every stream continues from its own arena afterwards
    nxt = (torch.randn(1, 5, 3584, generator=g) * 0.5).to(torch.bfloat16)
    for o, r in zip(out, refs):
        a = m(inputs_embeds=nxt.cuda(), past_key_values=o['cache']); b = om(inputs_embeds=nxt, past_key_values=r.past_key_values)
        assert maxerr(a.informative_logits[0, -1], b.informative_logits[0, -1]) < TOL
