"""End-to-end parity of the HIP model (through the Python boundary over the C ABI) with the golden vectors recorded
from the reference's own classes (fp32, tight) and with the oracle run in bf16 (stated bf16 tolerance)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from oracle import duet_oracle as O
from conftest import load_npz, load_golden_weights
from helpers import hip_model, oracle_model, product_config

F32_TOL = 3e-4           # fp32 HIP path vs reference fp32 (accumulation order only)
BF16_TOL = 6e-2          # bf16 HIP path vs oracle executed in bf16 on the same weights (logits are O(1))


def maxerr(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


@pytest.fixture(scope='module', params=['A', 'B'])
def f32(request):
    m, cfgd, w = hip_model(request.param, torch.float32)
    m.config.all_position_logits = True
    ops = {k: torch.from_numpy(v) for k, v in load_npz(f'cfg{request.param}_ops.npz').items()}
    return request.param, m, cfgd, w, ops


def test_vision_stages_fp32(f32):
    tag, m, cfgd, w, ops = f32
    px = ops['pixel_values'].cuda()
    ve_default = m.visual_embed(px)                      # shipped path: last layer + projector on the tokens the bilinear pool reads (where that is fewer than all)
    m.set_full_tower(True)                               # the taps want tower / connector outputs for every token
    try:
        ve = m.visual_embed(px)
        assert maxerr(m.vit_debug_tap(0, px.shape[0]), ops['tower'].flatten(0, 1)) < F32_TOL
        assert maxerr(m.vit_debug_tap(1, px.shape[0]), ops['connector'].flatten(0, 1)) < F32_TOL
    finally:
        m.set_full_tower(False)
    assert maxerr(ve, ops['visual_embed']) < F32_TOL
    assert maxerr(ve_default, ops['visual_embed']) < F32_TOL
    assert ve.shape == ops['visual_embed'].shape


@pytest.mark.parametrize('mode', ['average', 'max'])
def test_other_pooling_modes_fp32(mode):
    m, cfgd, w = hip_model('A', torch.float32, mm_spatial_pool_mode=mode)
    ops = {k: torch.from_numpy(v) for k, v in load_npz('cfgA_ops.npz').items()}
    ve = m.visual_embed(ops['pixel_values'].cuda())
    assert maxerr(ve, ops['pool_' + mode].flatten(0, 1)) < F32_TOL


def test_llm_step_sequence_fp32(f32):
    tag, m, cfgd, w, ops = f32
    cache = None
    for i in range(6):
        out = m(inputs_embeds=ops[f'step{i}_in'][None].cuda(), past_key_values=cache, use_cache=True, return_dict=True)
        cache = out.past_key_values
        assert out.logits.shape[1] == ops[f'step{i}_in'].shape[0]
        assert maxerr(out.logits[0], ops[f'step{i}_logits']) < F32_TOL, f'step {i} logits'
        assert maxerr(out.informative_logits[0], ops[f'step{i}_inf']) < F32_TOL
        assert maxerr(out.relevance_logits[0], ops[f'step{i}_rel']) < F32_TOL
        assert len(cache) == int(ops[f'step{i}_kvlen']) and cache.get_seq_length() == len(cache)
    emb = m.get_input_embeddings()
    for j in range(2):                                           # long context -> split-KV path of the attention kernel
        out = m(inputs_embeds=emb(ops[f'long{j}_ids'].cuda()), past_key_values=cache)
        cache = out.past_key_values
        assert maxerr(out.logits[0, -1], ops[f'long{j}_logits_last']) < F32_TOL
        assert maxerr(out.informative_logits[0], ops[f'long{j}_inf']) < F32_TOL
    nt = cfgd['frame_num_tokens']
    out = m(inputs_embeds=ops['visual_embed'][nt:2 * nt][None].cuda(), past_key_values=cache)
    assert maxerr(out.logits[0, -1], ops['long2_logits_last']) < F32_TOL
    assert maxerr(out.relevance_logits[0], ops['long2_rel']) < F32_TOL
    assert len(out.past_key_values) == int(ops['long2_kvlen'])


def test_last_position_logits_default(f32):
    tag, m, cfgd, w, ops = f32
    m.config.all_position_logits = False
    try:
        out = m(inputs_embeds=ops['step0_in'][None].cuda())
        assert out.logits.shape == (1, 1, cfgd['vocab_size'])
        assert maxerr(out.logits[:, -1:][0, 0], ops['step0_logits'][-1]) < F32_TOL
    finally:
        m.config.all_position_logits = True


def test_embeddings_and_joint_embed(f32):
    tag, m, cfgd, w, ops = f32
    emb = m.get_input_embeddings()
    assert emb(torch.zeros(1, 0, dtype=torch.long)).shape == (1, 0, cfgd['hidden_size'])          # k = 0 (test/inference.py:234)
    assert torch.equal(emb(ops['ids0']).cpu()[0], w['model.embed_tokens.weight'][ops['ids0'][0]])
    m.config.v_placeholder_id = int(ops['v_placeholder_id'])
    je = m.joint_embed(ops['joint_ids'].cuda(), ops['pixel_values'][:2].cuda())
    assert maxerr(je[0], ops['joint_embed']) < F32_TOL


def test_cache_handles_are_functional_and_guarded(f32):
    tag, m, cfgd, w, ops = f32
    assert not m.new_cache()                                      # falsy when empty (test/inference.py:229)
    o1 = m(inputs_embeds=ops['step0_in'][None].cuda())
    h = o1.past_key_values
    o2 = m(inputs_embeds=ops['step1_in'][None].cuda(), past_key_values=h)
    newer = o2.past_key_values
    ref = o2.informative_logits.clone()
    assert len(h) == ops['step0_in'].shape[0] and len(newer) == len(h) + ops['step1_in'].shape[0]
    # continue again from the OLDER handle: O(1) truncate, same numbers, and the newer handle becomes stale
    o3 = m(inputs_embeds=ops['step1_in'][None].cuda(), past_key_values=h)
    assert torch.equal(o3.informative_logits, ref)
    with pytest.raises(RuntimeError):
        m(inputs_embeds=ops['step1_in'][None].cuda(), past_key_values=newer)
    with pytest.raises(TypeError):
        m(inputs_embeds=ops['step1_in'][None].cuda(), past_key_values=object())
    with pytest.raises(ValueError):
        m(inputs_embeds=ops['step1_in'].cuda())                   # not [1,S,H]


def test_kv_arena_grows_without_changing_results():
    m, cfgd, w = hip_model('A', torch.float32)
    ops = {k: torch.from_numpy(v) for k, v in load_npz('cfgA_ops.npz').items()}
    m.kv_initial_tokens = 256
    emb = m.get_input_embeddings()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, cfgd['vocab_size'], (1, 700), generator=g)
    a = m(inputs_embeds=emb(ids.cuda()))                          # one forward (arena sized up front by growth)
    cache = None
    for s0 in range(0, 700, 100):                                 # many forwards crossing the 256 -> 512 -> 1024 growth
        o = m(inputs_embeds=emb(ids[:, s0:s0 + 100].cuda()), past_key_values=cache); cache = o.past_key_values
    assert len(cache) == 700
    assert maxerr(o.informative_logits[0, -1], a.informative_logits[0, -1]) < 1e-4


@pytest.mark.parametrize('penalty', [None, 1.15])
def test_native_generate_equals_python_loop(f32, penalty):
    from mmduet_amd.modeling_live import fast_greedy_generate
    tag, m, cfgd, w, ops = f32
    x = ops['step0_in'][None].cuda()
    out_a = torch.zeros(1, 10, dtype=torch.long, device='cuda'); out_b = torch.zeros_like(out_a)
    seen_a, seen_b = [3, 5], [3, 5]
    ids_a, ca, la = fast_greedy_generate(model=m, inputs_embeds=x, past_key_values=None, eos_token_id=-1, inplace_output_ids=out_a,
                                         repetition_penalty=penalty, generated_token_ids=seen_a)
    m.python_generate_loop = True
    try:
        ids_b, cb, lb = fast_greedy_generate(model=m, inputs_embeds=x, past_key_values=None, eos_token_id=-1, inplace_output_ids=out_b,
                                             repetition_penalty=penalty, generated_token_ids=seen_b)
    finally:
        m.python_generate_loop = False
    assert ids_a.tolist() == ids_b.tolist() and la == lb and len(ca) == len(cb) == x.shape[1] + 9
    # and against the oracle's restatement of the reference loop
    om, _, _ = oracle_model(tag)
    out_c = torch.zeros(1, 10, dtype=torch.long)
    ids_c, cc, lc = O.fast_greedy_generate(model=om, inputs_embeds=ops['step0_in'][None], past_key_values=None, eos_token_id=-1,
                                           inplace_output_ids=out_c, repetition_penalty=penalty, generated_token_ids=[3, 5])
    assert ids_a.cpu().tolist() == ids_c.tolist() and la == lc


def test_penalty_list_is_unbounded(f32):
    """models/modeling_live.py:60-66 penalises ALL ids generated so far in the video: a list beyond the device buffer's first 16 384 slots grows the buffer
    (no clamp); native loop == Python loop == oracle."""
    from mmduet_amd.modeling_live import fast_greedy_generate
    tag, m, cfgd, w, ops = f32
    x = ops['step0_in'][None].cuda()
    V = cfgd['vocab_size']
    base = [(7 * i + 3) % (V // 2) for i in range(20000)]            # 20 000 entries, half the vocabulary penalised
    outs = []
    for loop in (False, True):
        out = torch.zeros(1, 6, dtype=torch.long, device='cuda'); seen = list(base)
        m.python_generate_loop = loop
        try:
            ids, _, l = fast_greedy_generate(model=m, inputs_embeds=x, past_key_values=None, eos_token_id=-1, inplace_output_ids=out, repetition_penalty=1.3, generated_token_ids=seen)
        finally:
            m.python_generate_loop = False
        assert len(l) == 20006
        outs.append((ids.tolist(), l[-6:]))
    assert outs[0] == outs[1]
    om, _, _ = oracle_model(tag)
    ids_c, _, lc = O.fast_greedy_generate(model=om, inputs_embeds=ops['step0_in'][None], past_key_values=None, eos_token_id=-1,
                                          inplace_output_ids=torch.zeros(1, 6, dtype=torch.long), repetition_penalty=1.3, generated_token_ids=list(base))
    assert outs[0][0] == ids_c.tolist()


def test_eos_stops_generation_and_is_not_penalised(f32):
    from mmduet_amd.modeling_live import fast_greedy_generate
    tag, m, cfgd, w, ops = f32
    x = ops['step0_in'][None].cuda()
    out = torch.zeros(1, 10, dtype=torch.long, device='cuda')
    ids, _, _ = fast_greedy_generate(model=m, inputs_embeds=x, past_key_values=None, eos_token_id=-1, inplace_output_ids=out)
    eos = int(ids[0, 3])
    out2 = torch.zeros(1, 10, dtype=torch.long, device='cuda'); seen = []
    ids2, cache, seen = fast_greedy_generate(model=m, inputs_embeds=x, past_key_values=None, eos_token_id=eos, inplace_output_ids=out2,
                                             repetition_penalty=1.0, generated_token_ids=seen)
    first = ids[0].tolist().index(eos)
    assert ids2[0].tolist() == ids[0, :first + 1].tolist()         # EOS is written ...
    assert eos not in seen and len(seen) == first                  # ... but not added to the penalty list
    assert len(cache) == x.shape[1] + first                        # ... and not fed back (models/modeling_live.py:68-75)


def test_bf16_model_tracks_bf16_oracle():
    """bf16 storage/MFMA path vs the oracle executed with bf16 tensors (same rounding points)."""
    for tag in ('A', 'B'):
        m, cfgd, w = hip_model(tag, torch.bfloat16)
        m.config.all_position_logits = True
        om, _, _ = oracle_model(tag, torch.bfloat16)
        ops = {k: torch.from_numpy(v) for k, v in load_npz(f'cfg{tag}_ops.npz').items()}
        ve = m.visual_embed(ops['pixel_values'].cuda())
        vr = om.visual_embed(ops['pixel_values'])
        assert maxerr(ve, vr) < BF16_TOL * max(1.0, vr.float().abs().max().item())
        cache = ocache = None
        for i in range(5):
            x = ops[f'step{i}_in'][None]
            out = m(inputs_embeds=x.cuda(), past_key_values=cache); cache = out.past_key_values
            oo = om(inputs_embeds=x, past_key_values=ocache); ocache = oo.past_key_values
            assert maxerr(out.informative_logits, oo.informative_logits) < BF16_TOL
            assert maxerr(out.relevance_logits, oo.relevance_logits) < BF16_TOL
            assert maxerr(out.logits, oo.logits) < 2 * BF16_TOL
            # and within a looser band of the fp32 reference vectors
            assert maxerr(out.informative_logits[0], ops[f'step{i}_inf']) < 2 * BF16_TOL


def test_checkpoint_loader_and_lora_merge(tmp_path):
    """safetensors checkpoint (LLaVA names incl. the deleted last ViT layer and the pooling head) -> loader -> same
    numbers; LoRA adapter merged as W + (alpha/r) B A."""
    from safetensors.torch import save_file
    import json
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.weights import load_pretrained_into, load_lora_into, save_checkpoint
    cfgd, w = load_golden_weights('A')
    ops = {k: torch.from_numpy(v) for k, v in load_npz('cfgA_ops.npz').items()}
    extra = dict(w)
    C = cfgd['vit_hidden_size']
    extra[O.VT + f"encoder.layers.{cfgd['vit_layers']}.layer_norm1.weight"] = torch.ones(C)        # the layer LLaVA deletes
    extra[O.VT + 'head.probe'] = torch.zeros(1, 1, C)
    extra['model.image_newline'] = torch.zeros(cfgd['hidden_size'])
    ck = tmp_path / 'ckpt'
    config = product_config(cfgd)
    save_checkpoint(extra, str(ck), config)
    # LoRA on layer 0 q_proj and layer 1 down_proj + a modules_to_save head
    g = torch.Generator().manual_seed(9)
    r, alpha = 4, 8
    H, I = cfgd['hidden_size'], cfgd['intermediate_size']
    A1, B1 = torch.randn(r, H, generator=g) * 0.1, torch.randn(H, r, generator=g) * 0.1
    A2, B2 = torch.randn(r, I, generator=g) * 0.1, torch.randn(H, r, generator=g) * 0.1
    new_head = torch.randn(2, H, generator=g) * 0.1
    lo = tmp_path / 'lora'; lo.mkdir()
    save_file({'base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight': A1, 'base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight': B1,
               'base_model.model.model.layers.1.mlp.down_proj.lora_A.default.weight': A2, 'base_model.model.model.layers.1.mlp.down_proj.lora_B.default.weight': B2,
               'base_model.model.informative_head.modules_to_save.default.weight': new_head,
               'base_model.model.informative_head.original_module.weight': w['informative_head.weight']}, str(lo / 'adapter_model.safetensors'))
    json.dump({'r': r, 'lora_alpha': alpha}, open(lo / 'adapter_config.json', 'w'))
    m = VideoHeadLiveLlavaQwenForCausalLM(config, torch_dtype=torch.float32, max_vit_batch=4, max_step_tokens=128, kv_initial_tokens=256)
    missing = load_pretrained_into(m, str(ck))
    assert missing == []
    assert load_lora_into(m, str(lo)) == 2
    m.finalize()
    w2 = dict(w)
    w2['model.layers.0.self_attn.q_proj.weight'] = w['model.layers.0.self_attn.q_proj.weight'] + (alpha / r) * B1 @ A1
    w2['model.layers.1.mlp.down_proj.weight'] = w['model.layers.1.mlp.down_proj.weight'] + (alpha / r) * B2 @ A2
    w2['informative_head.weight'] = new_head
    om = O.OracleModel(O.OracleConfig(**cfgd), w2)
    x = ops['step0_in'][None]
    out = m(inputs_embeds=x.cuda()); ref = om(inputs_embeds=x)
    assert maxerr(out.informative_logits, ref.informative_logits) < F32_TOL
    assert maxerr(out.logits[0, -1], ref.logits[0, -1]) < F32_TOL
    assert maxerr(m.visual_embed(ops['pixel_values'].cuda()), ops['visual_embed']) < F32_TOL


def test_missing_weight_is_reported():
    from mmduet_amd._lib import MmduetError
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    cfgd, w = load_golden_weights('A')
    m = VideoHeadLiveLlavaQwenForCausalLM(product_config(cfgd), torch_dtype=torch.float32, max_vit_batch=2, max_step_tokens=64)
    for k, v in w.items():
        if k != 'model.layers.1.mlp.up_proj.weight':
            m.load_tensor(k, v)
    with pytest.raises(MmduetError, match='up_proj'):
        m.finalize()
    with pytest.raises(MmduetError):
        m.visual_embed(torch.zeros(1, 3, 56, 56))


def test_preprocess_336_to_384_bit_exact():
    """BASELINE frames are 336 px; the tower runs at 384: device resampler == Pillow (fixture from the reference stack)."""
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    z = load_npz('preprocess.npz')
    cfg = VideoHeadLiveLlavaQwenConfig(vocab_size=64, hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=4,
                                       num_key_value_heads=2, frame_num_tokens=49, vit_hidden_size=64, vit_intermediate_size=64,
                                       vit_num_hidden_layers=2, vit_num_attention_heads=4)
    m = VideoHeadLiveLlavaQwenForCausalLM(cfg, torch_dtype=torch.float32, max_vit_batch=1, max_step_tokens=64)
    pv = m.get_vision_tower().image_processor.preprocess(torch.from_numpy(z['up336_frames']))['pixel_values']
    u8 = torch.round((pv.cpu() * 0.5 + 0.5) * 255).to(torch.uint8)
    assert torch.equal(u8, torch.from_numpy(z['up336_resized_u8']))
    ref = (torch.from_numpy(z['up336_resized_u8']).float() * np.float32(1 / 255) - 0.5) / 0.5
    assert torch.equal(pv.cpu(), ref)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('tag', ['same', 'up', 'down'])
def test_fused_preprocess_patch_embed_is_bit_identical(dtype, tag):
    """mmd_vit_encode_frames (resampler output written straight into the patch-embed operand) == preprocess -> visual_embed, bit for bit;
    the Pillow-exact pixel fixture stays the pin of the arithmetic (test_gpu_ops.py::test_preprocess_bit_exact_with_pillow)."""
    from conftest import load_npz
    z = load_npz('preprocess.npz')
    m = hip_model('A', dtype)[0]
    fr = torch.from_numpy(z[f'{tag}_frames'])
    two = m.visual_embed(m.get_vision_tower().image_processor.preprocess(fr)['pixel_values'])
    one = m.visual_embed_frames(fr)
    assert one.shape == two.shape and torch.equal(one, two)
    big = torch.cat([fr] * 5)[:11]                       # more frames than one tower batch (max_vit_batch = 8 in the test model)
    assert torch.equal(m.visual_embed_frames(big), m.visual_embed(m.get_vision_tower().image_processor.preprocess(big)['pixel_values']))


def test_kv_stash_is_dropped_when_its_context_goes_away():
    """mmd_kv_stash / mmd_kv_unstash: a stash continues the context [0, from).  Unstash restores the tokens bit for bit when the arena stands at
    `from`; a truncate below `from` or a stream reset invalidates the stash (unstash then fails instead of resurrecting stale KV)."""
    import ctypes as C
    from mmduet_amd._lib import lib, MmduetError
    m, _, _ = hip_model('A', torch.float32)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 24, m.config.hidden_size, generator=g).cuda()
    probe = torch.randn(1, 3, m.config.hidden_size, generator=g).cuda()
    full = m(inputs_embeds=x)
    want = m(inputs_embeds=probe, past_key_values=full.past_key_values).logits.clone()
    base = m.cache_prefix(full.past_key_values, 24)
    stash = m.kv_stash(base, 10)                                  # tokens [10, 24) set aside
    other = m(inputs_embeds=probe * 2, past_key_values=m.cache_prefix(base, 10))      # something else uses slots 10..12
    assert len(other.past_key_values) == 13
    back = m.kv_unstash(stash)                                    # wrapper truncates to 10, the library restores [10, 24)
    assert len(back) == 24
    assert torch.equal(m(inputs_embeds=probe, past_key_values=back).logits, want)
    arena = back.arena
    # a stash whose context is truncated away is dropped
    stash = m.kv_stash(m.cache_prefix(back, 24), 10)
    arena.truncate(6)
    with pytest.raises(MmduetError, match='stash'):
        m.kv_unstash(stash)
    # ... and so is one across a stream reset
    h = m(inputs_embeds=x).past_key_values
    st2 = m.kv_stash(h, 8)
    assert lib().mmd_stream_reset(h.arena.h) == 0
    with pytest.raises(MmduetError, match='stash'):
        m.kv_unstash(st2)
    # the arena must stand exactly at `from`: the raw entry point refuses otherwise
    h = m(inputs_embeds=x).past_key_values
    assert lib().mmd_kv_stash(h.arena.h, 12, 24) == 0
    assert lib().mmd_kv_unstash(h.arena.h) != 0                   # len is 24, the stash continues 12


@pytest.mark.parametrize('grid,B', [(12, 1), (12, 3), (16, 1), (16, 5)])
def test_fp16_tower_small_grids_fall_back_to_the_full_last_layer(grid, B):
    """the half-precision forms of the kernels exist for GEMMs with M > 64 and attention with S >= 64 only, so the sparse last tower layer
    (U = (2 out)^2 query rows per frame) must not be taken when B * U <= 64 or U < 64 -- grid 12 / stride 4 (U = 36) and grid 16 / stride 4 at B = 1 (B * U = 64)
    failed with an opaque EHIP.  The encode must work, equal the full-tower path bit for bit, and track the oracle to the bf16 bound."""
    from oracle import duet_oracle as O
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.weights import synthetic_weights
    img = grid * 14
    out = -(-grid // 4)
    pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2,
                                        vit_hidden_size=64, vit_intermediate_size=128, vit_num_hidden_layers=3, vit_layers_removed=1, vit_num_attention_heads=4,
                                        vit_image_size=img, vit_patch_size=14, video_pooling_stride=4, frame_num_tokens=out * out, frame_resolution=img, v_placeholder='<image>')
    m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=8, max_step_tokens=256, kv_initial_tokens=512)
    assert m.tower_dtype == 'fp16'
    w = {}
    for name, t in synthetic_weights(pcfg, seed=7, device=m.device, dtype=torch.bfloat16, scale='unit'):
        m.load_tensor(name, t); w[name] = t.float()
    m.finalize()
    ocfg = O.OracleConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2, vit_hidden_size=64,
                          vit_intermediate_size=128, vit_layers=2, vit_heads=4, vit_image_size=img, vit_patch_size=14, video_pooling_stride=4, frame_num_tokens=out * out, frame_resolution=img)
    px = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(grid + B)).to(torch.bfloat16).cuda()
    ve = m.visual_embed(px)
    assert ve.shape == (B * out * out, 128) and torch.isfinite(ve.float()).all()
    m.set_full_tower(True)
    full = m.visual_embed(px)
    tap = m.vit_debug_tap(0, B)                         # legal now: the last encode ran the full last layer
    m.set_full_tower(False)
    assert tap.shape == (B * grid * grid, 64)
    ref = O.visual_embed(w, ocfg, px.float())
    scale = ref.abs().max().item()
    assert (ve.float() - ref).abs().max().item() <= 4e-2 * max(1.0, scale)
    assert (full.float() - ref).abs().max().item() <= 4e-2 * max(1.0, scale)
    if 4 * out * out < 64 or B * 4 * out * out <= 64:    # the guarded cases: both calls ran the SAME (full) schedule
        assert torch.equal(ve, full)


def test_tower_features_keeps_a_callers_full_tower_setting(f32):
    """tower_features() flips mmd_vit_set_full_tower for its own encode and must restore what it FOUND, not 0; and the debug tap refuses only when the
    last encode really ran the sparse last layer."""
    m = f32[1]
    from mmduet_amd._lib import lib
    cfg = m.config
    px = torch.randn(2, 3, cfg.vit_image_size, cfg.vit_image_size, generator=torch.Generator().manual_seed(3)).cuda()
    m.set_full_tower(True)
    m.tower_features(px)
    assert lib().mmd_vit_get_full_tower(m._ctx) == 1
    m.visual_embed(px)
    m.vit_debug_tap(0, 2)                                # still legal
    m.set_full_tower(False)
    m.tower_features(px)
    assert lib().mmd_vit_get_full_tower(m._ctx) == 0
