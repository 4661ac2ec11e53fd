"""Pin oracle/ against vectors produced by the reference's own classes (tests/golden/make_golden.py).  CPU only."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from oracle import duet_oracle as O
from oracle.preprocess import siglip_preprocess, pil_resize_bicubic_u8
from conftest import load_npz

TOL = dict(atol=2e-5, rtol=2e-5)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.float() - b.float()).abs().max().item()
    ok = torch.allclose(a.float(), b.float(), **kw)
    assert ok, f'max abs err {err}'


def test_vision_stages(golden_model):
    tag, cfgd, w, ops = golden_model
    cfg = O.OracleConfig(**cfgd)
    tower = O.vit_forward(w, cfg, ops['pixel_values'])
    close(tower, ops['tower'])
    proj = O.connector(w, ops['tower'])
    close(proj, ops['connector'])
    for mode in ('bilinear', 'average', 'max'):
        cfg.mm_spatial_pool_mode = mode
        close(O.post_projector_pooling(cfg, ops['connector']), ops['pool_' + mode])
    cfg.mm_spatial_pool_mode = cfgd['mm_spatial_pool_mode']
    close(O.visual_embed(w, cfg, ops['pixel_values']), ops['visual_embed'])


def test_llm_step_sequence(golden_model):
    tag, cfgd, w, ops = golden_model
    cfg = O.OracleConfig(**cfgd)
    m = O.OracleModel(cfg, w)
    cache = None
    for i in range(6):
        out = m(inputs_embeds=ops[f'step{i}_in'][None], past_key_values=cache, use_cache=True, return_dict=True)
        cache = out.past_key_values
        close(out.logits[0], ops[f'step{i}_logits'], atol=5e-5)
        close(out.informative_logits[0], ops[f'step{i}_inf'])
        close(out.relevance_logits[0], ops[f'step{i}_rel'])
        assert len(cache) == int(ops[f'step{i}_kvlen'])
    emb = m.get_input_embeddings()
    for j in range(2):
        out = m(inputs_embeds=emb(ops[f'long{j}_ids']), past_key_values=cache)
        cache = out.past_key_values
        close(out.logits[0, -1], ops[f'long{j}_logits_last'], atol=5e-5)
        close(out.informative_logits[0], ops[f'long{j}_inf'])
    nt = cfg.frame_num_tokens
    out = m(inputs_embeds=ops['visual_embed'][nt:2 * nt][None], past_key_values=cache)
    close(out.logits[0, -1], ops['long2_logits_last'], atol=5e-5)
    close(out.relevance_logits[0], ops['long2_rel'])
    assert len(out.past_key_values) == int(ops['long2_kvlen'])


def test_handles_are_functional(golden_model):
    """A handle held before a forward still denotes the old context afterwards (remove_assistant_turns semantic)."""
    tag, cfgd, w, ops = golden_model
    m = O.OracleModel(O.OracleConfig(**cfgd), w)
    o1 = m(inputs_embeds=ops['step0_in'][None])
    h = o1.past_key_values
    n = len(h)
    o2 = m(inputs_embeds=ops['step1_in'][None], past_key_values=h)
    assert len(h) == n and len(o2.past_key_values) == n + ops['step1_in'].shape[0]
    o3 = m(inputs_embeds=ops['step1_in'][None], past_key_values=h)
    close(o2.informative_logits, o3.informative_logits, atol=0, rtol=0)
    assert not O.KVHandle() and bool(h)


def test_joint_embed(golden_model):
    tag, cfgd, w, ops = golden_model
    cfg = O.OracleConfig(**cfgd); cfg.v_placeholder_id = int(ops['v_placeholder_id'])
    m = O.OracleModel(cfg, w)
    close(m.joint_embed(ops['joint_ids'], ops['pixel_values'][:2])[0], ops['joint_embed'])


def test_bilinear_taps_match_torch():
    for n_in, n_out in ((27, 7), (4, 2), (5, 3), (24, 6), (9, 4)):
        x = torch.randn(1, 3, n_in, n_in)
        ref = F.interpolate(x, size=[n_out, n_out], mode='bilinear')
        taps = O.bilinear_taps(n_in, n_out)
        out = torch.zeros(1, 3, n_out, n_out)
        for oy, (y0, y1, ly) in enumerate(taps):
            for ox, (x0, x1, lx) in enumerate(taps):
                top = x[..., y0, x0] * (1 - lx) + x[..., y0, x1] * lx
                bot = x[..., y1, x0] * (1 - lx) + x[..., y1, x1] * lx
                out[..., oy, ox] = top * (1 - ly) + bot * ly
        close(out, ref, atol=1e-5)


def test_adaptive_avg_pool_matches_torch():
    x = torch.randn(2, 24 * 24, 8)
    ref = F.adaptive_avg_pool2d(x.reshape(2, 24, 24, 8).permute(0, 3, 1, 2), (7, 7)).flatten(2, 3).permute(0, 2, 1)
    close(O.adaptive_avg_pool_tokens(x, (7, 7)), ref, atol=1e-6)


def test_preprocess():
    z = load_npz('preprocess.npz')
    for tag, size in (('same', 56), ('up', 56), ('down', 56)):
        pv = siglip_preprocess(torch.from_numpy(z[f'{tag}_frames']), size)
        assert np.array_equal(pv.numpy(), z[f'{tag}_pixel_values'])
    img = z['up336_frames'][0].transpose(1, 2, 0)
    assert np.array_equal(pil_resize_bicubic_u8(img, 384).transpose(2, 0, 1), z['up336_resized_u8'][0])
    small = z['up_frames'][0].transpose(1, 2, 0)
    mine = pil_resize_bicubic_u8(small, 56).astype(np.float32) * np.float32(1 / 255)
    assert np.array_equal(((mine - np.float32(.5)) / np.float32(.5)).transpose(2, 0, 1), z['up_pixel_values'][0])


def test_repetition_penalty_rule():
    s = torch.tensor([1.0, -2.0, 3.0, 0.5])
    out = O.repetition_penalty_(s, [0, 1], 2.0)
    assert out.tolist() == [0.5, -4.0, 3.0, 0.5]
