"""A CPU stand-in for libmmduet_hip.so's C ABI, backed by the oracle.  TEST INFRASTRUCTURE (build container only).

Purpose: run the REFERENCE's own stream driver classes, unchanged, over the PRODUCT's Python surface
(mmduet_amd.modeling_live exactly as shipped: KV handles, arena pool, lazy logits, the generation loop, the checkpoint / tokenizer
loaders) in a container that has no GPU.  Only the shared library is replaced: `FakeLib` implements the entry points of
include/mmduet.h that this path calls, with the same argument meaning (raw pointers, sizes, 0 = ok), computing with oracle/duet_oracle.py.
Nothing here is importable by the product; tests/ref_conformance_runner.py installs it with `mmduet_amd._lib.lib` patched."""
import ctypes as C
import torch
import torch.nn.functional as F
from oracle import duet_oracle as O
from oracle.preprocess import siglip_preprocess

_DT = {0: torch.float32, 1: torch.bfloat16}
_POOL = {0: 'bilinear', 1: 'average', 2: 'max'}


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    return p.value or 0


def _view(p, numel, dtype):
    """torch view (no copy) of `numel` elements of `dtype` at raw address p."""
    if numel == 0:
        return torch.empty(0, dtype=dtype)
    nbytes = numel * torch.empty(0, dtype=dtype).element_size()
    return torch.frombuffer((C.c_char * nbytes).from_address(_addr(p)), dtype=dtype)


class _Ctx:
    def __init__(self, cfg):
        self.s = cfg
        self.dtype = _DT[cfg.dtype]
        self.w = {}
        self.cfg = O.OracleConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                  num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                                  rope_theta=float(cfg.rope_theta), rms_norm_eps=float(cfg.rms_norm_eps), vit_hidden_size=cfg.vit_hidden,
                                  vit_intermediate_size=cfg.vit_intermediate, vit_layers=cfg.vit_layers, vit_heads=cfg.vit_heads,
                                  vit_image_size=cfg.vit_image, vit_patch_size=cfg.vit_patch, vit_layer_norm_eps=float(cfg.vit_ln_eps),
                                  vit_post_layernorm=bool(cfg.vit_post_layernorm), video_pooling_stride=cfg.pool_stride,
                                  mm_spatial_pool_mode=_POOL[cfg.pool_mode], frame_num_tokens=cfg.frame_num_tokens, frame_resolution=cfg.vit_image)
        self.err = b''
        self.final = False


class _Stream:
    """One KV arena: contiguous context of `len` tokens, O(1) truncate (mmd_kv_truncate), append by mmd_llm_step."""

    def __init__(self, ctx, cap):
        self.ctx, self.cap, self.handle, self.len = ctx, cap, O.KVHandle(), 0

    def prefix(self):
        if self.len == 0:
            return None
        return O.KVHandle([k[:, :self.len] for k in self.handle.k], [v[:, :self.len] for v in self.handle.v])


class FakeLib:
    """Entry points of include/mmduet.h used by mmduet_amd.modeling_live on the reference driver's path.  Handles are small integers."""

    def __init__(self):
        self._objs, self._next = {}, 1
        self.calls = {}

    def _new(self, obj):
        h = self._next; self._next += 1
        self._objs[h] = obj
        return h

    def _get(self, h):
        return self._objs[_addr(h)]

    def __getattr__(self, name):
        if name.startswith('mmd_'):
            raise AttributeError(f'cabi_oracle_shim.FakeLib: {name} is not on the reference driver path (add it to the shim if the path grew)')
        raise AttributeError(name)

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # ---- context / weights -------------------------------------------------------------------------------------------------------
    def mmd_create(self, cfg_ref, device, out_ref):
        self._count('mmd_create')
        out_ref._obj.value = self._new(_Ctx(cfg_ref._obj))
        return 0

    def mmd_destroy(self, h):
        self._objs.pop(_addr(h), None)

    def mmd_last_error(self, h):
        try:
            return self._get(h).err
        except Exception:
            return b'?'

    def mmd_set_stream(self, h, stream):
        return 0

    def mmd_set_rope_inv_freq(self, h, ptr, n):
        c = self._get(h)
        inv = _view(ptr, n, torch.float32).clone()
        d = c.cfg.head_dim
        want = 1.0 / (c.cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        assert torch.equal(inv, want), 'rope table handed to the library differs from the reference expression'
        return 0

    def mmd_load_tensor(self, h, name, ptr, dtype, shape, rank, on_dev):
        self._count('mmd_load_tensor')
        c = self._get(h)
        shp = [int(shape[i]) for i in range(rank)]
        n = 1
        for s in shp:
            n *= s
        c.w[name.decode()] = _view(ptr, n, _DT[dtype]).clone().view(shp).to(c.dtype)
        return 0

    def mmd_finalize_weights(self, h):
        c = self._get(h)
        missing = [k for k in O.weight_shapes(c.cfg) if k not in c.w and 'post_layernorm' not in k]
        if missing:
            c.err = f'missing tensors: {missing[:3]}'.encode()
            return -1
        c.final = True
        return 0

    def mmd_weight_bytes(self, h):
        return sum(t.numel() * t.element_size() for t in self._get(h).w.values())

    # ---- vision --------------------------------------------------------------------------------------------------------------------
    def mmd_preprocess_frames(self, h, frames, T, R, out):
        self._count('mmd_preprocess_frames')
        c = self._get(h)
        fr = _view(frames, T * 3 * R * R, torch.uint8).view(T, 3, R, R)
        size = c.cfg.vit_image_size
        _view(out, T * 3 * size * size, c.dtype).view(T, 3, size, size).copy_(siglip_preprocess(fr.numpy(), size).to(c.dtype))
        return 0

    def mmd_vit_encode(self, h, px, B, out):
        self._count('mmd_vit_encode')
        c = self._get(h)
        size = c.cfg.vit_image_size
        x = _view(px, B * 3 * size * size, c.dtype).view(B, 3, size, size)
        e = O.visual_embed(c.w, c.cfg, x)
        _view(out, e.numel(), c.dtype).view_as(e).copy_(e)
        return 0

    # ---- language ------------------------------------------------------------------------------------------------------------------
    def mmd_embed_tokens(self, h, ids, k, out):
        self._count('mmd_embed_tokens')
        c = self._get(h)
        i = _view(ids, k, torch.int64)
        H = c.cfg.hidden_size
        _view(out, k * H, c.dtype).view(k, H).copy_(c.w['model.embed_tokens.weight'][i])
        return 0

    def mmd_stream_create(self, h, tokens, out_ref):
        self._count('mmd_stream_create')
        out_ref._obj.value = self._new(_Stream(self._get(h), int(tokens)))
        return 0

    def mmd_stream_destroy(self, h):
        self._objs.pop(_addr(h), None)

    def mmd_kv_len(self, h):
        return self._get(h).len

    def mmd_kv_capacity(self, h):
        return self._get(h).cap

    def mmd_kv_truncate(self, h, n):
        self._count('mmd_kv_truncate')
        s = self._get(h)
        if n < 0 or n > s.len:
            s.ctx.err = b'kv_truncate outside the context'
            return -34
        s.len = int(n)
        return 0

    def mmd_llm_step(self, h, sh, x, S, hidden):
        self._count('mmd_llm_step')
        c, s = self._get(h), self._get(sh)
        H = c.cfg.hidden_size
        xin = _view(x, S * H, c.dtype).view(S, H)
        hid, cache = O.llm_forward(c.w, c.cfg, xin, s.prefix())
        s.handle, s.len = cache, s.len + S
        _view(hidden, S * H, c.dtype).view(S, H).copy_(hid)
        return 0

    def mmd_lm_head(self, h, rows, M, out):
        self._count('mmd_lm_head')
        c = self._get(h)
        r = _view(rows, M * c.cfg.hidden_size, c.dtype).view(M, -1)
        _view(out, M * c.cfg.vocab_size, torch.float32).view(M, -1).copy_(F.linear(r, c.w['lm_head.weight']).float())
        return 0

    def mmd_video_heads(self, h, rows, M, out):
        self._count('mmd_video_heads')
        c = self._get(h)
        r = _view(rows, M * c.cfg.hidden_size, c.dtype).view(M, -1)
        o = torch.cat([F.linear(r, c.w['informative_head.weight']), F.linear(r, c.w['relevance_head.weight'])], -1).float()
        _view(out, M * 4, torch.float32).view(M, 4).copy_(o)
        return 0

    # ---- entry points of the PRODUCT driver's path (mmduet_amd.inference / multistream / __main__): tests/cli_shim_runner.py ----------
    def mmd_set_tower_share(self, h, n):
        return 0

    def mmd_stream_reset(self, sh):
        self._get(sh).len = 0
        return 0

    def mmd_vit_encode_frames(self, h, frames, B, R, out):
        self._count('mmd_vit_encode_frames')
        c = self._get(h)
        fr = _view(frames, B * 3 * R * R, torch.uint8).view(B, 3, R, R)
        e = O.visual_embed(c.w, c.cfg, siglip_preprocess(fr.numpy(), c.cfg.vit_image_size).to(c.dtype))
        _view(out, e.numel(), c.dtype).view_as(e).copy_(e)
        return 0

    def _step(self, c, s, xin):
        hid, cache = O.llm_forward(c.w, c.cfg, xin, s.prefix())
        s.handle, s.len = cache, s.len + xin.shape[0]
        return hid

    def _heads(self, c, rows):
        return torch.cat([F.linear(rows, c.w['informative_head.weight']), F.linear(rows, c.w['relevance_head.weight'])], -1).float()

    def mmd_frame_step(self, h, sh, x, S, rows, n_rows, res):
        self._count('mmd_frame_step')
        c, s = self._get(h), self._get(sh)
        H = c.cfg.hidden_size
        hid = self._step(c, s, _view(x, S * H, c.dtype).view(S, H))
        o = self._heads(c, hid[[int(rows[i]) for i in range(n_rows)]]).reshape(-1).tolist()
        for i, v in enumerate(o):
            res[i] = v
        return 0

    def mmd_frame_step_multi(self, h, streams, seg_rows, n_segs, x, head_rows, n_head, res, hid_rows, n_hid, hidden_out, logits_out):
        self._count('mmd_frame_step_multi')
        c = self._get(h)
        H = c.cfg.hidden_size
        total = sum(int(seg_rows[j]) for j in range(n_segs))
        xin = _view(x, total * H, c.dtype).view(total, H)
        hid, at = [], 0
        for j in range(n_segs):
            S = int(seg_rows[j])
            hid.append(self._step(c, self._get(streams[j]), xin[at:at + S])); at += S
        hid = torch.cat(hid)
        if n_head:
            o = self._heads(c, hid[[int(head_rows[i]) for i in range(n_head)]]).reshape(-1).tolist()
            for i, v in enumerate(o):
                res[i] = v
        if n_hid:
            rows = hid[[int(hid_rows[i]) for i in range(n_hid)]]
            _view(hidden_out, n_hid * H, c.dtype).view(n_hid, H).copy_(rows)
            if _addr(logits_out):
                _view(logits_out, n_hid * c.cfg.vocab_size, torch.float32).view(n_hid, -1).copy_(F.linear(rows, c.w['lm_head.weight']).float())
        return 0

    # ---- scheduler rounds with the sampling "on the device" (mmd_sampler_* / mmd_round_multi, include/mmduet.h) -----------------------------------
    def mmd_sampler_create(self, h, out_ref):
        out_ref._obj.value = self._new(dict(ctx=_addr(h), tok=0, prev=[], eos=-1, pen=0.0))
        return 0

    def mmd_sampler_destroy(self, sp):
        self._objs.pop(_addr(sp), None)

    def mmd_sampler_begin(self, sp, eos, pen, prev, n_prev, max_new):
        s = self._get(sp)
        s.update(eos=int(eos), pen=float(pen), prev=[int(prev[i]) for i in range(n_prev)] if pen > 0 else [])
        return 0

    def mmd_sampler_prev_len(self, sp):
        return len(self._get(sp)['prev'])

    def mmd_round_multi(self, h, streams, seg_rows, n_segs, seg_embeds, samplers, seg_flags, head_rows, n_head, res, toks):
        """One merged forward: FEED rows are the embedding of the sampler's last token; SAMPLE segments draw the next token from their last row
        (models/modeling_live.py:60-72: repetition penalty over the sampler's list, first arg-max, EOS neither fed back nor penalised)."""
        self._count('mmd_round_multi')
        c = self._get(h)
        H = c.cfg.hidden_size
        hid, at = [], 0
        for j in range(n_segs):
            S, fl = int(seg_rows[j]), int(seg_flags[j])
            sp = self._get(samplers[j]) if fl else None
            if fl & 1:
                assert S == 1
                xin = c.w['model.embed_tokens.weight'][sp['tok']][None].to(c.dtype)
            else:
                xin = _view(seg_embeds[j], S * H, c.dtype).view(S, H)
            hj = self._step(c, self._get(streams[j]), xin)
            hid.append(hj); at += S
            toks[j] = -1
            if fl & 2:
                scores = F.linear(hj[-1:], c.w['lm_head.weight']).float()[0]
                if sp['pen'] > 0 and sp['prev']:
                    idx = torch.as_tensor(sp['prev'], dtype=torch.long)
                    picked = scores[idx]
                    scores[idx] = torch.where(picked < 0, picked * sp['pen'], picked / sp['pen'])
                tok = int(scores.argmax(-1))
                sp['tok'] = tok
                if sp['pen'] > 0 and tok != sp['eos']:
                    sp['prev'].append(tok)
                toks[j] = tok
        hid = torch.cat(hid)
        if n_head:
            o = self._heads(c, hid[[int(head_rows[i]) for i in range(n_head)]]).reshape(-1).tolist()
            for i, v in enumerate(o):
                res[i] = v
        return 0

    def mmd_greedy_generate(self, h, sh, x, S, eos, pen, prev, n_prev_ref, cap, out_ids, max_new, n_out_ref):
        """models/modeling_live.py:51-77: the last token is written, never fed; HF repetition penalty over `prev` (grown in place)."""
        self._count('mmd_greedy_generate')
        c, s = self._get(h), self._get(sh)
        H = c.cfg.hidden_size
        n_prev = n_prev_ref._obj
        seen = [int(prev[i]) for i in range(n_prev.value)]
        xin = _view(x, S * H, c.dtype).view(S, H)
        n = 0
        for n in range(max_new):
            hid = self._step(c, s, xin)
            scores = F.linear(hid[-1:], c.w['lm_head.weight']).float()[0]
            if pen > 0 and seen:
                idx = torch.as_tensor(seen, dtype=torch.long)
                picked = scores[idx]
                scores[idx] = torch.where(picked < 0, picked * pen, picked / pen)
            tok = int(scores.argmax(-1))
            if pen > 0 and tok != eos:
                seen.append(tok)
            out_ids[n] = tok
            if tok == eos:
                break
            xin = c.w['model.embed_tokens.weight'][tok][None].to(c.dtype)
        n_out_ref._obj.value = n + 1 if max_new > 0 else 0
        if pen > 0:
            assert len(seen) <= cap
            for i, t in enumerate(seen):
                prev[i] = t
            n_prev.value = len(seen)
        return 0

    def mmd_kv_stash(self, sh, start, end):
        self._count('mmd_kv_stash')
        s = self._get(sh)
        if not (0 <= start <= end <= s.len):
            return -34
        s.stash = (int(start), int(end), [k[:, start:end].clone() for k in s.handle.k], [v[:, start:end].clone() for v in s.handle.v])
        return 0

    def mmd_kv_unstash(self, sh):
        self._count('mmd_kv_unstash')
        s = self._get(sh)
        st = getattr(s, 'stash', None)
        if st is None or s.len != st[0]:
            return -22
        start, end, ks, vs = st
        s.handle = O.KVHandle([torch.cat([k[:, :start], t], 1) for k, t in zip(s.handle.k, ks)], [torch.cat([v[:, :start], t], 1) for v, t in zip(s.handle.v, vs)])
        s.len, s.stash = end, None
        return 0
