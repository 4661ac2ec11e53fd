"""Several video streams per GPU, batched into shared LLM forwards.

The reference runs one video per process with batch 1 (test/inference.py:341 `DataLoader(batch_size=1)`); its only seam for
more is launching more processes.  On MI355X a single stream leaves the machine badly used while it is *generating*: every
token streams the 14 GB of weights for one row.  With B streams resident (57 KB of KV per token: thousands of streams fit in
288 GB) the GEMMs of one forward can carry rows of all of them -- frame chunks of the streams that are watching, one row
of every stream that is talking -- while RoPE / KV append / attention stay per stream (mmd_frame_step_multi).  A generating
stream then rides on the other streams' chunk GEMMs for the price of its own attention.

`MultiStreamInfer` runs B unmodified `LiveInferForBenchmark` state machines ("slots").  Each slot lives in its own Python
thread, but exactly ONE thread runs at any time (semaphore hand-off, i.e. coroutines): a slot runs its driver loop until the
driver calls the model, posts that call as a request and parks; when every live slot is parked the scheduler merges the
requests into one multi-stream forward, hands the results back and resumes the slots one after the other.  The driver code
(prompt-prefix rules, thresholds, speculative chunks + replay, KV keep/drop) is the single-stream code, line for line, so
per-stream results equal the single-stream run up to GEMM accumulation order (tests/test_gpu_multistream.py).
"""
import collections
import threading
import time
import torch
from .inference import LiveInferForBenchmark
from .modeling_live import VideoHeadCausalLMOutputWithPast


MAX_TALKERS_PER_ROUND = 32          # mmd_round_multi samples for at most 32 streams per forward (include/mmduet.h); more talking slots than that are dealt over two forwards


class _Request:
    __slots__ = ('kind', 'x', 'cache', 'head_rows', 'result', 'gen')

    def __init__(self, kind, x, cache, head_rows=(), gen=None):
        self.kind, self.x, self.cache, self.head_rows, self.result, self.gen = kind, x, cache, list(head_rows), None, gen

    @property
    def rows(self):
        if self.kind == 'generate' and self.gen['ids']:          # a response in progress: the row of the token drawn last
            return 1
        return self.x.reshape(-1, self.x.shape[-1]).shape[0]


class _ModelProxy:
    """What a slot's driver sees as `model`: the real model, except that LLM forwards are posted to the scheduler."""

    def __init__(self, real, slot, max_step_tokens):
        self.__dict__.update(_real=real, _slot=slot, _rows=max_step_tokens)

    @property
    def max_step_tokens(self):
        """Rows this slot's next forward may carry (the driver sizes its frame chunk by it, inference._chunk_size): the slot's fixed share of the step, or -- `dynamic_chunks`
        -- the share of the streams that are WATCHING right now (the scheduler sets it before it resumes the slot)."""
        return self._slot.row_cap if self._slot.row_cap else self._rows

    def __getattr__(self, name):
        return getattr(self._real, name)

    def eval(self):
        return self

    def frame_step(self, inputs_embeds, past_key_values, head_rows):
        r = self._slot.post(_Request('frames', inputs_embeds, past_key_values, head_rows))
        return r['heads'], r['cache']

    def __call__(self, input_ids=None, past_key_values=None, inputs_embeds=None, **kw):
        if inputs_embeds is None:
            inputs_embeds = self._real.joint_embed(input_ids, kw.get('frames'))
        r = self._slot.post(_Request('forward', inputs_embeds, past_key_values))
        return VideoHeadCausalLMOutputWithPast(self._real, r['hidden'], r['cache'])

    forward = __call__

    def greedy_generate(self, inputs_embeds, past_key_values, eos_token_id, max_new_tokens, repetition_penalty=None, generated_token_ids=None):
        """models/modeling_live.py:51-77 with one scheduler round per token (same rule as mmd_greedy_generate: the EOS token is
        written but neither fed back nor penalised; HF repetition penalty over every token generated so far in this video)."""
        if hasattr(self._real, 'round_multi') and not self._slot.sched.python_decode:
            # the whole response is ONE request: the scheduler advances it a token per round (mmd_round_multi samples on the device) and wakes this slot when it is complete
            gen = dict(slot=self._slot, eos=eos_token_id, max_new=int(max_new_tokens), ids=[], penalty=repetition_penalty,
                       prev=list(generated_token_ids) if (generated_token_ids is not None and repetition_penalty is not None) else None)
            r = self._slot.post(_Request('generate', inputs_embeds, past_key_values, gen=gen))
            if generated_token_ids is not None and repetition_penalty is not None:
                generated_token_ids.extend(t for t in r['ids'] if t != eos_token_id)
            return r['ids'], r['cache']
        pen = float(repetition_penalty) if repetition_penalty is not None else 0.0
        seen = generated_token_ids if (generated_token_ids is not None and pen > 0) else None
        x, cache, ids = inputs_embeds, past_key_values, []
        for _ in range(max_new_tokens):
            r = self._slot.post(_Request('decode', x, cache))
            cache = r['cache']
            scores = r['logits'][0]
            if seen:
                idx = torch.as_tensor(seen, dtype=torch.long, device=scores.device)
                picked = scores[idx]
                scores = scores.clone()
                scores[idx] = torch.where(picked < 0, picked * pen, picked / pen)
            tok = int(scores.argmax(-1))
            if seen is not None and tok != eos_token_id:
                seen.append(tok)
            ids.append(tok)
            if tok == eos_token_id:
                break
            x = self._real.get_input_embeddings()(torch.tensor([[tok]], device=getattr(self._real, 'device', 'cpu')))
        return ids, cache


class _Slot(threading.Thread):
    def __init__(self, sched, index):
        super().__init__(daemon=True, name=f'mmduet-slot{index}')
        self.sched, self.index = sched, index
        self.go = threading.Semaphore(0)
        self.request, self.finished, self.error = None, False, None
        self.driver = None
        self.sampler = None                  # mmd_sampler of this slot (created at its first response)
        self.row_cap = 0                     # dynamic_chunks: rows this slot's next forward may carry (0 = the fixed share)

    # -- called on the slot's thread ---------------------------------------------------------------------------------
    def post(self, request):
        self.request = request
        self.sched.parked.release()          # hand the baton to the scheduler ...
        self.go.acquire()                    # ... and wait for the result
        self.request = None
        if isinstance(request.result, BaseException):
            raise request.result
        return request.result

    def run(self):
        self.go.acquire()
        try:
            if getattr(self.sched.model.device, 'type', 'cpu') == 'cuda':
                torch.cuda.set_device(self.sched.model.device)
            with torch.no_grad():
                while self.sched.todo:
                    n, video = self.sched.todo.popleft()
                    if callable(video):                      # lazy entry: the clip is read when a slot takes it, not when the list is built
                        video = video()
                        if video is None:                    # unreadable clip: skipped like the reference does (test/datasets.py:102-104)
                            continue
                    d = self.driver = self.sched._make_driver(self, video.get('args') or self.sched.args)
                    for k, v in (video.get('driver_attrs') or {}).items():
                        setattr(d, k, v)
                    if video.get('fps'):
                        d.set_fps(fps=video['fps'])
                    if video['frames'].dtype == torch.uint8:
                        d.input_video_stream(video['frames'])
                    else:                                    # a pre-extracted feature tensor [T, tokens, C] (mmduet_amd/features.py)
                        d.input_feature_stream(video['frames'])
                    d.input_query_stream(video['conversation'])
                    responses = d.inference()
                    self.sched.results[n] = dict(responses=responses, debug_data=list(d.debug_data_list), forward_calls=d.forward_calls,
                                                 replayed_frames=d.replayed_frames, response_token_ids=list(d.response_token_ids),
                                                 generated_token_ids=[int(t) for t in d.generated_token_ids],
                                                 final_kv_len=len(d.past_key_values) if d.past_key_values else 0)
                    if self.sched.on_result is not None:
                        self.sched.on_result(n, video, self.sched.results[n])
                    # give the stream's KV arena back to the model's pool NOW: the slot <-> driver <-> proxy cycle otherwise keeps it alive until the cyclic collector
                    # runs, and the next video of this slot maps a fresh arena (mmd_stream_create: ~90 ms of address reservation + page mapping each)
                    if not self.sched.keep_drivers:
                        d.past_key_values = None
                        self.driver = d = None
        except BaseException as e:          # surfaces in MultiStreamInfer.run()
            self.error = e
        finally:
            self.finished = True
            self.sched.parked.release()


class MultiStreamInfer:
    """B concurrent `LiveInferForBenchmark` streams on one GPU.

    args: the driver's LiveTestArguments (every slot gets the same flags); model / tokenizer as for the single-stream driver.
    `run(videos)`: videos = list of dict(frames=uint8 [T,3,R,R], conversation=[turns], fps=optional, args=optional per-video
    LiveTestArguments, driver_attrs=optional dict set on the video's driver); returns, in input order, dict(responses,
    debug_data, response_token_ids, generated_token_ids, final_kv_len, forward_calls, replayed_frames) per video."""

    def __init__(self, args, model=None, tokenizer=None, n_slots=4, driver_cls=LiveInferForBenchmark, vit_lookahead_batches=2, dynamic_chunks=False, max_chunk_frames=39):
        if n_slots < 1:
            raise ValueError('n_slots must be >= 1')
        if model is None:
            first = driver_cls(args)
            model, tokenizer = first.model, first.tokenizer
        self.model, self.tokenizer, self.n_slots = model, tokenizer, n_slots
        self.args, self.driver_cls = args, driver_cls
        self.vit_lookahead_batches = vit_lookahead_batches
        self.per_slot_rows = max(256, model.max_step_tokens // n_slots)
        # dynamic_chunks: while some streams talk (one row each) the streams that watch take their share of the step as well -- frames_per_forward x n_slots / n_watching frames
        # per forward, at most max_chunk_frames.  A merged forward costs ~8 ms + 11 us per row (weights streamed once, one launch set): the fewer, fuller rounds the watchers
        # need, the more of a response's 32 rounds are the cheap all-talking kind; against it stands the replay of a chunk's tail behind a response, which grows with the chunk.
        self.dynamic_chunks = bool(dynamic_chunks)
        self.max_chunk_frames = int(max_chunk_frames)
        self.base_frames = max(1, int(getattr(args, 'frames_per_forward', 1) or 1))
        self.keep_drivers = False             # True: a finished video's driver (and its KV handle) stays reachable as slot.driver (tests that read the arena afterwards)
        self.rounds = self.merged_rows = 0
        self.round_log = None                 # set to a list to record (kinds, rows, t_start, t_end) per merged forward (kinds: one letter per segment: f = frame chunk, f of 'forward' = query turn, g = a talking stream's row(s), d = a decode row of the python_decode loop)
        self.exec_seconds = 0.0               # time inside the merged forwards (launch + the one sync), the rest is driver host work
        self.python_decode = False            # True: responses decode through per-token Python (logits tensor, host arg-max, embedding call per slot and token) -- the
                                              # round-5 form, kept as the cross-check of the native rounds and for duck-typed models without `round_multi`
        self._vit_stream = None

    def _make_driver(self, slot, args):
        d = self.driver_cls(args, model=_ModelProxy(self.model, slot, self.per_slot_rows), tokenizer=self.tokenizer)
        if self.dynamic_chunks:
            d.frames_per_forward = max(d.frames_per_forward, self.max_chunk_frames)          # the cap now comes from the slot's row share (`_ModelProxy.max_step_tokens`)
            slot.row_cap = self._row_cap(self.n_slots)
        if getattr(d, 'overlap_vision', False):
            # ONE tower stream for all slots (the tower's workspace is per context), batches issued just ahead of their use
            if self._vit_stream is None and self.model.device.type == 'cuda':
                self._vit_stream = torch.cuda.Stream(device=self.model.device, priority=0)
            d._vit_stream = self._vit_stream
            d.vit_lookahead_batches = self.vit_lookahead_batches
        return d

    def _row_cap(self, n_watching):
        """Rows a watching slot may post when n_watching streams watch: the step's frame budget (base frames x slots) dealt to the watchers, capped; + room for a text prefix."""
        nt = int(getattr(self.model.config, 'frame_num_tokens', 49) or 49)
        k = min(self.max_chunk_frames, max(self.base_frames, (self.base_frames * self.n_slots) // max(1, n_watching)))
        k = min(k, max(1, (self.model.max_step_tokens - 128 * max(1, n_watching) - self.n_slots) // (nt * max(1, n_watching))))
        return 128 + nt * k

    def _execute(self, requests):
        """Merge the parked requests into as few multi-stream forwards as the row budget allows."""
        budget = self.model.max_step_tokens
        group, rows, talkers = [], 0, 0
        groups = []
        for r in requests:
            t = 1 if r.kind == 'generate' else 0
            if group and (rows + r.rows > budget or talkers + t > MAX_TALKERS_PER_ROUND):
                groups.append(group); group, rows, talkers = [], 0, 0
            group.append(r); rows += r.rows; talkers += t
        if group:
            groups.append(group)
        t0 = time.perf_counter()
        native = hasattr(self.model, 'round_multi') and not self.python_decode
        if native:
            # the talking streams' single rows go BEHIND the watching streams' chunks: consecutive equal-row segments share one attention launch per layer
            groups = [sorted(g, key=lambda r: r.kind == 'generate' and bool(r.gen['ids'])) for g in groups]
        for group in groups:
            tg = time.perf_counter()
            if native:
                self._round_native(group)
                self.rounds += 1
                self.merged_rows += sum(r.rows for r in group)
                if self.round_log is not None:
                    self.round_log.append((''.join(r.kind[0] for r in group), [r.rows for r in group], tg, time.perf_counter()))
                continue
            segs = [dict(x=r.x, cache=r.cache, head_rows=r.head_rows, hidden={'frames': 'none', 'forward': 'all', 'decode': 'last'}[r.kind]) for r in group]
            try:
                out = self.model.multi_step(segs)
                for r, o in zip(group, out):
                    r.result = o
            except BaseException as e:
                for r in group:
                    r.result = e
            self.rounds += 1
            self.merged_rows += sum(r.rows for r in group)
            if self.round_log is not None:
                self.round_log.append((''.join(r.kind[0] for r in group), [r.rows for r in group], tg, time.perf_counter()))
        self.exec_seconds += time.perf_counter() - t0

    def _round_native(self, group):
        """One merged forward through mmd_round_multi.  A 'generate' request stays posted for the whole response: its first round carries the prompt rows, every later
        round the ONE row of the token drawn in the round before (gathered on the device); the request completes -- `result` set, `gen['done']` -- at EOS or the cap."""
        segs = []
        for r in group:
            if r.kind == 'generate':
                g = r.gen
                if 'sampler' not in g:           # first round of this response
                    slot = g['slot']
                    if slot.sampler is None:
                        slot.sampler = self.model.new_sampler()
                    g['sampler'] = slot.sampler
                    g['sampler'].begin(g['eos'], g['penalty'], g['prev'], g['max_new'])
                    segs.append(dict(x=r.x, cache=r.cache, sampler=g['sampler'], sample=True))
                else:
                    segs.append(dict(x=None, cache=g['cache'], sampler=g['sampler'], feed=True, sample=True))
            else:
                segs.append(dict(x=r.x, cache=r.cache, head_rows=r.head_rows))
        try:
            if any(r.kind == 'forward' for r in group):          # a query turn wants every hidden row back (lazy logits): the general entry point, rare
                out = self._round_with_hidden(group, segs)
            else:
                out = self.model.round_multi(segs)
        except BaseException as e:
            for r in group:
                r.result = e
                if r.kind == 'generate':
                    r.gen['done'] = True
            return
        for r, o in zip(group, out):
            if r.kind == 'generate':
                g = r.gen
                g['cache'] = o['cache']
                g['ids'].append(o['token'])
                if o['token'] == g['eos'] or len(g['ids']) >= g['max_new']:
                    g['done'] = True
                    r.result = dict(ids=g['ids'], cache=g['cache'])
            elif r.kind == 'forward':
                r.result = dict(hidden=o['hidden'], cache=o['cache'])
            else:
                r.result = dict(heads=o['heads'], cache=o['cache'])

    def _round_with_hidden(self, group, segs):
        """A round that contains a 'forward' request (all hidden rows of a query turn wanted -- lazy logits): the watching streams' rows go through multi_step, which
        returns hidden states; the talking streams' rows follow in a round of their own."""
        out = [None] * len(group)
        idx = [i for i, r in enumerate(group) if r.kind != 'generate']
        res = self.model.multi_step([dict(x=group[i].x, cache=group[i].cache, head_rows=group[i].head_rows, hidden='all' if group[i].kind == 'forward' else 'none') for i in idx],
                                    want_logits=False)
        for i, o in zip(idx, res):
            out[i] = o
        gi = [i for i, r in enumerate(group) if r.kind == 'generate']
        if gi:
            res = self.model.round_multi([segs[i] for i in gi])
            for i, o in zip(gi, res):
                out[i] = o
        return out

    def run(self, videos, on_result=None):
        """`videos` entries may be callables returning the dict (or None to skip): they are evaluated when a slot takes them, so a long test
        file is never resident at once; `on_result(index, video, result)` is called as each video completes (records can be flushed then)."""
        self.on_result = on_result
        self.todo = collections.deque(enumerate(videos))
        self.results = [None] * len(videos)
        self.parked = threading.Semaphore(0)
        slots = [_Slot(self, i) for i in range(min(self.n_slots, max(1, len(videos))))]
        for s in slots:
            s.start()
        running = list(slots)
        for s in running:                      # first leg of every slot: up to its first model call
            s.go.release(); self.parked.acquire()
        while True:
            for s in running:
                if s.error is not None:
                    raise s.error
            running = [s for s in running if not s.finished]
            if not running:
                break
            self._execute([s.request for s in running])
            talking = lambda s: s.request is not None and s.request.kind == 'generate' and s.request.result is None and not s.request.gen.get('done')
            for s in running:                  # strictly one thread at a time: resume, wait until it parks again or ends
                if talking(s):
                    continue                   # mid-response: the slot stays parked, the scheduler feeds its next token itself
                if self.dynamic_chunks:        # (slots resumed before this one have posted already: the ones that started a response count as talking)
                    s.row_cap = self._row_cap(sum(1 for o in running if not o.finished and not (o.request is not None and o.request.kind == 'generate' and o.request.result is None)))
                s.go.release(); self.parked.acquire()
        return self.results
