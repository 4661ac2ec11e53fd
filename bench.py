#!/usr/bin/env python
"""bench.py -- frames/sec of the streaming video-text-duet forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: bench.py launches its own
                                                            N ranks through torch.distributed.run before touching a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic video stream per GPU.  `--config`:
  stream300  (default; BASELINE.json configs[1]) T = 300 1-fps uint8 frames [T,3,336,336] resident in HBM -> device preprocess
             (Pillow-exact bicubic to 384, normalise) -> SigLIP tower + projector + bilinear pooling (batches of 35 frames) ->
             causal LLaVA-OV-Qwen2-7B steps over the growing interleaved KV arena (one user query at t=0) -> informative /
             relevance head logits -> greedy per-frame response decision on the host -> greedy text generation when a frame fires.
  ground600  (configs[2]) 600 frames, grounding mode (`--stream_end_prob_threshold 1`: scores only, never generates,
             scripts/inference/charades.sh:8-12).
  qvh        (configs[3]) 150-frame grounding streams (QVHighlights clips, scripts/inference/qvh.sh:12), `--streams-per-gpu`
             concurrent streams per rank in shared forwards, scores of all ranks met by ONE RCCL all-gather.
Weights: seeded random init at the true shapes, bf16 (no checkpoints exist offline).  With N > 1 every rank runs its own
streams (weak scaling) and the per-frame scores are all-gathered over RCCL inside the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events on the launch stream) and `cpu_baseline`
(the oracle on the host cores, bounded sample).
"""
import argparse, json, math, os, random, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0      # dense bf16 MFMA peak
PMC_CLOCK = ('profiles/r06_pmc_clock.json',)          # tools/pmc_clock.py over the committed SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass of this workload
PMC_TRAFFIC = ('profiles/r06_pmc_traffic.json', 'profiles/history/r05_pmc_traffic.json', 'profiles/history/r04_pmc_traffic.json', 'profiles/history/r03_pmc_traffic.json')      # newest first

CONFIGS = {
    # name: (frames, responses, frames_per_forward, streams_per_gpu, workload text)
    'stream300': dict(frames=300, responses=4, k=26, streams=1, mode='response',
                      text='query at t=0, greedy per-frame response decision'),
    'ground600': dict(frames=600, responses=0, k=39, streams=1, mode='grounding',          # k re-swept for grounding mode in round 3 (profiles/r03_k_sweep_*.json): 39 > 26 by 1.6 %, flat beyond
                      text='grounding mode (threshold 1: scores only, no generation), KV grows to 29.4 k tokens'),
    'qvh': dict(frames=150, responses=0, k=30, streams=1, mode='grounding',
                text='QVHighlights-style 150-frame grounding streams, scores all-gathered over RCCL'),
    # configs[4]: scripts/inference/youcook2.sh:12-14 (--stream_end_score_sum_threshold 2 --remove_assistant_turns true, 0.5 fps sampling of
    # ~5-10 min cooking videos), fp8 e4m3 weights.  Responses are pinned to 12 seeded frames (a YouCook2 video has ~8 annotated steps).
    'youcook2': dict(frames=600, responses=12, k=26, streams=1, mode='response', remove=True, weights='fp8',
                     text='YouCook2-style dense captioning: running-sum decision rule, assistant turns removed from the context, fp8 e4m3 LLM weights'),
    # SURVEY.md section 8(d) "native-336 variant": the secondary encoder of models/vision_live.py:61 at its real shape (CLIP-L/14-336: 24 layers, width 1024, 577 tokens),
    # 336-px frames with no resize, pooled to 6x6 (+CLS) tokens -- the offline feature-extraction path (data/utils.py:99-117).  Vision side only; reported separately,
    # never mixed with the 384 path
    'native336': dict(frames=300, responses=0, k=1, streams=1, mode='vision',
                      text='CLIP-L/14-336 secondary encoder (models/vision_live.py:34-64), native 336-px frames, adaptive pooling to 6x6 + CLS tokens; vision encode only (feature extraction)'),
}


def effective_cpus():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a pool box shows all
    256 hardware threads but grants a fraction; torch would otherwise start one spinning worker per visible thread)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def thread_cpu_times():
    """{tid: (comm, cpu seconds)} of this process's threads (/proc/self/task): which host thread burns the cores a rank is granted."""
    out = {}
    tick = os.sysconf('SC_CLK_TCK')
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                st = open(f'/proc/self/task/{tid}/stat').read()
                comm = st[st.index('(') + 1:st.rindex(')')]
                f_ = st[st.rindex(')') + 2:].split()
                out[int(tid)] = (comm, (int(f_[11]) + int(f_[12])) / tick)
            except Exception:
                pass
    except Exception:
        pass
    return out


def recorded_parity(args):
    """The second half of BASELINE.json's metric ("resp-head logit delta"): NOT measured by this run -- the recorded result of tests/test_gpu_fullsize.py
    (this workload through the product driver against the fp32 oracle on the GPU, same weights, same frames), copied to profiles/ when the test last ran on an MI355X."""
    for path in ('profiles/r06_parity_full_size.json', 'profiles/history/r05_parity_full_size.json', 'profiles/history/r04_parity_full_size.json'):
        try:
            rec = json.load(open(os.path.join(ROOT, path))).get(args.config)
            if not rec or rec.get('weights') != ('fp8' if args.weights == 'fp8' else 'bf16'):
                continue
            ll, tk = rec['llm_side'], rec.get('tokens', {})
            return {'source': f'{path} (tests/test_gpu_fullsize.py, recorded; not re-measured here)', 'frames': rec['frames'], 'max_abs_vs_fp32_oracle': round(ll['ours_vs_fp32'], 4),
                    'mean_abs_vs_fp32_oracle': round(ll['ours_vs_fp32_mean'], 4), 'bf16_oracle_max_abs_vs_fp32': round(ll['bf16_oracle_vs_fp32'], 4), 'logit_scale': round(ll['logit_scale'], 2),
                    'response_tokens_equal_fp32_argmax': f"{tk['equal_fp32_argmax']}/{tk['n']}" if tk else None, 'kv_len_equal': rec['kv_len']['ours'] == rec['kv_len']['oracle'],
                    # the in-run check below reads ONE 26-frame prefix (104 logits): this is that statistic over every 26-frame window of the recorded stream -- the spread it has to be read against
                    'early_26_frame_windows': ({k: round(v, 4) if isinstance(v, float) else v for k, v in rec['early_stream'].items()} if 'early_stream' in rec else None)}
        except Exception:
            pass
    return None


def recorded_idle():
    """GPU idle share of a timed-like pass from the committed rocprofv3 kernel trace (tools/gap_trace.sh: union of both HIP streams, no per-launch events) -- the figure
    `gpu_idle_frac` (every launch bracketed with HIP events, tower inline) is an upper bound of, because the event pairs stretch the pass they measure.  Not re-measured here."""
    import re
    for path in ('profiles/r06_gpu_idle_gaps.txt', 'profiles/history/r05_gpu_idle_gaps.txt'):
        try:
            m = re.search(r'span ([\d.]+) ms, GPU busy \(union over both streams\) ([\d.]+) ms, idle ([\d.]+) ms = ([\d.]+) %', open(os.path.join(ROOT, path)).read())
            if m:
                return {'frac': round(float(m.group(4)) / 100, 4), 'span_ms': float(m.group(1)), 'busy_ms': float(m.group(2)), 'source': path + ' (rocprofv3 --kernel-trace of one timed-like pass of this workload; recorded)'}
        except Exception:
            pass
    return None


def measured_parity(args, model, tok, cfg, frames, query, forced, device):
    """The second half of BASELINE.json's metric, MEASURED BY THIS RUN (after the timed region; checker only -- nothing here is timed): a prefix of the timed stream
    (the first `frames_per_forward` frames: the system prompt, the query at t = 0, one full chunk and, for the default workload, the response pinned to frame 21 with the
    replay of the frames behind it) goes through the product driver once more with the raw head logits recorded, and through oracle/stream_check.StreamOracle on the GPU
    -- plain torch fp32 on the same bf16-rounded weights (regenerated from the seed), (A) fed this build's frame embeddings, (B) end to end from the uint8 frames with its
    own PIL preprocess + fp32 tower, and once in bf16 (the reference's eager rounding points: the yardstick).  Same code as tests/test_gpu_fullsize.py::_check_stream."""
    from oracle import duet_oracle as O
    from oracle.stream_check import StreamOracle, head_logits, run_oracle_stream, dequantised_fp8
    from mmduet_amd.weights import synthetic_weights
    n = min(args.frames, max(1, args.frames_per_forward))
    sub = argparse.Namespace(**{**vars(args), 'frames': n})
    f_in = [f for f in forced if f <= n]
    d = make_driver(sub, model, tok, 1.0, f_in)
    d.record_head_logits = True
    fr = frames[:n]
    run_stream(d, fr, query)
    torch.cuda.synchronize(device)
    lg_h, ids_h, kv_h = head_logits(d), [list(x) for x in d.response_token_ids], len(d.past_key_values)
    feats = d._vit_out.view(n, cfg.frame_num_tokens, -1).clone()
    w16 = {name: t for name, t in synthetic_weights(cfg, seed=0, device=device, dtype=torch.bfloat16, scale='init02')}
    w32 = dequantised_fp8(w16) if args.weights == 'fp8' else {k: v.float() for k, v in w16.items()}
    if args.weights == 'fp8':
        w16 = {k: v.to(torch.bfloat16) for k, v in w32.items()}
    ocfg = O.OracleConfig()
    out = {}

    def oracle_run(w, dtype, **src):
        a = driver_args(sub, 1.0)
        a.bf16, a.overlap_vision = dtype == torch.bfloat16, False
        o = StreamOracle(ocfg, w, device)
        od = bench_driver_class()(a, model=o, tokenizer=tok)
        od.forced_frames, od.eos_token_id, od.record_head_logits = frozenset(f_in), -1, True
        run_oracle_stream(od, o, ids_h, query, **src)
        return head_logits(od), o.tf, len(od.past_key_values)

    t0 = time.perf_counter()
    lg_32, tf32, kv_32 = oracle_run(w32, torch.float32, feats=feats)
    lg_16, _, _ = oracle_run(w16, torch.bfloat16, feats=feats)
    lg_e, _, _ = oracle_run(w32, torch.float32, frames=fr)
    flat = [v for r in tf32 for v in r['agree']]
    out = {'frames': n, 'response_frames': f_in, 'max_abs_vs_fp32_oracle': round((lg_h - lg_32).abs().max().item(), 4), 'mean_abs_vs_fp32_oracle': round((lg_h - lg_32).abs().mean().item(), 4),
           'bf16_oracle_max_abs_vs_fp32': round((lg_16 - lg_32).abs().max().item(), 4), 'bf16_oracle_mean_abs_vs_fp32': round((lg_16 - lg_32).abs().mean().item(), 4),
           'end_to_end_max_abs_vs_fp32_oracle': round((lg_h - lg_e).abs().max().item(), 4), 'end_to_end_mean_abs_vs_fp32_oracle': round((lg_h - lg_e).abs().mean().item(), 4),
           'logit_scale': round(lg_32.abs().max().item(), 2), 'response_tokens_equal_fp32_argmax': f'{sum(flat)}/{len(flat)}' if flat else None,
           'kv_len_equal': kv_h == kv_32, 'checker_seconds': round(time.perf_counter() - t0, 1),
           'how': 'oracle/stream_check.StreamOracle on the GPU (torch fp32 kernels, same bf16-rounded weights) over a prefix of the timed stream, teacher-forced with the product ids; after the timed region'}
    del w16, w32
    torch.cuda.empty_cache()
    return out


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=2)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--config', choices=sorted(CONFIGS), default='stream300')
    p.add_argument('--frames', type=int, default=None)
    p.add_argument('--resolution', type=int, default=336)
    p.add_argument('--frames-per-forward', type=int, default=None, help='frames per causal LLM forward (1 = the reference schedule; results agree up to fp reduction order)')
    p.add_argument('--responses', type=int, default=None, help='responses per stream, forced at frames drawn once from random.Random(0) (random-init heads carry no signal)')
    p.add_argument('--streams-per-gpu', type=int, default=None, help='concurrent streams per rank inside the timed region (shared forwards, mmduet_amd/multistream.py)')
    p.add_argument('--max-new-tokens', type=int, default=32)
    p.add_argument('--vit-batch', type=int, default=35, help='frames per tower batch (scheduling only)')
    p.add_argument('--tiny', action='store_true', help='tiny model (plumbing check, not a valid measurement)')
    p.add_argument('--no-overlap', action='store_true', help='run the vision tower and the LLM steps on one stream')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-parity-check', action='store_true', help='skip the post-timing oracle check of a stream prefix (resp_head_logit_delta.measured_in_run)')
    p.add_argument('--prof-stride', type=int, default=31, help='bracket every n-th launch of the dominant kernel class with HIP events inside the timed region (round 4: 7 -> 31 -- the event pairs, not the kernels, '
                   'kept a runtime thread 75 %% busy: 0.65 CPU s per 0.85 s step and -2 %% frames/s, profiles/history/r04_host_thread_probe.txt)')
    p.add_argument('--no-prof', action='store_true', help='do not bracket the dominant kernel with HIP events in the timed region')
    p.add_argument('--multi-stream', type=int, default=4, help='also measure S streams per GPU in shared forwards (reported under "multi_stream"; never the headline value; 0 = skip)')
    p.add_argument('--multi-frames-per-forward', type=int, default=13)
    p.add_argument('--weights', choices=['bf16', 'fp8'], default=None, help='fp8 = e4m3 per-output-channel scaled LLM weights (BASELINE configs[4]); reported with dtype fp8, never the bf16 headline')
    p.add_argument('--phase', choices=['ab', 'b'], default='ab', help="'b': Phase B alone -- the frame embeddings come from a feature file written before the timed region (mmduet_amd/features.py); LLM-only frames/s, never the headline")
    p.add_argument('--prof-steps', type=int, default=2, help='sample kernel durations (HIP events, --prof-stride) in the first N timed steps only (every step runs the same workload; the HIP runtime keeps an event thread busy for as long as events are being recorded: 0.6 CPU-s per step); 0 = all timed steps')
    p.add_argument('--tower-dtype', choices=['auto', 'bf16', 'fp16', 'fp16_resid16'], default='auto', help="vision tower arithmetic: fp16 = the reference's torch.cuda.amp.autocast() tower (models/modeling_live.py:28), bf16 = the model dtype; auto = the product default")
    p.add_argument('--host-sync', choices=['auto', 'spin', 'yield', 'blocking'], default='auto', help='how this rank waits for the GPU (hipSetDeviceFlags before the first HIP call): blocking frees the host core a spinning wait burns -- matters when 8 ranks share 16 cores')
    p.add_argument('--frames-on-host', action='store_true', help='the uint8 frames start in PINNED HOST memory and cross to the GPU inside the timed region (per tower batch, on the tower stream) -- the end-to-end form of the path (the reference: pixel_values.to(cuda), test/inference.py:203); without the flag the frames are resident in HBM (the headline, per the bench contract) and the host-frames rate is reported next to it under "host_frames"')
    p.add_argument('--host-frames-steps', type=int, default=3, help='steps of the secondary host-frames leg (0 = skip)')
    p.add_argument('--layers', type=int, default=None, help='debug: override LLM layer count (INVALID as a measurement)')
    a = p.parse_args(argv)
    c = CONFIGS[a.config]
    if a.frames is None: a.frames = c['frames']
    if a.responses is None: a.responses = c['responses']
    if a.frames_per_forward is None: a.frames_per_forward = c['k']
    if a.streams_per_gpu is None: a.streams_per_gpu = c['streams']
    a.mode = c['mode']
    a.remove_turns = bool(c.get('remove', False))
    if a.weights is None: a.weights = c.get('weights', 'bf16')
    if a.mode == 'grounding':
        a.responses = 0
    return a


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` (no torchrun environment): start N ranks as a CHILD process tree and exit with its code.  Done before
    anything touches a GPU (`device_count()` does not initialise HIP on this image); never an exec from a process that has."""
    if args.gpus <= 1 or 'RANK' in os.environ or 'WORLD_SIZE' in os.environ:
        return
    n = torch.cuda.device_count()
    if n < args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but only {n} GPU(s) are visible')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


def build(args, device):
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.tokenization_live import build_live_tokenizer_and_update_config
    from mmduet_amd.weights import synthetic_weights
    if args.tiny:
        cfg = VideoHeadLiveLlavaQwenConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                           num_key_value_heads=2, vit_hidden_size=64, vit_intermediate_size=128, vit_num_hidden_layers=3,
                                           vit_num_attention_heads=4, vit_image_size=56, vit_patch_size=14, video_pooling_stride=2,
                                           frame_num_tokens=4, frame_resolution=56, v_placeholder='<image>')
    else:
        cfg = VideoHeadLiveLlavaQwenConfig(frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>')
        if args.layers:
            cfg.num_hidden_layers = args.layers
    if args.weights == 'fp8':
        cfg.weight_dtype = 'fp8_e4m3'
    if getattr(args, 'tower_dtype', 'auto') != 'auto':
        cfg.tower_dtype = args.tower_dtype
    k = max(1, args.frames_per_forward)
    step_tokens = max(256, max(1, args.streams_per_gpu) * (k * cfg.frame_num_tokens + 192))
    if getattr(args, 'multi_stream', 0):
        step_tokens = max(step_tokens, args.multi_stream * (args.multi_frames_per_forward * cfg.frame_num_tokens + 192))
    model = VideoHeadLiveLlavaQwenForCausalLM(cfg, torch_dtype=torch.bfloat16, device=device, max_vit_batch=args.vit_batch, max_step_tokens=step_tokens,
                                              kv_initial_tokens=args.frames * cfg.frame_num_tokens + 4096)
    tok = build_live_tokenizer_and_update_config('synthetic:bench', cfg)
    for name, t in synthetic_weights(cfg, seed=0, device=device, dtype=torch.bfloat16, scale='init02'):
        model.load_tensor(name, t)
    model.finalize()
    return model, tok, cfg


def bench_driver_class():
    from mmduet_amd.inference import LiveInferForBenchmark

    class BenchDriver(LiveInferForBenchmark):
        """The per-frame decision rule runs unchanged (score vs threshold on the host, every frame); because random-init
        heads make the number of firing frames arbitrary, the frames that respond are pinned to a fixed schedule so the
        workload (T frame steps + R responses x max_new_tokens tokens) is the same for every build and schedule."""
        forced_frames = frozenset()

        def _decide(self, video_scores):
            fired = super()._decide(video_scores)
            return fired or (self.frame_idx in self.forced_frames)
    return BenchDriver


def driver_args(args, threshold, frames_per_forward=None):
    from mmduet_amd.arguments_live import LiveTestArguments
    rule = dict(stream_end_score_sum_threshold=1e9) if getattr(args, 'remove_turns', False) else dict(stream_end_prob_threshold=threshold)     # youcook2: the running-sum rule is the one evaluated
    return LiveTestArguments(llm_pretrained='synthetic:bench', frame_fps=1.0, bf16=True, remove_assistant_turns=getattr(args, 'remove_turns', False), **rule,
                             score_heads='informative_score', max_new_tokens=args.max_new_tokens, grounding_mode=(args.mode == 'grounding'),
                             frames_per_forward=frames_per_forward or args.frames_per_forward, overlap_vision=not args.no_overlap,
                             system_prompt='A multimodal AI assistant is helping users with some activities.')


def make_driver(args, model, tok, threshold, forced=()):
    d = bench_driver_class()(driver_args(args, threshold), model=model, tokenizer=tok)
    d.forced_frames = frozenset(forced)
    d.eos_token_id = -1            # random weights: let every response run to the cap so the work per response is fixed
    return d


class MultiRunner:
    """S streams per GPU through mmduet_amd.multistream (shared forwards); every stream responds at its own seeded frames."""

    def __init__(self, args, model, tok, frames, query, n_streams, frames_per_forward):
        from mmduet_amd.multistream import MultiStreamInfer
        T = frames.shape[0]
        a = driver_args(args, 1.0, frames_per_forward)
        self.videos = [dict(frames=frames, conversation=[{'role': 'user', 'content': query, 'time': 0.0}],
                            driver_attrs=dict(forced_frames=frozenset(random.Random(s).sample(range(1, T + 1), args.responses)) if args.responses > 0 else frozenset(),
                                              eos_token_id=-1)) for s in range(n_streams)]
        self.ms = MultiStreamInfer(a, model=model, tokenizer=tok, n_slots=n_streams, driver_cls=bench_driver_class())
        self.T, self.n, self.responses = T, n_streams, args.responses

    def run(self):
        res = self.ms.run(self.videos)
        assert all(len(r['debug_data']) == self.T and len(r['response_token_ids']) == self.responses for r in res)
        self.last = res
        return [torch.tensor([[x['informative_score'], x['relevance_score']] for x in r['debug_data']], dtype=torch.float32) for r in res]


def run_multi_stream(args, model, tok, frames, query, n_streams, frames_per_forward, steps, warmup, device):
    """Secondary measurement: returns (frames/s, ms per step, scheduler rounds per step, replayed frames, time-in-forwards fraction)."""
    r = MultiRunner(args, model, tok, frames, query, n_streams, frames_per_forward)
    for _ in range(warmup):
        r.run()
    torch.cuda.synchronize(device)
    r.ms.rounds = 0; r.ms.exec_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        r.run()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return n_streams * steps * r.T / dt, dt / steps * 1e3, r.ms.rounds // steps, sum(x['replayed_frames'] for x in r.last), r.ms.exec_seconds / dt


def run_stream(driver, frames, query):
    driver.reset()
    if isinstance(frames, str) or frames.dtype != torch.uint8:
        driver.input_feature_stream(frames)               # Phase B alone: [T, tokens, C] features (a path or a resident tensor)
    else:
        driver.input_video_stream(frames)
    driver.input_query_stream([{'role': 'user', 'content': query, 'time': 0.0}])
    responses = driver.inference()
    scores = torch.tensor([[x['informative_score'], x['relevance_score']] for x in driver.debug_data_list], dtype=torch.float32)
    n_resp = sum(r['role'] == 'assistant' for r in responses)
    return scores, n_resp


def cpu_baseline():
    """The oracle (CPU restatement of the reference path, sdpa) on the host cores, bounded sample (~15-25 s):
      (a) BASELINE configs[0]: the 30-frame 336-px clip through the oracle stream driver on the tiny plumbing model, fp32;
      (b) the true layer shapes: 3 frames through patch-embed + 6 of 26 SigLIP layers + projector + pooling and three 49-token steps
          through 6 of 28 Qwen2-7B decoder layers (KV context 7 350 tokens = the mean of a 300-frame stream), timed in fp32 AND bf16
          and extrapolated linearly in layer count.  `value` is the faster of the two."""
    from oracle import duet_oracle as O
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    out = dict(unit='frames/s', cores=cores, kind='port', cpu_model=cpu_model(), nproc=os.cpu_count())
    # (a) configs[0] plumbing run
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from helpers import oracle_model, make_args, tokenizer_for, stream_cases
        from mmduet_amd.inference import LiveInferForBenchmark
        om, _, _ = oracle_model('A')
        meta = stream_cases()
        R = om.config.frame_resolution if hasattr(om.config, 'frame_resolution') else 56
        frames = torch.randint(0, 256, (30, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
        frames = torch.nn.functional.interpolate(frames.float(), size=(R, R), mode='nearest').to(torch.uint8)     # the tiny tower's resolution
        d = LiveInferForBenchmark(make_args(system_prompt=meta['system_prompt'], stream_end_prob_threshold=1.0), model=om, tokenizer=tokenizer_for(om.config))
        t0 = time.perf_counter()
        d.input_video_stream(frames); d.inference()
        out['config1_tiny_fp32_frames_per_s'] = round(30 / (time.perf_counter() - t0), 1)
    except Exception as e:                                   # the plumbing leg never blocks the line
        out['config1_tiny_fp32_frames_per_s'] = f'error: {type(e).__name__}: {e}'
    # (b) true layer shapes: NL of the 26 / 28 layers, NF frames, two repetitions per dtype (~10-15 s of CPU work on 16 threads)
    per = {}
    NL, NF = 6, 3
    cfg = O.OracleConfig(num_hidden_layers=NL, vit_layers=NL, vocab_size=1024)
    g = torch.Generator().manual_seed(0)
    w32 = {name: ((torch.randn(shape, generator=g) * 0.02) if len(shape) >= 2 else (torch.ones(shape) if name.endswith('weight') else torch.zeros(shape)))
           for name, shape in O.weight_shapes(cfg).items()}          # drawn once, cast per dtype
    for dt_name, dt in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
        w = {k: v.to(dt) for k, v in w32.items()}
        px = torch.randn(NF, 3, 384, 384, generator=g).to(dt)
        # KV context of 7350 tokens (the mean over a 300-frame stream) as random K / V: computing it would be a 7350-row prefill
        cache = O.KVHandle([(torch.randn(cfg.num_key_value_heads, 7350, cfg.head_dim, generator=g) * 0.5).to(dt) for _ in range(NL)],
                           [(torch.randn(cfg.num_key_value_heads, 7350, cfg.head_dim, generator=g) * 0.5).to(dt) for _ in range(NL)])
        best = None
        for rep in range(2):                                # first repetition pays page faults / kernel selection
            t0 = time.perf_counter(); O.vit_patch_embed(w, cfg, px); t_embed = (time.perf_counter() - t0) / NF
            t0 = time.perf_counter(); h = O.vit_forward(w, cfg, px); t_vit = max((time.perf_counter() - t0) / NF - t_embed, 0.0) / NL          # per frame and layer
            t0 = time.perf_counter(); e = O.post_projector_pooling(cfg, O.connector(w, h)); t_proj = (time.perf_counter() - t0) / NF
            x = e.reshape(NF, -1, cfg.hidden_size)
            t0 = time.perf_counter()
            for f in range(NF): O.llm_forward(w, cfg, x[f], cache)                         # the reference's schedule: one 49-token step per frame
            t_llm = (time.perf_counter() - t0) / NF / NL
            pf = t_embed + t_vit * 26 + t_proj + t_llm * 28
            best = pf if best is None else min(best, pf)
        per[dt_name] = best
        del w, cache
    pf = min(per.values())
    out.update(value=round(1.0 / pf, 4), per_frame_s={k: round(v, 3) for k, v in per.items()},
               sample=(f'oracle (oracle/duet_oracle.py, torch CPU sdpa): (a) configs[0] 30-frame clip through the oracle stream driver, tiny plumbing model, fp32; '
                       f'(b) true layer shapes, {NF} frames x (patch-embed + {NL} of 26 ViT layers + projector + pool) + {NF} 49-token steps x {NL} of 28 decoder layers over a '
                       f'7350-token KV context, fp32 and bf16, two repetitions each, per-frame time extrapolated linearly in layer count; value = 1 / min(per_frame_s)'))
    return out


def bench_native336(args):
    """--config native336: frames/s of the CLIP-L/14-336 encoder (LiveVisionEncoder) over a 300-frame 336-px stream in batches of 32, random-init weights at the true shape."""
    import ctypes as C
    from mmduet_amd.vision_live import LiveVisionEncoder, KNOWN
    from mmduet_amd import _lib
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    kind, vc = KNOWN['openai/clip-vit-large-patch14-336']
    Cw, CI, P, L = vc['hidden_size'], vc['intermediate_size'], vc['patch_size'], vc['num_hidden_layers']
    n_tok = (vc['image_size'] // P) ** 2 + 1
    g = torch.Generator(device=device).manual_seed(0)
    rn = lambda *sh: (torch.randn(*sh, generator=g, device=device) * 0.02).to(torch.bfloat16)
    sd = {'embeddings.patch_embedding.weight': rn(Cw, 3, P, P), 'embeddings.position_embedding.weight': rn(n_tok, Cw), 'embeddings.class_embedding': rn(Cw),
          'pre_layrnorm.weight': torch.ones(Cw, device=device, dtype=torch.bfloat16), 'pre_layrnorm.bias': torch.zeros(Cw, device=device, dtype=torch.bfloat16)}
    for i in range(L):
        p = f'encoder.layers.{i}.'
        for ln in ('layer_norm1', 'layer_norm2'):
            sd[p + ln + '.weight'] = torch.ones(Cw, device=device, dtype=torch.bfloat16); sd[p + ln + '.bias'] = torch.zeros(Cw, device=device, dtype=torch.bfloat16)
        for lin in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            sd[p + f'self_attn.{lin}.weight'] = rn(Cw, Cw); sd[p + f'self_attn.{lin}.bias'] = torch.zeros(Cw, device=device, dtype=torch.bfloat16)
        sd[p + 'mlp.fc1.weight'] = rn(CI, Cw); sd[p + 'mlp.fc1.bias'] = torch.zeros(CI, device=device, dtype=torch.bfloat16)
        sd[p + 'mlp.fc2.weight'] = rn(Cw, CI); sd[p + 'mlp.fc2.bias'] = torch.zeros(Cw, device=device, dtype=torch.bfloat16)
    enc = LiveVisionEncoder.from_state_dict(kind, vc, sd, torch_dtype=torch.bfloat16, frame_token_cls=True, frame_token_pooled=(6, 6), max_batch=32)
    frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(device)
    L_ = _lib.lib()
    n = len(_lib.K_NAMES)

    def prof(on):
        if on:
            _lib.check(L_.mmd_prof_reset(enc._ctx), enc._ctx); _lib.check(L_.mmd_prof_set_stride(enc._ctx, 1), enc._ctx)
        _lib.check(L_.mmd_prof_enable(enc._ctx, (1 << n) - 1 if on else 0), enc._ctx)

    for _ in range(max(1, args.warmup)):
        out = enc(frames)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = enc(frames)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    assert out.shape == (args.frames, 37, Cw) and bool(torch.isfinite(out.float()).all())
    prof(True); enc(frames); torch.cuda.synchronize(device); prof(False)
    ms, cnt, by, fl = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)()
    _lib.check(L_.mmd_prof_read(enc._ctx, ms, cnt, by, fl), enc._ctx)
    i = _lib.K_NAMES.index('gemm_tile')
    ach = fl[i] / max(1, cnt[i]) / (ms[i] / max(1, cnt[i]) * 1e-3) / 1e12
    roof = dict(bound='mfma', kernel='gemm_tile', achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TF, unit='TFLOP/s', frac=round(ach / MFMA_BF16_PEAK_TF, 4), traffic=None,
                avg_launch_us=round(ms[i] / max(1, cnt[i]) * 1e3, 2), launches_timed=int(cnt[i]), per_class_ms_untimed_pass={k: round(ms[j], 1) for j, k in enumerate(_lib.K_NAMES)},
                source='one more pass of the same workload, every launch bracketed with HIP events (untimed)')
    gf = 2 * 577 * (4 * Cw * Cw + 2 * Cw * CI) * L + 2 * 576 * 588 * Cw + 4 * 577 * 577 * Cw * L          # per frame: encoder GEMMs + patch embed + attention
    line = {'metric': 'video frames/sec (vision encode only, native 336 px, CLIP-L/14-336 secondary encoder)', 'value': round(args.steps * args.frames / dt, 2), 'unit': 'frames/s',
            'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'{args.frames}-frame 336px stream, ' + CONFIGS['native336']['text'], 'name': 'native336', 'tower_batch': 32, 'tokens_per_frame_out': 37,
                       'algorithmic_gflop_per_frame': round(gf / 1e9, 1), 'weights': 'random init N(0,0.02), true shape (24 layers)', 'note': 'a separate configuration: never mixed with the 384-px LLaVA path'},
            'roofline': roof, 'cpu_baseline': None}
    _emit(line)


_REAL_STDOUT = None


def _quiet_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (librccl prints a five-line version banner when its first communicator is created): file
    descriptor 1 is pointed at stderr for the run and the result line goes to the saved descriptor (_emit)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    sys.stdout.flush()
    data = (json.dumps(line) + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.buffer.write(data); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main():
    args = parse()
    if args.mode == 'vision':
        _quiet_stdout()
        return bench_native336(args)
    launch_ranks_if_needed(args)          # (the parent of a self-launched multi-rank run leaves its stdout to the ranks)
    _quiet_stdout()
    torch.set_num_threads(max(1, effective_cpus() // max(1, int(os.environ.get('WORLD_SIZE', '1')))))   # host-side torch ops (and the CPU baseline) use the cores this process really has
    if args.host_sync != 'auto':                            # must precede the first HIP call of the process (so before init_distributed, which initialises the device for RCCL)
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        flag = {'spin': 1, 'yield': 2, 'blocking': 4}[args.host_sync]          # hipDeviceScheduleSpin / Yield / BlockingSync
        hip.hipSetDevice(int(os.environ.get('LOCAL_RANK', '0')))
        rc = hip.hipSetDeviceFlags(flag)
        if rc != 0:
            raise SystemExit(f'bench.py: hipSetDeviceFlags({flag}) failed with {rc}')
    from mmduet_amd.distributed import init_distributed, gather_scores, NativeScoreGather
    import torch.distributed as dist
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launch environment has WORLD_SIZE={world}')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    rccl_ranks = 1
    if world > 1:
        one = torch.ones(1, device=device)
        dist.all_reduce(one)                                # RCCL all-reduce: every rank really is in the communicator
        rccl_ranks = int(one.item())
        if rccl_ranks != args.gpus:
            raise SystemExit(f'bench.py: RCCL communicator has {rccl_ranks} ranks, --gpus {args.gpus}')
    model, tok, cfg = build(args, device)
    R = args.resolution if not args.tiny else 48
    g = torch.Generator().manual_seed(1 + rank)
    frames_host = torch.randint(0, 256, (args.frames, 3, R, R), dtype=torch.uint8, generator=g).pin_memory()
    frames = frames_host if args.frames_on_host else frames_host.to(device)     # default: resident in HBM when the timed region starts
    query = 'Please narrate the video in real time.'[:24]
    T, S = args.frames, max(1, args.streams_per_gpu)
    forced = sorted(random.Random(0).sample(range(1, T + 1), args.responses)) if args.responses > 0 else []   # fixed pseudo-random frames
    threshold = 1.0          # informative probability never exceeds 1: the rule is evaluated every frame but responses follow `forced`
    driver = make_driver(args, model, tok, threshold, forced)
    if args.phase == 'b':
        # Phase A once, outside every timed region: extract, write the reference's file layout ([T, tokens, C] bf16), read it back into HBM
        import tempfile
        from mmduet_amd.features import extract_features, save_frame_features, load_frame_features
        fpath = os.path.join(tempfile.gettempdir(), f'mmduet_bench_features_rank{rank}.pt')
        save_frame_features(fpath, extract_features(model, frames, 'embed'), to_bf16=True)
        frames = load_frame_features(fpath, device=device, dtype=torch.bfloat16)
        os.remove(fpath)
    multi_runner = MultiRunner(args, model, tok, frames, query, S, args.frames_per_forward) if S > 1 else None

    def one_step():
        if multi_runner is not None:
            return multi_runner.run(), args.responses
        sc, n_resp = run_stream(driver, frames, query)
        return [sc], n_resp

    # untimed pass with every kernel class bracketed: finds the dominant kernel class of this schedule
    model.prof_reset(); model.prof_enable(True)
    one_step()
    model.prof_enable(False)
    prof_all = model.prof_read()
    # the same with tower and LLM on ONE HIP stream and a wall clock around it: what fraction of the wall is no kernel of ours running (host scheduling,
    # launch gaps, syncs).  Diagnostic for an N-rank run on a CPU-quota'd box: a rank that gets fewer cores shows a larger idle fraction, not slower kernels.
    gpu_idle = None
    if multi_runner is None:
        d1 = make_driver(argparse.Namespace(**{**vars(args), 'no_overlap': True}), model, tok, threshold, forced)
        run_stream(d1, frames, query)                           # (workspaces / arenas warm)
        model.prof_reset(); model.prof_set_stride(1); model.prof_enable(True)
        torch.cuda.synchronize(device); tw = time.perf_counter()
        run_stream(d1, frames, query)
        torch.cuda.synchronize(device); tw = time.perf_counter() - tw
        model.prof_enable(False)
        busy = sum(v['ms'] for v in model.prof_read().values()) * 1e-3
        gpu_idle = dict(frac=round(max(0.0, 1.0 - busy / tw), 4), wall_ms=round(tw * 1e3, 1), kernel_ms=round(busy * 1e3, 1),
                        note='one stream, every launch bracketed with HIP events (untimed pass; the event pairs themselves add ~2 us per launch to the wall)')
    dom = max(prof_all, key=lambda k: prof_all[k]['ms'])
    # the tile-GEMM class (MFMA-bound: tower + LLM chunk GEMMs) and the weight-streaming class (HBM-bound: decode GEMV) are within a few per cent
    # of each other on this workload; the roofline object stays on the MFMA class whenever it is within 20 % of the largest one (so the figure
    # tracks one kernel family from round to round), the other class is reported under `roofline_secondary`
    second = None
    if dom != 'gemm_tile' and prof_all['gemm_tile']['ms'] >= 0.8 * prof_all[dom]['ms']:
        dom, second = 'gemm_tile', dom
    elif dom == 'gemm_tile' and prof_all['gemm_skinny']['ms'] >= 0.5 * prof_all[dom]['ms']:
        second = 'gemm_skinny'

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # The timed region gathers through the C ABI (mmd_gather_block: ONE ncclAllGather issued by libmmduet_hip on the current stream); torch.distributed's
    # all_gather_into_tensor is the cross-check, outside the timed region.  Constructing the communicator is collective and agrees on success across ranks.
    ng, native = None, None
    try:
        ng = NativeScoreGather(device)
    except Exception as e:
        native = f'unavailable, timed region uses torch.distributed: {type(e).__name__}: {e}'

    def gather(scs):
        return ng.gather_streams(scs, T, S) if ng is not None else gather_scores(scs, t_max=T, n_max=S)

    n_resp = 0
    for w in range(max(1, args.warmup) if world > 1 else args.warmup):
        scs, n_resp = one_step()
        allsc, lens = gather(scs)
    prof_on = not args.no_prof
    model.prof_reset()
    model.prof_set_stride(args.prof_stride)                 # every 7th launch of the class carries the two HIP events (sampling)
    model.prof_enable([dom] + ([second] if second else []) if prof_on else False)        # only the dominant class(es) are bracketed inside the timed region
    sync()
    t0 = time.perf_counter()
    cpu0 = time.process_time(); thr0 = thread_cpu_times()
    fwd = 0
    step_blocks = []
    for si in range(args.steps):
        scs, n_resp = one_step()
        allsc, lens = gather(scs)                                          # ONE RCCL all-gather of the padded score block
        step_blocks.append(allsc)
        fwd += driver.forward_calls if multi_runner is None else multi_runner.ms.rounds
        if prof_on and args.prof_steps > 0 and si + 1 == args.prof_steps:
            model.prof_enable(False)                                       # the sampled launches of the first `prof_steps` timed steps are the sample: every step runs the same workload
    sync()
    dt = time.perf_counter() - t0
    thr1 = thread_cpu_times()
    host_threads = sorted(((round((c - thr0.get(t, (n, 0.0))[1]) / max(1, args.steps), 3), n, 'main' if t == os.getpid() else 'tid') for t, (n, c) in thr1.items()), reverse=True)[:5]
    host_cpu_s = time.process_time() - cpu0                                # user + system CPU seconds of this rank's process (all its threads) inside the timed region
    model.prof_enable(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    prof = model.prof_read()
    assert allsc.shape == (world, S, T, 2) and int(lens.min()) == T
    if ng is not None:          # cross-check of the native transport against torch.distributed's, same scores, outside the timed region
        a2, l2 = gather_scores(scs, t_max=T, n_max=S)
        same = torch.equal(l2.cpu(), lens.cpu()) and torch.equal(torch.nan_to_num(a2.cpu()), torch.nan_to_num(allsc.cpu()))
        native = 'ok' if same else 'MISMATCH vs torch.distributed'
        ng.close()
    # what was timed is checked (outside the timed region): every gathered score of every step is a finite probability, and -- same frames, same weights,
    # fixed reduction orders, no atomics -- every timed step produced the SAME bits whatever the tower / decode overlap did
    blocks = [b.cpu() for b in step_blocks]
    finite = all(bool(torch.isfinite(b).all()) and bool(((b >= 0) & (b <= 1)).all()) for b in blocks)
    identical = all(torch.equal(b, blocks[0]) for b in blocks[1:])
    step_maxdiff = max([float((b - blocks[0]).abs().max()) for b in blocks[1:]], default=0.0)
    if not finite:
        raise SystemExit('bench.py: non-finite / out-of-range scores in the timed region')
    verified = dict(scores_finite_in_0_1=finite, steps_bit_identical=identical, max_abs_step_to_step_diff=step_maxdiff, steps_compared=len(blocks),
                    note='all per-frame scores of all timed steps; parity of this workload against the fp32 oracle: tests/test_gpu_fullsize.py')

    def roof_from(p, name):
        avg_ms = p['ms'] / p['launches']
        if name in ('gemm_tile', 'attn_vit'):
            ach = p['flops'] / p['launches'] / (avg_ms * 1e-3) / 1e12
            r = dict(bound='mfma', kernel=name, achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TF, unit='TFLOP/s', frac=round(ach / MFMA_BF16_PEAK_TF, 4), traffic=None)
        else:
            ach = p['bytes'] / p['launches'] / (avg_ms * 1e-3) / 1e9
            r = dict(bound='hbm', kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4), traffic=None)
        r['avg_launch_us'] = round(avg_ms * 1e3, 2)
        r['launches_timed'] = int(p['launches'])
        return r

    roof = None
    if prof_on and prof[dom]['launches'] > 0:
        roof = roof_from(prof[dom], dom)
        roof['sampling_stride'] = args.prof_stride
        # the tower runs on a side stream next to the LLM steps: two MFMA-bound kernels then share the CUs and every launch of the
        # class reads longer than it would alone.  One more untimed pass on ONE stream gives the per-kernel figure.
        if not args.no_overlap and multi_runner is None:
            d2 = make_driver(argparse.Namespace(**{**vars(args), 'no_overlap': True}), model, tok, threshold, forced)
            model.prof_reset(); model.prof_set_stride(1); model.prof_enable([dom])
            run_stream(d2, frames, query)
            model.prof_enable(False)
            p2 = model.prof_read()[dom]
            if p2['launches'] > 0:
                # the per-kernel figure is the headline of the roofline object ("the timed-region frac should come from a no-overlap
                # pass"); what the same class reads inside the overlapped timed region -- tower GEMMs sharing the chip with LLM GEMMs, or running on a
                # capped grid beside a response's decoding (burst schedule) -- is kept next to it
                r2 = roof_from(p2, dom)
                overl = dict(achieved=roof['achieved'], frac=roof['frac'], avg_launch_us=roof['avg_launch_us'], launches_timed=roof['launches_timed'],
                             sampling_stride=args.prof_stride, sampled_steps=(args.prof_steps if args.prof_steps > 0 else args.steps), note='inside the timed region: the tower runs on a side HIP stream next to the LLM steps (and on half the CUs beside decode bursts), '
                             'so launches of the class share the chip and read longer although the step is shorter')
                roof.update(achieved=r2['achieved'], frac=r2['frac'], avg_launch_us=r2['avg_launch_us'], launches_timed=r2['launches_timed'], sampling_stride=1,
                            source='one more pass of the same workload in this run with tower and LLM on ONE HIP stream, every launch of the class bracketed with HIP events (untimed)')
                roof['timed_region_overlapped'] = overl
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # runs of this same workload; gfx950 FETCH_SIZE correction applied) -- not re-measured here
        for path in PMC_TRAFFIC:
            try:
                tr = json.load(open(os.path.join(ROOT, path))).get(dom)
                if tr:
                    roof['traffic'] = round(tr['hbm_bytes_per_launch'])
                    roof['traffic_source'] = path
                    break
            except Exception:
                pass
        # the clock the chip granted and the share of its cycles the matrix pipe was busy (same committed pass; MI355X guide "DVFS give-back": the FLOP fraction of the
        # 2.4 GHz peak mixes the two) -- not re-measured here
        for path in PMC_CLOCK:
            try:
                ck = json.load(open(os.path.join(ROOT, path))).get(dom)
                if ck and 'effective_clock_ghz' in ck and ck.get('clock_reliable', True):
                    roof['effective_clock_ghz'] = ck['effective_clock_ghz']
                    roof['mfma_busy_frac_of_cycles'] = ck['mfma_busy_frac_of_cycles']
                    roof['clock_source'] = path + ' (GRBM_GUI_ACTIVE / 8 XCDs / duration; SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles; frac is quoted against the 2.4 GHz peak)'
                    break
            except Exception:
                pass
        roof['per_class_ms_untimed_pass'] = {k: round(v['ms'], 1) for k, v in prof_all.items()}
    roof2 = None
    if prof_on and second and prof[second]['launches'] > 0:
        roof2 = roof_from(prof[second], second)
        roof2['sampling_stride'] = args.prof_stride
    cpu = None
    if rank == 0 and not (args.no_cpu_baseline or args.tiny or world > 1):
        cpu = cpu_baseline()
    multi = None
    if args.multi_stream > 1 and multi_runner is None and args.mode == 'response':
        # secondary measurement, outside the timed region and never the headline: S streams per GPU in shared forwards
        fps, ms_step, rounds, replay, frac = run_multi_stream(args, model, tok, frames, query, args.multi_stream, args.multi_frames_per_forward,
                                                              steps=1, warmup=1, device=device)
        multi = dict(streams_per_gpu=args.multi_stream, frames_per_forward=args.multi_frames_per_forward, value=round(fps, 2), unit='frames/s (this GPU)',
                     ms_per_step=round(ms_step, 1), forwards_per_step=rounds, replayed_frames=replay, time_in_forwards_frac=round(frac, 3),
                     note='mmduet_amd.multistream: one LLM forward carries frame chunks and decode rows of all streams; per-stream results as single-stream')
    host_leg = None
    if multi_runner is None and args.phase == 'ab' and args.host_frames_steps > 0 and not args.frames_on_host:
        # secondary measurement, never the headline: the same workload with the frames starting in pinned host memory; every tower batch uploads its own frames
        # on the tower's side stream inside the timed steps (mmduet_amd/inference.py input_video_stream / _issue_vit)
        run_stream(driver, frames_host, query)
        sync(); th = time.perf_counter()
        for _ in range(args.host_frames_steps):
            sc_h, _ = run_stream(driver, frames_host, query)
        sync(); th = time.perf_counter() - th
        tm = torch.tensor([th], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        th = float(tm.item())
        host_leg = dict(value=round(world * args.host_frames_steps * args.frames / th, 2), unit='frames/s', steps=args.host_frames_steps, ms_per_step=round(th / args.host_frames_steps * 1e3, 2),
                        rel_to_resident=round((world * args.host_frames_steps * args.frames / th) / (world * S * args.steps * args.frames / dt), 4),
                        scores_equal_resident=bool(torch.equal(sc_h, scs[0])) if not args.frames_on_host else None,
                        note='frames in pinned host memory, %.1f MB per stream uploaded per tower batch inside the timed steps; `value` of the line is the HBM-resident form' % (frames_host.numel() / 1e6))
    parity = recorded_parity(args)
    if rank == 0 and not (args.tiny or args.no_parity_check or args.phase == 'b' or args.layers):
        try:
            measured = measured_parity(args, model, tok, cfg, frames, query, forced, device)
        except Exception as e:                               # the checker never blocks the line
            measured = {'error': f'{type(e).__name__}: {e}'}
        parity = {'measured_in_run': measured, 'recorded_full_stream': parity}
    if rank == 0:
        total_frames = world * S * args.steps * args.frames
        value = total_frames / dt
        kv_end = int(len(driver.past_key_values)) if multi_runner is None else int(multi_runner.last[0]['final_kv_len'])
        line = {
            'metric': 'video frames/sec (stream decode, 1fps 336px)', 'value': round(value, 2), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.weights == 'bf16' else 'fp8_e4m3 weights x bf16 activations',
            'data': 'synthetic', 'rccl_ranks': rccl_ranks,
            'host_cpu_s_per_step': round(host_cpu_s / max(1, args.steps), 3), 'host_sync': args.host_sync, 'host_threads_cpu_s_per_step': [dict(comm=n, cpu_s=c, thread=w) for c, n, w in host_threads if c > 0.005], 'nproc_granted': effective_cpus(), 'nproc_visible': os.cpu_count(),
            'torch_threads': torch.get_num_threads(), 'gpu_idle_frac': gpu_idle, 'gpu_idle_frac_kernel_trace': recorded_idle(),
            'config': {'workload': ('tiny-plumbing' if args.tiny else 'llava-onevision-qwen2-7b + siglip-so400m-384') +
                       f', {args.frames}-frame 1fps {R}px stream{"s" if S > 1 else ""} ({S} per GPU), ' + CONFIGS[args.config]['text'],
                       'name': args.config, 'frames_per_forward': args.frames_per_forward, 'streams_per_gpu': S, 'responses_per_stream': int(n_resp),
                       'max_new_tokens': args.max_new_tokens, 'response_frames': forced, 'llm_forwards_per_step': fwd // max(1, args.steps),
                       'kv_tokens_end': kv_end, 'weights': ('random init N(0,0.02), true shapes' if not args.tiny else 'tiny') + ('' if args.weights == 'bf16' else ', LLM matrices quantised to fp8 e4m3 per output channel'),
                       'parallelism': f'dp{world} ({S} stream(s) per GPU, one RCCL all-gather of the [{world},{S},{T}+1,2] score block per step, issued by libmmduet_hip (mmd_gather_block))',
                       'native_gather_check': native, 'frames_start_in': 'pinned host memory (uploaded inside the timed region)' if args.frames_on_host else 'HBM', 'tower_overlap': not args.no_overlap, 'tower_dtype': getattr(model, 'tower_dtype', None), 'phase': 'A+B' if args.phase == 'ab' else 'B only (frame embeddings pre-extracted to a feature file; LLM side alone)', 'layers_override': args.layers},
            'verified': verified, 'resp_head_logit_delta': parity, 'roofline': roof, 'roofline_secondary': roof2, 'cpu_baseline': cpu, 'multi_stream': multi, 'host_frames': host_leg,
        }
        import ctypes
        ctypes.CDLL(None).fflush(None)          # RCCL's banner sits in C stdio's buffer when stdout is a pipe: push it out BEFORE the JSON line
        _emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
