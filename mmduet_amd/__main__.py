"""`python -m mmduet_amd ...` -- benchmark inference over pre-decoded videos (mirror of `python -m test.inference`,
test/inference.py:332-363).

The reference decodes videos with OpenCV inside its dataset class (test/datasets.py:32-85); OpenCV is not part of this
package, so `--test_fname` is a JSON list whose entries carry the already sampled, letter-boxed frames:
    {"question_id": ..., "frames": "clip0.npy" (uint8 [T,3,R,R], relative to --input_dir), "fps": 1.0,
     "video_duration": 30.0, "conversation": [{"role": "user", "content": "...", "time": 0.0}, ...]}
An entry may carry "video": "clip0.avi" -- a Motion-JPEG or uncompressed AVI, decoded by mmduet_amd/video_decode.py (what cv2.VideoCapture
provides to the reference's load_video) -- or the decoder's raw output -- "decoded": "clip0_raw.npy" (uint8 [N,H,W,3] BGR, decode order),
"input_fps": 29.97[, "frame_count": header value] -- and the reference's sampling + letterbox (test/datasets.py:32-85) runs on the
GPU (video_input.load_video_frames), `--time_instruction_format` included.  With `--features_dir DIR` an entry may carry
"features": "clip0.pt" instead: a pre-extracted [T, tokens, C] feature file (mmduet_amd/features.py; the reference's offline extraction format,
data/utils.py:99-117) -- Phase A then comes from disk and only the LLM side runs.
`--evaluator_format true` writes debug_data in the deprecated shape `test/evaluate.py --func grounding|qvh_highlight` reads.
Everything else (flags, JSONL output format, `--start_idx/--end_idx` sharding, skip-on-unreadable) follows the reference.
`--streams_per_gpu S` runs S videos at a time through shared LLM forwards (mmduet_amd/multistream.py), same records.
With torchrun, entries are sharded over the ranks (`i % world == rank`) and every rank writes `<output_fname>.rank<r>`; the per-frame head
scores of all ranks are then met by ONE all-gather (mmduet_amd.distributed.gather_scores: RCCL over xGMI on the GPUs) and rank 0 writes them, keyed by
question_id in dataset order, to `<output_fname>.scores.json` (and, in response mode, the generated token ids to `<output_fname>.responses.json`) -- the reference's seam for this is N hand-launched `--start_idx/--end_idx` processes
(test/inference.py:335-338) whose files the user concatenates.
"""
import json, os, sys
import numpy as np
import torch


def main(argv=None):
    from .arguments_live import parse_args
    from .inference import LiveInferForBenchmark
    from .results import result_record
    from .distributed import init_distributed, shard_indices, shard_shape, gather_scores, gather_responses
    from .prefetch import ClipPrefetcher, pin
    args = parse_args('test', argv)
    rank, world, local = init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    data = json.load(open(args.test_fname))[args.start_idx:args.end_idx]
    mine = shard_indices(len(data), rank, world)
    infer = LiveInferForBenchmark(args)
    out_name = args.output_fname if world == 1 else f'{args.output_fname}.rank{rank}'
    # ---- a clip's way to the driver, in two stages (mmduet_amd/prefetch.py; the reference: DataLoader(num_workers=4), test/inference.py:341) ----
    def load_host(i):
        """HOST stage (loader threads, no GPU call): test entry i -> dict(kind, tensors in pinned host memory, fps / duration / conversation), or None when
        unreadable (test/datasets.py:102-104)."""
        ex = data[i]
        conv = [dict(t) for t in ex['conversation']]
        try:
            if 'features' in ex and args.features_dir:
                # Phase A from disk (mmduet_amd/features.py): the driver is fed [T, tokens, C] features instead of frames
                from .features import load_frame_features
                feats = load_frame_features(os.path.join(args.features_dir, ex['features']), infer.frame_num_tokens)
                fps = ex.get('fps', args.frame_fps)
                if args.max_num_frames:
                    feats = feats[:args.max_num_frames]
                return dict(kind='features', frames=pin(feats), fps=fps, duration=ex.get('video_duration', len(feats) / fps), conv=conv, index=i)
            if 'video' in ex or 'decoded' in ex:
                from .video_input import sampling_selector, frame_sampling_plan
                if 'video' in ex:            # a container this package can read without a codec library (Motion-JPEG / uncompressed AVI, video_decode.py):
                    from .video_decode import read_avi          # only the frames the sampling schedule keeps are decoded
                    sel = sampling_selector(args.frame_fps, args.max_num_frames)
                    raw, in_fps, count, kept = read_avi(os.path.join(args.input_dir, ex['video']), select=sel)
                    n_dec = sel.n_decodable
                else:
                    full = np.load(os.path.join(args.input_dir, ex['decoded']), mmap_mode='r')
                    in_fps, count = ex['input_fps'], ex.get('frame_count')
                    kept = frame_sampling_plan(in_fps, len(full) if count is None else count, args.frame_fps, args.max_num_frames, n_decodable=len(full))[0]
                    raw = torch.from_numpy(np.ascontiguousarray(full[kept])) if kept else torch.empty((0,) + full.shape[1:], dtype=torch.uint8)
                    n_dec = len(full)
                return dict(kind='raw', raw=pin(raw), in_fps=in_fps, count=count, kept=kept, n_decodable=n_dec, conv=conv, index=i)
            frames = torch.from_numpy(np.load(os.path.join(args.input_dir, ex['frames'])))
            if args.max_num_frames:
                frames = frames[:args.max_num_frames]
            fps = ex.get('fps', args.frame_fps)
            return dict(kind='frames', frames=pin(frames), fps=fps, duration=ex.get('video_duration', len(frames) / fps), conv=conv, index=i)
        except Exception as e:
            print(f"error loading {ex.get('question_id')} due to exception {e}, this example will be skipped", file=sys.stderr)
            return None

    copy_stream = torch.cuda.Stream(device=infer.device) if infer.device.type == 'cuda' else None

    def stage_device(h):
        """DEVICE stage, early half: start the upload of a raw clip on the copy stream (asynchronous: the buffer is pinned), to be met by `finish`."""
        if h is None or h['kind'] != 'raw' or copy_stream is None or 'dev' in h:
            return h
        with torch.cuda.stream(copy_stream):
            h['dev'] = h['raw'].to(infer.device, non_blocking=True)
            h['ready'] = torch.cuda.Event(); h['ready'].record(copy_stream)
        return h

    def finish(h):
        """DEVICE stage, late half (main thread, the driver's stream) -> (frames uint8 [T,3,R,R] | features, fps, duration, conversation) or None."""
        if h is None:
            return None
        ex, conv = data[h['index']], h['conv']
        try:
            if h['kind'] == 'raw':
                from .video_input import load_video_frames
                stage_device(h)
                raw = h.get('dev', h['raw'])
                if 'ready' in h:
                    torch.cuda.current_stream(infer.device).wait_event(h['ready'])
                    raw.record_stream(torch.cuda.current_stream(infer.device))
                out = load_video_frames(infer.model, raw, h['in_fps'], h['count'], output_fps=args.frame_fps, resolution=args.frame_resolution,
                                        max_num_frames=args.max_num_frames, time_instruction_format=args.time_instruction_format,
                                        presampled=(h['kept'], h['n_decodable']))
                frames, fps, duration = out[0], out[1], out[2]
                if args.time_instruction_format is not None:      # test/datasets.py:97-98
                    conv[0]['content'] = out[3] + '\n' + conv[0]['content']
                if args.max_num_frames:
                    frames = frames[:args.max_num_frames]
            else:
                frames, fps, duration = h['frames'], h['fps'], h['duration']
        except Exception as e:
            print(f"error loading {ex.get('question_id')} due to exception {e}, this example will be skipped", file=sys.stderr)
            return None
        return frames, fps, duration, [{'role': 'system', 'content': args.system_prompt}] + conv

    local_scores = {}          # dataset index -> [[informative, relevance] per frame] of the videos this rank ran

    local_ids = {}             # dataset index -> generated token ids of every response of that video

    def keep_scores(i, debug_data, token_ids=()):
        local_scores[i] = [[d['informative_score'], d['relevance_score']] for d in debug_data]
        local_ids[i] = [[int(t) for t in r] for r in token_ids]

    failure = None
    pf = ClipPrefetcher(load_host, mine, workers=args.num_workers, device=infer.device)
    try:
        with open(out_name, 'w') as f_out:
            if args.streams_per_gpu > 1:
                # several videos share every LLM forward.  A slot takes its next clip from the prefetcher (in dataset order) and every record is written (and
                # flushed) as its video completes, in completion order -- a crash loses nothing, a long test file is never resident at once
                from .multistream import MultiStreamInfer
                ms = MultiStreamInfer(args, model=infer.model, tokenizer=infer.tokenizer, n_slots=args.streams_per_gpu)

                def entry(i):
                    def make():
                        got = pf.take()
                        assert got is not None and got[0] == i, (got and got[0], i)
                        v = finish(got[1])
                        return None if v is None else dict(frames=v[0], fps=v[1], conversation=v[3], ex=data[i], duration=v[2], index=i)
                    return make

                def on_result(n, video, res):
                    rec = result_record(video['ex']['question_id'], res['responses'], video['duration'], res['debug_data'], evaluator_format=args.evaluator_format)
                    f_out.write(json.dumps(rec) + '\n')
                    f_out.flush()
                    keep_scores(video['index'], res['debug_data'], res.get('response_token_ids', ()))
                ms.run([entry(i) for i in mine], on_result=on_result)
            else:
                n = 0
                cur = pf.take()
                while cur is not None:
                    i, h = cur
                    nxt = None
                    v = finish(h)
                    if v is not None:
                        frames, fps, duration, conversation = v
                        infer.reset()
                        infer.set_fps(fps=fps)
                        if frames.dtype == torch.uint8:
                            infer.input_video_stream(frames)
                        else:
                            infer.input_feature_stream(frames)
                        infer.input_query_stream(conversation)
                        if pf.next_ready():           # the next clip is decoded already: its upload runs under this clip's LLM steps
                            nxt = pf.take()
                            stage_device(nxt[1])
                        responses = infer.inference()
                        rec = result_record(data[i]['question_id'], responses, duration, infer.debug_data_list, evaluator_format=args.evaluator_format)
                        f_out.write(json.dumps(rec) + '\n')
                        keep_scores(i, infer.debug_data_list, getattr(infer, 'response_token_ids', ()))
                        if n % 5 == 0:
                            f_out.flush()
                        n += 1
                    cur = nxt if nxt is not None else pf.take()
    except Exception as e:              # (world > 1: this rank still joins the collectives below with the scores it has, then re-raises -- the other ranks must not hang on it.
        failure = e                     #  KeyboardInterrupt / SystemExit are not caught: the launcher tears the job down)
        if world == 1:
            raise
    finally:
        pf.close()
    if world > 1:
        try:
            _gather_and_write(args, data, mine, world, rank, local_scores, local_ids, failure, infer.device)
        finally:
            if failure is not None:     # the original error wins over anything the collectives raised after it
                raise failure


def _gather_and_write(args, data, mine, world, rank, local_scores, local_ids, failure, device):
    """The collectives of a multi-rank run; rank 0 writes the merged files.  A rank that failed takes part with what it has and says so: the ranks exchange a failure flag
    first, and when any is set the merged files carry the suffix `.partial` and list the failed ranks under "__failed_ranks__" (a failed rank's unprocessed clips would
    otherwise look like unreadable ones: zero-length streams)."""
    import torch.distributed as dist
    from .distributed import gather_scores, gather_responses, shard_shape
    flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device if dist.get_backend() == 'nccl' else 'cpu')
    flags = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(flags, flag)
    failed = [r for r, f in enumerate(flags) if int(f.item())]
    suffix = '.partial' if failed else ''
    # the ONE collective of the path.  n_max is known without communication (deterministic assignment); the longest stream is not (frame counts come from
    # the clips), so gather_scores precedes the block by its 16-byte shape exchange.  A skipped (unreadable) clip travels as a zero-length stream.
    n_max, _ = shard_shape(len(data), world)
    allsc, lens = gather_scores([torch.tensor(local_scores.get(i, []), dtype=torch.float32).view(-1, 2) for i in mine], n_max=n_max)
    if rank == 0:
        allsc, lens = allsc.cpu(), lens.cpu()
        merged = {}
        for i, ex in enumerate(data):
            r, slot = i % world, i // world
            key = str(ex['question_id'])
            if key in merged:           # duplicate question ids must not collapse silently: later ones carry their dataset index
                key = f'{key}#{i}'
            merged[key] = allsc[r, slot, :int(lens[r, slot])].tolist()
        if failed:
            merged['__failed_ranks__'] = failed
        with open(f'{args.output_fname}.scores.json{suffix}', 'w') as f:
            json.dump(merged, f)
    # response mode: the generated token ids travel too (their own padded block; skipped on every rank alike when nothing was generated)
    allids = gather_responses([local_ids.get(i, []) for i in mine], n_max=n_max)
    if rank == 0 and any(t for w in allids for t in w):
        ids = {}
        for i, ex in enumerate(data):
            key = str(ex['question_id'])
            ids[key if key not in ids else f'{key}#{i}'] = allids[i % world][i // world]
        if failed:
            ids['__failed_ranks__'] = failed
        with open(f'{args.output_fname}.responses.json{suffix}', 'w') as f:
            json.dump(ids, f)
    dist.barrier()


if __name__ == '__main__':
    main()
