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
question_id in dataset order, to `<output_fname>.scores.json` -- the reference's seam for this is N hand-launched `--start_idx/--end_idx` processes
(test/inference.py:335-338) whose files the user concatenates.
"""
import json, os, sys
import numpy as np
import torch


def main(argv=None):
    from .arguments_live import parse_args
    from .inference import LiveInferForBenchmark
    from .results import result_record
    from .distributed import init_distributed, shard_indices, shard_shape, gather_scores
    args = parse_args('test', argv)
    rank, world, local = init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    data = json.load(open(args.test_fname))[args.start_idx:args.end_idx]
    mine = shard_indices(len(data), rank, world)
    infer = LiveInferForBenchmark(args)
    out_name = args.output_fname if world == 1 else f'{args.output_fname}.rank{rank}'
    def load(ex):
        """One test entry -> (frames uint8 [T,3,R,R], fps, duration, conversation) or None when unreadable (test/datasets.py:102-104)."""
        conv = [dict(t) for t in ex['conversation']]
        try:
            if 'features' in ex and args.features_dir:
                # Phase A from disk (mmduet_amd/features.py): the driver is fed [T, tokens, C] features instead of frames
                from .features import load_frame_features
                feats = load_frame_features(os.path.join(args.features_dir, ex['features']), infer.frame_num_tokens)
                fps = ex.get('fps', args.frame_fps)
                if args.max_num_frames:
                    feats = feats[:args.max_num_frames]
                return feats, fps, ex.get('video_duration', len(feats) / fps), [{'role': 'system', 'content': args.system_prompt}] + conv
            if 'video' in ex or 'decoded' in ex:
                from .video_input import load_video_frames
                if 'video' in ex:            # a container this package can read without a codec library (Motion-JPEG / uncompressed AVI, video_decode.py)
                    from .video_decode import read_avi
                    raw, in_fps, count = read_avi(os.path.join(args.input_dir, ex['video']))
                else:
                    raw, in_fps, count = torch.from_numpy(np.load(os.path.join(args.input_dir, ex['decoded']))), ex['input_fps'], ex.get('frame_count')
                out = load_video_frames(infer.model, raw, in_fps, count, output_fps=args.frame_fps,
                                        resolution=args.frame_resolution, max_num_frames=args.max_num_frames,
                                        time_instruction_format=args.time_instruction_format)
                frames, fps, duration = out[0], out[1], out[2]
                if args.time_instruction_format is not None:      # test/datasets.py:97-98
                    conv[0]['content'] = out[3] + '\n' + conv[0]['content']
            else:
                frames = torch.from_numpy(np.load(os.path.join(args.input_dir, ex['frames'])))
                fps = ex.get('fps', args.frame_fps)
                duration = ex.get('video_duration', len(frames) / fps)
        except Exception as e:
            print(f"error loading {ex.get('question_id')} due to exception {e}, this example will be skipped", file=sys.stderr)
            return None
        if args.max_num_frames:
            frames = frames[:args.max_num_frames]
        return frames, fps, duration, [{'role': 'system', 'content': args.system_prompt}] + conv

    local_scores = {}          # dataset index -> [[informative, relevance] per frame] of the videos this rank ran

    def keep_scores(i, debug_data):
        local_scores[i] = [[d['informative_score'], d['relevance_score']] for d in debug_data]

    with open(out_name, 'w') as f_out:
        if args.streams_per_gpu > 1:
            # several videos share every LLM forward.  Clips are loaded when a slot takes them and every record is written (and flushed)
            # as its video completes, in completion order -- a crash loses nothing, a long test file is never resident at once
            from .multistream import MultiStreamInfer
            ms = MultiStreamInfer(args, model=infer.model, tokenizer=infer.tokenizer, n_slots=args.streams_per_gpu)

            def entry(i):
                ex = data[i]

                def make():
                    v = load(ex)
                    return None if v is None else dict(frames=v[0], fps=v[1], conversation=v[3], ex=ex, duration=v[2], index=i)
                return make

            def on_result(n, video, res):
                rec = result_record(video['ex']['question_id'], res['responses'], video['duration'], res['debug_data'], evaluator_format=args.evaluator_format)
                f_out.write(json.dumps(rec) + '\n')
                f_out.flush()
                keep_scores(video['index'], res['debug_data'])
            ms.run([entry(i) for i in mine], on_result=on_result)
        for n, i in enumerate([] if args.streams_per_gpu > 1 else mine):
            ex = data[i]
            v = load(ex)
            if v is None:
                continue
            frames, fps, duration, conversation = v
            infer.reset()
            infer.set_fps(fps=fps)
            if frames.dtype == torch.uint8:
                infer.input_video_stream(frames)
            else:
                infer.input_feature_stream(frames)
            infer.input_query_stream(conversation)
            responses = infer.inference()
            rec = result_record(ex['question_id'], responses, duration, infer.debug_data_list, evaluator_format=args.evaluator_format)
            f_out.write(json.dumps(rec) + '\n')
            keep_scores(i, infer.debug_data_list)
            if n % 5 == 0:
                f_out.flush()
    if world > 1:
        # the ONE collective of the path.  n_max is known without communication (deterministic assignment); the longest stream is not (frame counts come from
        # the clips), so gather_scores precedes the block by its 16-byte shape exchange.  A skipped (unreadable) clip travels as a zero-length stream.
        n_max, _ = shard_shape(len(data), world)
        allsc, lens = gather_scores([torch.tensor(local_scores.get(i, []), dtype=torch.float32).view(-1, 2) for i in mine], n_max=n_max)
        if rank == 0:
            allsc, lens = allsc.cpu(), lens.cpu()
            merged = {}
            for i, ex in enumerate(data):
                r, slot = i % world, i // world
                merged[str(ex['question_id'])] = allsc[r, slot, :int(lens[r, slot])].tolist()
            with open(f'{args.output_fname}.scores.json', 'w') as f:
                json.dump(merged, f)
        import torch.distributed as dist
        dist.barrier()


if __name__ == '__main__':
    main()
