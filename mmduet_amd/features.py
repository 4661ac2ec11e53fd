"""Offline frame-feature files: Phase A cached on disk, Phase B run alone.

The reference's extractor (`distributed_encode`, data/utils.py:99-117) walks a directory of clips, runs `vision_encode(model, batch)`
over each, and writes one `.pt` per clip holding the tensor [T, tokens, C] (optionally cast to bf16), sharded over tasks by
`i % num_tasks == rank`.  `LiveMixin.visual_embed` then starts at the connector when the model carries no tower
(models/modeling_live.py:26-33).  Here:

  level 'tower'  = what `vision_encode` returns for the LLaVA tower: [T, vit_tokens, vit_hidden]  (1.68 MB per frame in bf16 at so400m);
                   consumed through `mmd_connector_pool` (mm_projector + post_projector_pooling on the GPU);
  level 'embed'  = the pooled LLM-side embeddings [T, frame_num_tokens, hidden] (351 KB per frame): what the stream driver queues per frame
                   (test/inference.py:211-212); consumed as is -- Phase B then runs with no vision work at all.

`python -m mmduet_amd.features --input_dir clips/ --output_dir feats/ [--level tower|embed] [--save_bf16 true]` extracts from `.npy` clips
(uint8 [T,3,R,R], what the benchmark dataset class hands the driver); with torchrun the clips are sharded over the ranks like the reference.
"""
import os
import sys
import numpy as np
import torch


def save_frame_features(path, feats, tokens_per_frame=None, to_bf16=True):
    """feats: [T, tokens, C] or [T*tokens, C] (with tokens_per_frame) tensor; file layout [T, tokens, C] (data/utils.py:114-117)."""
    t = feats.detach().to('cpu')
    if t.ndim == 2 and tokens_per_frame:
        t = t.reshape(-1, tokens_per_frame, t.shape[-1])
    if t.ndim not in (2, 3):
        raise ValueError(f'features must be [T, tokens, C] (or flat [T*tokens, C]), got {tuple(t.shape)}')
    torch.save(t.to(torch.bfloat16) if to_bf16 else t, path)


def load_frame_features(path, tokens_per_frame=None, device='cpu', dtype=None):
    """-> [T, tokens, C].  `.pt` (torch.save) or `.npy`; a flat [T*tokens, C] file is reshaped with tokens_per_frame."""
    t = torch.from_numpy(np.load(path)) if str(path).endswith('.npy') else torch.load(path, map_location='cpu')
    if t.ndim == 2:
        if not tokens_per_frame:
            raise ValueError(f'{path}: flat feature file needs tokens_per_frame')
        t = t.reshape(-1, tokens_per_frame, t.shape[-1])
    if t.ndim != 3:
        raise ValueError(f'{path}: expected [T, tokens, C], got {tuple(t.shape)}')
    return t.to(device=device, dtype=dtype or t.dtype)


def feature_level(model, feats):
    """'tower' or 'embed' from the shape of a [T, tokens, C] feature tensor (raises on anything else)."""
    cfg = model.config
    tokens, C = int(feats.shape[1]), int(feats.shape[2])
    if tokens == cfg.vit_grid ** 2 and C == cfg.vit_hidden_size:
        return 'tower'
    if tokens == getattr(model, 'tokens_per_frame', cfg.frame_num_tokens) and C == cfg.hidden_size:
        return 'embed'
    raise ValueError(f'feature file of shape [T, {tokens}, {C}] matches neither the tower output [T, {cfg.vit_grid ** 2}, {cfg.vit_hidden_size}] '
                     f'nor the pooled embeddings [T, {cfg.frame_num_tokens}, {cfg.hidden_size}] of this model')


@torch.no_grad()
def extract_features(model, frames_u8, level='embed'):
    """uint8 frames [T,3,R,R] -> [T, tokens, C] on the model's device (device preprocess, tower, and for 'embed' connector + pooling)."""
    px = model.get_vision_tower().image_processor.preprocess(frames_u8, return_tensors='pt')['pixel_values']
    if level == 'tower':
        return model.tower_features(px)
    if level == 'embed':
        return model.visual_embed(px).view(len(px), -1, model.config.hidden_size)
    raise ValueError(f"level must be 'tower' or 'embed', got {level!r}")


def main(argv=None):
    import argparse
    from .arguments_live import parse_args, _str2bool
    from .distributed import init_distributed, shard_indices
    from .inference import LiveInferForBenchmark
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--output_dir', required=True)
    ap.add_argument('--level', choices=['tower', 'embed'], default='embed')
    ap.add_argument('--save_bf16', type=_str2bool, default=True)
    own, rest = ap.parse_known_args(argv)
    args = parse_args('test', rest + ['--stream_end_prob_threshold', '1'] if not any(a.startswith('--stream_end') or a.startswith('--threshold_z') for a in rest) else rest)
    rank, world, local = init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    infer = LiveInferForBenchmark(args)
    files = sorted(f for f in os.listdir(args.input_dir) if f.endswith('.npy'))
    os.makedirs(own.output_dir, exist_ok=True)
    for i in shard_indices(len(files), rank, world):          # data/utils.py:108-109: i % num_tasks == rank
        frames = torch.from_numpy(np.load(os.path.join(args.input_dir, files[i])))
        feats = extract_features(infer.model, frames, own.level)
        save_frame_features(os.path.join(own.output_dir, os.path.splitext(files[i])[0] + '.pt'), feats, to_bf16=own.save_bf16)
        print(f'{files[i]}: {tuple(feats.shape)} -> {own.output_dir}', file=sys.stderr)


if __name__ == '__main__':
    main()
