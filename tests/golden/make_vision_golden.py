"""Golden vectors for the secondary frame encoders (SURVEY.md section 8 a3'): the REFERENCE's own `_siglip_vision_encode` /
`_clip_vision_encode` (models/vision_live.py:11-54) run on HF `SiglipVisionModel` / `CLIPVisionModel` tiny seeded configs.

    python tests/golden/make_vision_golden.py          (build container only: imports /root/reference)

Writes tests/golden/vision_live.npz: the towers' state dicts (HF names), uint8 input frames, and the encoder outputs for every
(frame_token_cls, frame_token_pooled) combination the reference functions can execute.  (`_clip_vision_encode` with cls AND pooled
concatenates a [B,C] with a [B,hw,C] tensor and raises in the reference; that combination has no reference output.)
"""
import json, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness

SIGLIP = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=64, patch_size=16, layer_norm_eps=1e-6,
              hidden_act='gelu_pytorch_tanh')
CLIP = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14, layer_norm_eps=1e-5,
            hidden_act='quick_gelu')


def reseed(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.7 / (p[0].numel() ** 0.5))
            elif ('norm' in name or 'layrnorm' in name) and name.endswith('weight'):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    return model.eval()


def main():
    ref_harness.install()
    import warnings
    warnings.filterwarnings('ignore')
    from models.vision_live import _siglip_vision_encode, _clip_vision_encode
    from transformers import SiglipVisionConfig, SiglipVisionModel, CLIPVisionConfig, CLIPVisionModel
    out = {}
    g = torch.Generator().manual_seed(0)
    sig = reseed(SiglipVisionModel(SiglipVisionConfig(**SIGLIP, attn_implementation='eager')), 1)      # transformers 5: the vision model IS what 4.x exposed as .vision_model
    clip = reseed(CLIPVisionModel(CLIPVisionConfig(**CLIP, attn_implementation='eager')), 2)
    for tag, vm, cfg in (('siglip', sig, SIGLIP), ('clip', clip, CLIP)):
        for k, v in vm.state_dict().items():
            if 'position_ids' not in k:
                out[f'{tag}.w.{k}'] = v.detach().numpy()
        frames = torch.randint(0, 256, (3, 3, cfg['image_size'], cfg['image_size']), dtype=torch.uint8, generator=g)
        out[f'{tag}.frames'] = frames.numpy()
        fn = _siglip_vision_encode if tag == 'siglip' else _clip_vision_encode
        combos = [(False, (2, 2)), (False, (3, 3)), (True, None)] + ([(True, (2, 2))] if tag == 'siglip' else [])
        for cls, pooled in combos:
            with torch.no_grad():
                y = fn(vm, frames, frame_token_cls=cls, frame_token_pooled=pooled)
            out[f'{tag}.out.cls{int(cls)}.pool{0 if pooled is None else pooled[0]}'] = y.float().numpy()
    out['__config__'] = np.frombuffer(json.dumps(dict(siglip=SIGLIP, clip=CLIP)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'vision_live.npz'), **out)
    print({k: v.shape for k, v in out.items() if '.out.' in k})


if __name__ == '__main__':
    main()
