"""Load the *reference's own* model/driver classes on CPU so they can serve as the pin for the oracle.

Only usable in the build container (needs /root/reference).  Never imported by product code, bench.py or the
`-m gpu` tests: it exists to (a) validate oracle/ against the reference and (b) generate tests/golden/*.npz.

How it works:
  * the reference imports `peft`, `torchvision`, `cv2`, `llava` which are not installed -> we register stub modules.
    The `llava` stub carries the [3P-recalled] behaviour of LLaVA-NeXT that matters for this path:
      - LlavaMetaModel creates `vision_tower` (SigLIP tower, last encoder layer deleted, returns the hidden state
        *before* post_layernorm) and `mm_projector` (mlp2x_gelu) and `get_vision_tower()`;
      - the tower owns an `image_processor` whose preprocess = PIL-bicubic resize to 384, 1/255, (x-.5)/.5.
  * transformers>=5 aliases `rope_scaling` to `rope_parameters`; the reference sets `config.rope_scaling = None`
    (models/live_llava/video_head_live_llava_qwen.py:73) -> shim the setter to ignore None.
  * the driver hard-codes device 'cuda' (test/inference.py:42,61-63,176,...) -> a TorchFunctionMode maps it to cpu.
"""
import sys, types, importlib, contextlib
import torch, torch.nn as nn

REF = '/root/reference'
_installed = False


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class TinyVisionCfg:
    """Vision-side hyper-parameters handed to the llava stub (set before building the model)."""
    hidden_size = 32
    intermediate_size = 64
    num_hidden_layers = 3      # as stored in the checkpoint; LLaVA deletes the last one -> 2 run
    num_attention_heads = 4
    image_size = 56
    patch_size = 14
    layer_norm_eps = 1e-6
    hidden_act = 'gelu_pytorch_tanh'


def install():
    global _installed
    if _installed:
        return
    import transformers  # noqa  (must be imported before spec-less stubs exist)
    from transformers import (AutoModelForCausalLM, Qwen2Config, Qwen2Model, Qwen2ForCausalLM, LlamaConfig,
                              LlamaModel, LlamaForCausalLM, AutoConfig, AutoModel, AutoTokenizer)  # noqa
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor, Cache, HfArgumentParser, TrainingArguments  # noqa
    from transformers.models.siglip.modeling_siglip import SiglipVisionModel
    from transformers.models.siglip.configuration_siglip import SiglipVisionConfig
    from transformers.configuration_utils import PreTrainedConfig

    # --- rope_scaling = None shim -------------------------------------------------------------------
    prop = PreTrainedConfig.__dict__.get('rope_scaling')
    if isinstance(prop, property) and prop.fset is not None:
        orig_set = prop.fset
        def _set(self, value):
            if value is None:
                return
            orig_set(self, value)
        PreTrainedConfig.rope_scaling = property(prop.fget, _set)

    # --- peft / torchvision / cv2 --------------------------------------------------------------------
    class _Nope:
        def __init__(self, *a, **k): raise RuntimeError('peft stub')
        @classmethod
        def from_pretrained(cls, *a, **k): raise RuntimeError('peft stub')
    _stub('peft', LoraConfig=_Nope, get_peft_model=lambda *a, **k: (_ for _ in ()).throw(RuntimeError('peft stub')), PeftModel=_Nope)
    tv = _stub('torchvision'); tvt = _stub('torchvision.transforms'); tvio = _stub('torchvision.io')
    def _normalize(t, mean, std):
        mean = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1); std = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std
    tvf = _stub('torchvision.transforms.functional', normalize=_normalize)
    tv.transforms = tvt; tvt.functional = tvf; tv.io = tvio; tvio.read_video = None
    _stub('cv2')

    # --- llava ([3P-recalled] behaviour, see module docstring) ----------------------------------------
    class SigLipImageProcessor:
        def __init__(self, size=384):
            self.size = (size, size); self.image_mean = (0.5, 0.5, 0.5); self.image_std = (0.5, 0.5, 0.5)
            self.rescale_factor = 1 / 255
        def preprocess(self, images, return_tensors='pt'):
            import numpy as np
            from PIL import Image
            out = []
            for im in images:                       # uint8 [3,R,R] tensors (test/datasets.py:85)
                a = np.asarray(im)
                if a.shape[0] == 3: a = a.transpose(1, 2, 0)
                if a.shape[:2] != self.size:
                    a = np.asarray(Image.fromarray(a).resize((self.size[1], self.size[0]), resample=Image.BICUBIC))
                a = a.astype(np.float32) * np.float32(self.rescale_factor)     # HF rescale: image * scale (float32 after astype)
                a = (a - np.float32(0.5)) / np.float32(0.5)
                out.append(a.transpose(2, 0, 1))
            return {'pixel_values': torch.from_numpy(np.stack(out))}

    class SigLipVisionTower(nn.Module):
        def __init__(self):
            super().__init__()
            c = TinyVisionCfg
            cfg = SiglipVisionConfig(hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                                     num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                     image_size=c.image_size, patch_size=c.patch_size, layer_norm_eps=c.layer_norm_eps,
                                     hidden_act=c.hidden_act, attn_implementation='sdpa')
            self.vision_tower = SiglipVisionModel(cfg)
            del self.vision_tower.encoder.layers[-1:]      # LLaVA: `del vision_model.encoder.layers[-1:]`
            self.vision_tower.head = nn.Identity()
            self.image_processor = SigLipImageProcessor(c.image_size)
            self.num_patches_per_side = c.image_size // c.patch_size
            self.hidden_size = c.hidden_size
        def forward(self, images):
            vm = self.vision_tower
            h = vm.embeddings(images.to(vm.embeddings.patch_embedding.weight.dtype))
            for layer in vm.encoder.layers:                # == output_hidden_states[-1], i.e. no post_layernorm
                h = layer(h, None)
            return h.to(images.dtype)

    class LlavaMetaModel:
        def __init__(self, config):
            super().__init__(config)
            self.vision_tower = SigLipVisionTower()
            vh = TinyVisionCfg.hidden_size
            self.mm_projector = nn.Sequential(nn.Linear(vh, config.hidden_size), nn.GELU(), nn.Linear(config.hidden_size, config.hidden_size))
        def get_vision_tower(self):
            return self.vision_tower

    globals().update(SigLipImageProcessor=SigLipImageProcessor, SigLipVisionTower=SigLipVisionTower)
    _stub('llava'); _stub('llava.model'); _stub('llava.model.llava_arch', LlavaMetaModel=LlavaMetaModel)
    _stub('llava.mm_utils', tokenizer_image_token=None); _stub('llava.model.builder', load_pretrained_model=None)
    _stub('llava.constants', IMAGE_TOKEN_INDEX=-200, DEFAULT_IMAGE_TOKEN='<image>'); _stub('llava.conversation', conv_templates={})

    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True


class CudaToCpu(torch.overrides.TorchFunctionMode):
    """Rewrite device='cuda' / .to('cuda') into cpu so the reference driver runs in this GPU-less container."""
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get('device', None) in ('cuda', torch.device('cuda')):
            kwargs['device'] = 'cpu'
        args = tuple('cpu' if (isinstance(a, str) and a == 'cuda') else a for a in args)
        return func(*args, **kwargs)


def build_reference_model(llm_cfg: dict, dtype=torch.float32, seed=0, attn='sdpa'):
    """Instantiate the reference's VideoHeadLiveLlavaQwenForCausalLM with random weights (no from_pretrained)."""
    install()
    from models.live_llava.video_head_live_llava_qwen import VideoHeadLiveLlavaQwenForCausalLM, VideoHeadLiveLlavaQwenConfig
    cfg = VideoHeadLiveLlavaQwenConfig(**llm_cfg)
    cfg._attn_implementation = attn
    torch.manual_seed(seed)
    model = VideoHeadLiveLlavaQwenForCausalLM(cfg)
    # post_init leaves heads/projector at default init; re-draw everything reproducibly with a visible scale
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / (p.shape[-1] ** 0.5) if 'embed' not in name else 0.5))
            elif 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    model = model.to(dtype).eval()
    return model
