"""ctypes binding of libmmduet_hip.so (C ABI in include/mmduet.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing the model raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C mmduet_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmmduet_hip.so')

MMD_F32, MMD_BF16 = 0, 1
POOL_MODES = {'bilinear': 0, 'average': 1, 'max': 2, 'adaptive_avg': 3}
K_NAMES = ('gemm_skinny', 'gemm_tile', 'attn_llm', 'attn_vit', 'norm_rope', 'other')
EPI = {'none': 0, 'gelu_tanh': 1, 'gelu_erf': 2, 'resid': 3, 'swiglu': 4}


class MmdConfig(C.Structure):
    _fields_ = [('struct_size', C.c_int32), ('dtype', C.c_int32),
                ('vocab_size', C.c_int32), ('hidden_size', C.c_int32), ('intermediate_size', C.c_int32),
                ('num_layers', C.c_int32), ('num_heads', C.c_int32), ('num_kv_heads', C.c_int32), ('head_dim', C.c_int32),
                ('rope_theta', C.c_float), ('rms_norm_eps', C.c_float),
                ('vit_hidden', C.c_int32), ('vit_intermediate', C.c_int32), ('vit_layers', C.c_int32), ('vit_heads', C.c_int32),
                ('vit_image', C.c_int32), ('vit_patch', C.c_int32), ('vit_ln_eps', C.c_float), ('vit_post_layernorm', C.c_int32),
                ('pool_mode', C.c_int32), ('pool_stride', C.c_int32), ('frame_num_tokens', C.c_int32),
                ('max_vit_batch', C.c_int32), ('max_step_tokens', C.c_int32), ('weight_dtype', C.c_int32),
                ('vision_only', C.c_int32), ('vit_class_token', C.c_int32), ('vit_pre_layernorm', C.c_int32), ('vit_act', C.c_int32), ('vit_pool_head', C.c_int32), ('tower_f16', C.c_int32)]


class MmduetError(RuntimeError):
    pass


_lib = None

_VP, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    'mmd_create': (_I, [C.POINTER(MmdConfig), _I, C.POINTER(_VP)]),
    'mmd_destroy': (None, [_VP]),
    'mmd_last_error': (C.c_char_p, [_VP]),
    'mmd_set_stream': (_I, [_VP, _VP]),
    'mmd_get_stream': (_VP, [_VP]),
    'mmd_synchronize': (_I, [_VP]),
    'mmd_load_tensor': (_I, [_VP, C.c_char_p, _VP, _I, C.POINTER(_I64), _I, _I]),
    'mmd_merge_lora': (_I, [_VP, C.c_char_p, _VP, _VP, _I, _F]),
    'mmd_set_rope_inv_freq': (_I, [_VP, _VP, _I]),
    'mmd_set_tower_share': (_I, [_VP, _I]),
    'mmd_finalize_weights': (_I, [_VP]),
    'mmd_weight_bytes': (_I64, [_VP]),
    'mmd_vit_encode': (_I, [_VP, _VP, _I, _VP]),
    'mmd_vit_set_full_tower': (_I, [_VP, _I]),
    'mmd_vit_get_full_tower': (_I, [_VP]),
    'mmd_vit_debug_tap': (_I, [_VP, _I, _VP, _I64]),
    'mmd_connector_pool': (_I, [_VP, _VP, _I, _VP]),
    'mmd_vit_encode_frames': (_I, [_VP, _VP, _I, _I, _VP]),
    'mmd_normalize_frames': (_I, [_VP, _VP, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _F, _VP]),
    'mmd_vision_tower': (_I, [_VP, _VP, _I, _VP]),
    'mmd_vision_pool_tokens': (_I, [_VP, _VP, _I, _I, _I, _VP]),
    'mmd_vision_pool_head': (_I, [_VP, _VP, _I, _VP]),
    'mmd_preprocess_frames': (_I, [_VP, _VP, _I, _I, _VP]),
    'mmd_letterbox_geometry': (_I, [_I, _I, _I] + [C.POINTER(_I)] * 6),
    'mmd_letterbox_frames': (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _I, _VP]),
    'mmd_embed_tokens': (_I, [_VP, _VP, _I, _VP]),
    'mmd_stream_create': (_I, [_VP, _I64, C.POINTER(_VP)]),
    'mmd_stream_destroy': (None, [_VP]),
    'mmd_kv_len': (_I64, [_VP]),
    'mmd_kv_capacity': (_I64, [_VP]),
    'mmd_kv_stride': (_I64, [_VP]),
    'mmd_kv_truncate': (_I, [_VP, _I64]),
    'mmd_kv_debug_set_len': (_I, [_VP, _I64]),
    'mmd_stream_reset': (_I, [_VP]),
    'mmd_kv_stash': (_I, [_VP, _I64, _I64]),
    'mmd_kv_unstash': (_I, [_VP]),
    'mmd_comm_unique_id': (_I, [_VP]),
    'mmd_comm_create': (_I, [_VP, _I, _I, _I, _VP, C.POINTER(_VP)]),
    'mmd_comm_destroy': (None, [_VP]),
    'mmd_comm_world': (_I, [_VP]),
    'mmd_comm_last_error': (C.c_char_p, [_VP]),
    'mmd_gather_scores': (_I, [_VP, _VP, _I, _I, _VP]),
    'mmd_comm_probe': (_I, []),
    'mmd_comm_set_stream': (_I, [_VP, _VP]),
    'mmd_gather_block': (_I, [_VP, _VP, C.c_int64, _VP]),
    'mmd_llm_step': (_I, [_VP, _VP, _VP, _I, _VP]),
    'mmd_video_heads': (_I, [_VP, _VP, _I, _VP]),
    'mmd_lm_head': (_I, [_VP, _VP, _I, _VP]),
    'mmd_frame_step': (_I, [_VP, _VP, _VP, _I, _VP, _I, _VP]),
    'mmd_llm_step_multi': (_I, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), _I, _VP, _VP]),
    'mmd_frame_step_multi': (_I, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), _I, _VP, C.POINTER(C.c_int32), _I, C.POINTER(_F), C.POINTER(C.c_int32), _I, _VP, _VP]),
    'mmd_sampler_create': (_I, [_VP, C.POINTER(_VP)]),
    'mmd_sampler_destroy': (None, [_VP]),
    'mmd_sampler_begin': (_I, [_VP, _I64, _F, _VP, _I, _I]),
    'mmd_sampler_prev_len': (_I, [_VP]),
    'mmd_round_multi': (_I, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), _I, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(_F), C.POINTER(_I64)]),
    'mmd_greedy_generate': (_I, [_VP, _VP, _VP, _I, _I64, _F, _VP, C.POINTER(_I), _I, _VP, _I, C.POINTER(_I)]),
    'mmd_prof_enable': (_I, [_VP, _I]),
    'mmd_prof_set_stride': (_I, [_VP, _I]),
    'mmd_prof_read': (_I, [_VP, _VP, _VP, _VP, _VP]),
    'mmd_prof_reset': (_I, [_VP]),
    'mmd_op_gemm': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I]),
    'mmd_op_gemm_bench': (_I, [_VP, _I, _I, _I, _I, _I, _I, C.POINTER(_F), _VP, _VP]),
    'mmd_op_gemm_last_plan': (_I, [_VP, C.POINTER(_I)]),
    'mmd_op_gemm_pair': (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, C.POINTER(_I)]),
    'mmd_op_gemm_slabs': (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _I, C.POINTER(_I)]),
    'mmd_op_quantize_fp8': (_I, [_VP, _VP, _I, _I, _VP, _VP]),
    'mmd_op_gemm_w8': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I]),
    'mmd_op_rmsnorm': (_I, [_VP, _VP, _VP, _VP, _I, _I, _F]),
    'mmd_op_layernorm': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _F]),
    'mmd_op_resid32_layernorm': (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _I, _I, _F]),
    'mmd_op_rope_append': (_I, [_VP, _VP, _I, _I, _I, _I, _F, _I64, _VP, _VP, _VP, _I64]),
    'mmd_op_attention': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I64, _I64, _I, _I]),
    'mmd_op_attention_bench': (_I, [_VP, _I, _I, _I, _I, _I64, _I, _I, C.POINTER(_F)]),
    'mmd_op_attention_last_form': (_I, [_VP, C.POINTER(_I)]),
    'mmd_op_pool': (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load (once) and return the native library; raises MmduetError if it is absent -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmduetError(f'{LIB_PATH} not found: the MI355X HIP library is required (no CPU fallback). '
                          f'Build it with `make -C {os.path.dirname(LIB_PATH)}` or __graft_entry__.build().')
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise MmduetError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise MmduetError(f'{LIB_PATH} does not export {name}') from e
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(rc, ctx=None, what=''):
    if rc == 0:
        return
    msg = lib().mmd_last_error(ctx)
    raise MmduetError(f'{what or "libmmduet_hip"} failed ({rc}): {msg.decode() if msg else "?"}')
