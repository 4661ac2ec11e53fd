"""ctypes access to the raw operator entry points (mmd_op_*) for the GPU parity tests."""
import ctypes as C
import torch
from mmduet_amd._lib import lib, check, EPI
from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM, _ptr
from helpers import product_config
from conftest import load_golden_weights


class RawOps:
    def __init__(self, dtype, max_step_tokens=64):
        """max_step_tokens > 2048: the context gets the large (192 MB) split-K slab workspace of a model that runs several streams' merged chunks."""
        cfgd, _ = load_golden_weights('A')
        self.m = VideoHeadLiveLlavaQwenForCausalLM(product_config(cfgd), torch_dtype=dtype, max_vit_batch=2, max_step_tokens=max_step_tokens, kv_initial_tokens=256)
        self.ctx, self.dtype, self.dev = self.m._ctx, dtype, self.m.device

    def t(self, x):
        return x.to(device=self.dev, dtype=self.dtype).contiguous()

    def gemm(self, X, W, bias=None, R=None, epi='none', out_f32=False, variant=0):
        M, K = X.shape; N = W.shape[0]
        NO = N // 2 if epi == 'swiglu' else N
        Y = torch.empty(M, NO, device=self.dev, dtype=torch.float32 if out_f32 else self.dtype)
        self.m._bind_stream()
        Xd, Wd = self.t(X), self.t(W)                      # keep the device copies alive across the launch
        bd = self.t(bias) if bias is not None else None
        Rd = self.t(R) if R is not None else None
        check(lib().mmd_op_gemm(self.ctx, _ptr(Xd), _ptr(Wd), _ptr(bd), _ptr(Rd), _ptr(Y), M, N, K, EPI[epi], int(out_f32), variant), self.ctx, 'gemm')
        torch.cuda.synchronize()
        return Y

    def gemm_slabs(self, X, W, variant=2, max_splits=16):
        """-> (sum of the fp32 partial slabs [M, N], number of slabs)"""
        M, K = X.shape; N = W.shape[0]
        Xd, Wd = self.t(X), self.t(W)
        slabs = torch.zeros(max_splits, M, N, device=self.dev, dtype=torch.float32)
        n = C.c_int(0)
        self.m._bind_stream()
        check(lib().mmd_op_gemm_slabs(self.ctx, _ptr(Xd), _ptr(Wd), M, N, K, variant, _ptr(slabs), max_splits, C.byref(n)), self.ctx, 'gemm_slabs')
        torch.cuda.synchronize()
        return slabs[:max(1, n.value)].sum(0), n.value

    def rmsnorm(self, x, w, eps):
        xd, wd = self.t(x), self.t(w)
        y = torch.empty_like(xd); self.m._bind_stream()
        check(lib().mmd_op_rmsnorm(self.ctx, _ptr(xd), _ptr(wd), _ptr(y), x.shape[0], x.shape[1], eps), self.ctx)
        torch.cuda.synchronize()
        return y

    def layernorm(self, x, w, b, eps):
        xd, wd, bd = self.t(x), self.t(w), self.t(b)
        y = torch.empty_like(xd); self.m._bind_stream()
        check(lib().mmd_op_layernorm(self.ctx, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), x.shape[0], x.shape[1], eps), self.ctx)
        torch.cuda.synchronize()
        return y

    def rope_append(self, qkv, nh, nkv, d, theta, pos0, Kc, Vc):
        S = qkv.shape[0]
        q = torch.empty(S, nh * d, device=self.dev, dtype=self.dtype); self.m._bind_stream()
        qd = self.t(qkv)
        check(lib().mmd_op_rope_append(self.ctx, _ptr(qd), S, nh, nkv, d, theta, pos0, _ptr(q), _ptr(Kc), _ptr(Vc), Kc.shape[1]), self.ctx)
        return q

    def attention(self, q, Kc, Vc, nh, nkv, d, n_ctx, causal=True, variant=0):
        S = q.shape[0]
        o = torch.empty(S, nh * d, device=self.dev, dtype=self.dtype); self.m._bind_stream()
        qd = self.t(q)
        check(lib().mmd_op_attention(self.ctx, _ptr(qd), _ptr(Kc), _ptr(Vc), _ptr(o), S, nh, nkv, d, n_ctx, Kc.shape[1], int(causal), variant), self.ctx)
        torch.cuda.synchronize()
        return o

    def pool(self, x, grid, mode, stride):
        B, _, H = x.shape
        out = -(-grid // stride) if mode == 0 else stride if mode == 3 else grid // stride
        y = torch.empty(B, out * out, H, device=self.dev, dtype=self.dtype); self.m._bind_stream()
        xd = self.t(x)
        check(lib().mmd_op_pool(self.ctx, _ptr(xd), _ptr(y), B, grid, H, mode, stride), self.ctx)
        torch.cuda.synchronize()
        return y
