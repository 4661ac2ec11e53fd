"""`python -m mmduet_amd ...` (mmduet_amd.__main__.main, the product CLI as shipped) on the CPU.  TEST INFRASTRUCTURE, started as one subprocess per rank
by tests/test_cli_world2.py.

Real: the CLI, the product stream driver (mmduet_amd.inference / multistream), the model's Python layer (handles, arena pool, native-loop bindings), the
checkpoint / tokenizer loaders, torch.distributed (gloo, world from the torchrun-style environment), `i % world` sharding, the `<output>.rank<r>` files and
the score all-gather.  Replaced: libmmduet_hip.so by tests/cabi_oracle_shim.FakeLib (same C entry points, computed by the oracle -- there is no GPU in the
build container), torch.cuda.* / 'cuda' device strings."""
import os, sys, types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import torch
import torch.distributed as dist


def main():
    from ref_conformance_runner import CudaIsCpu
    import mmduet_amd._lib as L, mmduet_amd.modeling_live as ML
    from cabi_oracle_shim import FakeLib
    fake = FakeLib()
    L.lib = ML.lib = lambda: fake
    stream = types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None)
    torch.cuda.is_available = lambda: True
    torch.cuda.current_device = lambda: 0
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda device=None: stream
    torch.cuda.empty_cache = lambda: None
    torch.cuda.synchronize = lambda device=None: None
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)          # (init_distributed would pick 'nccl' behind the faked torch.cuda)
    from mmduet_amd.__main__ import main as cli
    with CudaIsCpu(), torch.no_grad():
        cli(sys.argv[1:])
    print('CLI_CALLS ' + ' '.join(f'{k}={v}' for k, v in sorted(fake.calls.items())), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
