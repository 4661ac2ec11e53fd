"""Can tower batches hide under a decode burst?  Times (a) a 64-token decode at ~8 k context alone, (b) N tower batches alone, (c) both at once
(tower on a low-priority side stream, issued first).  python tools/probes/overlap_probe.py [n_tower_batches]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    args = bench.parse(['--frames', '300', '--tower-dtype', os.environ.get('OVERLAP_TOWER_DTYPE', 'auto')])
    dev = torch.device('cuda:0')
    model, tok, cfg = bench.build(args, dev)
    from mmduet_amd.modeling_live import fast_greedy_generate
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (35, 3, 336, 336), generator=g, dtype=torch.uint8).to(dev)
    H = cfg.hidden_size
    cache = None
    for _ in range(6):                      # ~7.6 k tokens of context
        x = (torch.randn(1, 1274, H, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        cache = model(inputs_embeds=x, past_key_values=cache, use_cache=True, return_dict=True).past_key_values
    prompt = (torch.randn(1, 4, H, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    out_ids = torch.zeros(1, 64, dtype=torch.long, device=dev)
    lo, hi = torch.cuda.Stream.priority_range()
    side = torch.cuda.Stream(device=dev, priority=lo)
    vout = torch.empty(35 * 49, H, dtype=torch.bfloat16, device=dev)

    # OVERLAP_DECODE_PRIO=-1: the decode loop runs on a HIGH-priority stream (torch's default stream and the tower's `lo` are both priority 0)
    dprio = os.environ.get('OVERLAP_DECODE_PRIO')
    dstream = torch.cuda.Stream(device=dev, priority=int(dprio)) if dprio is not None else None

    def decode():
        if dstream is None:
            fast_greedy_generate(model=model, inputs_embeds=prompt, past_key_values=cache, eos_token_id=-1, inplace_output_ids=out_ids)
            return
        dstream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(dstream):
            fast_greedy_generate(model=model, inputs_embeds=prompt, past_key_values=cache, eos_token_id=-1, inplace_output_ids=out_ids)

    def tower(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                model.visual_embed_frames(frames, out=vout)

    def timed(fn):
        torch.cuda.synchronize(dev); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(dev); return (time.perf_counter() - t0) * 1e3, r

    decode(); tower(1); torch.cuda.synchronize(dev)
    for rep in range(2):
        td, _ = timed(decode)
        tt, _ = timed(lambda: tower(nb))

        def both():
            tower(nb)
            t0 = time.perf_counter(); decode(); return (time.perf_counter() - t0) * 1e3     # generate returns after its last token is read back
        tb, tdec = timed(both)
        print(f'decode alone {td:7.1f} ms | {nb} tower batches alone {tt:7.1f} ms | both {tb:7.1f} ms (decode inside: {tdec:7.1f} ms) | serial sum {td + tt:7.1f}', flush=True)


if __name__ == '__main__':
    main()
