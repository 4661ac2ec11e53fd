"""oracle/ -- CPU restatement of the MMDuet streaming video-text-duet forward path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import anything from this
package, and there only as the *checker* (or the reported CPU baseline) -- never as the thing measured or shipped.
`mmduet_amd/` must not import it; the product path fails loudly when the HIP library is missing.

What it restates (plain torch-CPU ops, no transformers / llava / peft imports):
  duet_oracle.py   ViT tower -> projector -> pooling -> Qwen2 decoder with a functional KV handle ->
                   lm/informative/relevance heads -> greedy generation (reference file:line cited per function)
  preprocess.py    the LLaVA SigLipImageProcessor behaviour (PIL bicubic resize, 1/255, (x-.5)/.5)

Pinning status: the reference repository holds NO golden vectors / known-answer tests for this path,
so the oracle is pinned against outputs of the reference's own classes run in the build container:
`tests/golden/make_golden.py` imports /root/reference (through `tests/golden/ref_harness.py`), runs the reference
model + driver on tiny seeded configs and commits the results as `tests/golden/*.npz|*.json`;
`tests/test_oracle_vs_golden.py` checks this package against those fixtures.  The arithmetic itself lives in
third-party packages the reference depends on (transformers==4.44.2 pinned in requirements.txt:41 -- 5.15.0 is what
is installed and was used for the fixtures; LLaVA-NeXT un-pinned and absent -> its tower / projector / image
processor behaviour is restated from the published source and marked [3P-recalled] where used).
"""
