"""Diagnostic: is the tree-vs-dense gradient gap of the Qwen3-4B test rounding noise?  Same comparison in bf16 and in (loss-scaled) fp16,
which keeps three more mantissa bits: rounding noise must shrink with them."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, bench
from dynamictreeattn_amd import synth, dense
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine
from oracle import model_oracle as mo
cfg = synth.QWEN3_4B
for dtype, ls in ((torch.bfloat16, 1.0), (torch.float16, 4096.0)):
    m = bench.build_model(cfg, torch.device("cuda:0"), dtype)
    seqs = synth.as_tensors(synth.tau2(3, cfg["vocab_size"], G=4, sys_len=1000, turns=4, lo=100, hi=500))
    att = lambda: [{"w_logprobs": -1.0 * ls, "w_entropy": 0.1 * ls} for _ in seqs]      # fp16 needs a loss scale: small gradients underflow
    t = TokenTrie(seqs, att()); t.backward_permute()
    lt = TreeTrainingEngine(m.config, "cuda:0", dtype, 16384).backward(m, t, mo.default_loss, 2048)
    gt = {n: p.grad.float().clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    ld = dense.backward(m, seqs, att(), mo.default_loss)
    r = np.array([mo.grad_ratio(p.grad.float(), gt[n]) for n, p in m.named_parameters()])
    print(dtype, "loss", lt, ld, "worst", r.max(), "median", np.median(r), "finite", np.isfinite(r).all(), flush=True)
    del m, gt; torch.cuda.empty_cache()
