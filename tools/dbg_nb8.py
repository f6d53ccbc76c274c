import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
import test_model_gpu as T
from oracle import whisper_ref as R
cfg = T.hf_cfg()
for dtype in ("f32", "f16x3", "f16m6"):
    rc, sd, eng = T.make(cfg, dtype)
    x = T.feats(3)
    for nb in (4, 5, 8):
        gp = T.gen_params(nb, 16)
        want_seq, want = R.generate(sd, rc, x, gp, return_first_logits=True)
        toks, lens, got = eng.generate(x.cuda(), T.PROMPT, T.EOS, T.EOS, max_length=16, num_beams=nb, suppress_tokens=gp.suppress_tokens,
                                       begin_suppress_tokens=gp.begin_suppress_tokens, return_first_logits=True)
        err = (got.cpu() - want).abs().max().item()
        same = [R.canonical(want_seq[i].tolist(), 3, T.EOS, T.PROMPT) == R.canonical(toks[i, :lens[i]].cpu().tolist(), 3, T.EOS, T.PROMPT) for i in range(3)]
        print(dtype, nb, "first-logit err %.2e" % err, "seq equal", same, flush=True)
