"""Bit-equality of two builds on a decode-heavy case: python tools/engine_equal.py (expects lib/libwseg_old.so and _new.so)."""
import os, subprocess, sys, tempfile, torch
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2])
from whisperseg_amd.engine import Engine
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=1, decoder_layers=3,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", "bf16", seed=3)
x = torch.randn(12, 80, 1000, generator=torch.Generator().manual_seed(5)) * 0.5
out = {}
for nb in (1, 4):
    t, l, fl = eng.generate(x.cuda(), [50258, 50259, 50363], 50257, 50257, max_length=48, num_beams=nb, suppress_tokens=[50257], return_first_logits=True)
    out[nb] = (t.cpu(), l.cpu(), fl.cpu())
torch.save(out, sys.argv[1])
'''
res = {}
for tag in ("old", "new"):
    subprocess.check_call(["cp", f"{root}/whisperseg_amd/lib/libwseg_{tag}.so", f"{root}/whisperseg_amd/lib/libwseg.so"])
    subprocess.check_call([sys.executable, "-c", code, f"/tmp/eq_{tag}.pt", root])
    res[tag] = torch.load(f"/tmp/eq_{tag}.pt")
for nb in (1, 4):
    print("beams", nb, "tokens (45 generated) equal:", torch.equal(res["old"][nb][0], res["new"][nb][0]), " first logits bit-equal:", torch.equal(res["old"][nb][2], res["new"][nb][2]))
