import os, sys, subprocess, torch
sys.path.insert(0, "/root/repo")
# bit-equality of the packed cross-attention with the fp32-FMA kernel: same engine, same inputs, env knob in a child process
code = r'''
import sys, torch
sys.path.insert(0, ".")
from whisperseg_amd.engine import Engine
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=2, decoder_layers=4,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", "bf16", seed=3)
x = torch.randn(24, 80, 1000, generator=torch.Generator().manual_seed(5)) * 0.5
for nb in (1, 2, 4):
    t, l, fl = eng.generate(x.cuda(), [50258, 50259, 50363], 50257, 50257, max_length=12, num_beams=nb, return_first_logits=True)
    torch.save((t.cpu(), l.cpu(), fl.cpu()), sys.argv[1] + str(nb))
'''
for tag, env in (("/tmp/eq_pk_", {}), ("/tmp/eq_old_", {"WSEG_CROSS_NO_PK": "1"})):
    subprocess.check_call([sys.executable, "-c", code, tag], env={**os.environ, **env}, cwd=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for nb in (1, 2, 4):
    a, b = torch.load("/tmp/eq_pk_%d" % nb), torch.load("/tmp/eq_old_%d" % nb)
    print("beams", nb, "tokens equal", torch.equal(a[0], b[0]), "first logits bit-equal", torch.equal(a[2], b[2]))
for nb in (2, 4):
    a, b = torch.load("/tmp/eq_pk_%d" % nb), torch.load("/tmp/eq_old_%d" % nb)
    d = (a[2] - b[2]).abs().amax(dim=1).view(-1, nb)
    print("beams", nb, "max |diff| per beam slot:", d.amax(dim=0).tolist(), "windows with a difference:", int((d.amax(dim=1) > 0).sum()), "of", d.shape[0])
