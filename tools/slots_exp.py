import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from whisperseg_amd.engine import Engine
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=32, decoder_layers=32, encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
import numpy as np
for dt in sys.argv[1:]:
    eng = Engine.random(cfg, "cuda:0", dt)
    feats = torch.randn(1024, 80, 1000, device="cuda") * 0.5
    rng = np.random.default_rng(3)
    caps = (rng.integers(4, 65, size=1024) + 3).astype(np.int32)
    for slots, refill in ((256, 0), (256, 128), (1024, 0)):
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            eng.generate(feats, [50258, 50259, 50363], 50257, 50257, max_length=67, num_beams=4, suppress_tokens=[50257, 1, 2], begin_suppress_tokens=[220], n_slots=slots, window_max_length=caps, refill_min=refill)
            torch.cuda.synchronize(); dt_ = time.time() - t0
        enc, ckv, dec, steps = eng.last_timing(); st = eng.last_stats()
        print(dt, "slots", slots, "refill", refill, "total %.0f ms enc %.0f ckv %.0f dec %.0f steps %d admissions %d occupancy %.2f" % (dt_ * 1e3, enc, ckv, dec, steps, st["n_admissions"], st["occupancy"]), flush=True)
    del eng; torch.cuda.empty_cache()
