import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import test_scheduler_gpu as T
eng = T.tiny_engine("f32")
x = T.tiny_feats(23)
ref_t, ref_l = T.gen(eng, x, 4, 448, kv_positions=448)
print("full pool lens", ref_l.tolist(), eng.last_stats())
for kw in (dict(), dict(n_slots=7), dict(n_slots=3), dict(kv_positions=16), dict(n_slots=5, kv_positions=16, refill_min=1)):
    t, l = T.gen(eng, x, 4, 448, **kw)
    st = eng.last_stats()
    bad = [i for i in range(23) if int(l[i]) != int(ref_l[i]) or not torch.equal(t[i], ref_t[i])]
    print(kw, "mismatching windows", bad, {k: st[k] for k in ("n_slots", "n_steps", "n_admissions", "kv_units_total", "kv_units_peak", "n_preemptions")})
# subsets decoded alone
for lo, hi in ((0, 3), (3, 5), (5, 7), (0, 7)):
    t, l = T.gen(eng, x[lo:hi], 4, 448)
    bad = [i for i in range(hi - lo) if int(l[i]) != int(ref_l[lo + i]) or not torch.equal(t[i], ref_t[lo + i])]
    print((lo, hi), "mismatch", bad, eng.last_stats()["kv_units_total"], eng.last_stats()["n_preemptions"])
