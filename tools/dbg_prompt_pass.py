"""Debug: the parity sweep with the prompt pass vs stepping through the prompt (WSEG_NO_PROMPT_PASS), per dtype; rows of the runs that differ."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402
from tools import tiny_model as TM  # noqa: E402
from tools.parity_sweep import MODEL_DIR, score  # noqa: E402
from whisperseg_amd.model import WhisperSegmenter  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "tiny_sweep.json")) as f:
    sweep = json.load(f)
for dtype in sys.argv[1:] or ["f16x3"]:
    seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dtype)
    for mode in ("pass", "step"):
        if mode == "step":
            os.environ["WSEG_NO_PROMPT_PASS"] = "1"
        else:
            os.environ.pop("WSEG_NO_PROMPT_PASS", None)
        res = score(seg, sweep)
        bad = res["structure_mismatch_runs"] + res["beyond_one_frame_runs"]
        print(dtype, mode, "exact", res["exact_runs"], "bad", [(b["index"], b.get("max_dev_frames")) for b in bad], flush=True)
        for b in bad[:3]:
            run = sweep[b["index"]]
            got = seg.segment(GI.tiny_recording(run["seed"], run["n_windows"]), TM.SR, **run["kwargs"])
            print("  run", b["index"], run["kwargs"], "windows", run["n_windows"])
            print("   got ", [round(x, 4) for x in got["onset"]], [round(x, 4) for x in got["offset"]], got["cluster"])
            print("   want", [round(x, 4) for x in run["expected"]["onset"]], [round(x, 4) for x in run["expected"]["offset"]], run["expected"]["cluster"])
