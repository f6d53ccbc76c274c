"""Drives bench.main() with a CPU stand-in for the engine so that the bench's LAUNCH and DISTRIBUTED logic (self-launch
of N ranks, weight broadcast, token all_gather, barrier / max-over-ranks timing, the per-rank report) can be exercised
under gloo without a GPU (tests/test_bench_dist_cpu.py).  Test infrastructure: the numbers it prints mean nothing."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class StubExtractor:
    def extract_windows(self, audio, starts, win_len):
        return torch.stack([audio[int(s):int(s) + win_len].abs().sum().reshape(1) for s in starts])


class StubEngine:
    def __init__(self, device):
        rank = int(os.environ.get("RANK", "0"))
        self.device = device
        self.weights = {"w": torch.full((4,), float(100 + rank))}     # differs per rank until rank 0's is broadcast

    def generate(self, feats, prompt, eos, pad, max_length=8, num_beams=4, return_first_logits=False, **kw):
        n = feats.shape[0]
        toks = torch.full((n, max_length), pad, dtype=torch.int32)
        toks[:, :3] = torch.tensor(prompt, dtype=torch.int32)
        toks[:, 3] = (feats[:, 0] * 1000).to(torch.int32) % 1000 + 50364
        toks[:, 4] = 15 + int(self.weights["w"][0].item()) % 10           # 15 + 0 on every rank once the broadcast happened
        toks[:, 5] = toks[:, 3] + 7
        lens = torch.full((n,), 6, dtype=torch.int32)
        if return_first_logits:      # [n * beams, vocab]: the beams of a window agree at the first step
            logits = (feats[:, :1] % 7.0 + torch.arange(32)[None, :] * 0.25).repeat_interleave(num_beams, dim=0)
            return toks, lens, logits
        return toks, lens

    def exact_reference(self):
        return self

    def last_timing(self):
        return (0.0, 0.0, 0.0, 0.0)


def stub_segmenter():
    """dist_configs (BASELINE configs[3] / configs[4] through whisperseg_amd.dist): the CPU stand-in segmenter of tests/test_dist_cpu.py —
    window table, sharding, collectives and parse are the product's, only the decode is a pure function of the window content."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dist_cpu import FakeSegmenter
    return FakeSegmenter()


def backend(args, device):
    return None, StubEngine(device), lambda sr, sts: StubExtractor(), stub_segmenter


if __name__ == "__main__":
    bench.main(backend=backend)
