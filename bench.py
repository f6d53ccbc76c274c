#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: audio-seconds segmented per wall-second.

    python bench.py [--gpus N --steps K --warmup W]            (N>1: launched by torch.distributed.run)

One "step" = one pass of the whole hot path over one batch of synthetic windows per GPU: PCM already
resident in HBM -> log-mel kernels -> Whisper encoder -> cross-K/V -> beam-search decode (libwseg) ->
token ids to the host -> detokenise + regex parse (the CPU epilogue).  Workload (BASELINE.json metric):
whisperseg-large geometry (1550 M), bf16, 30 s windows (spec_time_step 0.03 @ 16 kHz, 480 000 samples,
SURVEY §8d), by default 256 concurrent windows (2 h 8 min of audio) per GPU per step — the concurrency of
BASELINE.json configs[4], sharded weakly: every GPU gets its own 256 windows; `--windows 120` is the one-hour
recording of configs[3] — all windows of a step decoded as one batch, seeded random weights (no checkpoint exists
offline), synthetic 16 kHz sine+noise, beams 4,
decode length pinned to --gen-tokens with EOS suppressed (random weights never emit a meaningful EOS).
Windows are independent, so ranks shard them with no data-path collective ("weak" scaling: fixed
windows per GPU); the only exchange is the all_gather of token ids to every rank.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEOMETRY = {
    "large": dict(d_model=1280, heads=20, layers=32, ffn=5120),
    "base": dict(d_model=512, heads=8, layers=6, ffn=2048),
    "tiny": dict(d_model=128, heads=2, layers=2, ffn=512),
}
PROMPT, EOS = [50258, 50259, 50363], 50257
MFMA_PEAK_BF16 = 2.5e15          # dense, /opt/skills/guides/MI355X_MICROARCH.md
SUPPRESS = [1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542,
            873, 893, 902, 918, 922, 931, 50258, EOS]


def hf_config(model):
    g = GEOMETRY[model]
    return dict(d_model=g["d_model"], encoder_attention_heads=g["heads"], decoder_attention_heads=g["heads"],
                encoder_layers=g["layers"], decoder_layers=g["layers"], encoder_ffn_dim=g["ffn"], decoder_ffn_dim=g["ffn"],
                vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448,
                total_spec_columns=1000)


def flops_per_window(model, beams, gen):
    """Algorithmic FLOPs: ENC + CROSSKV + DEC(beams, gen tokens)."""
    g = GEOMETRY[model]
    d, f, L, T, V, P = g["d_model"], g["ffn"], g["layers"], 500, 51865, 3
    conv = 2 * 80 * 3 * d * 1000 + 2 * d * 3 * d * 500
    enc = conv + L * (8 * T * d * d + 4 * T * T * d + 4 * T * d * f)
    crosskv = L * 4 * T * d * d
    dec = 0
    for t in range(1, P + gen + 1):
        dec += L * (8 * d * d + 4 * t * d + 4 * d * d + 4 * T * d + 4 * d * f) + (2 * d * V if t >= P else 0)
    return enc, crosskv, beams * dec


def synth_pcm(n_windows, win_len, sr, seed):
    rng = np.random.default_rng(seed)
    n = n_windows * win_len
    t = np.arange(n, dtype=np.float64) / sr
    return (0.1 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)


def cpu_baseline(args, sr, sts, win_len):
    """The oracle (CPU restatement of the reference's HF path, torch fp32) timed on this box's host cores on
    a bounded sample of the same workload."""
    from oracle import frontend as OF
    from oracle import whisper_ref as OW
    # threads actually used: the affinity mask (not os.cpu_count(): a container may see far more CPUs than
    # it can run on, and hundreds of OpenMP threads on 8-row decode GEMMs only spin), capped at 16.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    cfg = hf_config(args.model)
    rc = OW.RefConfig.from_hf_dict(cfg)
    sd = OW.random_state_dict(rc, seed=0, fast=True)
    n = args.cpu_windows
    pcm = synth_pcm(n, win_len, sr, 1000)
    gp = OW.GenParams(prompt=PROMPT, eos_token_id=EOS, pad_token_id=EOS, max_length=3 + args.gen_tokens,
                      num_beams=args.beams, suppress_tokens=SUPPRESS, begin_suppress_tokens=[220, EOS])
    t0 = time.perf_counter()
    feats = np.stack([OF.logmel_window(pcm[i * win_len:(i + 1) * win_len], sr, sts)[:, :1000] for i in range(n)])
    OW.generate(sd, rc, torch.from_numpy(feats), gp)
    dt = time.perf_counter() - t0
    return {"value": n * 1000 * sts / dt, "unit": "audio-sec/s", "cores": cores, "kind": "port",
            "sample": f"{n} x {1000 * sts:.0f} s windows, oracle/whisper_ref.py torch-fp32 on {cores} threads, "
                      f"{args.model} geometry, beams {args.beams}, {args.gen_tokens} generated tokens, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large", choices=sorted(GEOMETRY))
    ap.add_argument("--windows", type=int, default=256,
                    help="30 s windows per GPU per step (default 256 concurrent windows, BASELINE configs[4]; 120 = one 1-hour "
                         "recording, configs[3])")
    ap.add_argument("--batch", type=int, default=0, help="windows per generate call (0 = all windows of the step)")
    ap.add_argument("--gen-tokens", type=int, default=32)
    ap.add_argument("--beams", type=int, default=4)
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--spec-time-step", type=float, default=0.03)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-windows", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from whisperseg_amd import _lib, dist as wdist, postprocess
    from whisperseg_amd.audio_utils import get_feature_extractor
    from whisperseg_amd.engine import Engine

    rank, world, local_rank = wdist.init_from_env()
    if world != args.gpus:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    lib = _lib.load(require_device=True)
    distributed = world > 1 or torch.distributed.is_initialized()

    sr, sts = args.sr, args.spec_time_step
    win_len = int(1000 * sts * sr)
    W = args.windows
    batch = args.batch or W
    cfg = hf_config(args.model)
    eng = Engine.random(cfg, device, args.dtype, seed=0)
    if distributed:   # one copy of the weights is authoritative: broadcast rank 0's over RCCL/xGMI
        wdist.broadcast_weights(eng.weights, src=0)
    extractor = get_feature_extractor(sr, sts, 0, 30, 1000, device)
    pcm = torch.from_numpy(synth_pcm(W, win_len, sr, seed=rank)).to(device)          # resident in HBM
    starts = (torch.arange(W, dtype=torch.int64) * win_len).to(device)
    max_length = 3 + args.gen_tokens
    codebook = {str(i): i for i in range(10)}

    def step():
        feats = extractor.extract_windows(pcm, starts, win_len)
        toks, lens = [], []
        for lo in range(0, W, batch):
            t, l = eng.generate(feats[lo:lo + batch], PROMPT, EOS, EOS, max_length=max_length, num_beams=args.beams,
                                suppress_tokens=SUPPRESS, begin_suppress_tokens=[220, EOS])
            toks.append(t)
            lens.append(l)
        toks, lens = torch.cat(toks), torch.cat(lens)
        if distributed:
            toks, lens = wdist.gather_rows(toks, lens, W * world)
        toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
        # CPU epilogue: ids -> text -> segments (added tokens of real checkpoints sit at 50364 + i)
        n_seg = 0
        for row, ln in zip(toks, lens):
            text = "".join("<|%d|>" % (t - 50364) if t >= 50364 else (str(t - 15) if 15 <= t <= 24 else "") for t in row[3:ln])
            n_seg += len(postprocess.extract_segments(text, sts, codebook))
        return n_seg

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    audio_seconds = args.steps * W * world * 1000 * sts
    value = audio_seconds / dt
    enc_f, ckv_f, dec_f = flops_per_window(args.model, args.beams, args.gen_tokens)
    windows_per_s = args.steps * W * world / dt
    enc_ms, ckv_ms, dec_ms, n_steps = eng.last_timing()

    roofline = None
    if not args.no_roofline and args.dtype == "bf16":
        # dominant kernel: the large-tile bf16 MFMA GEMM (encoder + cross-K/V projections).  Timed live, per
        # launch, with HIP events on the launching stream over one more step of the same workload.
        _lib.check(lib.wseg_profile_begin())
        step()
        fl, ms, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(lib.wseg_profile_end(C.byref(fl), C.byref(ms), C.byref(n)))
        traffic, traffic_note = None, None
        # PMC counters cannot be collected live next to the timing (separate rocprofv3 passes): use the committed pass
        # over the same GEMM shapes at the same window count (tools/pmc_traffic.sh), averaged over the four per-layer
        # encoder GEMMs.
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemm_hbm_traffic.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            if args.model != "large" or tj.get("windows", 120) != W:
                continue
            pl = tj["per_launch"]
            keys = [k for k in ("qkv", "o-proj", "fc1", "fc2") if k in pl]
            traffic = sum(pl[k]["hbm_bytes"] for k in keys) / len(keys)
            traffic_note = ("bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, committed pass "
                            "profiles/%s; algorithmic bytes per launch %.3g"
                            % (os.path.basename(tpath), sum(pl[k]["algorithmic_bytes"] for k in keys) / len(keys)))
            break
        if n.value:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel<*> (256x256 ping-pong tiles; + the 128x128 persistent kernel for narrow problems)", "achieved": achieved,
                        "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": achieved / (MFMA_PEAK_BF16 / 1e12),
                        "traffic": traffic, "traffic_note": traffic_note, "launches_per_step": int(n.value),
                        "avg_launch_us": ms.value * 1e3 / n.value, "flops_per_step": fl.value,
                        "end_to_end_frac": windows_per_s / world * (enc_f + ckv_f + dec_f) / MFMA_PEAK_BF16}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sr, sts, win_len)

    if rank == 0:
        out = {
            "metric": "audio-sec/s segmented (whisperseg-large, 30 s windows)" if args.model == "large"
                      else f"audio-sec/s segmented (whisperseg-{args.model}, 30 s windows)",
            "value": value, "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic 16 kHz sine+noise PCM resident in HBM; seeded random weights",
            "config": {"workload": f"whisperseg-{args.model} geometry, {W} x {1000 * sts:.0f} s windows per GPU per step "
                                   f"(spec_time_step {sts}, sr {sr}), beams {args.beams}, {args.gen_tokens} generated tokens "
                                   f"(EOS suppressed), decode batch {batch}",
                       "windows_per_gpu": W, "decode_batch": batch, "beams": args.beams, "gen_tokens": args.gen_tokens,
                       "parallelism": f"clip-sharded x{world}"},
            "windows_per_s": windows_per_s, "realtime_factor_per_gpu": value / world,
            "stage_ms_last_call": {"encoder": enc_ms, "cross_kv": ckv_ms, "decode": dec_ms, "decode_steps": n_steps},
            "flops_per_window": {"encoder": enc_f, "cross_kv": ckv_f, "decoder": dec_f},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
