"""Segmenter API — host-side mirror of reference model.py (SegmenterBase :118-470, WhisperSegmenterForEval
:572-622, WhisperSegmenter :625-676, WhisperSegmenterFast :678-746) on top of the MI355X engine.

Same class names, constructor signatures, `segment()` keyword arguments / defaults and return value
({"onset": [...], "offset": [...], "cluster": [...]}); same backend seam
(`generate_segment_text_core(...)` writing `generated_texts_dict[thread_id]`).  What differs is where the
work happens: windows are cut, transformed to log-mel and decoded on the GPU by libwseg; this file only
does bookkeeping.  There is no CPU execution path — constructing a segmenter without a gfx950 device or
without libwseg.so raises.
"""
import itertools
import json
import os
import threading

import numpy as np
import torch

from . import _lib, postprocess, scoring
from .audio_utils import get_feature_extractor, get_n_fft_given_sr
from .checkpoint import checkpoint_files
from .engine import DEFAULT_BEGIN_SUPPRESS_TOKENS, DEFAULT_SUPPRESS_TOKENS, Engine
from .tokenizer import WhisperSegTokenizer
from .utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP
from .windows import shard_bounds, window_table

PROMPT_TOKENS = ["<|startoftranscript|>", "<|en|>", "<|notimestamps|>"]   # reference model.py:656
# Engine mode when neither the constructor nor $WHISPERSEG_AMD_DTYPE names one: "f16x3", split precision — the reference computes in
# fp32 (model.py:655-666); here GEMM operands (activations AND weights) travel as hi + lo IEEE-half pairs and every product is taken as
# hi*hi + hi*lo + lo*hi on the half matrix cores with fp32 accumulation, fp32 everywhere else; first-step logits within 3.3e-6 of the logit
# scale of the exact mode at 32 + 32 layers.  Its rows are identical to the reference's on all 6 200 recordings of the seven parity
# sweeps (200 + 3 x 1 000 + 3 x 1 000 recordings of three fixture models; profiles/r06_parity_sweeps.json — measured, not structural:
# "f32" is the mode that reproduces the reference by construction), "bf16x3" on 6 198 (bfloat16 pairs: 1.3e-5; no
# fp16 range limit on the GEMM operands — the encoder attention's Q / K / V^T are IEEE-half pairs in every split mode and saturate at
# +-65 504).  "f16m6" — hi*hi on the half matrix cores, both cross terms on the block-scaled fp6 MX matrix cores, 27 % faster — was the
# default of r04-r05: it reproduces the 200 recordings its formats were chosen on, and the held-out sweeps of r06 found it outside the
# north-star tolerance (clusters exact, boundaries within +-1 mel frame) on 10 of 6 000 held-out recordings: a fast mode, like "f16" / "bf16"
# (plain 16-bit: 2x faster, 98 % / 91 % of the held-out recordings inside the tolerance).  "f32" is the exact-parity mode.
DEFAULT_DTYPE = "f16x3"
POOL_WINDOWS = 8192      # windows per engine call / per pooled group of files (2.6 GB of log-mel features)


def _per_item(value):
    """Iterator over a per-recording parameter: a list / tuple / array has one entry per recording, anything else (incl.
    None and strings) is the value for every recording."""
    if isinstance(value, (list, tuple, np.ndarray)):
        def entries():
            yield from value
            raise ValueError("a per-recording parameter list has fewer entries than there are recordings")
        return entries()
    return itertools.repeat(value)


def _read_json(path, default=None):
    if not os.path.exists(path):
        return default
    with open(path) as f:
        return json.load(f)


def resolve_model_dir(model_path):
    """Directory holding config.json + weights.  Accepts the reference's two layouts: a plain HF
    checkpoint dir (model.py:633) or a converted dir with an `hf_model/` subfolder (model.py:694-702)."""
    if os.path.exists(os.path.join(model_path, "config.json")):
        return model_path
    sub = os.path.join(model_path, "hf_model")
    if os.path.exists(os.path.join(sub, "config.json")):
        return sub
    raise FileNotFoundError(f"{model_path}: no config.json (model download is not available offline; pass a local directory)")


def load_generation_settings(model_dir, vocab_size):
    """suppress_tokens / begin_suppress_tokens as HF would take them from the checkpoint's
    generation_config.json (falling back to config.json, then to the multilingual-Whisper defaults)."""
    for name in ("generation_config.json", "config.json"):
        cfg = _read_json(os.path.join(model_dir, name), {}) or {}
        if "suppress_tokens" in cfg or "begin_suppress_tokens" in cfg:
            return list(cfg.get("suppress_tokens") or []), list(cfg.get("begin_suppress_tokens") or [])
    if vocab_size == 51865:
        return list(DEFAULT_SUPPRESS_TOKENS), list(DEFAULT_BEGIN_SUPPRESS_TOKENS)
    return [], []


class SegmenterBase:
    def __init__(self):
        self.segment_matcher = postprocess.SEGMENT_PATTERN
        self.total_spec_columns = None
        self.precision_bits = 3
        self.cluster_codebook = None
        self.default_segmentation_config = {}
        self.device_list = []
        self.suppress_tokens, self.begin_suppress_tokens = [], []
        # Cap on the window slots the engine decodes concurrently (None: $WSEG_SLOTS, else 1024, halved until the workspace
        # fits 80 % of the free device memory).  The reference bounds memory with `batch_size`; here `batch_size` no longer
        # caps the concurrency, this attribute does.
        self.max_slots = None

    # ---- slicing + features (reference model.py:127-166), batched on the first device ------------
    def get_sliced_audios_features(self, audio, sr, min_frequency, spec_time_step, num_trials):
        """-> list of (trial_id, offset_time, features, clip_seconds); `features` is a float32 [80, 1000]
        DEVICE tensor (a view into one batch tensor) instead of a numpy array."""
        device = self.device_list[0]
        if torch.is_tensor(audio):      # already a tensor (e.g. straight out of whisperseg_amd.resample): no host round trip
            pcm = audio.to(device=device, dtype=torch.float32).contiguous()
        else:
            pcm = torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32)).to(device, non_blocking=True)
        out = self.sliced_features_from_device_pcm(pcm, sr, min_frequency, spec_time_step, num_trials)
        return out["shard"]

    def sliced_features_from_device_pcm(self, pcm, sr, min_frequency, spec_time_step, num_trials, rank=0, world=1,
                                        window_range=None):
        """Window table of a recording already resident in HBM + log-mel features of THIS rank's contiguous
        shard of it (world == 1: everything; window_range = (lo, hi): exactly those rows of the table, for callers that
        shard a pooled multi-recording window list).  Returns {"table": all rows without features, "shard": this
        rank's rows with device features, "n_total", "lo", "hi"}."""
        cols = self.total_spec_columns
        chunk_length = max(30, int(np.ceil(spec_time_step * cols)))
        extractor = get_feature_extractor(sr, spec_time_step, min_frequency, chunk_length, cols, pcm.device)
        table = window_table(int(pcm.numel()), sr, spec_time_step, num_trials, cols)
        if window_range is not None:
            lo, hi = max(0, int(window_range[0])), min(len(table), int(window_range[1]))
            hi = max(lo, hi)
        else:
            bounds = shard_bounds(len(table), world)
            lo, hi = bounds[rank] if rank < len(bounds) else (len(table), len(table))
        clip_len = int(cols * spec_time_step * sr)
        starts = torch.tensor([w.start for w in table[lo:hi]], dtype=torch.int64).to(pcm.device, non_blocking=True)
        feats = extractor.extract_windows(pcm, starts, clip_len)
        shard = [(w.trial_id, w.offset_time, feats[i], w.clip_seconds) for i, w in enumerate(table[lo:hi])]
        rows = [(w.trial_id, w.offset_time, None, w.clip_seconds) for w in table]
        return {"table": rows, "shard": shard, "n_total": len(table), "lo": lo, "hi": hi}

    # ---- device fan-out (reference model.py:169-189) ---------------------------------------------
    def generate_segment_text(self, sliced_audios_features, batch_size, max_length, num_beams, top_k=1, top_p=1.0,
                              length_penalty=1.0, status_monitor=None):
        texts_by_shard, errors, threads = {}, {}, []

        def run(shard, thread_id, monitor):
            try:
                self.generate_segment_text_core(shard, batch_size, max_length, num_beams, top_k, top_p, length_penalty,
                                                texts_by_shard, thread_id, monitor)
            except BaseException as exc:      # the reference loses worker errors; we re-raise them
                errors[thread_id] = exc

        bounds = shard_bounds(len(sliced_audios_features), len(self.device_list))
        for thread_id, (lo, hi) in enumerate(bounds):
            args = (sliced_audios_features[lo:hi], thread_id, status_monitor if thread_id == 0 else None)
            if len(bounds) == 1:
                run(*args)
            else:
                t = threading.Thread(target=run, args=args)
                t.start()
                threads.append(t)
        for t in threads:
            t.join()
        if errors:
            raise errors[min(errors)]
        out = []
        for thread_id in sorted(texts_by_shard):
            out += texts_by_shard[thread_id]
        return out

    def generate_segment_text_core(self, sliced_audios_features, batch_size, max_length, num_beams, top_k, top_p,
                                   length_penalty, generated_texts_dict, thread_id, status_monitor=None):
        raise NotImplementedError

    def _decode_token_batches(self, engine, tokenizer, sliced, batch_size, max_length, num_beams, top_k, top_p,
                              length_penalty, status_monitor=None):
        """-> list of (tokens int32 [b, L] device, lengths int32 [b] device).

        The reference decodes `batch_size` windows per `generate` call (model.py:653) and every batch runs until its slowest
        window ends.  Here ALL windows go to the engine, which decodes them through its window slots with in-flight refill
        (wseg_generate); `batch_size` is accepted for API compatibility and does not cap the concurrency (the engine bounds
        its slots by device memory; $WSEG_SLOTS overrides).  Calls are chunked at POOL_WINDOWS windows only to bound the
        stacked feature tensor."""
        sample = {}
        if num_beams == 1 and top_k != 1:
            # reference model.py:615-616 / 662-663: do_sample = (num_beams == 1), i.e. multinomial sampling among the top_k
            # (then top_p) tokens.  The draw comes from the engine's counter-based generator, seeded from torch's global
            # generator (torch.manual_seed makes a run reproducible; the stream is not HF's).
            if not 2 <= int(top_k) <= 16:
                raise NotImplementedError("sampling is implemented for top_k in 2..16 (top_k=1, the reference's default, "
                                          "is the deterministic argmax)")
            sample = dict(top_k=int(top_k), top_p=float(top_p), seed=int(torch.randint(0, 2 ** 62, ()).item()))
        prompt = tokenizer.convert_tokens_to_ids(PROMPT_TOKENS)
        out = []
        n = len(sliced)
        for pos in range(0, n, POOL_WINDOWS):
            batch = torch.stack([item[2].to(engine.device) for item in sliced[pos:pos + POOL_WINDOWS]])
            out.append(engine.generate(batch, prompt, tokenizer.eos_token_id, tokenizer.pad_token_id,
                                       max_length=max_length, num_beams=num_beams, length_penalty=length_penalty,
                                       suppress_tokens=self.suppress_tokens,
                                       begin_suppress_tokens=self.begin_suppress_tokens, n_slots=self.max_slots, **sample))
            if sample:
                sample["seed"] += 1
            if status_monitor is not None:
                status_monitor["progress"] = int(100 * min(1, (pos + POOL_WINDOWS) / n))
        return out

    def _decode_batches(self, engine, tokenizer, sliced, batch_size, max_length, num_beams, top_k, top_p,
                        length_penalty, status_monitor):
        texts = []
        for tokens, lengths in self._decode_token_batches(engine, tokenizer, sliced, batch_size, max_length, num_beams,
                                                          top_k, top_p, length_penalty, status_monitor):
            tokens, lengths = tokens.cpu().numpy(), lengths.cpu().numpy()
            texts += tokenizer.batch_decode([row[:ln] for row, ln in zip(tokens, lengths)], skip_special_tokens=False)
        return texts

    # ---- hooks used by whisperseg_amd.dist.segment_distributed (one process per GPU) ----------------
    def _first_engine(self):
        if hasattr(self, "model_list"):
            return self.model_list[0], self.tokenizer_list[0]
        return self.model, self.tokenizer

    def decode_shard_tokens(self, shard, batch_size=4, max_length=448, num_beams=4, length_penalty=1.0, top_k=1, top_p=1.0):
        engine, tokenizer = self._first_engine()
        L = int(min(max_length, engine.geo["dec_positions"]))
        parts = self._decode_token_batches(engine, tokenizer, shard, batch_size, max_length, num_beams, top_k, top_p, length_penalty)
        if not parts:
            return (torch.zeros((0, L), dtype=torch.int32, device=engine.device),
                    torch.zeros((0,), dtype=torch.int32, device=engine.device))
        return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)

    def tokens_to_texts(self, tokens, lengths):
        _, tokenizer = self._first_engine()
        return tokenizer.batch_decode([row[:ln] for row, ln in zip(tokens, lengths)], skip_special_tokens=False)

    # ---- text -> segments (reference model.py:191-394) -------------------------------------------
    def extract_segments(self, text, spec_time_step):
        return postprocess.extract_segments(text, spec_time_step, self.cluster_codebook, self.segment_matcher)

    def parse_generation(self, generated_text_list, sliced_audios_features, min_segment_length, audio_duration,
                         spec_time_step, num_trials, eps, time_per_frame_for_voting, consolidation_method):
        return postprocess.parse_generation(generated_text_list, sliced_audios_features, self.cluster_codebook,
                                            min_segment_length, audio_duration, spec_time_step, num_trials, eps,
                                            time_per_frame_for_voting, consolidation_method, self.precision_bits,
                                            self.segment_matcher)

    def custom_distance(self, segment1, segment2):
        return postprocess.custom_distance(segment1, segment2)

    def consolidate_trials_by_clustering(self, trials, eps, min_samples):
        return postprocess.consolidate_by_clustering(trials, eps, min_samples)

    def consolidate_trials_by_voting(self, trials, time_per_frame_for_voting):
        return postprocess.consolidate_by_voting(trials, time_per_frame_for_voting, self.cluster_codebook)

    def resolve_segmentation_params(self, min_frequency=None, spec_time_step=None, min_segment_length=None, eps=None,
                                    time_per_frame_for_voting=None):
        """The None-defaults of segment() (reference model.py:416-432): checkpoint defaults, then multiples of spec_time_step.
        Only None is a default: an explicit 0 stays 0."""
        defaults = self.default_segmentation_config
        if min_frequency is None:
            min_frequency = defaults.get("min_frequency", 0)
        if spec_time_step is None:
            spec_time_step = defaults.get("spec_time_step", 0.0025)
        if min_segment_length is None:
            min_segment_length = spec_time_step * RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP
        if eps is None:
            eps = spec_time_step * RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP * 4
        if time_per_frame_for_voting is None:
            time_per_frame_for_voting = spec_time_step
        return min_frequency, spec_time_step, min_segment_length, eps, time_per_frame_for_voting

    # ---- the public entry point (reference model.py:397-470) -------------------------------------
    @torch.no_grad()
    def segment(self, audio, sr, min_frequency=None, spec_time_step=None, min_segment_length=None, eps=None,
                time_per_frame_for_voting=None, consolidation_method="clustering", max_length=448, batch_size=4,
                num_trials=1, num_beams=4, top_k=1, top_p=1.0, length_penalty=1.0, status_monitor=None):
        min_frequency, spec_time_step, min_segment_length, eps, time_per_frame_for_voting = self.resolve_segmentation_params(
            min_frequency, spec_time_step, min_segment_length, eps, time_per_frame_for_voting)
        sliced = self.get_sliced_audios_features(audio, sr, min_frequency, spec_time_step, num_trials)
        texts = self.generate_segment_text(sliced, batch_size, max_length, num_beams, top_k, top_p, length_penalty,
                                           status_monitor)
        prediction = self.parse_generation(texts, sliced, min_segment_length, len(audio) / sr, spec_time_step,
                                           num_trials, eps, time_per_frame_for_voting, consolidation_method)
        prediction = postprocess.correct_fft_blur(prediction, get_n_fft_given_sr(sr), sr)
        return postprocess.drop_consecutive_duplicates(prediction)

    # ---- many recordings at once (continuous batching across files) -----------------
    @torch.no_grad()
    def segment_batch(self, audios, srs=None, min_frequency=None, spec_time_step=None, min_segment_length=None, eps=None,
                      time_per_frame_for_voting=None, consolidation_method="clustering", max_length=448, batch_size=4,
                      num_trials=1, num_beams=4, top_k=1, top_p=1.0, length_penalty=1.0, status_monitor=None):
        """segment() for a list of recordings with their windows POOLED into shared decode batches.

        The reference batches only inside one file (model.py:653) and its folder mode is a serial loop
        (scripts/segment.py:39-56, evaluate.py:15-24), so short clips decode with 1-2 windows per launch.  Windows are
        independent, hence pooling changes nothing but the batch a window is decoded in: the per-recording results equal
        segment()'s (bit-identical in f32 mode).  `srs` is one int, a list, or None when `audios` yields (audio, sr) pairs —
        `audios` may be a generator that loads files lazily: recordings are consumed group by group.
        PER-RECORDING parameters: `min_frequency`, `spec_time_step`, `min_segment_length`, `eps`, `time_per_frame_for_voting`,
        `num_trials` and `consolidation_method` may each be one value for all recordings or a list with one entry per
        recording (None entries take segment()'s defaults) — what the reference passes per file (evaluate.py:15-24) or per
        species (config/segment_config.json:1-49: sr 16 k-300 k, spec_time_step 0.0005-0.01, min_frequency 0 / 35 000,
        1 or 3 trials).  The front-end runs per recording with its own (sr, spec_time_step, min_frequency) filterbank; the
        windows of all recordings — every one a [80, 1000] log-mel image whatever its rate — share ONE pooled decode.
        The decode parameters (max_length, num_beams, top_k, top_p, length_penalty) are per call.
        Returns a list of prediction dicts."""
        if srs is None:
            pairs = iter(audios)
        elif isinstance(srs, (int, float)):
            pairs = ((a, srs) for a in audios)
        else:
            pairs = zip(audios, srs)
        per_item = [_per_item(v) for v in (min_frequency, spec_time_step, min_segment_length, eps, time_per_frame_for_voting,
                                          num_trials, consolidation_method)]
        out, group, pending = [], [], 0

        def flush():
            nonlocal group, pending
            pooled = [w for g in group for w in g["windows"]]
            texts = self.generate_segment_text(pooled, batch_size, max_length, num_beams, top_k, top_p, length_penalty,
                                               status_monitor) if pooled else []
            pos = 0
            for g in group:
                mine = texts[pos:pos + len(g["windows"])]
                pos += len(g["windows"])
                pred = self.parse_generation(mine, g["windows"], g["min_segment_length"], g["duration"], g["spec_time_step"],
                                             g["num_trials"], g["eps"], g["frame"], g["method"])
                pred = postprocess.correct_fft_blur(pred, get_n_fft_given_sr(g["sr"]), g["sr"])
                out.append(postprocess.drop_consecutive_duplicates(pred))
            group, pending = [], 0

        # recordings are pooled in bounded groups (~POOL_WINDOWS windows): features of a group are freed before the next
        # group is cut, so a large folder needs no more device memory than a small one
        for (audio, sr), mf, sts, msl, e, tpf, nt, method in zip(pairs, *per_item):
            mf, sts, msl, e, tpf = self.resolve_segmentation_params(mf, sts, msl, e, tpf)
            windows = self.get_sliced_audios_features(audio, sr, mf, sts, nt)
            group.append(dict(duration=len(audio) / sr, sr=sr, windows=windows, spec_time_step=sts, min_segment_length=msl, eps=e,
                              frame=tpf, num_trials=nt, method=method))
            pending += len(windows)
            if pending >= POOL_WINDOWS:
                flush()
        if group:
            flush()
        return out

    # ---- scoring helpers (reference model.py:474-569) --------------------------------------------
    def segment_score(self, prediction, label, target_cluster=None, tolerance=None):
        if tolerance is None:
            tolerance = self.default_segmentation_config.get("spec_time_step", 0.0025) * 4
        return scoring.segment_score(prediction, label, target_cluster, tolerance)

    def frame_score(self, prediction, label, target_cluster=None, time_per_frame_for_scoring=None):
        if time_per_frame_for_scoring is None:
            time_per_frame_for_scoring = min(0.001, self.default_segmentation_config.get("spec_time_step", 0.0025))
        return scoring.frame_score(prediction, label, target_cluster, time_per_frame_for_scoring)

    # ---- shared construction helpers ---------------------------------------------------------------
    def _adopt_config(self, hf_config):
        self.total_spec_columns = hf_config["total_spec_columns"]
        self.cluster_codebook = hf_config["cluster_codebook"]
        self.inverse_cluster_codebook = {cid: name for name, cid in self.cluster_codebook.items()}
        if "default_segmentation_config" in hf_config:
            self.default_segmentation_config.update(hf_config["default_segmentation_config"])


def _resolve_devices(device, device_ids):
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if str(device).startswith("cpu"):
        raise _lib.WsegError("whisperseg_amd has no CPU execution path: an MI355X (gfx950) device is required")
    _lib.load(require_device=True)
    return [torch.device("cuda", int(i)) for i in device_ids]


class WhisperSegmenter(SegmenterBase):
    """reference model.py:625-676: one model replica (+ tokenizer) per entry of device_ids."""

    def __init__(self, model_path, device=None, device_ids=[0, ], dtype=None):
        super().__init__()
        self.device_list = _resolve_devices(device, device_ids)
        model_dir = resolve_model_dir(model_path)
        dtype = dtype or os.environ.get("WHISPERSEG_AMD_DTYPE", DEFAULT_DTYPE)
        self.model_list = [Engine.from_pretrained(model_dir, device=dev, dtype=dtype) for dev in self.device_list]
        self.tokenizer_list = [WhisperSegTokenizer.from_pretrained(model_dir, language="english") for _ in self.device_list]
        hf_config = _read_json(os.path.join(model_dir, "config.json"))
        self._adopt_config(hf_config)
        self.suppress_tokens, self.begin_suppress_tokens = load_generation_settings(model_dir, hf_config["vocab_size"])

    def generate_segment_text_core(self, sliced_audios_features, batch_size, max_length, num_beams, top_k, top_p,
                                   length_penalty, generated_texts_dict, thread_id, status_monitor=None):
        generated_texts_dict[thread_id] = self._decode_batches(
            self.model_list[thread_id], self.tokenizer_list[thread_id], sliced_audios_features, batch_size, max_length,
            num_beams, top_k, top_p, length_penalty, status_monitor)


class WhisperSegmenterFast(WhisperSegmenter):
    """reference model.py:678-746 is the CTranslate2 backend (float16 on a GPU, model.py:691), which the reference's CLI and
    evaluation try first (scripts/segment.py:34-37, evaluate.py:62-65).  Here it is the same MI355X engine as WhisperSegmenter and
    — because those call sites make it the DEFAULT path of the CLI — it defaults to the same split-precision mode ("f16x3": rows
    identical to the reference's on all 6 200 sweep recordings).  `dtype="f16"` (or $WHISPERSEG_AMD_DTYPE=f16) selects what CT2
    computes in: plain IEEE half, 1.9x faster, 98 % of the held-out recordings within +-1 mel frame.  A CTranslate2-converted
    directory (binary `model.bin` + `hf_model/` without HF weights) cannot be read and raises, which makes the try-Fast-then-
    fallback idiom behave as it does upstream when ctranslate2 is missing."""

    def __init__(self, model_path, device=None, device_ids=[0, ], dtype=None):
        model_dir = resolve_model_dir(model_path)
        checkpoint_files(model_dir)      # raises FileNotFoundError for a CTranslate2-only directory
        super().__init__(model_path, device=device, device_ids=device_ids,
                         dtype=dtype or os.environ.get("WHISPERSEG_AMD_DTYPE", DEFAULT_DTYPE))


class WhisperSegmenterForEval(SegmenterBase):
    """reference model.py:572-622: single device, optionally built from in-memory objects.

    `model` may be an `Engine`, or any object exposing `.state_dict()` and `.config` in HF layout
    (e.g. the HF model the reference's train.py hands over, train.py:134) — its weights are converted."""

    def __init__(self, model_path=None, device=None, model=None, tokenizer=None, dtype=None):
        super().__init__()
        dtype = dtype or os.environ.get("WHISPERSEG_AMD_DTYPE", DEFAULT_DTYPE)
        if model_path is not None:
            self.device_list = _resolve_devices(device, [0]) if device is None or str(device) in ("cuda", "cpu") \
                else _resolve_devices("cuda", [torch.device(device).index or 0])
            model_dir = resolve_model_dir(model_path)
            self.model = Engine.from_pretrained(model_dir, device=self.device_list[0], dtype=dtype)
            self.tokenizer = WhisperSegTokenizer.from_pretrained(model_dir, language="english")
            hf_config = _read_json(os.path.join(model_dir, "config.json"))
            self.suppress_tokens, self.begin_suppress_tokens = load_generation_settings(model_dir, hf_config["vocab_size"])
        else:
            model = getattr(model, "module", model)
            if isinstance(model, Engine):
                self.model, hf_config = model, getattr(model, "hf_config", None)
                if hf_config is None:
                    raise ValueError("Engine passed as model= must carry .hf_config (total_spec_columns, cluster_codebook)")
            else:
                hf_config = model.config if isinstance(model.config, dict) else model.config.to_dict()
                dev = device if device is not None else "cuda:0"
                self.model = Engine.from_state_dict({k: v.detach() for k, v in model.state_dict().items()}, hf_config,
                                                    device=dev, dtype=dtype)
                gen = getattr(model, "generation_config", None)
                if gen is not None:
                    self.suppress_tokens = list(getattr(gen, "suppress_tokens", None) or [])
                    self.begin_suppress_tokens = list(getattr(gen, "begin_suppress_tokens", None) or [])
            self.device_list = [self.model.device]
            self.tokenizer = tokenizer
        self.device = self.device_list[0]
        self.hf_config = hf_config
        self._adopt_config(hf_config)

    def update_cluster_codebook(self, cluster_codebook):
        self.hf_config["cluster_codebook"] = cluster_codebook
        self.cluster_codebook = cluster_codebook
        self.inverse_cluster_codebook = {cid: name for name, cid in cluster_codebook.items()}

    def generate_segment_text(self, sliced_audios_features, batch_size, max_length, num_beams, top_k=1, top_p=1.0,
                              length_penalty=1.0, status_monitor=None):
        return self._decode_batches(self.model, self.tokenizer, sliced_audios_features, batch_size, max_length,
                                    num_beams, top_k, top_p, length_penalty, status_monitor)
