#!/usr/bin/env python3
"""Command-line segmentation — counterpart of reference scripts/segment.py (same flags, same CSV).

    python scripts/segment.py --model_path DIR --audio_path a.wav --csv_save_path out.csv
    python scripts/segment.py --model_path DIR --audio_folder wavs/ --csv_save_path out.csv
    cat a.wav | python scripts/segment.py --model_path DIR --audio_path - --csv_save_path buffer
"""
import argparse
import csv
import glob
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from model import WhisperSegmenter, WhisperSegmenterFast  # noqa: E402  (root-level shim, as upstream imports it)
from whisperseg_amd.wavio import load_wav  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--model_path")
    p.add_argument("--audio_path", default=None, help="one .wav file, or '-' to read a wav from stdin")
    p.add_argument("--audio_folder", default=None, help="directory of .wav/.WAV files (when --audio_path is absent)")
    p.add_argument("--csv_save_path")
    p.add_argument("--device", default="cuda", help="'cuda' (an MI355X is required; 'cpu' raises)")
    p.add_argument("--device_ids", type=int, nargs="+", default=[0, ], help="GPU indices, one model replica each")
    p.add_argument("--batch_size", default=8, type=int)
    p.add_argument("--min_frequency", default=None, type=int)
    p.add_argument("--spec_time_step", default=None, type=float)
    p.add_argument("--num_trials", default=1, type=int)
    return p


def write_csv(columns, rows, dest):
    """Same text pandas' DataFrame.to_csv(index=False) produces for these columns (repr-shortest floats)."""
    w = csv.writer(dest, lineterminator="\n")
    w.writerow(columns)
    for row in rows:
        w.writerow([repr(v) if isinstance(v, float) else v for v in row])


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.csv_save_path.endswith(".csv") or args.csv_save_path == "buffer", \
        "csv_save_path must ends with .csv or be 'buffer'"
    try:
        segmenter = WhisperSegmenterFast(args.model_path, device=args.device, device_ids=args.device_ids)
    except Exception:
        segmenter = WhisperSegmenter(args.model_path, device=args.device, device_ids=args.device_ids)
    kwargs = dict(min_frequency=args.min_frequency, spec_time_step=args.spec_time_step, num_trials=args.num_trials,
                  batch_size=args.batch_size)
    if args.audio_path is None:
        assert args.audio_folder is not None, "Either audio_path or audio_folder needs to be specified!"
        columns, rows = ["filename", "onset", "offset", "cluster"], []
        paths = glob.glob(args.audio_folder + "/*.wav") + glob.glob(args.audio_folder + "/*.WAV")
        # same rows as the reference's serial loop, but the windows of many files share the engine's decode slots; files
        # are read lazily, group by group, so a large folder needs no more memory than a small one
        results = segmenter.segment_batch((load_wav(path) for path in paths), **kwargs)
        for path, res in zip(paths, results):
            name = os.path.basename(path)
            rows += [(name, on, off, c) for on, off, c in zip(res["onset"], res["offset"], res["cluster"])]
    else:
        audio, sr = load_wav(io.BytesIO(sys.stdin.buffer.read())) if args.audio_path == "-" else load_wav(args.audio_path)
        res = segmenter.segment(audio, sr, **kwargs)
        columns = ["onset", "offset", "cluster"]
        rows = list(zip(res["onset"], res["offset"], res["cluster"]))
    if args.csv_save_path == "buffer":
        buf = io.StringIO()
        write_csv(columns, rows, buf)
        print(buf.getvalue())
    else:
        with open(args.csv_save_path, "w", newline="") as f:
            write_csv(columns, rows, f)


if __name__ == "__main__":
    main()
