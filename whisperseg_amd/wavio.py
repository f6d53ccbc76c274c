"""Minimal RIFF/WAVE reader standing in for `librosa.load(path, sr=None)` (reference scripts/segment.py:48,61;
librosa/soundfile are not in the image): native sampling rate, float32 in [-1, 1), channels averaged to mono.
PCM 8/16/24/32-bit, IEEE float 32/64 and WAVE_FORMAT_EXTENSIBLE are handled; no resampling here."""
import io
import struct

import numpy as np


def _read_chunks(f):
    header = f.read(12)
    if len(header) < 12 or header[:4] not in (b"RIFF", b"RF64") or header[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    while True:
        head = f.read(8)
        if len(head) < 8:
            return
        cid, size = head[:4], struct.unpack("<I", head[4:])[0]
        data = f.read(size)
        if size % 2:
            f.read(1)
        yield cid, data


def load_wav(path_or_file):
    """-> (float32 mono ndarray, sampling_rate)."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        if not hasattr(f, "read"):
            f = io.BytesIO(f)
        fmt, raw = None, None
        for cid, data in _read_chunks(f):
            if cid == b"fmt ":
                tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", data[:16])
                if tag == 0xFFFE and len(data) >= 26:
                    tag = struct.unpack("<H", data[24:26])[0]
                fmt = (tag, ch, sr, bits)
            elif cid == b"data":
                raw = data
        if fmt is None or raw is None:
            raise ValueError("missing fmt or data chunk")
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(raw[: len(raw) // 2 * 2], "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw[: len(raw) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (np.frombuffer(raw[: len(raw) // 4 * 4], "<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(raw, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE format tag {tag}")
    if ch > 1:
        x = x[: len(x) // ch * ch].reshape(-1, ch).mean(axis=1).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), int(sr)
