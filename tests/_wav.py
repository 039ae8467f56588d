"""Test helper: build RIFF / WAVE file images (the inputs of the host WAV reader and the crawler pipeline)."""
import struct

import numpy as np


def wav_bytes(data, channels, bits, is_float=False, rate=44100, extra_chunks=False, extensible=False):
    """data: samples (interleaved, any shape) as uint8 (8 bit), int16, int32 (24 bit: low three bytes are stored;
    32 bit), float32 or float64.  extra_chunks: a LIST chunk of odd size before `data` (word alignment) and a
    trailing chunk after it.  extensible: WAVE_FORMAT_EXTENSIBLE header (40-byte fmt chunk)."""
    flat = np.ascontiguousarray(data).reshape(-1)
    if bits == 24:
        payload = b"".join(struct.pack("<i", int(v))[:3] for v in flat)
    else:
        payload = flat.tobytes()
    tag = 0xFFFE if extensible else (3 if is_float else 1)
    fmt = struct.pack("<HHIIHH", tag, channels, rate, channels * rate * bits // 8, channels * bits // 8, bits)
    if extensible:
        sub = (3 if is_float else 1).to_bytes(2, "little") + bytes.fromhex("000000001000800000aa00389b71")
        fmt += struct.pack("<HHI", 22, bits, 0) + sub
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunks:
        odd = b"INFOISFT" + struct.pack("<I", 5) + b"afec\x00"          # 17 bytes: odd
        chunks += b"LIST" + struct.pack("<I", len(odd)) + odd + b"\x00"
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    if extra_chunks:
        chunks += b"cue " + struct.pack("<I", 4) + struct.pack("<I", 0)
    body = b"WAVE" + chunks
    return b"RIFF" + struct.pack("<I", len(body)) + body


def parse_wav(image):
    """Minimal RIFF / WAVE parse for the tests' own use (PCM only): -> (channels, rate, bits, frames, payload bytes).
    Raises ValueError for anything that is not a PCM WAVE file."""
    if len(image) < 12 or image[:4] != b"RIFF" or image[8:12] != b"WAVE":
        raise ValueError("not a RIFF / WAVE image")
    i, fmt, payload = 12, None, None
    while i + 8 <= len(image):
        name, size = image[i:i + 4], struct.unpack("<I", image[i + 4:i + 8])[0]
        if name == b"fmt ":
            fmt = struct.unpack("<HHIIHH", image[i + 8:i + 24])
        elif name == b"data":
            payload = image[i + 8:i + 8 + size]
        i += 8 + size + (size & 1)
    if fmt is None or payload is None or fmt[0] != 1:
        raise ValueError("no PCM fmt / data chunk")
    channels, rate, bits = fmt[1], fmt[2], fmt[5]
    frames = len(payload) // (channels * bits // 8)
    return channels, rate, bits, frames, payload[:frames * channels * bits // 8]
