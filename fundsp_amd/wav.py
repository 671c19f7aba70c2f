"""WAV sink with the byte layout of the reference's writer (src/write.rs:26-116: 44-byte RIFF/WAVE header, fmt block of
16 bytes, format 1 = 16-bit PCM or 3 = IEEE float, interleaved little-endian frames), for audible A/B checks of rendered
banks.  Data format on the output side of the path; pure numpy, no device."""
import struct

import numpy as np


def _header(data_length, fmt, channels, sample_rate):
    sample_bytes = 2 if fmt == 1 else 4
    return b"".join([
        b"RIFF", struct.pack("<I", data_length + 36), b"WAVE", b"fmt ", struct.pack("<I", 16),
        struct.pack("<HHI", fmt, channels, sample_rate),
        struct.pack("<I", sample_rate * channels * sample_bytes),          # data rate in bytes per second
        struct.pack("<HH", channels * sample_bytes, sample_bytes * 8),     # frame length, bits per sample
        b"data", struct.pack("<I", data_length),
    ])


def wav32_bytes(data, sample_rate):
    """Wave::write_wav32 (write.rs:81-100). data: [channels][frames] float32."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(data, dtype=np.float32)))
    ch, n = x.shape
    assert ch > 0
    return _header(4 * ch * n, 3, ch, int(round(sample_rate))) + x.T.astype("<f4").tobytes()


def wav16_bytes(data, sample_rate):
    """Wave::write_wav16 (write.rs:59-79): round(clamp11(x) * 32767.49) as i16."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(data, dtype=np.float32)))
    ch, n = x.shape
    assert ch > 0
    s = np.clip(x, np.float32(-1), np.float32(1)) * np.float32(32767.49)
    s = np.where(s >= 0, np.floor(s + np.float32(0.5)), np.ceil(s - np.float32(0.5)))  # libm roundf: half away from zero
    return _header(2 * ch * n, 1, ch, int(round(sample_rate))) + s.T.astype("<i2").tobytes()


def save_wav32(path, data, sample_rate):
    with open(path, "wb") as f:
        f.write(wav32_bytes(data, sample_rate))


def save_wav16(path, data, sample_rate):
    with open(path, "wb") as f:
        f.write(wav16_bytes(data, sample_rate))
