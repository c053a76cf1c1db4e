"""CPU tests of the host-side decode: load_audio returns what librosa.load(path, sr=16000) hands the reference
(kws/libs/audio_processor.py:145) -- float32 mono in [-1, 1) -- for every WAV encoding libsndfile reads as PCM / float."""
import struct
import wave

import numpy as np
import pytest

from kws.common.errors import AudioProcessingError
from kws.libs.audio_processor import fix_length, load_audio, load_pcm16


def write_wav(path, payload: bytes, tag: int, channels: int, rate: int, bits: int, extensible: bool = False):
    block = channels * bits // 8
    if extensible:
        guid_tail = bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, 0, tag) + guid_tail
    else:
        fmt = struct.pack("<HHIIHH", tag, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd"
    body += b"data" + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_integer_pcm_scaling(tmp_path):
    rng = np.random.default_rng(0)
    i16 = rng.integers(-32768, 32768, 1000, dtype=np.int16)
    write_wav(tmp_path / "a16.wav", i16.tobytes(), 1, 1, 16000, 16)
    assert np.array_equal(load_audio(tmp_path / "a16.wav"), i16.astype(np.float32) / np.float32(32768))
    assert np.array_equal(load_pcm16(tmp_path / "a16.wav"), i16)
    u8 = rng.integers(0, 256, 999, dtype=np.uint8)                      # odd payload length: pad byte after the chunk
    write_wav(tmp_path / "a8.wav", u8.tobytes(), 1, 1, 16000, 8)
    assert np.array_equal(load_audio(tmp_path / "a8.wav"), (u8.astype(np.float32) - 128) / 128)
    i24 = rng.integers(-(1 << 23), 1 << 23, 500)
    raw = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in i24)
    write_wav(tmp_path / "a24.wav", raw, 1, 1, 16000, 24, extensible=True)
    assert np.array_equal(load_audio(tmp_path / "a24.wav"), (i24 / 8388608.0).astype(np.float32))
    i32 = rng.integers(-(1 << 31), 1 << 31, 400, dtype=np.int64).astype("<i4")
    write_wav(tmp_path / "a32.wav", i32.tobytes(), 1, 1, 16000, 32)
    assert np.array_equal(load_audio(tmp_path / "a32.wav"), (i32.astype(np.float64) / 2147483648.0).astype(np.float32))


def test_float_and_stereo(tmp_path):
    rng = np.random.default_rng(1)
    f32 = rng.uniform(-1, 1, (300, 2)).astype("<f4")
    write_wav(tmp_path / "f32s.wav", f32.tobytes(), 3, 2, 16000, 32)
    assert np.array_equal(load_audio(tmp_path / "f32s.wav"), f32.mean(axis=1, dtype=np.float32))   # librosa.to_mono
    f64 = rng.uniform(-1, 1, 200).astype("<f8")
    write_wav(tmp_path / "f64.wav", f64.tobytes(), 3, 1, 16000, 64)
    assert np.array_equal(load_audio(tmp_path / "f64.wav"), f64.astype(np.float32))
    with wave.open(str(tmp_path / "std.wav"), "wb") as w:              # a file the standard library writes
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        st = rng.integers(-3000, 3000, (100, 2), dtype=np.int16)
        w.writeframes(st.tobytes())
    assert np.array_equal(load_audio(tmp_path / "std.wav"), (st.astype(np.float32) / np.float32(32768)).mean(axis=1, dtype=np.float32))


def test_refusals_and_resampling(tmp_path):
    x = (np.sin(2 * np.pi * 440 * np.arange(8000) / 8000.0) * 12000).astype(np.int16)
    write_wav(tmp_path / "r8k.wav", x.tobytes(), 1, 1, 8000, 16)
    with pytest.raises(AudioProcessingError):
        load_audio(tmp_path / "r8k.wav")                                  # librosa would resample with soxr: refused by default
    y = load_audio(tmp_path / "r8k.wav", resample=True)                   # polyphase resampler: parity unpinned, but sane
    assert y.dtype == np.float32 and len(y) == 16000
    ref = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000.0) * (12000 / 32768)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(AudioProcessingError):
        load_audio(tmp_path / "junk.wav")
    write_wav(tmp_path / "adpcm.wav", b"\0" * 64, 2, 1, 16000, 4)
    with pytest.raises(AudioProcessingError):
        load_audio(tmp_path / "adpcm.wav")
    assert fix_length(np.arange(5, dtype=np.float32), 8).tolist() == [0, 1, 2, 3, 4, 0, 0, 0]
