"""TFRecord + tf.train.Example reader/writer for the reference's audio datasets, in plain
Python/numpy -- TensorFlow is not a dependency of this package.

What the reference does with these files:
  data.py:25-43                  TFRecordDataset(f"{datadir}/{dataset}.tfrecords") -> parse_single_example with
                                 {"audio": FixedLenFeature([sample_duration], float32)} -> batch(minibatch_size)
                                 -> shuffle(buffer_size=24) -> repeat() -> one-shot iterator -> batch["audio"]
  training_estimators.py:76-95   the same parse with [2**16], then shuffle(24).repeat().batch(batch_size)
  make-small-dataset.py:18-32    writer: Example(features={"audio": Feature(float_list=FloatList(value=datum))})

File format (TensorFlow's record writer): for each record
    uint64 length | uint32 masked_crc32c(length bytes) | bytes data[length] | uint32 masked_crc32c(data)
with masked_crc = ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32 and crc = CRC-32C (Castagnoli).
`data` is a serialized tf.train.Example:  Example{1: Features{1: map<string, Feature>}},
Feature{1: BytesList | 2: FloatList | 3: Int64List}, FloatList{1: repeated float (packed)}.
"""
from __future__ import annotations

import struct
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np

_MASK_DELTA = 0xA282EAD8


def _make_crc_table():
    poly = 0x82F63B78
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab[i] = c
    return tab


_CRC_TABLE = _make_crc_table()


_native_crc = False      # False: not looked for yet; None: unavailable; else the C function


def _find_native_crc():
    """cmps_crc32c of libcmps.so (SSE4.2 crc32 instruction, GB/s); None when the library is not built -- the pure-Python
    loop below computes the same value at ~1 MB/s, which only matters for verify=True on large files."""
    global _native_crc
    if _native_crc is False:
        try:
            from . import _capi
            _native_crc = _capi.load().cmps_crc32c
        except Exception:
            _native_crc = None
    return _native_crc


def crc32c(data: bytes) -> int:
    fn = _find_native_crc()
    if fn is not None:
        return int(fn(bytes(data), len(data), 0))
    crc = 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _crc32c_python(data: bytes) -> int:
    crc = 0xFFFFFFFF
    for b in data:
        crc = int(_CRC_TABLE[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    crc = crc32c(data)
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------
# record framing
# ---------------------------------------------------------------------------------------------------
def read_records(path: str, verify: bool = False) -> Iterator[bytes]:
    """Yields the payload of every record.  verify=True checks both CRCs (slow in pure Python: test use)."""
    with open(path, "rb") as fh:
        while True:
            head = fh.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError(f"{path}: truncated record header")
            (length,) = struct.unpack("<Q", head[:8])
            if verify and struct.unpack("<I", head[8:])[0] != masked_crc32c(head[:8]):
                raise IOError(f"{path}: corrupt record length")
            data = fh.read(length)
            foot = fh.read(4)
            if len(data) < length or len(foot) < 4:
                raise IOError(f"{path}: truncated record")
            if verify and struct.unpack("<I", foot)[0] != masked_crc32c(data):
                raise IOError(f"{path}: corrupt record data")
            yield data


def write_records(path: str, payloads) -> None:
    with open(path, "wb") as fh:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            fh.write(head)
            fh.write(struct.pack("<I", masked_crc32c(head)))
            fh.write(data)
            fh.write(struct.pack("<I", masked_crc32c(data)))


# ---------------------------------------------------------------------------------------------------
# minimal protobuf wire format
# ---------------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: bytes):
    """Yields (field_number, wire_type, value) with value = int (varint / fixed) or bytes (length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def parse_example(data: bytes) -> Dict[str, np.ndarray]:
    """tf.train.Example -> {feature name: float32 / int64 array or list of bytes}."""
    out: Dict[str, np.ndarray] = {}
    for num, wt, features in _fields(data):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(features):          # map<string, Feature> entries
            if fnum != 1 or fwt != 2:
                continue
            key, feat = None, b""
            for enum_, ewt, ev in _fields(entry):
                if enum_ == 1:
                    key = ev.decode("utf-8")
                elif enum_ == 2:
                    feat = ev
            if key is None:
                continue
            for knum, kwt, lst in _fields(feat):
                if kwt != 2:
                    continue
                if knum == 2:                                  # FloatList
                    parts: List[np.ndarray] = []
                    for vnum, vwt, vv in _fields(lst):
                        if vnum == 1 and vwt == 2:             # packed
                            parts.append(np.frombuffer(vv, dtype="<f4"))
                        elif vnum == 1 and vwt == 5:           # one value
                            parts.append(np.frombuffer(vv, dtype="<f4"))
                    out[key] = np.concatenate(parts) if parts else np.zeros(0, np.float32)
                elif knum == 3:                                # Int64List
                    vals: List[int] = []
                    for vnum, vwt, vv in _fields(lst):
                        if vnum == 1 and vwt == 2:
                            p = 0
                            while p < len(vv):
                                x, p = _varint(vv, p)
                                vals.append(x - (1 << 64) if x >= (1 << 63) else x)
                        elif vnum == 1 and vwt == 0:
                            vals.append(vv)
                    out[key] = np.asarray(vals, dtype=np.int64)
                elif knum == 1:                                # BytesList
                    out[key] = [vv for vnum, vwt, vv in _fields(lst) if vnum == 1 and vwt == 2]
    return out


def make_example(audio: np.ndarray) -> bytes:
    """Serialised Example with one float_list feature "audio" (make-small-dataset.py:24-32)."""
    vals = np.ascontiguousarray(audio, dtype="<f4").tobytes()
    float_list = _enc_ld(1, vals)
    feature = _enc_ld(2, float_list)
    entry = _enc_ld(1, b"audio") + _enc_ld(2, feature)
    features = _enc_ld(1, entry)
    return _enc_ld(1, features)


def write_audio_tfrecord(path: str, clips) -> None:
    write_records(path, (make_example(c) for c in clips))


# ---------------------------------------------------------------------------------------------------
# the reference's pipelines
# ---------------------------------------------------------------------------------------------------
def _shuffle(source: Iterator, buffer_size: int, rng: np.random.Generator) -> Iterator:
    """tf.data shuffle: keep a buffer of `buffer_size` elements, emit a random one, refill from the source."""
    buf = []
    for item in source:
        buf.append(item)
        if len(buf) >= buffer_size:
            j = int(rng.integers(len(buf)))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield buf.pop()
    while buf:
        j = int(rng.integers(len(buf)))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()


def _audio_records(path: str, length: int, verify: bool) -> Iterator[np.ndarray]:
    for rec in read_records(path, verify=verify):
        ex = parse_example(rec)
        if "audio" not in ex:
            raise KeyError(f"{path}: record without an 'audio' feature")
        a = np.asarray(ex["audio"], dtype=np.float32)
        if a.shape[0] != length:          # FixedLenFeature([length]) rejects other sizes (data.py:32)
            raise ValueError(f"{path}: 'audio' has {a.shape[0]} values, FixedLenFeature expects {length}")
        yield a


def audio_batches(path: str, minibatch_size: int, sample_duration: int, seed: int = 0, shuffle_buffer: int = 24,
                  order: str = "data", verify: bool = False) -> Callable[[], np.ndarray]:
    """A zero-argument callable returning the next float32 [minibatch_size, sample_duration] batch, forever.
    order="data":       batch -> shuffle(24) -> repeat      (data.py:37-40; the shuffle is over BATCHES; a final
                        short batch is kept, as tf.data's batch() does by default)
    order="estimator":  shuffle(24) -> repeat -> batch      (training_estimators.py:92)"""
    rng = np.random.default_rng(seed)

    def gen_data():
        while True:                                            # repeat()
            def batches():
                cur = []
                for a in _audio_records(path, sample_duration, verify):
                    cur.append(a)
                    if len(cur) == minibatch_size:
                        yield np.stack(cur)
                        cur = []
                if cur:
                    yield np.stack(cur)
            yield from _shuffle(batches(), shuffle_buffer, rng)

    def gen_estimator():
        def records():
            while True:                                        # shuffle(24).repeat(): reshuffled every epoch
                yield from _shuffle(_audio_records(path, sample_duration, verify), shuffle_buffer, rng)
        cur = []
        for a in records():
            cur.append(a)
            if len(cur) == minibatch_size:
                yield np.stack(cur)
                cur = []

    it = gen_data() if order == "data" else gen_estimator()
    return lambda: next(it)
