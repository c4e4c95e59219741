"""Host-side reader / writer of the ship detector's TFRecord files
(shipdetection/data_extract.py:33-50 writes them, shipdetection/train.py:28-63 reads them).

A record is a tf.train.Example whose five features are byte strings:
  image        float32 pixels, raw bytes          image_shape   int32 [3], raw bytes
  bbox         float32 [n,4] boxes, raw bytes     bbox_shape    int32 [2], raw bytes
  filename     utf-8
TFRecord framing: uint64 length, uint32 masked crc32c of the length, payload, uint32 masked
crc32c of the payload.  Only what these files use of the protobuf wire format is handled
(length-delimited fields); the checksums are verified on request (pure Python, slow).
File IO stays on the host; `collate` hands the batch to the device in the layout
frcnn_pad_gt takes.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List

import numpy as np

_CRC_TABLE = None


def _crc32c(data: bytes) -> int:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    c = 0xFFFFFFFF
    for byte in data:
        c = _CRC_TABLE[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(buf: bytes, p: int):
    v, shift = 0, 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, p
        shift += 7


def _fields(buf: bytes) -> Iterator:
    """(field number, payload) of the length-delimited fields of one message."""
    p = 0
    while p < len(buf):
        tag, p = _varint(buf, p)
        wire = tag & 7
        if wire == 2:
            n, p = _varint(buf, p)
            yield tag >> 3, buf[p:p + n]
            p += n
        elif wire == 0:
            _, p = _varint(buf, p)
        elif wire == 1:
            p += 8
        elif wire == 5:
            p += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")


def parse_example(payload: bytes) -> Dict[str, bytes]:
    """tf.train.Example -> {feature name: first bytes_list value}."""
    out = {}
    for f1, features in _fields(payload):                 # Example.features = 1
        if f1 != 1:
            continue
        for f2, entry in _fields(features):               # Features.feature (map) = 1
            if f2 != 1:
                continue
            key, value = None, None
            for f3, item in _fields(entry):               # map entry: key = 1, value = 2
                if f3 == 1:
                    key = item.decode("utf-8")
                elif f3 == 2:
                    for f4, blist in _fields(item):       # Feature.bytes_list = 1
                        if f4 == 1:
                            for f5, v in _fields(blist):  # BytesList.value = 1
                                if f5 == 1 and value is None:
                                    value = v
            if key is not None and value is not None:
                out[key] = value
    return out


def decode_record(example: Dict[str, bytes]) -> dict:
    """shipdetection/train.py:28-45 without the padding: arrays of one record."""
    image_shape = np.frombuffer(example["image_shape"], np.int32)
    bbox_shape = np.frombuffer(example["bbox_shape"], np.int32)
    return dict(image=np.frombuffer(example["image"], np.float32).reshape(image_shape),
                bbox=np.frombuffer(example["bbox"], np.float32).reshape(bbox_shape),
                filename=example["filename"].decode("utf-8"))


def read_tfrecord(path: str, verify: bool = False) -> Iterator[dict]:
    with open(path, "rb") as fh:
        while True:
            head = fh.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("truncated TFRecord header")
            n, len_crc = struct.unpack("<QI", head)
            payload = fh.read(n)
            tail = fh.read(4)
            if len(payload) < n or len(tail) < 4:
                raise ValueError("truncated TFRecord payload")
            if verify:
                if _masked_crc(head[:8]) != len_crc:
                    raise ValueError("TFRecord length checksum mismatch")
                if _masked_crc(payload) != struct.unpack("<I", tail)[0]:
                    raise ValueError("TFRecord payload checksum mismatch")
            yield decode_record(parse_example(payload))


def _ld(field: int, payload: bytes) -> bytes:
    n, out = len(payload), bytearray([(field << 3) | 2])
    while True:
        out.append((n & 0x7F) | (0x80 if n > 0x7F else 0))
        n >>= 7
        if not n:
            break
    return bytes(out) + payload


def serialize_example(dic: dict) -> bytes:
    """shipdetection/data_extract.py:33-50: dict(image, image_shape, bbox, bbox_shape,
    filename) -> serialized tf.train.Example."""
    feats = {
        "image": np.asarray(dic["image"], np.float32).tobytes(),
        "bbox": np.asarray(dic["bbox"], np.float32).tobytes(),
        "filename": dic["filename"].encode("utf-8"),
        "image_shape": np.asarray(dic["image_shape"], np.int32).tobytes(),
        "bbox_shape": np.asarray(dic["bbox_shape"], np.int32).tobytes(),
    }
    entries = b"".join(
        _ld(1, _ld(1, k.encode("utf-8")) + _ld(2, _ld(1, _ld(1, v)))) for k, v in feats.items())
    return _ld(1, entries)


def write_tfrecord(path: str, records: List[dict]) -> None:
    with open(path, "wb") as fh:
        for r in records:
            payload = serialize_example(r)
            head = struct.pack("<Q", len(payload))
            fh.write(head + struct.pack("<I", _masked_crc(head)) + payload +
                     struct.pack("<I", _masked_crc(payload)))


def collate(records: List[dict]):
    """Records -> (images [B,H,W,C] float32, boxes [total,4] float32, offsets [B+1] int32):
    the ragged layout frcnn_pad_gt takes.  Images must share a shape."""
    images = np.stack([r["image"] for r in records]).astype(np.float32)
    boxes = np.concatenate([np.asarray(r["bbox"], np.float32).reshape(-1, 4) for r in records])
    offsets = np.zeros(len(records) + 1, np.int32)
    offsets[1:] = np.cumsum([np.asarray(r["bbox"]).reshape(-1, 4).shape[0] for r in records])
    return images, boxes, offsets
