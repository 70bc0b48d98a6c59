"""TensorBoard event files without the tensorboard package.

The reference logs through `torch.utils.tensorboard.SummaryWriter(exp_root).add_scalar(tag, value, global_step)`
(Trainer/base_trainer.py:8, 39, 103; Trainer/trainer11.py:8, 38, 59, 112).  tensorboard is not installable here, and the
only thing the trainers use is scalar logging, so this module writes the same files by hand:

  file      events.out.tfevents.<unix time>.<host>  =  a sequence of TFRecords
  TFRecord  uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload)      (little endian)
  payload   Event proto:  1: wall_time (double)   2: step (int64)   3: file_version (string, first record: "brain.Event:2")
                          5: summary { 1: value { 1: tag (string)   2: simple_value (float) } }

`read_events` parses them back (checksums verified); TensorBoard reads them as it reads the reference's.  Host code only.
"""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1      # CRC-32C (Castagnoli), reflected
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1          # int64 two's complement on the wire
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag, value, step, wall_time):
    val = _field_bytes(1, tag.encode("utf-8")) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
    summary = _field_bytes(1, val)
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint(2 << 3) + _varint(int(step)) + _field_bytes(5, summary)


def encode_version_event(wall_time):
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _field_bytes(3, b"brain.Event:2")


def record(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


class EventFileWriter:
    """`SummaryWriter(logdir)` as far as the reference's trainers use it: add_scalar / flush / close."""

    def __init__(self, logdir, flush_every=64):
        os.makedirs(logdir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(now)}.{socket.gethostname()}.{os.getpid()}")
        self._f = open(self.path, "ab")
        self._f.write(record(encode_version_event(now)))
        self._pending, self._flush_every = 0, flush_every

    def add_scalar(self, tag, value, global_step=0, walltime=None):
        if hasattr(value, "item"):
            value = value.item()
        self._f.write(record(encode_scalar_event(tag, value, global_step, time.time() if walltime is None else walltime)))
        self._pending += 1
        if self._pending >= self._flush_every:
            self.flush()

    def flush(self):
        if not self._f.closed:
            self._f.flush()
        self._pending = 0

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _read_varint(buf, i):
    n, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _parse(buf):
    """{field number: [raw values]} of one message (varint -> int, 64-bit / 32-bit -> bytes, length-delimited -> bytes)."""
    out, i = {}, 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        elif wt == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        else:
            raise ValueError(f"unsupported wire type {wt}")
        out.setdefault(num, []).append(v)
    return out


def read_events(path):
    """[{'wall_time', 'step', 'file_version' | ('tag', 'value')}] of an event file; raises on a checksum mismatch."""
    data, i, out = open(path, "rb").read(), 0, []
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != masked_crc(head):
            raise ValueError(f"{path}: corrupt record length at byte {i}")
        payload = data[i + 12:i + 12 + n]
        if struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] != masked_crc(payload):
            raise ValueError(f"{path}: corrupt record payload at byte {i}")
        i += 16 + n
        ev = _parse(payload)
        rec = {"wall_time": struct.unpack("<d", ev[1][0])[0], "step": ev.get(2, [0])[0]}
        if rec["step"] >= 1 << 63:
            rec["step"] -= 1 << 64
        if 3 in ev:
            rec["file_version"] = ev[3][0].decode()
        for s in ev.get(5, []):
            for v in _parse(s).get(1, []):
                f = _parse(v)
                out.append(dict(rec, tag=f[1][0].decode(), value=struct.unpack("<f", f[2][0])[0]))
        if 5 not in ev:
            out.append(rec)
    return out
