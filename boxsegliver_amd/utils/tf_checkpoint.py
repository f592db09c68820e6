"""TensorFlow V2 checkpoints ("tensor bundles": <prefix>.index + <prefix>.data-00000-of-00001) without TensorFlow.

The reference saves and restores its variables with tf.train.Saver (core/estimator.py:694-703, core/models.py:151-185:
`pt.NewCheckpointReader`, `tf.train.get_checkpoint_state`); a user switching over holds such files.  This module reads
them (and writes them, so variables trained here can go back) from the published on-disk format -- TensorFlow itself is
not installable in this image, so the codec is restated from the format, not exercised against TensorFlow:

  <prefix>.index   an SSTable in the LevelDB table format (tensorflow/core/lib/io/table*.{h,cc}, format.cc): data blocks of
                   prefix-compressed (key, value) entries + restart array, each block followed by a 1-byte compression tag
                   (0 none, 1 snappy) and a masked CRC32C; an index block of (separator key -> BlockHandle); a 48-byte
                   footer = metaindex handle, index handle, zero padding, magic 0xdb4775248b80fb57.
                   key ""      -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
                   key <name>  -> BundleEntryProto  {1: dtype, 2: shape, 3: shard_id, 4: offset, 5: size, 6: crc32c fixed32}
                   (tensorflow/core/protobuf/tensor_bundle.proto)
  <prefix>.data-SSSSS-of-NNNNN   the tensors' raw little-endian bytes at [offset, offset + size).
  <dir>/checkpoint a text CheckpointState proto: model_checkpoint_path: "<prefix>".

Host-side only (numpy); no device work.
"""
import os
import re
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------------------ CRC32C
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_TABLE = _make_table()
_TABLE_PY = [int(v) for v in _TABLE]


def _mat_apply(cols, v):
    """cols: 32 ints = images of the basis vectors; v: int or uint32 ndarray."""
    if isinstance(v, np.ndarray):
        out = np.zeros_like(v)
        for b in range(32):
            out ^= np.where((v >> np.uint32(b)) & np.uint32(1), np.uint32(cols[b]), np.uint32(0))
        return out
    out = 0
    for b in range(32):
        if (v >> b) & 1:
            out ^= cols[b]
    return out


def _zeros_operator(nbytes):
    """The linear map `feed nbytes zero bytes` on the raw CRC register, as 32 columns."""
    one = [(_TABLE_PY[(1 << b) & 0xff] ^ ((1 << b) >> 8)) for b in range(32)]
    result = [1 << b for b in range(32)]
    sq = one
    n = int(nbytes)
    while n:
        if n & 1:
            result = [_mat_apply(sq, c) for c in result]
        sq = [_mat_apply(sq, c) for c in sq]
        n >>= 1
    return result


def crc32c(data):
    """CRC-32C (Castagnoli, reflected 0x82F63B78) of a bytes-like object.  Large inputs run lane-parallel: the buffer is
    front-padded with zeros (free for a zero register) to K equal chunks whose raw registers advance in lock step in numpy
    and are then merged pairwise with the `feed zeros` operator."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
    n = buf.size
    if n < 4096:
        c = 0xFFFFFFFF
        for b in buf.tolist():
            c = _TABLE_PY[(c ^ b) & 0xff] ^ (c >> 8)
        return c ^ 0xFFFFFFFF
    lanes = 1
    while lanes < 65536 and lanes * 2048 < n:
        lanes *= 2
    chunk = -(-n // lanes)
    padded = np.zeros(lanes * chunk, np.uint8)
    padded[lanes * chunk - n:] = buf
    cols = padded.reshape(lanes, chunk)
    reg = np.zeros(lanes, np.uint32)
    for i in range(chunk):
        reg = _TABLE[(reg ^ cols[:, i]) & np.uint32(0xff)] ^ (reg >> np.uint32(8))
    op = _zeros_operator(chunk)
    while reg.size > 1:
        reg = _mat_apply(op, reg[0::2]) ^ reg[1::2]
        op = [_mat_apply(op, c) for c in op]
    raw = int(reg[0]) ^ _mat_apply(_zeros_operator(n), 0xFFFFFFFF)
    return raw ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------- varint / protobuf
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("malformed varint")


def _pb_fields(buf):
    """[(field number, wire type, value)]; value = int (varint / fixed) or bytes (length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wt))
        out.append((num, wt, v))
    return out


def _pb_varint(num, v):
    return _put_varint(num << 3) + _put_varint(v)


def _pb_bytes(num, payload):
    return _put_varint((num << 3) | 2) + _put_varint(len(payload)) + payload


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------------------------------------------------ snappy
def snappy_uncompress(buf):
    """Raw snappy block format (the index file's blocks may be compressed with it)."""
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("malformed snappy copy")
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:
            for i in range(ln):                      # overlapping copy: byte by byte
                out.append(out[start + i])
    if len(out) != n:
        raise ValueError("snappy length mismatch: {} != {}".format(len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------------------------------------- table
def _read_block(buf, offset, size, verify=True):
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    (stored,) = struct.unpack_from("<I", buf, offset + size + 1)
    if verify and masked_crc(buf[offset:offset + size + 1]) != stored:
        raise ValueError("checkpoint index: block checksum mismatch at offset {}".format(offset))
    if ctype == 1:
        contents = snappy_uncompress(contents)
    elif ctype != 0:
        raise ValueError("checkpoint index: unknown block compression {}".format(ctype))
    return contents


def _block_entries(block):
    (num_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path):
    """All (key, value) pairs of an SSTable file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("{} is not a TensorFlow checkpoint index (bad magic)".format(path))
    footer = buf[len(buf) - 48:]
    _, pos = _get_varint(footer, 0)                 # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize)):
        off, p = _get_varint(handle, 0)
        size, _ = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, off, size)))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (key, value) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_size=4096):
    """items: (key bytes, value bytes) sorted by key.  Uncompressed blocks."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack("<I", masked_crc(block + b"\x00")))
        return _put_varint(off) + _put_varint(len(block))

    index, cur, cur_bytes = [], [], 0
    for key, value in items:
        cur.append((key, value))
        cur_bytes += len(key) + len(value) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b"", emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------------------ bundle
def _data_file(prefix, shard, num_shards):
    return "{}.data-{:05d}-of-{:05d}".format(prefix, shard, num_shards)


def _parse_entry(value):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, wt, v in _pb_fields(value):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, dim in _pb_fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, s in _pb_fields(dim):
                        if n3 == 1:
                            size = _signed64(s)
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


class CheckpointReader(object):
    """pt.NewCheckpointReader's surface: get_variable_to_shape_map(), has_tensor(name), get_tensor(name)."""

    def __init__(self, prefix):
        self.prefix = str(prefix)
        index = self.prefix + ".index"
        if not os.path.exists(index):
            raise FileNotFoundError("{} (TensorFlow V2 checkpoint index) does not exist".format(index))
        self.num_shards = 1
        self.entries = {}
        for key, value in read_table(index):
            if key == b"":
                for num, _, v in _pb_fields(value):
                    if num == 1:
                        self.num_shards = v
                    elif num == 2 and v != 0:
                        raise ValueError("big-endian checkpoints are not supported")
                continue
            self.entries[key.decode("utf-8")] = _parse_entry(value)

    def get_variable_to_shape_map(self):
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def get_tensor(self, name, verify=None):
        """verify: check the tensor's CRC32C (default: only for tensors up to 4 MiB)."""
        e = self.entries[name]
        if e["sliced"]:
            raise ValueError("{}: partitioned (sliced) variables are not supported".format(name))
        if e["dtype"] not in _DTYPES:
            raise ValueError("{}: unsupported dtype enum {}".format(name, e["dtype"]))
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("{}: {} bytes for shape {} of {}".format(name, e["size"], e["shape"], dt))
        with open(_data_file(self.prefix, e["shard_id"], self.num_shards), "rb") as f:
            f.seek(e["offset"])
            raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("{}: data file truncated".format(name))
        if verify is None:
            verify = e["size"] <= (4 << 20)
        if verify and e["crc32c"] is not None and masked_crc(raw) != e["crc32c"]:
            raise ValueError("{}: tensor checksum mismatch".format(name))
        return np.frombuffer(raw, dtype=dt.newbyteorder("<")).astype(dt).reshape(e["shape"])


def read_checkpoint(prefix, verify=None):
    r = CheckpointReader(prefix)
    return {name: r.get_tensor(name, verify) for name in r.entries}


def write_checkpoint(prefix, tensors):
    """One-shard bundle of {name: ndarray}; returns the prefix."""
    prefix = str(prefix)
    items = []
    offset = 0
    with open(_data_file(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])                     # (ascontiguousarray would turn scalars into [1])
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("{}: dtype {} has no checkpoint encoding here".format(name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            shape = b"".join(_pb_bytes(2, _pb_varint(1, int(s))) for s in a.shape)
            entry = _pb_varint(1, _DTYPE_IDS[a.dtype]) + _pb_bytes(2, shape)
            if offset:
                entry += _pb_varint(4, offset)
            entry += _pb_varint(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack("<I", masked_crc(raw))
            items.append((name.encode("utf-8"), entry))
            offset += len(raw)
    header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))       # num_shards 1, little endian (default), version 1
    write_table(prefix + ".index", [(b"", header)] + items)
    return prefix


# ---------------------------------------------------------------------------------------------------- status files
def get_checkpoint_state(checkpoint_dir, latest_filename=None):
    """tf.train.get_checkpoint_state: the `model_checkpoint_path` of <dir>/<latest_filename or "checkpoint"> (TensorFlow's
    text proto, or this package's JSON status file), made absolute against the directory; None if there is none."""
    import json
    status = os.path.join(str(checkpoint_dir), latest_filename or "checkpoint")
    if not os.path.exists(status):
        return None
    with open(status) as f:
        text = f.read()
    path = None
    try:
        path = json.loads(text).get("model_checkpoint_path")
    except ValueError:
        m = re.search(r'^\s*model_checkpoint_path:\s*"(.*)"\s*$', text, re.M)
        if m:
            path = m.group(1)
    if not path:
        return None
    return path if os.path.isabs(path) else os.path.join(str(checkpoint_dir), path)


def checkpoint_exists(prefix):
    """tf.train.checkpoint_exists: a V2 bundle prefix, or a file of this package."""
    return os.path.exists(str(prefix) + ".index") or os.path.isfile(str(prefix))
