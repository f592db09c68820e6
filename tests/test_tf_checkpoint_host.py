"""CPU: the TensorFlow V2 checkpoint ("tensor bundle") codec of utils/tf_checkpoint.py -- CRC-32C against its published
check values and a bit-serial definition, snappy against hand-assembled streams, the SSTable layer against a byte-level
hand-built file, bundle round trips, the status-file parser.  (TensorFlow is not installable here: the format is restated
from its specification, see the module docstring.)"""
import struct

import numpy as np
import pytest

from boxsegliver_amd.utils import tf_checkpoint as tfc


def _crc_bitwise(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xFFFFFFFF


def test_crc32c_check_values_and_lane_parallel_path():
    assert tfc.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert tfc.crc32c(b"\x00" * 32) == 0x8A9136AA and tfc.crc32c(b"\xff" * 32) == 0x62A8AB43     # RFC 3720 B.4
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfc.crc32c(b"") == 0
    rng = np.random.default_rng(0)
    for n in (4095, 4096, 4097, 10000, 65537, 300001):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tfc.crc32c(data) == _crc_bitwise(data), n
    # leveldb's mask: rotate right by 15, add the constant
    c = tfc.crc32c(b"foo")
    assert tfc.masked_crc(b"foo") == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_snappy_uncompress_literals_and_copies():
    # literal "abcd", 1-byte-offset copy (offset 4, len 11: 3 length bits) overlapping its own output, literal "XYZ"
    stream = bytes([18]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((11 - 4) << 2) | 1, 4]) + bytes([(3 - 1) << 2]) + b"XYZ"
    assert tfc.snappy_uncompress(stream) == b"abcdabcdabcdabcXYZ"
    # 2-byte-offset copy and a long literal (length byte form)
    lit = bytes(range(70))
    stream = tfc._put_varint(70 + 10) + bytes([60 << 2, 69]) + lit + bytes([((10 - 1) << 2) | 2, 70, 0])
    assert tfc.snappy_uncompress(stream) == lit + lit[:10]
    with pytest.raises(ValueError):
        tfc.snappy_uncompress(bytes([5, (4 - 1) << 2]) + b"abcd")     # announced 5 bytes, holds 4


def test_table_layout_bytes_and_reader(tmp_path):
    items = [(b"", b"H"), (b"UNet/a/weights", b"1"), (b"UNet/a/weights/Adam", b"22"), (b"UNet/b", b"333")]
    path = tmp_path / "t.index"
    tfc.write_table(path, items)
    raw = path.read_bytes()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) >= 48
    # first data block, by hand: entry = shared, non_shared, value_len, key delta, value; the 3rd key shares 14 bytes
    assert raw[:4] == bytes([0, 0, 1]) + b"H"
    assert raw[4:4 + 3 + 14 + 1] == bytes([0, 14, 1]) + b"UNet/a/weights" + b"1"
    assert raw[22:22 + 3 + 5 + 2] == bytes([14, 5, 2]) + b"/Adam" + b"22"
    assert tfc.read_table(path) == items
    # many blocks + a snappy-compressed block written by hand are read back
    big = [(("v%05d" % i).encode(), bytes([i % 251]) * (i % 40)) for i in range(3000)]
    tfc.write_table(path, big, block_size=512)
    assert tfc.read_table(path) == big
    bad = bytearray(path.read_bytes())
    bad[10] ^= 0xFF
    path.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        tfc.read_table(path)
    (tmp_path / "junk.index").write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError):
        tfc.read_table(tmp_path / "junk.index")


def test_snappy_block_in_table(tmp_path):
    block = tfc._build_block([(b"", b"hdr"), (b"k", b"v" * 8)])
    comp = tfc._put_varint(len(block)) + bytes([60 << 2, len(block) - 1]) + block          # one long literal
    out = bytearray(comp) + bytes([1])
    out += struct.pack("<I", tfc.masked_crc(bytes(comp) + b"\x01"))
    handle = tfc._put_varint(0) + tfc._put_varint(len(comp))
    meta_off = len(out)
    mb = tfc._build_block([])
    out += mb + b"\x00" + struct.pack("<I", tfc.masked_crc(mb + b"\x00"))
    idx_off = len(out)
    ib = tfc._build_block([(b"k", handle)], 1)
    out += ib + b"\x00" + struct.pack("<I", tfc.masked_crc(ib + b"\x00"))
    footer = tfc._put_varint(meta_off) + tfc._put_varint(len(mb)) + tfc._put_varint(idx_off) + tfc._put_varint(len(ib))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", tfc.MAGIC)
    (tmp_path / "s.index").write_bytes(bytes(out))
    assert tfc.read_table(tmp_path / "s.index") == [(b"", b"hdr"), (b"k", b"v" * 8)]


def test_bundle_roundtrip_and_reader_surface(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"UNet/Encode1/Repeat/convolution2d_1/weights": rng.standard_normal((3, 3, 3, 64)).astype(np.float32),
               "UNet/Encode1/Repeat/convolution2d_1/BatchNorm/gamma": rng.random(64).astype(np.float32),
               "Optimizer/UNet/Encode1/Repeat/convolution2d_1/weights/Adam": np.zeros((3, 3, 3, 64), np.float32),
               "Optimizer/beta1_power": np.float32(0.729), "global_step": np.int64(5000),
               "big/w": rng.standard_normal((256, 1024)).astype(np.float32)}
    prefix = tfc.write_checkpoint(tmp_path / "model.ckpt-5000", tensors)
    assert (tmp_path / "model.ckpt-5000.index").exists() and (tmp_path / "model.ckpt-5000.data-00000-of-00001").exists()
    r = tfc.CheckpointReader(prefix)
    shapes = r.get_variable_to_shape_map()
    assert shapes["global_step"] == [] and shapes["big/w"] == [256, 1024] and r.has_tensor("Optimizer/beta1_power")
    back = tfc.read_checkpoint(prefix, verify=True)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == np.asarray(v).dtype
        np.testing.assert_array_equal(back[k], v)
    assert int(back["global_step"]) == 5000
    # the entry proto, by hand: dtype DT_FLOAT, shape [64], size 256, masked crc
    raw_entry = dict(tfc.read_table(prefix + ".index"))[b"UNet/Encode1/Repeat/convolution2d_1/BatchNorm/gamma"]
    fields = {n: v for n, _, v in tfc._pb_fields(raw_entry)}
    g = tensors["UNet/Encode1/Repeat/convolution2d_1/BatchNorm/gamma"]
    assert fields[1] == 1 and fields[5] == 256 and fields[6] == tfc.masked_crc(g.tobytes())
    assert fields[2] == bytes([0x12, 2, 0x08, 64])
    # corrupt one byte of the data file: the tensor's checksum catches it
    data = bytearray((tmp_path / "model.ckpt-5000.data-00000-of-00001").read_bytes())
    data[r.entries["big/w"]["offset"] + 5] ^= 1
    (tmp_path / "model.ckpt-5000.data-00000-of-00001").write_bytes(bytes(data))
    with pytest.raises(ValueError):
        tfc.CheckpointReader(prefix).get_tensor("big/w", verify=True)
    with pytest.raises(FileNotFoundError):
        tfc.CheckpointReader(tmp_path / "nope")


def test_checkpoint_state_files(tmp_path):
    (tmp_path / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-5000"\nall_model_checkpoint_paths: "model.ckpt-5000"\n')
    assert tfc.get_checkpoint_state(tmp_path) == str(tmp_path / "model.ckpt-5000")
    (tmp_path / "checkpoint_best").write_text('{"model_checkpoint_path": "best.pt", "global_step": 3}')
    assert tfc.get_checkpoint_state(tmp_path, "checkpoint_best") == str(tmp_path / "best.pt")
    (tmp_path / "abs").write_text('model_checkpoint_path: "/x/y/model.ckpt-1"\n')
    assert tfc.get_checkpoint_state(tmp_path, "abs") == "/x/y/model.ckpt-1"
    assert tfc.get_checkpoint_state(tmp_path, "missing") is None
    assert not tfc.checkpoint_exists(tmp_path / "model.ckpt-5000")
    tfc.write_checkpoint(tmp_path / "model.ckpt-5000", {"a": np.zeros(3, np.float32)})
    assert tfc.checkpoint_exists(tmp_path / "model.ckpt-5000")


def test_padded_store_optimiser_slots_round_trip_in_tf_shapes(tmp_path):
    """channel-padded stores (UNet3D, SmallUNet_V2) used to drop the Adam slots on both TF import and
    export; they now travel in the logical (TF) shapes and scatter back into the padded layout, padding exactly zero."""
    import argparse
    import types

    import torch
    from boxsegliver_amd.NetworksV2.padded import PaddedParamStore
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.utils import tf_checkpoint as tfc
    specs = [("N/a/weights", (3, 3, 3, 30), "conv_w"), ("N/a/beta", (30,), "beta"), ("N/b/weights", (3, 3, 60, 30), "conv_w"),
             ("N/a/moving_mean", (30,), "moving_mean")]
    pads = {"N/a/weights": ((3, 3, 3, 32), {}), "N/a/beta": ((32,), {}), "N/a/moving_mean": ((32,), {}),
            "N/b/weights": ((3, 3, 64, 32), {2: [(0, 30, 0), (30, 30, 32)]})}

    def make():
        store = PaddedParamStore(specs, pads, torch.device("cpu"))
        store.initialize("xavier", seed=5)
        a = argparse.Namespace(learning_rate=1e-3, learning_policy="plateau", lr_decay_step=1000, lr_decay_rate=0.5,
                               num_of_total_steps=10, lr_power=0.9, lr_end=1e-6, lr_decay_boundaries=None,
                               lr_custom_values=None, optimizer="Adam")
        return types.SimpleNamespace(params=store, name="N"), Solver(a)

    model, solver = make()
    state = solver._ensure_state(model.params)
    g = torch.Generator().manual_seed(1)
    for grp in ("reg", "noreg"):
        for k in range(2):
            for name in model.params.trainable_names():
                if model.params.where[name][0] == grp:
                    model.params.write_slot(state[grp][k], name, torch.randn(model.params.logical_shape[name], generator=g))
    solver.global_step, solver.plateau_lr = 7, 2.5e-4
    prefix = est.save_tf_checkpoint(tmp_path / "m.ckpt-7", model, solver)
    shapes = tfc.CheckpointReader(prefix).get_variable_to_shape_map()
    assert shapes["Optimizer/N/b/weights/Adam_1"] == [3, 3, 60, 30] and shapes["Optimizer/N/a/beta/Adam"] == [30]
    model2, solver2 = make()
    est.restore_variables(prefix, model2, solver2)
    assert solver2.global_step == 7 and solver2.plateau_lr == pytest.approx(2.5e-4)
    for grp in ("reg", "noreg"):
        for k in range(2):
            assert torch.equal(solver2._state[grp][k], state[grp][k])          # incl. the zero padding
    _, off, n, shp, _ = model2.params.where["N/b/weights"]
    phys = solver2._state["reg"][0][off:off + n].view(shp)
    assert float(phys[:, :, 30:32].abs().max()) == 0 and float(phys[:, :, 62:].abs().max()) == 0 and float(phys[..., 30:].abs().max()) == 0
