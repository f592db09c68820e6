"""GPU: the loader of the resident slice store -- zlib inflate on host threads, the five PNG row filters undone on the device
(unetk_png_unfilter), pixels written straight into the store (data/lits.SliceStore) -- bit-exact against the host decoder
`png_decode` on files that use every filter type the way libpng does (the reference's slices were written by SimpleITK /
libpng with adaptive filters, DataLoader/Liver/extract.py:176-187, and are decoded by cv2, input_pipeline.py:243-284)."""
import json

import numpy as np

from oracle import lits_ops
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ct_like(rng, h, w):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    body = ((yy - h / 2) / (0.42 * h)) ** 2 + ((xx - w / 2) / (0.46 * w)) ** 2 <= 1
    hu = rng.normal(40, 25, size=(h, w)) * body + (-200) * (~body) + 60 * np.sin(yy / 9.0) * body
    return ((np.clip(hu, -200, 250) + 200) * 64).astype(np.uint16)


@pytest.mark.parametrize("h,w,depth", [(512, 512, 16), (512, 512, 8), (70, 37, 16), (64, 130, 8), (1, 5, 16), (129, 3, 8)])
def test_png_unfilter_all_five_filter_types_bit_exact(h, w, depth):
    from boxsegliver_amd import ops
    from boxsegliver_amd.data import lits
    rng = np.random.default_rng(h * 1000 + w + depth)
    plans = [(0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4), (4, 3, 4, 1, 2, 2, 3), tuple(int(v) for v in rng.integers(0, 5, 97))]
    imgs = []
    for _ in plans:
        a = _ct_like(rng, h, w)
        imgs.append(a if depth == 16 else (a >> 8).astype(np.uint8))
    files = [lits.png_encode(a, f) for a, f in zip(imgs, plans)]
    rows = [lits.png_inflate(b) for b in files]
    assert all((r[0], r[1], r[2]) == (w, h, depth) for r in rows)
    assert sorted(set(int(v) for v in rows[5][3].reshape(h, -1)[:, 0])) == sorted(set([0, 1, 2, 3, 4][:h]))
    filt = torch.from_numpy(np.stack([r[3] for r in rows])).cuda()
    out = torch.empty((len(files), h, w), dtype=torch.int16 if depth == 16 else torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.png_unfilter(filt, h, w, depth, out, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    got = out.cpu().numpy().view(np.uint16 if depth == 16 else np.uint8)
    for k, (b, a) in enumerate(zip(files, imgs)):
        np.testing.assert_array_equal(got[k], a)
        if h * w <= 70 * 37:                                  # the host decoder's per-byte loop: small images only
            np.testing.assert_array_equal(lits_ops.png_decode(b), a)
    # a corrupt filter byte is reported, not decoded as something
    bad = filt.clone()
    bad[0, 0] = 9
    ops.png_unfilter(bad, h, w, depth, out, status)
    assert int(status.item()) == 1


def test_png_written_by_a_third_party_encoder_decodes_on_the_device():
    """tests/golden/png/*.png come from Pillow (zlib + adaptive row filters, one file with its stream cut into several IDAT
    chunks; tests/golden/make_png_fixtures.py): png_inflate + unetk_png_unfilter give the committed pixels bit for bit."""
    import glob
    import os
    from boxsegliver_amd import ops
    from boxsegliver_amd.data import lits
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
    px = np.load(os.path.join(here, "pixels.npz"))
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(here, "*.png")))
    assert len(names) >= 5 and sorted(px.files) == names
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for name in names:
        want = px[name]
        w, h, depth, raw = lits.png_inflate(open(os.path.join(here, name + ".png"), "rb").read())
        assert (h, w, depth) == (want.shape[0], want.shape[1], 8 * want.dtype.itemsize)
        out = torch.empty((1, h, w), dtype=torch.int16 if depth == 16 else torch.uint8, device="cuda")
        ops.png_unfilter(torch.from_numpy(raw.copy())[None].cuda(), h, w, depth, out, status)
        np.testing.assert_array_equal(out.cpu().numpy().view(want.dtype)[0], want)
    assert int(status.item()) == 0


def _write_dataset(root, n_cases, depth, size, rng, filters):
    from boxsegliver_amd.data import lits
    meta, truth = [], {}
    for pid in range(n_cases):
        d = root / "png" / "volume-{:d}".format(pid)
        d.mkdir(parents=True)
        for z in range(depth):
            im = _ct_like(rng, size, size)
            lab = (np.abs(im.astype(np.int32) - 16000) < 900).astype(np.uint8) * (1 + (z % 2))
            f = tuple(int(v) for v in rng.integers(0, 5, 61)) if filters == "mixed" else filters
            (d / "{:03d}_im.png".format(z)).write_bytes(lits.png_encode(im, f))
            (d / "{:03d}_lb.png".format(z)).write_bytes(lits.png_encode((lab * 64).astype(np.uint8), f))
            truth[(pid, z)] = (im, lab * 64)
        meta.append({"PID": pid, "size": [depth, size, size], "spacing": [2.5, 0.8, 0.8], "bbox": [0, 4, 4, depth, size - 4, size - 4],
                     "tumors": [], "tumor_areas": [], "tumor_centers": [], "tumor_stddevs": [], "tumor_slices_from_to": [0],
                     "tumor_slices": [], "tumor_slices_index": [], "tumor_slices_centers": [], "tumor_slices_stddevs": [],
                     "tumor_slices_areas": [], "tumor_slices_tid": []})
    (root / "meta.json").write_text(json.dumps(meta))
    return meta, truth


def test_slice_store_streams_chunks_into_the_resident_store(tmp_path):
    """Several chunks through both staging buffers (chunk 5 over 21 slices), mixed filter types: every slot of the store
    holds exactly its file's pixels; staging memory is the two buffers, not the dataset."""
    from boxsegliver_amd.data import lits
    rng = np.random.default_rng(3)
    meta, truth = _write_dataset(tmp_path, 3, 7, 96, rng, "mixed")
    store = lits.SliceStore(tmp_path, meta, torch.device("cuda", 0), chunk=5, threads=4)
    assert tuple(store.im.shape) == (21, 96, 96) and store.im.dtype == torch.int16 and store.lb.dtype == torch.uint8
    im = store.im.cpu().numpy().view(np.uint16)
    lb = store.lb.cpu().numpy()
    for (pid, z), (a, l) in truth.items():
        np.testing.assert_array_equal(im[store.offset[pid] + z], a)
        np.testing.assert_array_equal(lb[store.offset[pid] + z], l)
    st = store.load_stats
    assert st["slices"] == 21 and st["decoded_here"] == 21 and st["staging_bytes"] == 2 * 5 * (96 * 193 + 96 * 97)
    # a slice of another size is refused (the store is one [n, h, w] tensor)
    meta[1]["size"] = [7, 64, 64]
    with pytest.raises(ValueError):
        lits.SliceStore(tmp_path, meta, torch.device("cuda", 0))


def _rank_main(rank, world, port, root, out):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from boxsegliver_amd.data import lits
    from boxsegliver_amd.utils.distribution_utils import DistributionStrategy
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    meta = json.load(open(os.path.join(root, "meta.json")))
    store = lits.SliceStore(root, meta, torch.device("cuda", 0), strategy=DistributionStrategy("mirrored", world, rank), chunk=4)
    torch.save({"im": store.im.cpu(), "lb": store.lb.cpu(), "stats": store.load_stats}, os.path.join(out, "r{}.pt".format(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_decode_half_each_and_exchange(tmp_path):
    """Data parallelism: rank r inflates / un-filters only its share of the slices and the shares are broadcast over the
    process group (gloo here: two ranks on the one GPU; RCCL on a node) -- both ranks end with the same full store."""
    import socket
    import torch.multiprocessing as mp
    from boxsegliver_amd.data import lits
    rng = np.random.default_rng(8)
    root = tmp_path / "data"
    root.mkdir()
    meta, truth = _write_dataset(root, 3, 5, 64, rng, "mixed")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "out"
    out.mkdir()
    mp.start_processes(_rank_main, args=(2, port, str(root), str(out)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(str(out / "r0.pt")), torch.load(str(out / "r1.pt"))
    assert r0["stats"]["decoded_here"] == 7 and r1["stats"]["decoded_here"] == 8 and r0["stats"]["slices"] == 15
    assert torch.equal(r0["im"], r1["im"]) and torch.equal(r0["lb"], r1["lb"])
    im = r0["im"].numpy().view(np.uint16)
    k = 0
    for pid in range(3):
        for z in range(5):
            np.testing.assert_array_equal(im[k], truth[(pid, z)][0])
            k += 1
