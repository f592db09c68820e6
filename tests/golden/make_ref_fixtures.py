"""Fixtures pinned on the REFERENCE ITSELF (build container only): the four reference modules that import without
TensorFlow -- config.py, utils/surface.py, DataLoader/misc.py, evaluators/evaluator_base.py -- are
imported from /root/reference, run on seeded inputs, and their outputs written to tests/golden/ref_*.json|npz.
tests/test_ref_fixtures.py then compares boxsegliver_amd against those files; nothing of the reference travels to the
GPU box (the fixtures are data: inputs and expected outputs).

    python tests/golden/make_ref_fixtures.py           # needs /root/reference; rewrites the ref_* fixtures

What this pins: the CLI flag surface of config.add_arguments / check_args / fill_default_args (drop-in boundary, 8b),
the surface-distance metrics ASSD / RMSD / MSD (8f1, at non-unit voxel spacing), the k-fold split and its file format
(8f2), and the metric table writer of EvaluateBase.  What it cannot pin: the TensorFlow-held arithmetic of the hot path
(TF 1.13 is not installable here) -- that part of the oracle stays "parity unpinned" (DESIGN.md 2)."""
import argparse
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _action_row(a):
    return {"flags": list(a.option_strings), "dest": a.dest, "default": a.default if not callable(a.default) else str(a.default),
            "choices": list(a.choices) if a.choices is not None else None, "nargs": a.nargs, "required": bool(a.required),
            "type": getattr(a.type, "__name__", None) if a.type is not None else None, "action": type(a).__name__,
            "const": a.const}


def argparse_surface(add_arguments):
    p = argparse.ArgumentParser()
    add_arguments(p)
    rows = [_action_row(a) for a in p._actions if a.dest != "help"]
    groups = [g.title for g in p._action_groups if g.title not in ("positional arguments", "optional arguments", "options")]
    return {"actions": rows, "groups": groups}


CHECK_CASES = [
    # (description, namespace fields) -- config.check_args / fill_default_args on each; outcome = attributes or the error
    ("plain train", dict(mode="train", tag="t1", model_dir="", classes=["Liver", "Tumor"], loss_weight_type="numerical",
                         loss_numeric_w=[0.2, 0.4, 4.4], primary_metric=None, secondary_metric=None, warm_start_from=None,
                         summary_prefix=None, metrics_eval=["Dice"])),
    ("numeric weights of the wrong length", dict(mode="train", tag="t2", model_dir="", classes=["Liver"],
                                                 loss_weight_type="numerical", loss_numeric_w=[1.0, 2.0, 3.0],
                                                 primary_metric=None, secondary_metric=None, warm_start_from=None,
                                                 summary_prefix=None, metrics_eval=["Dice"])),
    ("primary metric given", dict(mode="train", tag="t3", model_dir="", classes=["Liver", "Tumor"], loss_weight_type="none",
                                  loss_numeric_w=None, primary_metric="Tumor/Dice", secondary_metric="Liver/Dice",
                                  warm_start_from=None, summary_prefix="pre", metrics_eval=["Dice"])),
    ("primary metric of an unknown class", dict(mode="train", tag="t4", model_dir="", classes=["Liver"], loss_weight_type="none",
                                                loss_numeric_w=None, primary_metric="Tumor/Dice", secondary_metric=None,
                                                warm_start_from=None, summary_prefix=None, metrics_eval=["Dice"])),
    ("primary metric not evaluated", dict(mode="train", tag="t5", model_dir="", classes=["Liver"], loss_weight_type="none",
                                          loss_numeric_w=None, primary_metric="Liver/VOE", secondary_metric=None,
                                          warm_start_from=None, summary_prefix=None, metrics_eval=["Dice"])),
    ("explicit model_dir", dict(mode="eval", tag="t6", model_dir="some/dir", classes=["Liver"], loss_weight_type="none",
                                loss_numeric_w=None, primary_metric=None, secondary_metric=None, warm_start_from=None,
                                summary_prefix=None, metrics_eval=["Dice"])),
]


def run_check_cases(config):
    out = []
    for desc, fields in CHECK_CASES:
        ns = argparse.Namespace(**json.loads(json.dumps(fields)))
        p = argparse.ArgumentParser()
        res = {"desc": desc, "fields": fields}
        try:
            from contextlib import redirect_stderr
            with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()) as err:
                try:
                    config.check_args(ns, p)
                    config.fill_default_args(ns)
                except SystemExit:
                    msg = err.getvalue().strip().splitlines()[-1]
                    res["error"] = "parser.error:" + msg.split("error: ", 1)[-1]
                    out.append(res)
                    continue
            res["ok"] = {k: (str(v) if not isinstance(v, (int, float, str, list, type(None), bool)) else v)
                         for k, v in sorted(vars(ns).items())}
        except SystemExit as e:                      # parser.error()
            res["error"] = "SystemExit:{}".format(e.code)
        except Exception as e:                       # noqa: BLE001 -- the fixture records whatever the reference raises
            res["error"] = "{}:{}".format(type(e).__name__, e)
        out.append(res)
    return out


def make_masks(rng, shape):
    """Two overlapping blobby binary volumes (each one connected-ish object with holes / noise on the rim)."""
    z, y, x = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    out = []
    for _ in range(2):
        c = np.array(shape) * (0.35 + 0.3 * rng.random(3))
        r = np.array(shape) * (0.18 + 0.15 * rng.random(3))
        m = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2 < 1.0
        m ^= (rng.random(shape) < 0.02) & m                       # pepper inside
        out.append(m)
    return out


def main():
    sys.path.insert(0, REF)
    # The reference was written against SciPy < 1.0, which re-exported NumPy's `array` as `scipy.array` (removed since);
    # utils/surface.py calls it three times.  Restoring that one alias is the whole shim -- nothing reference-specific.
    import scipy
    import scipy.ndimage   # noqa: F401  (the reference relies on `import scipy` exposing the sub-packages)
    import scipy.spatial   # noqa: F401
    if not hasattr(scipy, "array"):
        scipy.array = np.array
    import config as ref_config                                    # noqa: E402  (reference modules, this container only)
    from DataLoader import misc as ref_misc                       # noqa: E402
    from evaluators import evaluator_base as ref_eb               # noqa: E402
    from utils.surface import Surface as RefSurface               # noqa: E402

    # 1. CLI surface
    cli = {"config.add_arguments": argparse_surface(ref_config.add_arguments), "check_cases": run_check_cases(ref_config),
           "CustomKeys": {k: v for k, v in vars(ref_config.CustomKeys).items() if not k.startswith("_")}}
    with open(os.path.join(HERE, "ref_config_surface.json"), "w") as f:
        json.dump(cli, f, indent=1, sort_keys=True)

    # 2. surface distances at non-unit spacing
    rng = np.random.default_rng(20260410)
    masks, refs, spacings, assd, rmsd, msd = [], [], [], [], [], []
    shapes = [(12, 20, 18), (9, 16, 24), (16, 16, 16), (7, 30, 22)]
    for i in range(24):
        shape = shapes[i % len(shapes)]
        a, b = make_masks(rng, shape)
        sp = [float(v) for v in np.round(0.5 + 2.0 * rng.random(3), 3)] if i % 6 else [1.0, 1.0, 1.0]
        s = RefSurface(a, b, physical_voxel_spacing=sp, mask_offset=[0., 0., 0.], reference_offset=[0., 0., 0.])
        pad = np.zeros((16, 30, 24), dtype=bool)
        pa, pb = pad.copy(), pad.copy()
        pa[:shape[0], :shape[1], :shape[2]] = a
        pb[:shape[0], :shape[1], :shape[2]] = b
        masks.append(np.packbits(pa)); refs.append(np.packbits(pb)); spacings.append(sp)
        assd.append(s.get_average_symmetric_surface_distance())
        rmsd.append(s.get_root_mean_square_symmetric_surface_distance())
        msd.append(s.get_maximum_symmetric_surface_distance())
    np.savez_compressed(os.path.join(HERE, "ref_surface.npz"), masks=np.stack(masks), refs=np.stack(refs),
                        shapes=np.array([shapes[i % len(shapes)] for i in range(24)]), spacings=np.array(spacings),
                        assd=np.array(assd), rmsd=np.array(rmsd), msd=np.array(msd))

    # 3. k-fold split: the shipped file, the generator (seed 1357) on the LiTS ids, and another list / k / seed
    kf = {"shipped_k_folds_txt": open(os.path.join(REF, "data/LiTS/k_folds.txt")).read()}
    with tempfile.TemporaryDirectory() as d:
        with redirect_stdout(io.StringIO()):
            kf["generated_131_k5_seed1357"] = ref_misc.read_or_create_k_folds(os.path.join(d, "a.txt"), list(range(131)), 5, 1357)
            kf["generated_131_file"] = open(os.path.join(d, "a.txt")).read()
            kf["generated_23_k4_seed7"] = ref_misc.read_or_create_k_folds(os.path.join(d, "b.txt"), list(range(100, 123)), 4, 7)
            kf["reread_131"] = ref_misc.read_or_create_k_folds(os.path.join(d, "a.txt"), [], None, None)
    kf["generated_131_k5_seed1357"] = [[int(x) for x in f] for f in kf["generated_131_k5_seed1357"]]
    kf["generated_23_k4_seed7"] = [[int(x) for x in f] for f in kf["generated_23_k4_seed7"]]
    with open(os.path.join(HERE, "ref_k_folds.json"), "w") as f:
        json.dump(kf, f, indent=1, sort_keys=True)

    # 4. EvaluateBase metric table
    ev = ref_eb.EvaluateBase()
    ev.clear_metrics()
    rows = [{"Liver/Dice": 0.95, "Tumor/Dice": 0.5, "Name": "volume-3"}, {"Liver/Dice": 0.9612345, "Tumor/Dice": 0.0, "Name": "volume-7"},
            {"Liver/Dice": 1, "Tumor/Dice": 0.25, "Name": "volume-11"}]
    rows = [sorted(r.items()) for r in rows]                       # key order is part of the output: store it explicitly
    for r in rows:
        ev.append_metrics(dict(r))
    with tempfile.TemporaryDirectory() as d:
        with redirect_stdout(io.StringIO()):
            ev.save_metrics("m.csv", d)
        table = open(os.path.join(d, "m.csv")).read()
    ev.clear_metrics()
    with open(os.path.join(HERE, "ref_save_metrics.json"), "w") as f:
        json.dump({"rows": rows, "file": table}, f, indent=1, sort_keys=True)
    write_meta_excerpt()
    print("wrote ref_config_surface.json, ref_surface.npz, ref_k_folds.json, ref_save_metrics.json, ref_meta_excerpt.json")


def write_meta_excerpt(pids=(3, 5, 32, 4)):
    """5. Four cases of the shipped DataLoader/Liver/prepare/meta.json, verbatim (data the reference holds for its input
    pipeline: volume extents, liver boxes, per-slice tumour boxes / moments): two ordinary tumour cases, one without tumours
    and the case with the most tumour slices.  `collect_datasets` / `parse_case` / `TrainSampler` must digest the real schema."""
    with open(os.path.join(REF, "DataLoader/Liver/prepare/meta.json")) as f:
        meta = json.load(f)
    pick = [c for c in meta if int(c["PID"]) in pids]
    assert len(pick) == len(pids)
    with open(os.path.join(HERE, "ref_meta_excerpt.json"), "w") as f:
        json.dump({"source": "DataLoader/Liver/prepare/meta.json (131 cases), cases " + ", ".join(str(p) for p in sorted(pids)),
                   "n_cases_in_file": len(meta), "cases": pick}, f, sort_keys=True)


if __name__ == "__main__":
    main()
