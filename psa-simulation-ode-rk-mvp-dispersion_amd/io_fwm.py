"""Result files, format-compatible with the reference's io_fwm.py.

* single run:  ``<name>.npz`` with keys ``z``, ``A``, ``metadata_json`` (io_fwm.py:73-134 / load :137-170),
  ``<name>.csv`` with columns z, P_<wave> x4, phi_<wave> x4 (:219-294), ``<name>.json`` metadata (:177-212),
  and the three together as a bundle (:297-328) -- files written here load with the reference's loaders and vice versa;
* sweeps (no reference counterpart: upstream never stores a sweep): ``save_sweep_npz`` / ``load_sweep_npz`` keep the
  per-point summary the HIP kernel produces (dbeta, A_end, |A3|^2 end/max, first_bad_step) plus optional axes and gains.

Nothing here touches the GPU; metadata is JSON (dataclasses, NumPy scalars/arrays and Paths are converted).
"""
from __future__ import annotations

import csv
import dataclasses
import datetime
import json
from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Tuple, Union

import numpy as np

from .sweep import SweepResult

PathLike = Union[str, Path]
WAVE_LABELS = ("pump 1", "pump 2", "signal", "idler")


def _target(path: PathLike, suffix: str, overwrite: bool) -> Path:
    p = Path(path).expanduser()
    if p.suffix.lower() != suffix:
        p = p.with_suffix(suffix)
    if p.exists() and not overwrite:
        raise FileExistsError(f"File already exists: {p}")
    p.parent.mkdir(parents=True, exist_ok=True)
    return p


def _jsonable(obj: Any) -> Any:
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return dataclasses.asdict(obj)
    if isinstance(obj, Path):
        return str(obj)
    if isinstance(obj, (np.integer, np.floating, np.bool_)):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, complex):
        return [obj.real, obj.imag]
    raise TypeError(f"Object of type {type(obj).__name__} is not JSON serializable")


def _stamped(metadata: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    md = dict(metadata or {})
    md.setdefault("timestamp_utc",
                  datetime.datetime.now(datetime.timezone.utc).replace(microsecond=0, tzinfo=None).isoformat() + "Z")
    return md


def _check_run(z, A, *, four_waves: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    z = np.asarray(z, dtype=float)
    A = np.asarray(A)
    if z.ndim != 1:
        raise ValueError("z must be a 1D array")
    if A.ndim != 2 or (four_waves and A.shape[1] != 4):
        raise ValueError("A must have shape (N, 4) for this summary function" if four_waves else "A must be a 2D array")
    if A.shape[0] != z.shape[0]:
        raise ValueError("A.shape[0] must match z.shape[0]")
    return z, A


# ---- single run ----------------------------------------------------------------------------------------------
def save_result_npz(path: PathLike, z, A, *, metadata: Optional[Mapping[str, Any]] = None, overwrite: bool = False) -> Path:
    z, A = _check_run(z, A)
    p = _target(path, ".npz", overwrite)
    np.savez_compressed(p, z=z, A=A, metadata_json=np.array(json.dumps(_stamped(metadata), ensure_ascii=False,
                                                                       default=_jsonable)))
    return p


def load_result_npz(path: PathLike) -> Tuple[np.ndarray, np.ndarray, Dict[str, Any]]:
    p = Path(path).expanduser()
    if not p.exists():
        raise FileNotFoundError(f"No such file: {p}")
    with np.load(p, allow_pickle=False) as data:
        if "z" not in data or "A" not in data:
            raise ValueError("NPZ file does not contain required keys: 'z' and 'A'")
        z, A = np.array(data["z"], dtype=float), np.array(data["A"])
        meta: Dict[str, Any] = {}
        if "metadata_json" in data:
            try:
                meta = json.loads(str(data["metadata_json"])) or {}
            except Exception:
                meta = {}
    return z, A, meta


def save_metadata_json(path: PathLike, metadata: Mapping[str, Any], *, overwrite: bool = False) -> Path:
    p = _target(path, ".json", overwrite)
    p.write_text(json.dumps(_stamped(metadata), ensure_ascii=False, indent=2, default=_jsonable), encoding="utf-8")
    return p


def load_metadata_json(path: PathLike) -> Dict[str, Any]:
    p = Path(path).expanduser()
    if not p.exists():
        raise FileNotFoundError(f"No such file: {p}")
    return json.loads(p.read_text(encoding="utf-8"))


def save_summary_csv(path: PathLike, z, A, *, wave_labels: Tuple[str, str, str, str] = WAVE_LABELS,
                     overwrite: bool = False) -> Path:
    z, A = _check_run(z, A, four_waves=True)
    if len(wave_labels) != 4:
        raise ValueError("wave_labels must have length 4")
    p = _target(path, ".csv", overwrite)
    table = np.column_stack([z, np.abs(A) ** 2, np.angle(A)])
    with p.open("w", encoding="utf-8", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["z"] + [f"P_{s}" for s in wave_labels] + [f"phi_{s}" for s in wave_labels])
        w.writerows([[float(v) for v in row] for row in table])
    return p


def save_run_bundle(output_dir: PathLike, run_name: str, z, A, *, metadata: Optional[Mapping[str, Any]] = None,
                    overwrite: bool = False) -> Dict[str, Path]:
    out = Path(output_dir).expanduser()
    md = _stamped(metadata)
    return {"npz": save_result_npz(out / f"{run_name}.npz", z, A, metadata=md, overwrite=overwrite),
            "csv": save_summary_csv(out / f"{run_name}.csv", z, A, overwrite=overwrite),
            "json": save_metadata_json(out / f"{run_name}.json", md, overwrite=overwrite)}


# ---- sweeps --------------------------------------------------------------------------------------------------------
def save_sweep_npz(path: PathLike, result: SweepResult, *, dbeta=None, x=None, gain=None,
                   metadata: Optional[Mapping[str, Any]] = None, overwrite: bool = False) -> Path:
    """Per-point summary of a sweep: a_end (N, n_waves), p_end, p_max, first_bad_step (+ dbeta, x axis, gain)."""
    n = result.a_end.shape[0]
    extra = {}
    for key, val in (("dbeta", dbeta), ("x", x), ("gain", gain)):
        if val is not None:
            arr = np.asarray(val, dtype=float)
            if arr.size != n:
                raise ValueError(f"{key} must have one entry per sweep point ({n}), got {arr.size}")
            extra[key] = arr
    md = _stamped(metadata)
    md.update(n_points=int(n), n_waves=int(result.a_end.shape[1]), n_steps=int(result.n_steps),
              save_every=int(result.save_every), kernel_ms=float(result.elapsed_ms))
    p = _target(path, ".npz", overwrite)
    np.savez_compressed(p, a_end=result.a_end, p_end=result.p_end, p_max=result.p_max,
                        first_bad_step=result.first_bad_step,
                        metadata_json=np.array(json.dumps(md, ensure_ascii=False, default=_jsonable)), **extra)
    return p


def load_sweep_npz(path: PathLike) -> Tuple[SweepResult, Dict[str, np.ndarray], Dict[str, Any]]:
    """-> (SweepResult, {"dbeta"/"x"/"gain": arrays that were stored}, metadata)."""
    p = Path(path).expanduser()
    if not p.exists():
        raise FileNotFoundError(f"No such file: {p}")
    with np.load(p, allow_pickle=False) as data:
        need = ("a_end", "p_end", "p_max", "first_bad_step")
        if any(k not in data for k in need):
            raise ValueError(f"NPZ file is not a sweep summary (needs {need})")
        meta = json.loads(str(data["metadata_json"])) if "metadata_json" in data else {}
        res = SweepResult(np.array(data["a_end"]), np.array(data["p_end"]), np.array(data["p_max"]),
                          np.array(data["first_bad_step"]), int(meta.get("n_steps", 0)), int(meta.get("save_every", 0)),
                          float(meta.get("kernel_ms", 0.0)))
        extra = {k: np.array(data[k]) for k in ("dbeta", "x", "gain") if k in data}
    return res, extra, meta
