"""fbin / ibin dataset files of the reference's benchmark crate (crates/benchmark/src/data/fbin.rs:30-45,69-148):
u32 count, u32 dim (little endian) followed by count*dim f32 (fbin) or i32 (ibin), row major.
Ids are row indices 0..count (fbin.rs:86); only the first `limit` neighbours of an ibin row are used."""
from __future__ import annotations

import os

import numpy as np


def write_fbin(path: str, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a, dtype="<f4")
    with open(path, "wb") as f:
        np.array(a.shape, dtype="<u4").tofile(f)
        a.tofile(f)


def write_ibin(path: str, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a, dtype="<i4")
    with open(path, "wb") as f:
        np.array(a.shape, dtype="<u4").tofile(f)
        a.tofile(f)


def _read(path: str, dtype: str) -> np.ndarray:
    with open(path, "rb") as f:
        count, dim = np.fromfile(f, dtype="<u4", count=2)
        a = np.fromfile(f, dtype=dtype, count=int(count) * int(dim))
    if a.size != int(count) * int(dim):
        raise ValueError(f"short payload in {path}")
    return a.reshape(int(count), int(dim))


def read_fbin(path: str) -> np.ndarray:
    return _read(path, "<f4")


def read_ibin(path: str) -> np.ndarray:
    return _read(path, "<i4")


def dataset_files(data_dir: str) -> dict:
    """[fbin] table of dataset.toml (fbin.rs:23-28); defaults when the file is absent."""
    cfg = {"data_fbin": "data.fbin", "query_fbin": "query.fbin", "query_ibin": "query.ibin"}
    p = os.path.join(data_dir, "dataset.toml")
    if os.path.exists(p):
        table = None
        for line in open(p):
            line = line.split("#")[0].strip()
            if line.startswith("["):
                table = line.strip("[] ")
            elif "=" in line and table == "fbin":
                k, v = (x.strip() for x in line.split("=", 1))
                if k in cfg:
                    cfg[k] = v.strip('"')
    return {k: os.path.join(data_dir, v) for k, v in cfg.items()}
