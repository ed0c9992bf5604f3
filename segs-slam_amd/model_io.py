"""On-disk interchange of the Scaffold model with the reference's tools (second half).

Writers for the two artefacts GaussianModel saves (src/gaussian_model.cpp):
  * savePly :1179-1261 -- binary little-endian PLY, one `vertex` element with float32 properties in this order:
        x y z | nx ny nz (zeros) | anchor_feat_0.. | offset_0.. (the (A, n_offsets, 3) tensor TRANSPOSED to (A, 3, n_offsets)
        and flattened) | opacity | scale_0..5 | rot_0..3
    -- the header tinyply writes for that call sequence (third_party/tinyply/tinyply.h write_header);
  * save_mlp_checkpoints :1262-1317 via saveTensorToTxt :262-286 -- one text file per Linear weight / bias, rows separated
    by newlines, values by single spaces, `std::fixed` with 5 decimals: {opacity,cov,color,feat}_{weight,bias}{1,2}.txt.
    (The reference also writes embedding_weight.txt, the nn::Embedding that its forward never uses, and does NOT write
    mlp_apperance; the latter is written here as appearance_{weight,bias}1.txt so that nothing trainable is lost.)
Host-side code: tensors are copied to the CPU first.  load_ply reads back what save_ply wrote (round-trip check).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np


def ply_property_names(feat_dim: int, n_offsets: int):
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"anchor_feat_{i}" for i in range(feat_dim)]
            + [f"offset_{i}" for i in range(3 * n_offsets)] + ["opacity"] + [f"scale_{i}" for i in range(6)]
            + [f"rot_{i}" for i in range(4)])


def save_ply(model, path: str):
    A, fd, no = model.A, model.dims.feat_dim, model.dims.n_offsets
    f = lambda t: t.detach().cpu().numpy().astype(np.float32)  # noqa: E731
    anchor = f(model.param("anchor"))
    cols = [anchor, np.zeros_like(anchor), f(model.param("anchor_feat")),
            f(model.param("offset").transpose(1, 2)).reshape(A, 3 * no), f(model.opacity[:A]), f(model.param("scaling")),
            f(model.rotation[:A])]
    rows = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = ply_property_names(fd, no)
    assert rows.shape == (A, len(names))
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {A}\n" + "".join(f"property float {n}\n" for n in names) \
        + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(rows.tobytes())


def load_ply(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as fh:
        names, n = [], 0
        while True:
            line = fh.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property float"):
                names.append(line.split()[-1])
            elif line == "end_header":
                break
        data = np.frombuffer(fh.read(), dtype="<f4").reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    pick = lambda prefix: data[:, [i for k, i in col.items() if k.startswith(prefix)]]  # noqa: E731
    no3 = pick("offset_").shape[1]
    return {"anchor": data[:, [col["x"], col["y"], col["z"]]], "anchor_feat": pick("anchor_feat_"),
            "offset": pick("offset_").reshape(n, 3, no3 // 3).transpose(0, 2, 1), "opacity": data[:, [col["opacity"]]],
            "scaling": pick("scale_"), "rotation": pick("rot_"), "names": names}


def _save_txt(t, path: str):
    a = t.detach().cpu().numpy().astype(np.float32)
    if a.ndim == 1:
        a = a[None, :]          # saveTensorToTxt indexes sizes[0] x sizes[1]; a bias is written as one row
    with open(path, "w") as fh:
        for row in a:
            fh.write(" ".join(f"{float(v):.5f}" for v in row) + "\n")


def save_mlp_checkpoints(model, result_dir: str):
    os.makedirs(result_dir, exist_ok=True)
    stems = {"mlp_opacity": "opacity", "mlp_cov": "cov", "mlp_color": "color", "mlp_feature_bank": "feat",
             "mlp_apperance": "appearance"}
    for name in model.dims.mlp_tensor_names():
        m, layer, kind = name.split(".")
        idx = {"0": 1, "2": 2}[layer]
        _save_txt(model.param(name), os.path.join(result_dir, f"{stems[m]}_{kind}{idx}.txt"))


_STEMS = {"mlp_opacity": "opacity", "mlp_cov": "cov", "mlp_color": "color", "mlp_feature_bank": "feat", "mlp_apperance": "appearance"}


def load_mlp_checkpoints(result_dir: str, dims) -> Dict[str, np.ndarray]:
    """Reads back the text files of save_mlp_checkpoints (5 decimals, as the reference writes them) into arrays of the shapes
    ModelDims.mlp_tensor_shape gives.  The reference has no reader for these files; this is the counterpart its own
    `save_mlp_checkpoints` output needs to be usable as an initial state here."""
    out = {}
    for name in dims.mlp_tensor_names():
        m, layer, kind = name.split(".")
        path = os.path.join(result_dir, f"{_STEMS[m]}_{kind}{ {'0': 1, '2': 2}[layer] }.txt")
        a = np.loadtxt(path, dtype=np.float32, ndmin=2)
        shape = dims.mlp_tensor_shape(name)
        if a.size != int(np.prod(shape)):
            raise ValueError(f"{path}: {a.shape} does not match {name} {shape}")
        out[name] = a.reshape(shape)
    return out


def load_model(ply_path: str, mlp_dir: str, dims, device, capacity=None):
    """GaussianModel::loadPly (src/gaussian_model.cpp:1054-1177) plus the MLP text checkpoints -> a ScaffoldModel on `device`:
    anchors, offsets (stored transposed in the file, :1160-1164), features, opacity, log-scales and rotations from the PLY
    the reference's savePly writes, MLP weights from load_mlp_checkpoints."""
    import torch
    from .neural_gaussians import ScaffoldModel
    d = load_ply(ply_path)
    A = d["anchor"].shape[0]
    if d["anchor_feat"].shape[1] != dims.feat_dim or d["offset"].shape[1] != dims.n_offsets:
        raise ValueError(f"{ply_path}: feat_dim {d['anchor_feat'].shape[1]} / n_offsets {d['offset'].shape[1]} do not match the model dimensions")
    model = ScaffoldModel(A, dims, device, capacity)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    mlp = {k: t(v) for k, v in load_mlp_checkpoints(mlp_dir, dims).items()}
    model.load(t(d["anchor"]), t(d["offset"]), t(d["anchor_feat"]), t(d["scaling"]), mlp)
    model.opacity[:A] = t(d["opacity"]).to(model.device)
    model.rotation[:A] = t(d["rotation"]).to(model.device)
    return model
