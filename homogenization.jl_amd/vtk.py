"""
VTK (.vtu) export of the checkerboard and of a level of the implicit grid -- the reference's `save` option
(src/examples/homogenized_coefficients.jl:69-87, 219, 303; construct_full_grid: src/implicit_fine_grid.jl:41-78).
Output side of the path only: level vectors are read back once per exported level.

Files are VTK XML UnstructuredGrid, inline base64 ("binary") arrays with a UInt64 length header, no compression;
ParaView / VTK / meshio read them.
"""
from __future__ import annotations

import base64

import numpy as np

_VTK_TYPE = {3: 5, 4: 10}     # triangle, tetrahedron
_NAMES = {np.dtype("float64"): "Float64", np.dtype("int64"): "Int64", np.dtype("uint8"): "UInt8",
          np.dtype("int32"): "Int32"}


def _array(f, name, a, ncomp=None):
    a = np.ascontiguousarray(a)
    raw = a.tobytes()
    head = np.array([len(raw)], dtype="<u8").tobytes()
    comp = f' NumberOfComponents="{ncomp}"' if ncomp else ""
    f.write(f'<DataArray type="{_NAMES[a.dtype]}" Name="{name}"{comp} format="binary">\n')
    f.write(base64.b64encode(head + raw).decode("ascii"))
    f.write("\n</DataArray>\n")


def write_vtu(path, points, cells, point_data=None, cell_data=None):
    """points (N, dim), cells (M, dim+1) 0-based; *_data: name -> (N,) / (N, c) array."""
    points = np.asarray(points, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    p3 = np.zeros((points.shape[0], 3))
    p3[:, : points.shape[1]] = points
    nv = cells.shape[1]
    if not path.endswith(".vtu"):
        path += ".vtu"
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" '
                'header_type="UInt64">\n<UnstructuredGrid>\n')
        f.write(f'<Piece NumberOfPoints="{p3.shape[0]}" NumberOfCells="{cells.shape[0]}">\n<Points>\n')
        _array(f, "Points", p3, 3)
        f.write("</Points>\n<Cells>\n")
        _array(f, "connectivity", cells.ravel())
        _array(f, "offsets", np.arange(1, cells.shape[0] + 1, dtype=np.int64) * nv)
        _array(f, "types", np.full(cells.shape[0], _VTK_TYPE[nv], dtype=np.uint8))
        f.write("</Cells>\n")
        for tag, data in (("PointData", point_data), ("CellData", cell_data)):
            if data:
                f.write(f"<{tag}>\n")
                for name, a in data.items():
                    a = np.asarray(a, dtype=np.float64)
                    _array(f, name, a, a.shape[1] if a.ndim == 2 else None)
                f.write(f"</{tag}>\n")
        f.write("</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
    return path


def read_vtu(path):
    """Minimal reader for files written by write_vtu (used by the tests)."""
    import xml.etree.ElementTree as ET
    types = {v: k for k, v in _NAMES.items()}
    root = ET.parse(path).getroot()
    out = {"point_data": {}, "cell_data": {}}

    def decode(el):
        raw = base64.b64decode(el.text.strip())
        n = int(np.frombuffer(raw[:8], dtype="<u8")[0])
        a = np.frombuffer(raw[8:8 + n], dtype=types[el.get("type")])
        c = el.get("NumberOfComponents")
        return a.reshape(-1, int(c)) if c else a

    piece = root.find("UnstructuredGrid/Piece")
    out["points"] = decode(piece.find("Points/DataArray"))
    for el in piece.find("Cells"):
        out[el.get("Name")] = decode(el)
    for tag, key in (("PointData", "point_data"), ("CellData", "cell_data")):
        node = piece.find(tag)
        if node is not None:
            for el in node:
                out[key][el.get("Name")] = decode(el)
    return out


def construct_full_grid(implicit, level: int):
    """All cells of refinement level `level` as one explicit mesh; interface nodes are duplicated per coarse cell
    (node n of coarse cell e has index e * Nf + n, n in the reference's hierarchical order).
    ref: src/implicit_fine_grid.jl:41-78"""
    base = implicit.base
    dim = base.dim
    nf = implicit.nf(level)
    m = 2 ** (level - 1)
    h2s = implicit.table_i32("hier2slot", level)
    ijk = implicit.table_i32("slot_ijk", level).reshape(-1, 3)[h2s][:, :dim].astype(np.float64) / m   # (nf, dim)
    ref_cells = implicit.table_i32("ref_cells", level).reshape(-1, dim + 1).astype(np.int64)
    ne = implicit.ncells()                                  # (a shrunk domain is a prefix of the base cells)
    X = base.nodes[base.elements[:ne] - 1]                  # (ne, dim+1, dim)
    J = X[:, 1:, :] - X[:, :1, :]                           # rows: X_a - X_0
    nodes = (X[:, :1, :] + np.einsum("na,eac->enc", ijk, J)).reshape(-1, dim)
    cells = (ref_cells[None, :, :] + (np.arange(ne, dtype=np.int64) * nf)[:, None, None]).reshape(-1, dim + 1)
    return nodes, cells


def export_domain(base, cond, name="checkerboard"):
    """ref: src/examples/homogenized_coefficients.jl:69-79 -- cell data "a" = the diagonal of the conductivity."""
    return write_vtu(name, base.nodes, base.elements - 1, cell_data={"a": np.asarray(cond, dtype=np.float64)})


def export_unknown(implicit, x, k: int, level: int, name=None, field="v"):
    """ref: src/examples/homogenized_coefficients.jl:81-87 -- point data "v" = x[1:nnodes(level), :][:] on the full
    grid of `level`.  `x` is a DeviceMatrix of the finest level or a host (Nf, Ne) array in hierarchical order."""
    nodes, cells = construct_full_grid(implicit, level)
    xh = x.to_host() if hasattr(x, "to_host") else np.asarray(x)
    nfl = implicit.nf(level)
    v = np.ascontiguousarray(xh[:nfl, :].T).ravel()
    return write_vtu(name or f"ahom_{k}", nodes, cells, point_data={field: v})
