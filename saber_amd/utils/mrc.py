"""MRC2014 files (what saber/utils/io.py:51-55 reads through the `mrcfile` package, absent from this image): the 1024-byte header,
an optional extended header, then the data block.  Restated from the published MRC2014 specification (Cheng et al. 2015, CCP-EM);
parity with `mrcfile` is unpinned here (no copy of it to read the files with), the tests check the header words at their published
offsets.

Header words used (byte offset): nx, ny, nz (0, 4, 8); mode (12: 0 int8, 1 int16, 2 float32, 6 uint16, 12 float16); mx, my, mz (28..36);
cella x, y, z in Angstrom (40..48); cellb (52..60); mapc, mapr, maps (64..72); dmin, dmax, dmean (76..84); ispg (88); nsymbt (92 =
bytes of extended header); exttyp (104); nversion (108); origin (196..204); 'MAP ' (208); machine stamp (212); rms (216); nlabl (220)."""
import struct
from typing import Optional, Tuple

import numpy as np

_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}
_MODE_OF = {np.dtype(v): k for k, v in _MODES.items()}


def read_mrc(path: str, permissive: bool = True) -> Tuple[np.ndarray, Tuple[float, float, float]]:
    """-> (data, (voxel_x, voxel_y, voxel_z) in Angstrom).  data is (ny, nx) when nz == 1 (a single image, as mrcfile presents it),
    else (nz, ny, nx).  permissive: a missing 'MAP ' id or an odd machine stamp is tolerated (mrcfile.open(permissive=True))."""
    with open(path, "rb") as f:
        head = f.read(1024)
        if len(head) < 1024:
            raise ValueError(f"{path}: shorter than an MRC header")
        stamp = head[212:214]
        end = ">" if stamp[:1] == b"\x11" else "<"
        nx, ny, nz, mode = struct.unpack_from(end + "4i", head, 0)
        if not permissive and head[208:212] != b"MAP ":
            raise ValueError(f"{path}: no 'MAP ' identifier")
        if mode not in _MODES or min(nx, ny, nz) <= 0:
            raise ValueError(f"{path}: unsupported mode {mode} or bad dimensions {(nx, ny, nz)}")
        mx, my, mz = struct.unpack_from(end + "3i", head, 28)
        cx, cy, cz = struct.unpack_from(end + "3f", head, 40)
        (nsymbt,) = struct.unpack_from(end + "i", head, 92)
        f.seek(1024 + max(nsymbt, 0))
        dt = np.dtype(_MODES[mode]).newbyteorder(end)
        data = np.fromfile(f, dtype=dt, count=nx * ny * nz)
    if data.size != nx * ny * nz:
        raise ValueError(f"{path}: data block holds {data.size} of {nx * ny * nz} values")
    data = data.astype(dt.newbyteorder("=")).reshape(nz, ny, nx)
    vox = tuple(float(c) / m if m > 0 else 0.0 for c, m in ((cx, mx), (cy, my), (cz, mz)))
    return (data[0] if nz == 1 else data), vox


def write_mrc(path: str, data: np.ndarray, voxel_size: Optional[float] = 1.0) -> None:
    """(ny, nx) image or (nz, ny, nx) volume, little-endian, no extended header; cella = voxel_size x sampling."""
    data = np.asarray(data)
    if data.dtype == np.float64:
        data = data.astype(np.float32)
    if data.dtype not in _MODE_OF or data.ndim not in (2, 3):
        raise ValueError(f"write_mrc: dtype {data.dtype} / {data.ndim}-D is not an MRC mode")
    vol = data[None] if data.ndim == 2 else data
    nz, ny, nx = vol.shape
    v = float(voxel_size or 0.0)
    head = bytearray(1024)
    struct.pack_into("<4i", head, 0, nx, ny, nz, _MODE_OF[data.dtype])
    struct.pack_into("<3i", head, 28, nx, ny, nz)
    struct.pack_into("<3f", head, 40, nx * v, ny * v, nz * v)
    struct.pack_into("<3f", head, 52, 90.0, 90.0, 90.0)
    struct.pack_into("<3i", head, 64, 1, 2, 3)
    f64 = vol.astype(np.float64)
    struct.pack_into("<3f", head, 76, float(f64.min()), float(f64.max()), float(f64.mean()))
    struct.pack_into("<i", head, 88, 1 if data.ndim == 3 else 0)
    struct.pack_into("<i", head, 108, 20140)
    head[208:212] = b"MAP "
    head[212:216] = b"\x44\x44\x00\x00"
    struct.pack_into("<f", head, 216, float(f64.std()))
    with open(path, "wb") as f:
        f.write(bytes(head))
        f.write(np.ascontiguousarray(vol.astype(vol.dtype.newbyteorder("<"))).tobytes())
