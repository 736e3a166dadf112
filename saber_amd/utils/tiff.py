"""Baseline TIFF 6.0, uncompressed grey-scale pages: what `skimage.io.imread / imsave` (absent from this image) do for the files the
reference touches - micrographs read by saber/utils/io.py:56-57 and the label volume written by mask3D_to_tiff (:151-155).  Classic TIFF
only (no BigTIFF, no compression, no tiles): anything else raises, naming the tag, rather than being decoded wrongly."""
import struct

import numpy as np

_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q"}


def _ifd(buf, off, e):
    (n,) = struct.unpack_from(e + "H", buf, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from(e + "HHI4s", buf, off + 2 + 12 * i)
        fmt = _TYPES.get(typ)
        if fmt is None or typ in (2, 5):
            continue
        size = struct.calcsize(fmt) * cnt
        src = val if size <= 4 else buf[struct.unpack(e + "I", val)[0]:][:size]
        tags[tag] = struct.unpack_from(e + fmt * cnt, src, 0)
    (nxt,) = struct.unpack_from(e + "I", buf, off + 2 + 12 * n)
    return tags, nxt


def imread(path: str) -> np.ndarray:
    """-> (H, W) for one page, (pages, H, W) for several pages of one shape"""
    with open(path, "rb") as f:
        buf = f.read()
    e = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if e is None or struct.unpack_from(e + "H", buf, 2)[0] != 42:
        raise ValueError(f"{path}: not a classic TIFF file")
    (off,) = struct.unpack_from(e + "I", buf, 4)
    pages = []
    while off:
        t, off = _ifd(buf, off, e)
        if t.get(259, (1,))[0] != 1:
            raise ValueError(f"{path}: compressed TIFF (tag 259 = {t[259][0]}) is not read here")
        if 322 in t or t.get(277, (1,))[0] != 1:
            raise ValueError(f"{path}: tiled or multi-sample TIFF is not read here")
        w, h, bits, fmt = t[256][0], t[257][0], t.get(258, (1,))[0], t.get(339, (1,))[0]
        kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
        if kind is None or bits not in (8, 16, 32, 64):
            raise ValueError(f"{path}: sample format {fmt} / {bits} bits is not read here")
        dt = np.dtype(f"{e}{kind}{bits // 8}")
        data = b"".join(buf[o:o + n] for o, n in zip(t[273], t[279]))
        pages.append(np.frombuffer(data, dtype=dt, count=w * h).reshape(h, w).astype(dt.newbyteorder("=")))
    return pages[0] if len(pages) == 1 else np.stack(pages)


def imsave(path: str, arr: np.ndarray) -> None:
    """(H, W) or (pages, H, W); bool is stored as uint8; one strip per page, little-endian"""
    arr = np.asarray(arr)
    if arr.dtype == bool:
        arr = arr.astype(np.uint8)
    if arr.ndim not in (2, 3) or arr.dtype.kind not in "uif" or arr.dtype.itemsize > 8:
        raise ValueError(f"imsave: {arr.dtype} {arr.ndim}-D is not written here")
    vol = arr[None] if arr.ndim == 2 else arr
    n, h, w = vol.shape
    fmt = {"u": 1, "i": 2, "f": 3}[arr.dtype.kind]
    page_bytes = h * w * arr.dtype.itemsize
    ifd_bytes = 2 + 12 * 9 + 4
    if 8 + n * (page_bytes + ifd_bytes) >= 1 << 32:
        raise ValueError("imsave: larger than classic TIFF's 4 GiB")
    out = bytearray(b"II" + struct.pack("<HI", 42, 8))
    off = 8
    for p in range(n):
        data_off = off + ifd_bytes
        nxt = data_off + page_bytes if p + 1 < n else 0
        ent = [(256, 4, w), (257, 4, h), (258, 3, arr.dtype.itemsize * 8), (259, 3, 1), (262, 3, 1), (273, 4, data_off), (277, 3, 1),
               (279, 4, page_bytes), (339, 3, fmt)]
        out += struct.pack("<H", len(ent))
        for tag, typ, val in ent:
            out += struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        out += struct.pack("<I", nxt)
        out += np.ascontiguousarray(vol[p].astype(arr.dtype.newbyteorder("<"))).tobytes()
        off = data_off + page_bytes
    with open(path, "wb") as f:
        f.write(bytes(out))
