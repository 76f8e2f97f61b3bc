"""OpenEXR scanline images without OpenCV -- SURVEY.md §8(f)-3, the on-disk format of every map the render path consumes.

The reference writes its position / normal maps with ``cv.imwrite('*.exr', float32 HxWx3)`` (``gen_data/gen_pos_maps.py:113-162``) and
reads them with ``cv.imread(path, cv.IMREAD_UNCHANGED)`` (``network/avatar.py:27-43``, ``dataset/dataset_mv_rgb.py:146``): single-part
scan-line files, channels ``B``, ``G``, ``R`` (OpenCV's channel order: array channel 0 is stored as ``B``), 32-bit float, ZIP
compression (16 scan lines per chunk).  ``imread`` returns exactly what ``cv.imread(..., IMREAD_UNCHANGED)`` returns for such files
(``[H, W, C]`` in B, G, R, [A] order, or ``[H, W]`` for a single ``Y`` channel); ``imwrite`` produces the same kind of file.

Format per the OpenEXR file-layout document (magic 20000630, attribute list, line-offset table, chunks ``y | size | data``) and the
ZIP codec of ImfZip.cpp: zlib over the byte-delta-predicted, even/odd de-interleaved block; a chunk that does not shrink is stored raw.
Supported: NO / RLE / ZIPS / ZIP compression, HALF / FLOAT / UINT channels with sampling 1.  PIZ, PXR24, B44, DWA, tiles and
multi-part files raise ``NotImplementedError``.  OpenCV / OpenEXR are not in this image: parity unpinned (round-trip and hand-built
files only, tests/test_formats_cpu.py)."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_MAGIC = 20000630
_NO, _RLE, _ZIPS, _ZIP, _PIZ = 0, 1, 2, 3, 4
_LINES = {_NO: 1, _RLE: 1, _ZIPS: 1, _ZIP: 16}
_DTYPES = {0: np.dtype('<u4'), 1: np.dtype('<f2'), 2: np.dtype('<f4')}
_ORDER = {'B': 0, 'G': 1, 'R': 2, 'A': 3}          # OpenCV's array channel of an EXR channel name


def _cstr(buf: bytes, pos: int):
    end = buf.index(b'\0', pos)
    return buf[pos:end].decode('latin1'), end + 1


def _unpredict(t: np.ndarray) -> np.ndarray:
    """ImfZip.cpp: t[i] = t[i-1] + t[i] - 128 (mod 256), then halves -> even / odd bytes."""
    d = t.astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) & 255).astype(np.uint8)
    n = t.size
    out = np.empty(n, np.uint8)
    out[0::2] = t[:(n + 1) // 2]
    out[1::2] = t[(n + 1) // 2:]
    return out


def _predict(raw: np.ndarray) -> np.ndarray:
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 255).astype(np.uint8)


def _unrle(src: bytes, n: int) -> np.ndarray:
    out = bytearray()
    i = 0
    while i < len(src):
        c = src[i] - 256 if src[i] > 127 else src[i]
        i += 1
        if c < 0:                                   # -c literal bytes
            out += src[i:i - c]
            i += -c
        else:                                       # c + 1 copies of the next byte
            out += bytes([src[i]]) * (c + 1)
            i += 1
    if len(out) != n:
        raise ValueError("EXR: corrupt RLE chunk")
    return np.frombuffer(bytes(out), np.uint8)


def imread(path: str) -> np.ndarray:
    with open(path, 'rb') as f:
        buf = f.read()
    magic, version = struct.unpack_from('<ii', buf, 0)
    if magic != _MAGIC:
        raise ValueError(f"{path}: not an OpenEXR file")
    if version & 0x1A00:                             # tiled (0x200), deep (0x800), multi-part (0x1000)
        raise NotImplementedError(f"{path}: tiled / deep / multi-part EXR files are not produced by the reference")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size, = struct.unpack_from('<i', buf, pos)
        attrs[name] = (typ, buf[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    channels, cp, cbuf = [], 0, attrs['channels'][1]
    while cbuf[cp] != 0:
        name, cp = _cstr(cbuf, cp)
        ptype, _plinear, xs, ys = struct.unpack_from('<iB3xii', cbuf, cp)
        cp += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError(f"{path}: sub-sampled channel {name}")
        channels.append((name, _DTYPES[ptype]))
    comp = attrs['compression'][1][0]
    if comp not in _LINES:
        raise NotImplementedError(f"{path}: EXR compression {comp} (PIZ / PXR24 / B44 / DWA); the reference's files are ZIP")
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines = _LINES[comp]
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f'<{n_chunks}Q', buf, pos)
    line_bytes = sum(dt.itemsize for _, dt in channels) * W
    planes = {name: np.empty((H, W), dt) for name, dt in channels}
    for off in offsets:
        y, size = struct.unpack_from('<ii', buf, off)
        rows = min(lines, y1 - y + 1)
        raw_n = rows * line_bytes
        data = buf[off + 8:off + 8 + size]
        if size == raw_n or comp == _NO:
            raw = np.frombuffer(data, np.uint8)
        elif comp == _RLE:
            raw = _unpredict(_unrle(data, raw_n))
        else:
            raw = _unpredict(np.frombuffer(zlib.decompress(data), np.uint8))
        if raw.size != raw_n:
            raise ValueError(f"{path}: chunk at line {y} has {raw.size} bytes, expected {raw_n}")
        p = 0
        for r in range(rows):                       # per scan line: the channels one after the other (alphabetical)
            for name, dt in channels:
                nb = W * dt.itemsize
                planes[name][y - y0 + r] = np.frombuffer(raw[p:p + nb].tobytes(), dt)
                p += nb
    names = [n for n, _ in channels]
    if len(names) == 1:
        return np.ascontiguousarray(planes[names[0]].astype(planes[names[0]].dtype.newbyteorder('=')))
    if not all(n in _ORDER for n in names):
        raise NotImplementedError(f"{path}: channels {names} (expected a subset of B, G, R, A)")
    names.sort(key=lambda n: _ORDER[n])
    dt = np.result_type(*[planes[n].dtype for n in names])
    return np.stack([planes[n].astype(dt.newbyteorder('=')) for n in names], -1)


def imwrite(path: str, img: np.ndarray, compression: str = 'zip') -> None:
    """``cv.imwrite(path, img)`` for float32 ``[H, W]`` / ``[H, W, 3]`` / ``[H, W, 4]`` arrays (channel order B, G, R, A)."""
    img = np.asarray(img)
    if img.dtype != np.float32 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] not in (1, 3, 4)):
        raise ValueError("EXR imwrite: float32 [H, W], [H, W, 1], [H, W, 3] or [H, W, 4]")
    comp = {'none': _NO, 'zips': _ZIPS, 'zip': _ZIP}[compression]
    H, W = img.shape[:2]
    if img.ndim == 2 or img.shape[2] == 1:
        chans = [('Y', img.reshape(H, W))]
    else:
        chans = sorted(((n, img[..., i]) for n, i in _ORDER.items() if i < img.shape[2]), key=lambda t: t[0])   # stored alphabetically

    def attr(name, typ, payload):
        return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(payload)) + payload

    chlist = b''.join(n.encode() + b'\0' + struct.pack('<iB3xii', 2, 0, 1, 1) for n, _ in chans) + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)
    head = struct.pack('<ii', _MAGIC, 2)
    head += attr('channels', 'chlist', chlist) + attr('compression', 'compression', bytes([comp]))
    head += attr('dataWindow', 'box2i', box) + attr('displayWindow', 'box2i', box) + attr('lineOrder', 'lineOrder', b'\0')
    head += attr('pixelAspectRatio', 'float', struct.pack('<f', 1.0)) + attr('screenWindowCenter', 'v2f', struct.pack('<2f', 0, 0))
    head += attr('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b'\0'
    lines = _LINES[comp]
    chunks = []
    for y in range(0, H, lines):
        rows = min(lines, H - y)
        raw = np.concatenate([np.ascontiguousarray(c[y + r]).astype('<f4').view(np.uint8) for r in range(rows) for _, c in chans])
        data = raw.tobytes()
        if comp != _NO:
            z = zlib.compress(_predict(raw).tobytes())
            if len(z) < len(data):
                data = z
        chunks.append(struct.pack('<ii', y, len(data)) + data)
    table, off = [], len(head) + 8 * len(chunks)
    for c in chunks:
        table.append(off)
        off += len(c)
    with open(path, 'wb') as f:
        f.write(head + struct.pack(f'<{len(table)}Q', *table) + b''.join(chunks))
