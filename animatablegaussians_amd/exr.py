"""OpenEXR scanline images without OpenCV -- SURVEY.md §8(f)-3, the on-disk format of every map the render path consumes.

The reference writes its position / normal maps with ``cv.imwrite('*.exr', float32 HxWx3)`` (``gen_data/gen_pos_maps.py:113-162``) and
reads them with ``cv.imread(path, cv.IMREAD_UNCHANGED)`` (``network/avatar.py:27-43``, ``dataset/dataset_mv_rgb.py:146``): single-part
scan-line files, channels ``B``, ``G``, ``R`` (OpenCV's channel order: array channel 0 is stored as ``B``), 32-bit float, ZIP
compression (16 scan lines per chunk).  ``imread`` returns exactly what ``cv.imread(..., IMREAD_UNCHANGED)`` returns for such files
(``[H, W, C]`` in B, G, R, [A] order, or ``[H, W]`` for a single ``Y`` channel); ``imwrite`` produces the same kind of file.

Format per the OpenEXR file-layout document (magic 20000630, attribute list, line-offset table, chunks ``y | size | data``) and the
ZIP codec of ImfZip.cpp: zlib over the byte-delta-predicted, even/odd de-interleaved block; a chunk that does not shrink is stored raw.
Supported: NO / RLE / ZIPS / ZIP and (round 6, read only) PIZ compression, HALF / FLOAT / UINT channels with sampling 1.  PXR24, B44, DWA,
tiles and multi-part files raise ``NotImplementedError``.  OpenCV / OpenEXR are not in this image: parity unpinned (round-trip and hand-built
files only, tests/test_formats_cpu.py)."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_MAGIC = 20000630
_NO, _RLE, _ZIPS, _ZIP, _PIZ = 0, 1, 2, 3, 4
_LINES = {_NO: 1, _RLE: 1, _ZIPS: 1, _ZIP: 16, _PIZ: 32}
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


# ---------------------------------------------------------------------------------------------------------------------------------
# PIZ (round 6; ImfPizCompressor.cpp / ImfHuf.cpp / ImfWav.cpp of OpenEXR 2.x / 3.x, restated): a chunk of 32 scan lines is stored as
#   u16 minNonZero, u16 maxNonZero, the bytes [minNonZero, maxNonZero] of a 8192-byte bitmap of the 16-bit values that occur,
#   i32 length, `length` bytes of Huffman-coded 16-bit symbols.
# Decoding: Huffman -> per channel (and per 16-bit half of a 32-bit channel) the inverse 2-D Haar-like wavelet -> the reverse lookup table of
# the bitmap -> scan-line order.  The reference's own files are ZIP (OpenCV's writer default); released datasets need not be.
# ---------------------------------------------------------------------------------------------------------------------------------
_HUF_ENCSIZE = (1 << 16) + 1
_SHORT_ZEROCODE_RUN, _LONG_ZEROCODE_RUN = 59, 63
_SHORTEST_LONG_RUN = 2 + _LONG_ZEROCODE_RUN - _SHORT_ZEROCODE_RUN
_DECBITS = 14


class _Bits:
    """MSB-first bit reader over a bytes object (ImfHuf.cpp getBits / getChar)."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n: int) -> int:
        while self.lc < n:
            self.c = ((self.c << 8) | self.d[self.p]) & 0xFFFFFFFFFFFFFFFF
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_unpack_table(data: bytes, pos: int, im: int, iM: int):
    """hufUnpackEncTable + hufCanonicalCodeTable: code lengths (6 bits each, runs of zeros packed) -> (length, code) per symbol."""
    lens = np.zeros(_HUF_ENCSIZE, np.int64)
    b = _Bits(data, pos)
    i = im
    while i <= iM:
        ln = b.get(6)
        if ln == _LONG_ZEROCODE_RUN:
            i += b.get(8) + _SHORTEST_LONG_RUN
        elif ln >= _SHORT_ZEROCODE_RUN:
            i += ln - _SHORT_ZEROCODE_RUN + 2
        else:
            lens[i] = ln
            i += 1
    if i > iM + 1:
        raise ValueError("EXR PIZ: corrupt Huffman table (a zero run passes the last symbol)")
    # canonical codes: the longest codes get the smallest values
    n = np.bincount(lens, minlength=59).astype(object)
    c = 0
    for ln in range(58, 0, -1):
        nc = (c + n[ln]) >> 1
        n[ln] = c
        c = nc
    codes = {}
    for sym in np.nonzero(lens)[0]:
        ln = int(lens[sym])
        codes[int(sym)] = (ln, int(n[ln]))
        n[ln] += 1
    return codes, b.p


def _huf_decode(data: bytes, n_raw: int) -> np.ndarray:
    """hufUncompress: header (im, iM, table length, bit count, reserved: 5 x u32), the packed code-length table, the code stream; symbol iM is
    the run-length code (followed by an 8-bit repeat count of the previous output)."""
    if len(data) < 20:
        if n_raw == 0:
            return np.zeros(0, np.uint16)
        raise ValueError("EXR PIZ: truncated Huffman block")
    im, iM, _tlen, nbits, _ = struct.unpack_from('<5I', data, 0)
    if im >= _HUF_ENCSIZE or iM >= _HUF_ENCSIZE:
        raise ValueError("EXR PIZ: corrupt Huffman header")
    codes, pos = _huf_unpack_table(data, 20, im, iM)
    if nbits > 8 * (len(data) - pos):
        raise ValueError("EXR PIZ: Huffman bit count exceeds the block")
    # decoding table over the first _DECBITS bits; longer codes are resolved bit by bit
    short_sym = np.full(1 << _DECBITS, -1, np.int64)
    short_len = np.zeros(1 << _DECBITS, np.int64)
    long_codes = {}
    for sym, (ln, code) in codes.items():
        if ln <= _DECBITS:
            lo = code << (_DECBITS - ln)
            short_sym[lo:lo + (1 << (_DECBITS - ln))] = sym
            short_len[lo:lo + (1 << (_DECBITS - ln))] = ln
        else:
            long_codes[(ln, code)] = sym
    max_len = max((ln for ln, _ in codes.values()), default=0)
    out = np.empty(n_raw, np.uint16)
    stream = int.from_bytes(data[pos:pos + (nbits + 7) // 8], 'big')
    total = 8 * ((nbits + 7) // 8)
    ssym, slen = short_sym.tolist(), short_len.tolist()
    bitpos, o = 0, 0                                  # bits consumed, symbols written

    def peek(k):                                      # the next k bits (zero-padded past the end)
        sh = total - bitpos - k
        return (stream >> sh) & ((1 << k) - 1) if sh >= 0 else (stream << -sh) & ((1 << k) - 1)

    while bitpos < nbits:
        idx = peek(_DECBITS)
        sym = ssym[idx]
        if sym >= 0:
            bitpos += slen[idx]
        else:
            for ln in range(_DECBITS + 1, max_len + 1):
                sym = long_codes.get((ln, peek(ln)), -1)
                if sym >= 0:
                    bitpos += ln
                    break
            else:
                raise ValueError("EXR PIZ: undecodable Huffman code")
        if bitpos > nbits:
            raise ValueError("EXR PIZ: Huffman stream ends inside a code")
        if sym == iM:                                 # run-length code
            rep = peek(8)
            bitpos += 8
            if o == 0 or o + rep > n_raw:
                raise ValueError("EXR PIZ: corrupt run")
            out[o:o + rep] = out[o - 1]
            o += rep
        else:
            if o >= n_raw:
                raise ValueError("EXR PIZ: too many symbols")
            out[o] = sym
            o += 1
    if o != n_raw:
        raise ValueError(f"EXR PIZ: {o} symbols decoded, {n_raw} expected")
    return out


def _wdec14(l, h):
    ls, hs = l.astype(np.int16).astype(np.int32), h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return (ai & 0xFFFF).astype(np.uint16), ((ai - hs) & 0xFFFF).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - (1 << 15)) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a: np.ndarray, mx: int) -> None:
    """ImfWav.cpp wav2Decode on a [ny, nx] uint16 array IN PLACE: coarsest level first, every level vectorised over its 2 x 2 cells."""
    ny, nx = a.shape
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny >= p2 else np.zeros(0, np.int64)
        xs = np.arange(0, nx - p2 + 1, p2) if nx >= p2 else np.zeros(0, np.int64)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing='ij')
            i00, i10 = dec(a[Y, X], a[Y + p, X])
            i01, i11 = dec(a[Y, X + p], a[Y + p, X + p])
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if (nx & p) and ys.size:                      # odd column: 1-D along y at the first x not covered by a cell
            x = (xs[-1] + p2) if xs.size else 0
            i00, a[ys + p, x] = dec(a[ys, x], a[ys + p, x])
            a[ys, x] = i00
        if (ny & p) and xs.size:                      # odd line: 1-D along x at the first y not covered
            y = (ys[-1] + p2) if ys.size else 0
            i00, a[y, xs + p] = dec(a[y, xs], a[y, xs + p])
            a[y, xs] = i00
        p2 = p
        p >>= 1


def _unpiz(data: bytes, channels, W: int, rows: int) -> np.ndarray:
    """One PIZ chunk -> the chunk's bytes in scan-line order (per line: the channels one after the other), as the other codecs return them."""
    mn, mx_nz = struct.unpack_from('<HH', data, 0)
    bitmap = np.zeros(8192, np.uint8)
    pos = 4
    if mn <= mx_nz:
        if mx_nz >= 8192:
            raise ValueError("EXR PIZ: corrupt bitmap range")
        bitmap[mn:mx_nz + 1] = np.frombuffer(data, np.uint8, mx_nz - mn + 1, pos)
        pos += mx_nz - mn + 1
    present = np.unpackbits(bitmap, bitorder='little').astype(bool)
    present[0] = True
    lut = np.zeros(65536, np.uint16)
    vals = np.nonzero(present)[0]
    lut[:vals.size] = vals
    max_value = vals.size - 1
    length, = struct.unpack_from('<i', data, pos)
    pos += 4
    if length < 0 or pos + length > len(data):
        raise ValueError("EXR PIZ: corrupt chunk length")
    sizes = [dt.itemsize // 2 for _, dt in channels]
    n_raw = sum(sz * W * rows for sz in sizes)
    tmp = _huf_decode(data[pos:pos + length], n_raw)
    planes, p = [], 0
    for sz in sizes:                                  # channel-major: [rows][W][size] 16-bit words per channel
        blk = tmp[p:p + rows * W * sz].reshape(rows, W, sz).copy()
        for j in range(sz):
            sub = np.ascontiguousarray(blk[:, :, j])
            _wav2_decode(sub, max_value)
            blk[:, :, j] = sub
        planes.append(lut[blk])
        p += rows * W * sz
    out = np.concatenate([planes[c][r].reshape(-1) for r in range(rows) for c in range(len(sizes))])
    return out.astype('<u2').view(np.uint8)


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
        raise NotImplementedError(f"{path}: EXR compression {comp} (PXR24 / B44 / DWA); the reference's files are ZIP")
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
        elif comp == _PIZ:
            raw = _unpiz(data, channels, W, rows)
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
