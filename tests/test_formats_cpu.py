"""On-disk formats either side of the render path: EXR maps, NPY skinning weights, checkpoints, PLY export,
pose-PCA files.  Host code only; everything here runs without a GPU."""
import os
import struct
import zlib

import numpy as np
import pytest


def test_exr_roundtrip_all_layouts(tmp_path):
    from animatablegaussians_amd import exr
    rng = np.random.default_rng(0)
    for shape, comp in (((37, 53, 3), 'zip'), ((16, 8, 3), 'zips'), ((5, 7), 'none'), ((33, 20, 4), 'zip'), ((40, 64, 1), 'zip')):
        img = rng.standard_normal(shape).astype(np.float32)
        img[: shape[0] // 2] = 0.0                         # the reference's maps are mostly empty: exercises real compression
        p = str(tmp_path / f"m_{len(shape)}_{comp}.exr")
        exr.imwrite(p, img, compression=comp)
        back = exr.imread(p)
        want = img[..., 0] if img.ndim == 3 and img.shape[2] == 1 else img
        assert back.dtype == np.float32 and back.shape == want.shape
        np.testing.assert_array_equal(back, want)
    big = np.zeros((64, 128, 3), np.float32)
    big[20:40, 30:90] = rng.standard_normal((20, 60, 3)).astype(np.float32)
    exr.imwrite(str(tmp_path / "pos.exr"), big)
    assert os.path.getsize(tmp_path / "pos.exr") < big.nbytes // 2          # ZIP chunks of 16 scan lines do compress


def test_exr_reads_a_hand_built_file(tmp_path):
    """A file assembled byte by byte from the OpenEXR layout document (not by our writer): HALF channels G and R plus a FLOAT
    channel B, uncompressed, 3 x 2 pixels, data window not at the origin."""
    from animatablegaussians_amd import exr
    W, H = 3, 2
    B = np.arange(6, dtype='<f4').reshape(H, W) + 0.5
    G = (np.arange(6).reshape(H, W) * 2).astype('<f2')
    R = (-np.arange(6).reshape(H, W)).astype('<f2')

    def attr(n, t, payload):
        return n.encode() + b'\0' + t.encode() + b'\0' + struct.pack('<i', len(payload)) + payload

    ch = b''.join(n + b'\0' + struct.pack('<iB3xii', pt, 0, 1, 1) for n, pt in ((b'B', 2), (b'G', 1), (b'R', 1))) + b'\0'
    head = struct.pack('<ii', 20000630, 2) + attr('channels', 'chlist', ch) + attr('compression', 'compression', b'\0')
    head += attr('dataWindow', 'box2i', struct.pack('<4i', 10, 20, 10 + W - 1, 20 + H - 1))
    head += attr('displayWindow', 'box2i', struct.pack('<4i', 0, 0, 63, 63)) + attr('lineOrder', 'lineOrder', b'\0') + b'\0'
    rows = [B[y].tobytes() + G[y].tobytes() + R[y].tobytes() for y in range(H)]
    off0 = len(head) + 8 * H
    table = struct.pack('<2Q', off0, off0 + 8 + len(rows[0]))
    body = b''.join(struct.pack('<ii', 20 + y, len(rows[y])) + rows[y] for y in range(H))
    p = tmp_path / "hand.exr"
    p.write_bytes(head + table + body)
    img = exr.imread(str(p))
    assert img.shape == (H, W, 3) and img.dtype == np.float32
    np.testing.assert_array_equal(img[..., 0], B)
    np.testing.assert_array_equal(img[..., 1], G.astype(np.float32))
    np.testing.assert_array_equal(img[..., 2], R.astype(np.float32))
    # ZIP chunk built with zlib directly from the documented predictor / reorder, 1 channel
    Y = np.linspace(-1, 1, 16 * 5, dtype='<f4').reshape(16, 5)
    raw = np.frombuffer(Y.tobytes(), np.uint8)
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    z = zlib.compress((d & 255).astype(np.uint8).tobytes())
    ch = b'Y\0' + struct.pack('<iB3xii', 2, 0, 1, 1) + b'\0'
    head = struct.pack('<ii', 20000630, 2) + attr('channels', 'chlist', ch) + attr('compression', 'compression', b'\3')
    head += attr('dataWindow', 'box2i', struct.pack('<4i', 0, 0, 4, 15)) + b'\0'
    p2 = tmp_path / "zip.exr"
    p2.write_bytes(head + struct.pack('<Q', len(head) + 8) + struct.pack('<ii', 0, len(z)) + z)
    np.testing.assert_array_equal(exr.imread(str(p2)), Y)
    with pytest.raises(ValueError):
        (tmp_path / "bad.exr").write_bytes(b'\0' * 64)
        exr.imread(str(tmp_path / "bad.exr"))
    pxr = head.replace(attr('compression', 'compression', b'\3'), attr('compression', 'compression', b'\5'))       # PXR24: not supported
    (tmp_path / "pxr.exr").write_bytes(pxr + struct.pack('<Q', 0))
    with pytest.raises(NotImplementedError):
        exr.imread(str(tmp_path / "pxr.exr"))


def test_position_map_and_lbs_files_feed_the_avatar_constructor_layout(tmp_path):
    """The asset files AvatarNet.__init__ reads (network/avatar.py:27-43): cano_smpl_pos_map.exr [S, 2S, 3] -> mask by norm,
    init_pts_lbs.npy [N, 55]; written here the way gen_data/gen_pos_maps.py:113-134 does, read back bit-exactly."""
    from animatablegaussians_amd import exr, synth
    S = 64
    m = synth.body_mask(S)
    mask = np.concatenate([m, m[:, ::-1]], 1)
    pos = np.zeros((S, 2 * S, 3), np.float32)
    rng = np.random.default_rng(1)
    pos[mask] = rng.standard_normal((int(mask.sum()), 3)).astype(np.float32) + 3.0
    exr.imwrite(str(tmp_path / "cano_smpl_pos_map.exr"), pos)
    lbs = rng.random((int(mask.sum()), 55)).astype(np.float32)
    np.save(tmp_path / "init_pts_lbs.npy", lbs)
    back = exr.imread(str(tmp_path / "cano_smpl_pos_map.exr"))
    got_mask = np.linalg.norm(back, axis=-1) > 0.                            # network/avatar.py:28
    np.testing.assert_array_equal(got_mask, mask)
    np.testing.assert_array_equal(back[got_mask], pos[mask])
    np.testing.assert_array_equal(np.load(tmp_path / "init_pts_lbs.npy"), lbs)


def test_avatar_net_from_data_dir_and_dropin_module(tmp_path, monkeypatch):
    """AvatarNet.from_data_dir reads the three asset files of network/avatar.py:27-43; dropin/avatar_module.AvatarNet(opt) is the
    reference trainer's constructor call (main_avatar.py:45-48) on top of it, with a stand-in for the reference's `config`."""
    import sys
    import types
    import torch
    from animatablegaussians_amd import exr, synth
    from animatablegaussians_amd.avatar import AvatarNet
    S = 64
    want = AvatarNet.synthetic({'with_viewdirs': True}, S=S, device='cpu')
    d = tmp_path / "subject" / "smpl_pos_map"
    d.mkdir(parents=True)
    mask = want.cano_smpl_mask.numpy()
    cano = np.zeros(mask.shape + (3,), np.float32)
    cano[mask] = want.init_points.numpy()
    nml = np.zeros_like(cano)
    nml[mask] = want.cano_nmls.numpy()
    exr.imwrite(str(d / "cano_smpl_pos_map.exr"), cano)
    exr.imwrite(str(d / "cano_smpl_nml_map.exr"), nml)
    np.save(d / "init_pts_lbs.npy", want.lbs.numpy())
    got = AvatarNet.from_data_dir({'with_viewdirs': True}, str(tmp_path / "subject"), device='cpu')
    assert torch.equal(got.init_points, want.init_points) and torch.equal(got.lbs, want.lbs) and torch.equal(got.cano_nmls, want.cano_nmls)
    assert torch.equal(got.cano_smpl_mask, want.cano_smpl_mask) and got.with_viewdirs
    cfg = types.ModuleType("config")
    cfg.opt, cfg.device = {'train': {'data': {'data_dir': str(tmp_path / "subject")}}}, 'cpu'
    monkeypatch.setitem(sys.modules, "config", cfg)
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "animatablegaussians_amd", "dropin"))
    import importlib
    mod = importlib.import_module("avatar_module")
    net = mod.AvatarNet({'with_viewdirs': False}).to(cfg.device)
    assert isinstance(net, AvatarNet) and not net.with_viewdirs and torch.equal(net.init_points, want.init_points)
    del synth


def test_ply_export_layout_and_roundtrip(tmp_path):
    import torch
    from animatablegaussians_amd import obj_io
    g = torch.Generator().manual_seed(2)
    N = 1000
    vals = {'positions': torch.randn(N, 3, generator=g), 'colors': torch.rand(N, 3, generator=g),
            'opacity': torch.rand(N, 1, generator=g) * 0.98 + 0.01, 'scales': torch.rand(N, 3, generator=g) * 0.02 + 1e-4,
            'rotations': torch.nn.functional.normalize(torch.randn(N, 4, generator=g)), 'max_sh_degree': 0}
    p = str(tmp_path / "posed_gaussians" / "00000012.ply")
    obj_io.save_gaussians_as_ply(p, vals)
    raw = open(p, 'rb').read()
    head = raw[:raw.index(b'end_header\n')].decode().split('\n')
    assert head[:3] == ['ply', 'format binary_little_endian 1.0', f'element vertex {N}']
    props = [ln.split()[2] for ln in head if ln.startswith('property')]
    assert props[:9] == ['x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2'] and props[9] == 'f_rest_0'
    assert props[54:] == ['opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3'] and len(props) == 62
    body = np.frombuffer(raw, '<f4', N * 62, raw.index(b'end_header\n') + 11).reshape(N, 62)
    np.testing.assert_allclose(body[:, 6], (vals['colors'][:, 2].numpy() - 0.5) / obj_io.C0, rtol=1e-6)     # f_dc_0 = SH of the B... swapped channel
    assert not body[:, 3:6].any() and not body[:, 9:54].any()
    back = obj_io.load_gaussians_from_ply(p, device='cpu')
    for k in ('positions', 'colors', 'opacity', 'scales', 'rotations'):
        np.testing.assert_allclose(back[k].numpy(), vals[k].numpy(), rtol=2e-5, atol=1e-6)
    assert back['features_extr'].shape == (N, 3, 15)


def test_pose_pca_matches_scikit_learn():
    sk = pytest.importorskip("sklearn.decomposition")
    import torch
    from animatablegaussians_amd.pose_pca import PosePCA
    rng = np.random.default_rng(3)
    n_pose, P = 40, 500
    basis = rng.standard_normal((6, P * 3))
    X = (rng.standard_normal((n_pose, 6)) * np.array([5, 4, 3, 2, 1, 0.5])) @ basis + rng.standard_normal((n_pose, P * 3)) * 0.01
    ref = sk.PCA(n_components=5).fit(X)
    ours = PosePCA(5).fit(torch.from_numpy(X))
    np.testing.assert_allclose(ours.explained_variance_.numpy(), ref.explained_variance_, rtol=1e-8)
    np.testing.assert_allclose(np.abs(ours.components_.numpy()), np.abs(ref.components_), rtol=1e-6, atol=1e-9)
    x = (rng.standard_normal(6) * 12) @ basis                                # far outside the training distribution
    low = ref.transform(x.reshape(1, -1))
    std = np.sqrt(ref.explained_variance_)
    want = ref.inverse_transform(np.minimum(np.maximum(low, -2 * std), 2 * std)).reshape(-1, 3)   # dataset_mv_rgb.py:312-321
    got = ours.transform_pca(torch.from_numpy(x.reshape(-1, 3)), sigma_pca=2.)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-7, atol=1e-9)
    assert float(np.abs(want - x.reshape(-1, 3)).max()) > 0.1                # the clamp was active
    # the test loop's use (main_avatar.py:722-733): front half of the pose map replaced at the mask, back half untouched
    S = 32
    mask = np.zeros((S, S), bool)
    mask.reshape(-1)[rng.choice(S * S, P, replace=False)] = True
    pos = rng.standard_normal((S, S, 6))
    pos[..., :3][mask] = x.reshape(-1, 3)
    live = pos.copy()
    front, back = np.split(live, [3], 2)
    front[mask] = want
    ref_map = np.concatenate([front, back], 2).transpose(2, 0, 1)
    got_map = ours.project_pose_map(torch.from_numpy(pos).permute(2, 0, 1), torch.from_numpy(mask), sigma_pca=2.)
    np.testing.assert_allclose(got_map.numpy(), ref_map, rtol=1e-7, atol=1e-9)


def test_checkpoint_files_use_the_reference_layout(tmp_path):
    import torch
    from animatablegaussians_amd import checkpoint
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(0)
    net = AvatarNet.synthetic({'with_viewdirs': True}, S=64, device='cpu')
    optm = torch.optim.Adam(net.parameters(), lr=5e-4)
    for p in net.parameters():
        p.grad = torch.full_like(p, 1e-3)
    optm.step()
    checkpoint.save_ckpt(str(tmp_path / "ck"), net, optm, epoch_idx=3, iter_idx=1234)
    d = torch.load(tmp_path / "ck" / "net.pt", map_location='cpu')
    assert set(d) == {'epoch_idx', 'iter_idx', 'avatar_net'} and (d['epoch_idx'], d['iter_idx']) == (3, 1234)
    keys = set(d['avatar_net'])
    for must in ("color_net.style.1.weight", "position_net.conv_in.1.weight", "other_net.to_rgbs1.5.bias", "color_net.noises.noise_0",
                 "viewdir_net.0.weight", "viewdir_net.2.bias", "color_net.convs2.10.conv.weight"):
        assert must in keys, must
    # exactly the reference AvatarNet's keys, in its order (fixture from the reference modules): three networks, then viewdir_net
    import json
    layout = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_layout.json")))
    want_keys = [f"{n}.{k}" for n, oc in (("color_net", 3), ("position_net", 3), ("other_net", 8)) for k, _, _ in layout[f"out_ch_{oc}"]["keys"]]
    want_keys += [f"viewdir_net.{k}" for k, _ in layout["avatar_net"]["viewdir_net"]]
    assert list(d['avatar_net']) == want_keys
    want_params = [f"{n}.{k}" for n, oc in (("color_net", 3), ("position_net", 3), ("other_net", 8)) for k in layout[f"out_ch_{oc}"]["param_order"]]
    want_params += [f"viewdir_net.{k}" for k, _ in layout["avatar_net"]["viewdir_net"]]
    assert [k for k, _ in net.named_parameters()] == want_params          # Adam state is indexed by this order
    want = {k: v.clone() for k, v in checkpoint.avatar_state_dict(net).items()}
    with torch.no_grad():
        for p in net.parameters():
            p.add_(1.0)
    optm2 = torch.optim.Adam(net.parameters(), lr=5e-4)
    assert checkpoint.load_ckpt(str(tmp_path / "ck"), net, optm2) == (3, 1234)
    for k, v in checkpoint.avatar_state_dict(net).items():
        assert torch.equal(v, want[k]), k
    assert optm2.state_dict()['state'][0]['step'] == optm.state_dict()['state'][0]['step']


def test_exr_writer_output_parses_byte_by_byte_per_the_file_layout_spec(tmp_path):
    """The file `exr.imwrite` produces, walked with an independent parser written here from the OpenEXR file-layout document and the ZIP
    codec of ImfZip.cpp (plain Python loops, nothing shared with exr.py): magic / version word, the attribute list with the types and
    values OpenCV's writer emits for a float32 BGR image, the line-offset table, the 16-scan-line chunks `y | size | zlib(predicted,
    de-interleaved bytes)`, channel-planar scan lines in alphabetical channel order.  OpenCV / OpenEXR are absent here, so this is the
    most the format can be pinned: the writer against the specification rather than against its own reader."""
    import struct
    import zlib
    from animatablegaussians_amd import exr
    H, W = 40, 24                                             # 3 chunks: 16 + 16 + 8 scan lines
    img = np.random.RandomState(5).standard_normal((H, W, 3)).astype(np.float32)
    path = str(tmp_path / "m.exr")
    exr.imwrite(path, img)
    buf = open(path, "rb").read()
    assert struct.unpack_from("<i", buf, 0)[0] == 20000630                        # magic 0x01312f76
    ver = struct.unpack_from("<i", buf, 4)[0]
    assert ver & 0xff == 2 and ver >> 8 == 0                                      # version 2, no tiled / long-name / deep / multipart flags
    pos, attrs = 8, {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos); name = buf[pos:e].decode(); pos = e + 1
        e = buf.index(b"\0", pos); typ = buf[pos:e].decode(); pos = e + 1
        size = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        attrs[name] = (typ, buf[pos:pos + size]); pos += size
    pos += 1                                                                      # end of header
    for need in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"):
        assert need in attrs, need
    typ, ch = attrs["channels"]
    assert typ == "chlist"
    names, p = [], 0
    while ch[p] != 0:
        e = ch.index(b"\0", p); names.append(ch[p:e].decode()); p = e + 1
        ptype, plinear, xs, ys = struct.unpack_from("<iB3xii", ch, p); p += 16
        assert ptype == 2 and xs == 1 and ys == 1                                 # FLOAT, sampling 1
    assert names == ["B", "G", "R"] and p + 1 == len(ch)                          # alphabetical; OpenCV's channel 0 is stored as B
    assert attrs["compression"] == ("compression", b"\x03")                       # ZIP_COMPRESSION: 16 scan lines per chunk
    assert attrs["dataWindow"] == ("box2i", struct.pack("<4i", 0, 0, W - 1, H - 1)) and attrs["displayWindow"][1] == attrs["dataWindow"][1]
    assert attrs["lineOrder"] == ("lineOrder", b"\0")                             # INCREASING_Y
    assert attrs["pixelAspectRatio"] == ("float", struct.pack("<f", 1.0)) and attrs["screenWindowWidth"] == ("float", struct.pack("<f", 1.0))
    assert attrs["screenWindowCenter"] == ("v2f", struct.pack("<2f", 0.0, 0.0))
    n_chunks = (H + 15) // 16
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos); pos += 8 * n_chunks
    assert offsets[0] == pos                                                      # the first chunk follows the table
    out = np.zeros((H, W, 3), np.float32)
    for ci, off in enumerate(offsets):
        y0, size = struct.unpack_from("<ii", buf, off)
        assert y0 == 16 * ci
        lines = min(16, H - y0)
        raw_n = lines * W * 3 * 4
        data = buf[off + 8: off + 8 + size]
        if size < raw_n:                                                          # ImfZip.cpp: inflate, then undo predictor and interleave
            t = bytearray(zlib.decompress(data))
            assert len(t) == raw_n
            for i in range(1, raw_n):
                t[i] = (t[i - 1] + t[i] - 128) & 255
            half = (raw_n + 1) // 2
            raw = bytearray(raw_n)
            raw[0::2] = t[:half]
            raw[1::2] = t[half:]
        else:
            raw = data                                                            # a chunk that does not shrink is stored raw
        blk = np.frombuffer(bytes(raw), "<f4").reshape(lines, 3, W)               # per scan line: the channels' rows in chlist order
        out[y0:y0 + lines] = blk.transpose(0, 2, 1)
        assert (offsets[ci + 1] if ci + 1 < n_chunks else len(buf)) == off + 8 + size     # chunks are back to back, the file ends with the last
    np.testing.assert_array_equal(out, img)


def test_pre_round2_checkpoints_load_by_name_and_their_optimizer_file_is_refused(tmp_path):
    """A `net.pt` written before the constant FIR / Haar buffers became part of the state (and with the old parameter order) still
    loads -- by name, through `load_reference_state_dict` -- while its `optm.pt`, whose Adam moments are indexed by the OLD parameter
    order, is skipped instead of being loaded onto the wrong parameters."""
    import torch
    from animatablegaussians_amd import checkpoint
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(1)
    net = AvatarNet.synthetic({'with_viewdirs': True}, S=64, device='cpu')
    optm = torch.optim.Adam(net.parameters(), lr=5e-4)
    for p in net.parameters():
        p.grad = torch.full_like(p, 1e-3)
    optm.step()
    const = ('.kernel', '.ll', '.lh', '.hl', '.hh')
    want = {k: v.clone() for k, v in checkpoint.avatar_state_dict(net).items()}
    legacy = {k: v.cpu() for k, v in want.items() if not k.endswith(const)}
    assert len(legacy) < len(want)
    os.makedirs(tmp_path / "old")
    torch.save({'epoch_idx': 1, 'iter_idx': 7, 'avatar_net': legacy}, tmp_path / "old" / "net.pt")
    torch.save({'avatar_net': optm.state_dict()}, tmp_path / "old" / "optm.pt")
    with torch.no_grad():
        for p in net.parameters():
            p.add_(1.0)
    optm2 = torch.optim.Adam(net.parameters(), lr=5e-4)
    assert checkpoint.load_ckpt(str(tmp_path / "old"), net, optm2) == (1, 7)
    for k, v in checkpoint.avatar_state_dict(net).items():
        assert torch.equal(v, want[k]), k
    assert len(optm2.state_dict()['state']) == 0                      # the optimizer file was not applied


# ---------------------------------------------------------------------------------------------------------------------------------
# PIZ (round 6).  No OpenEXR / OpenCV in the image, so the files are built HERE by an encoder written from the published algorithm
# (ImfPizCompressor.cpp / ImfWav.cpp / ImfHuf.cpp), sharing nothing with exr.py: scalar Python loops for the forward wavelet, a heap-built
# Huffman code whose LENGTHS go into the file (the decoder must derive the canonical codes from them), zero runs in the length table in both
# packed forms, run-length symbols in the stream.  Parity with a file OpenEXR itself wrote stays unpinned.
# ---------------------------------------------------------------------------------------------------------------------------------
def _piz_wenc14(a, b):
    a, b = (a ^ 0x8000) - 0x8000 if a & 0x8000 else a, (b ^ 0x8000) - 0x8000 if b & 0x8000 else b      # as signed shorts
    return ((a + b) >> 1) & 0xFFFF, (a - b) & 0xFFFF


def _piz_wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m, d = (ao + b) >> 1, ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m & 0xFFFF, d & 0xFFFF


def _piz_wav2_encode(a, nx, ny, mx):
    enc = _piz_wenc14 if mx < (1 << 14) else _piz_wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        for y in range(0, ny - p2 + 1, p2):
            x = 0
            for x in range(0, nx - p2 + 1, p2):
                i00, i01 = enc(a[y][x], a[y][x + p])
                i10, i11 = enc(a[y + p][x], a[y + p][x + p])
                a[y][x], a[y + p][x] = enc(i00, i10)
                a[y][x + p], a[y + p][x + p] = enc(i01, i11)
            if nx & p:
                xo = (nx - p2) // p2 * p2 + p2 if nx >= p2 else 0
                i00, a[y + p][xo] = enc(a[y][xo], a[y + p][xo])
                a[y][xo] = i00
        if ny & p:
            yo = (ny - p2) // p2 * p2 + p2 if ny >= p2 else 0
            for x in range(0, nx - p2 + 1, p2):
                i00, a[yo][x + p] = enc(a[yo][x], a[yo][x + p])
                a[yo][x] = i00
        p, p2 = p2, p2 << 1


def _piz_huffman(symbols, force_long=False):
    """(packed table + stream bytes as ImfHuf.cpp hufCompress lays them out).  Lengths from a heap-built Huffman tree over the symbols plus the
    run-length symbol iM; runs of >= 3 equal symbols are coded as symbol, iM, count."""
    import heapq
    from collections import Counter
    # run-length pre-pass: (sym) or (sym, RL, count)
    items, i = [], 0
    while i < len(symbols):
        j = i
        while j + 1 < len(symbols) and symbols[j + 1] == symbols[i] and j - i < 255:
            j += 1
        run = j - i                                # repeats after the first
        items.append((symbols[i], run if run >= 2 else 0))
        i += (run + 1) if run >= 2 else 1
    freq = Counter(s for s, _ in items)
    im, iM = min(freq), max(freq) + 1              # iM = the run-length symbol
    freq[iM] = max(1, sum(1 for _, r in items if r))
    heap = [(f, k, (k,)) for k, f in freq.items()]
    heapq.heapify(heap)
    lens = {k: 0 for k in freq}
    if len(heap) == 1:
        lens[heap[0][1]] = 1
    while len(heap) > 1:
        f1, k1, s1 = heapq.heappop(heap)
        f2, k2, s2 = heapq.heappop(heap)
        for k in s1 + s2:
            lens[k] += 1
        heapq.heappush(heap, (f1 + f2, min(k1, k2), s1 + s2))
    assert max(lens.values()) <= 58
    # canonical codes (the decoder's rule: longest codes get the smallest values, symbols in increasing order within a length)
    n = [0] * 59
    for l in lens.values():
        n[l] += 1
    c = 0
    for l in range(58, 0, -1):
        nc = (c + n[l]) >> 1
        n[l] = c
        c = nc
    code = {}
    for k in sorted(lens):
        code[k] = (lens[k], n[lens[k]])
        n[lens[k]] += 1
    bits = []

    def put(v, nb):
        bits.extend((v >> (nb - 1 - t)) & 1 for t in range(nb))

    k = im
    while k <= iM:                                 # the length table: 6 bits per symbol, zero runs as 59 + (run - 2) or 63 + 8-bit (run - 6)
        if lens.get(k, 0):
            put(lens[k], 6)
            k += 1
            continue
        z = k
        while z <= iM and not lens.get(z, 0):
            z += 1
        run = z - k
        while run:
            if run >= 6 or (force_long and run >= 6):
                r = min(run, 255 + 6)
                put(63, 6); put(r - 6, 8)
            elif run >= 2:
                r = min(run, 5)
                put(59 + r - 2, 6)
            else:
                r = 1
                put(0, 6)
            run -= r
            k += r
    while len(bits) % 8:
        bits.append(0)
    table = bytes(int(''.join(map(str, bits[t:t + 8])), 2) for t in range(0, len(bits), 8))
    bits = []
    for s_, r in items:
        put(code[s_][1], code[s_][0])
        if r:
            put(code[iM][1], code[iM][0]); put(r, 8)
    nbits = len(bits)
    while len(bits) % 8:
        bits.append(0)
    stream = bytes(int(''.join(map(str, bits[t:t + 8])), 2) for t in range(0, len(bits), 8))
    return struct.pack('<5I', im, iM, len(table), nbits, 0) + table + stream


def _piz_chunk(rows_by_channel, dtypes, W):
    """rows_by_channel[c]: [rows, W] array of dtypes[c]; returns the chunk's data bytes."""
    rows = rows_by_channel[0].shape[0]
    words = [np.ascontiguousarray(a.astype(dt)).view('<u2').reshape(rows, W, dt.itemsize // 2) for a, dt in zip(rows_by_channel, dtypes)]
    allw = np.concatenate([w.reshape(-1) for w in words])
    present = np.zeros(65536, bool)
    present[allw] = True
    present[0] = False                                         # zero is never stored
    nz = np.nonzero(present)[0]
    bitmap = np.packbits(present, bitorder='little')
    mn, mx = (int(nz[0]) >> 3, int(nz[-1]) >> 3) if nz.size else (8191, 0)
    fwd = np.zeros(65536, np.int64)
    vals = np.concatenate([[0], nz])
    fwd[vals] = np.arange(vals.size)
    max_value = vals.size - 1
    syms = []
    for w in words:
        q = fwd[w]
        for j in range(w.shape[2]):
            a = [[int(v) for v in row] for row in q[:, :, j]]
            _piz_wav2_encode(a, W, rows, max_value)
            q[:, :, j] = np.array(a)
        syms.extend(int(v) for v in q.reshape(-1))
    huf = _piz_huffman(syms)
    out = struct.pack('<HH', mn, mx) + (bitmap[mn:mx + 1].tobytes() if mn <= mx else b'') + struct.pack('<i', len(huf)) + huf
    return out, max_value


@pytest.mark.parametrize("case", ["float_bgr_37x45", "half_y_16bit_range", "constant_runs", "two_chunks_70_lines"])
def test_exr_piz_files_built_by_an_independent_encoder_decode_exactly(tmp_path, case):
    from animatablegaussians_amd import exr
    rng = np.random.default_rng(7)
    if case == "float_bgr_37x45":                  # odd sizes: the wavelet's odd-column / odd-line branches at several levels; 14-bit path
        H, W, names, dts = 37, 45, ['B', 'G', 'R'], [np.dtype('<f4')] * 3
        yy, xx = np.mgrid[0:H, 0:W]
        planes = [np.round(np.sin(0.2 * xx + c) + 0.1 * yy, 1).astype(np.float32) for c in range(3)]       # few distinct values: 14-bit wavelet
    elif case == "half_y_16bit_range":             # > 16383 distinct 16-bit values in a chunk: the 16-bit wavelet
        H, W, names, dts = 32, 700, ['Y'], [np.dtype('<f2')]
        planes = [rng.integers(0, 65536, (H, W)).astype(np.uint16).view(np.float16)]           # random bit patterns: ~19 k distinct values
    elif case == "constant_runs":                  # long runs -> run-length symbols in the Huffman stream, a one-symbol code
        H, W, names, dts = 20, 33, ['A', 'B', 'G', 'R'], [np.dtype('<f4')] * 4
        planes = [np.full((H, W), v, np.float32) for v in (1.0, 0.0, 0.5, -2.0)]
        planes[2][5:9, 7:20] = 3.25
    else:                                          # more than one chunk (32 lines each), UINT + HALF + FLOAT channels mixed
        H, W, names, dts = 70, 19, ['B', 'G', 'R'], [np.dtype('<u4'), np.dtype('<f2'), np.dtype('<f4')]
        planes = [rng.integers(0, 1000, (H, W)).astype(np.uint32), rng.standard_normal((H, W)).astype(np.float16), rng.standard_normal((H, W)).astype(np.float32)]

    def attr(name, typ, payload):
        return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(payload)) + payload

    ptype = {np.dtype('<u4'): 0, np.dtype('<f2'): 1, np.dtype('<f4'): 2}
    ch = b''.join(n.encode() + b'\0' + struct.pack('<iB3xii', ptype[dt], 0, 1, 1) for n, dt in zip(names, dts)) + b'\0'
    head = struct.pack('<ii', 20000630, 2) + attr('channels', 'chlist', ch) + attr('compression', 'compression', b'\4')
    head += attr('dataWindow', 'box2i', struct.pack('<4i', 0, 0, W - 1, H - 1)) + attr('lineOrder', 'lineOrder', b'\0') + b'\0'
    chunks, paths_used = [], set()
    for y in range(0, H, 32):
        rows = min(32, H - y)
        data, max_value = _piz_chunk([p[y:y + rows] for p in planes], dts, W)
        paths_used.add(max_value < (1 << 14))
        chunks.append(struct.pack('<ii', y, len(data)) + data)
    table, off = [], len(head) + 8 * len(chunks)
    for c in chunks:
        table.append(off)
        off += len(c)
    path = tmp_path / f"{case}.exr"
    path.write_bytes(head + struct.pack(f'<{len(table)}Q', *table) + b''.join(chunks))
    img = exr.imread(str(path))
    if case == "half_y_16bit_range":
        assert paths_used == {False}                # the 16-bit wavelet was exercised
        assert img.shape == (H, W) and np.array_equal(img.view(np.uint16), planes[0].view(np.uint16))
    else:
        if case != "two_chunks_70_lines":
            assert paths_used == {True}
        order = {'B': 0, 'G': 1, 'R': 2, 'A': 3}
        assert img.shape == (H, W, len(names))
        for n, p_ in zip(names, planes):
            got = img[..., order[n]]
            assert np.array_equal(got.astype(np.float64), p_.astype(np.float64)), (case, n)
    # a corrupted stream is refused, not mis-decoded silently into the wrong shape
    bad = bytearray(path.read_bytes())
    bad[table[0] + 8 + 4 + 30] ^= 0xFF
    (tmp_path / "bad.exr").write_bytes(bytes(bad))
    try:
        exr.imread(str(tmp_path / "bad.exr"))
    except (ValueError, IndexError):
        pass
