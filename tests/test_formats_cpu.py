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
    piz = head.replace(attr('compression', 'compression', b'\3'), attr('compression', 'compression', b'\4'))
    (tmp_path / "piz.exr").write_bytes(piz + struct.pack('<Q', 0))
    with pytest.raises(NotImplementedError):
        exr.imread(str(tmp_path / "piz.exr"))


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
