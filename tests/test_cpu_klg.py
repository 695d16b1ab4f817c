"""CPU suite: .klg log IO (GUI/Tools/KlgLogReader.cpp:22-87) -- container layout, zlib/raw depth, mm quantisation."""
import struct
import zlib

import numpy as np
import pytest

from co_fusion_amd import synth


@pytest.fixture(scope="module")
def klg():
    import __graft_entry__ as g
    g.build()
    from co_fusion_amd import klg as k
    return k


def _frames(n, W, H):
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=1)
    return [sc.render(cam, t, noise=True)[:2] for t in range(n)]


@pytest.mark.parametrize("compress", [True, False])
def test_klg_round_trip(tmp_path, klg, compress):
    W, H = 64, 48
    frames = _frames(3, W, H)
    path = tmp_path / "log.klg"
    with klg.KlgWriter(path, W, H, compress_depth=compress) as w:
        for t, (d, rgb) in enumerate(frames):
            w.write(1000 + 33333 * t, d, rgb)
    r = klg.KlgReader(path, W, H)
    assert r.num_frames == 3
    got = list(r)
    assert len(got) == 3
    for t, ((ts, d, rgb), (d0, rgb0)) in enumerate(zip(got, frames)):
        assert ts == 1000 + 33333 * t
        mm = np.where((d0 * 1000 > 0) & (d0 * 1000 < 65535), np.rint(d0 * np.float32(1000.0)), 0).astype(np.uint16)
        assert np.array_equal(d, mm.astype(np.float32) * np.float32(0.001))   # convertTo(CV_32FC1, 0.001)
        assert np.array_equal(rgb, rgb0)
    r.close()


def test_klg_container_layout_matches_the_reference_reader(tmp_path, klg):
    """Bytes as KlgLogReader::getCore walks them: int32 n; {int64 ts; int32 dsize; int32 rsize; depth; rgb}."""
    W, H = 32, 24
    (d, rgb), = _frames(1, W, H)
    path = tmp_path / "one.klg"
    with klg.KlgWriter(path, W, H, compress_depth=True) as w:
        w.write(42, d, rgb)
    raw = path.read_bytes()
    n, = struct.unpack_from("<i", raw, 0)
    ts, dsize, rsize = struct.unpack_from("<qii", raw, 4)
    assert (n, ts, rsize) == (1, 42, W * H * 3)
    depth = np.frombuffer(zlib.decompress(raw[20:20 + dsize]), np.uint16).reshape(H, W)
    assert np.array_equal(depth, np.rint(d * np.float32(1000.0)).astype(np.uint16))
    assert raw[20 + dsize:] == rgb.tobytes()

    # and a log written by somebody else (raw depth, hand-built) reads back
    other = tmp_path / "raw.klg"
    mm = (np.arange(W * H, dtype=np.uint16).reshape(H, W) * 7) % 5000
    other.write_bytes(struct.pack("<i", 1) + struct.pack("<qii", 7, W * H * 2, W * H * 3) + mm.tobytes() + rgb.tobytes())
    ts, dd, cc = next(iter(klg.KlgReader(other, W, H)))
    assert ts == 7 and np.array_equal(dd, mm.astype(np.float32) * np.float32(0.001)) and np.array_equal(cc, rgb)


def test_klg_rejects_jpeg_and_truncation(tmp_path, klg):
    W, H = 32, 24
    p = tmp_path / "jpeg.klg"
    mm = np.zeros((H, W), np.uint16)
    p.write_bytes(struct.pack("<i", 1) + struct.pack("<qii", 0, W * H * 2, 100) + mm.tobytes() + b"\xff\xd8" + b"\0" * 98)
    with pytest.raises(klg.KlgError, match="JPEG"):
        next(iter(klg.KlgReader(p, W, H)))
    q = tmp_path / "short.klg"
    q.write_bytes(struct.pack("<i", 2) + struct.pack("<qii", 0, W * H * 2, 0) + mm.tobytes())
    r = iter(klg.KlgReader(q, W, H))
    next(r)
    with pytest.raises(klg.KlgError):
        next(r)
    with pytest.raises(klg.KlgError):
        klg.KlgReader(tmp_path / "missing.klg", W, H)


def _jpeg_log(tmp_path, klg, rgb, d, name, **save_kw):
    import io
    from PIL import Image
    W, H = rgb.shape[1], rgb.shape[0]
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", **save_kw)
    jb = buf.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(jb)).convert("RGB"))
    mm = np.rint(d * np.float32(1000.0)).astype(np.uint16)
    path = tmp_path / name
    path.write_bytes(struct.pack("<i", 1) + struct.pack("<qii", 9, W * H * 2, len(jb)) + mm.tobytes() + jb)
    return path, ref


@pytest.mark.parametrize("subsampling,quality", [(0, 95), (1, 85), (2, 75)])
def test_klg_jpeg_colour_frames_decode_like_libjpeg(tmp_path, klg, subsampling, quality):
    """Real .klg logs store colour as JPEG (KlgLogReader.cpp:76-79 -> libjpeg).  The built-in baseline decoder must give
    libjpeg's pixels (Pillow bundles libjpeg-turbo: islow IDCT, fancy upsampling) -- exactly, for 4:4:4 / 4:2:2 / 4:2:0 --
    stored channel-reversed as JPEGLoader::readData does."""
    pytest.importorskip("PIL")
    W, H = 96, 80   # not a multiple of the 16x16 MCU: exercises the padded edge
    (d, rgb), = _frames(1, W, H)
    path, ref = _jpeg_log(tmp_path, klg, rgb, d, "j.klg", quality=quality, subsampling=subsampling)
    ts, dd, cc = next(iter(klg.KlgReader(path, W, H)))
    assert ts == 9
    assert np.array_equal(cc[..., ::-1], ref)
    # flip_colors undoes the reversal (LogReader::flipColors)
    _, _, cf = next(iter(klg.KlgReader(path, W, H, flip_colors=True)))
    assert np.array_equal(cf, ref)


def test_klg_jpeg_restart_intervals_and_progressive(tmp_path, klg):
    pytest.importorskip("PIL")
    W, H = 64, 48
    (d, rgb), = _frames(1, W, H)
    try:
        path, ref = _jpeg_log(tmp_path, klg, rgb, d, "r.klg", quality=90, subsampling=2, restart_marker_blocks=2)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    _, _, cc = next(iter(klg.KlgReader(path, W, H)))
    assert np.array_equal(cc[..., ::-1], ref)
    path, _ = _jpeg_log(tmp_path, klg, rgb, d, "p.klg", quality=90, progressive=True)
    with pytest.raises(klg.KlgError, match="progressive"):
        next(iter(klg.KlgReader(path, W, H)))
