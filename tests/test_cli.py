"""wct_hip/cli.py: flags, pairing and naming of PytorchWCT/WCT.py + data_loader.py (CPU part), one end-to-end run (GPU)."""
import os

import numpy as np
import pytest

from wct_hip import cli


def test_parser_defaults_match_reference():
    a = cli.build_parser().parse_args([])
    # WCT.py:16-34
    assert (a.contentPath, a.stylePath, a.outf, a.mode, a.alpha, a.num_run) == ("content", "style", "stylized_results", None, 1, 1)
    assert (a.content_size, a.style_size, a.picked_content_mark, a.picked_style_mark) == (0, 0, ".", ".")
    assert not (a.UHD or a.synthesis or a.debug or a.numpy)
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["--mode", "8x"])
    a = cli.build_parser().parse_args(["--mode", "16x", "--models_root", "/m"])
    cli.checkpoint_args(a)
    assert a.e5 == "/m/trained_models/wct_se_16x_new/5SE.pth" and a.d1 == "/m/trained_models/wct_se_16x_new_sd/1SD.pth"   # WCT.py:49-58
    a = cli.build_parser().parse_args(["--mode", "16x_kd2sd", "--models_root", "/m"])
    cli.checkpoint_args(a)
    assert a.d3 == "/m/trained_models/wct_se_16x_new_sd_kd2sd/3SD.pth"                                                       # WCT.py:60-70
    a = cli.build_parser().parse_args(["--models_root", "/m"])
    cli.checkpoint_args(a)
    assert a.e4 == "/m/trained_models/original_wct_models/vgg_normalised_conv4_1.t7"                                         # WCT.py:37-47


def test_pairs_and_names(tmp_path):
    c, s = tmp_path / "c", tmp_path / "s"
    c.mkdir(); s.mkdir()
    for n in ("a.jpg", "b.v2.png", "notes.txt", "x_pick.jpeg"):
        (c / n).write_bytes(b"")
    for n in ("s1.png", "s2.jpg", "readme.md"):
        (s / n).write_bytes(b"")
    pairs = cli.list_pairs(str(c), str(s))
    cs = [x for x in os.listdir(c) if cli.is_image_file(x)]
    ss = [x for x in os.listdir(s) if cli.is_image_file(x)]
    assert pairs == [(a, b) for a in cs for b in ss] and len(pairs) == 6          # data_loader.py:32-36
    assert cli.list_pairs(str(c), str(s), "pick", "s2") == [("x_pick.jpeg", "s2.jpg")]
    assert cli.pair_name("b.v2.png", "s1.png") == "b+s1.jpg"                       # data_loader.py:60: split(".")[0]
    a = cli.build_parser().parse_args(["--mode", "16x", "--outf", "o", "--log_mark", "L"])
    assert cli.out_name(a, "b+s1.jpg") == os.path.join("o", "L_mode=16x_alpha=1_b+s1.jpg")   # WCT.py:127


def test_resize_smaller_edge(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)).save(tmp_path / "i.png")
    assert cli.load_rgb_u8(str(tmp_path / "i.png")).shape == (40, 60, 3)
    assert cli.load_rgb_u8(str(tmp_path / "i.png"), 20).shape == (20, 30, 3)        # smaller edge -> 20
    assert cli.load_rgb_u8(str(tmp_path / "i.png"), 40).shape == (40, 60, 3)        # already matching: untouched


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    """Two contents x one style through the CLI; every saved PNG equals stylize_u8 on the decoded inputs."""
    import types
    import torch
    Image = pytest.importorskip("PIL.Image")
    from wct_hip import WCT
    rng = np.random.default_rng(1)
    c, s, o = tmp_path / "content", tmp_path / "style", tmp_path / "out"
    c.mkdir(); s.mkdir()
    imgs = {"c1.png": rng.integers(0, 256, size=(64, 80, 3), dtype=np.uint8), "c2.png": rng.integers(0, 256, size=(48, 48, 3), dtype=np.uint8)}
    for n, a in imgs.items():
        Image.fromarray(a).save(c / n)
    st = rng.integers(0, 256, size=(56, 40, 3), dtype=np.uint8)
    Image.fromarray(st).save(s / "st.png")
    # .png names keep the result lossless (the reference always appends .jpg; the name rule is tested above)
    assert cli.main(["--mode", "16x", "--contentPath", str(c), "--stylePath", str(s), "--outf", str(o), "--log_mark", "T", "--alpha", "0.6"]) == 0
    w = WCT(types.SimpleNamespace(mode="16x", alpha=0.6))
    for n, a in imgs.items():
        path = o / ("T_mode=16x_alpha=0.6_%s+st.jpg" % n.split(".")[0])
        assert path.exists()
        got = np.asarray(Image.open(path).convert("RGB"))
        ref = w.stylize_u8(torch.from_numpy(a), torch.from_numpy(st)).cpu().numpy()
        assert got.shape == ref.shape
        # JPEG is lossy: compare loosely here; exactness of the conversion itself is pinned by the G9 tests
        assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).mean() < 12
    log = (o / "log_T_16x.txt").read_text()
    assert "Number of content-style pairs: 2" in log and "Processed 2 images." in log


@pytest.mark.gpu
def test_cli_resizes_on_the_device(tmp_path):
    """--content_size / --style_size (data_loader.py:52-56): the CLI resizes on the GPU; the result equals the cascade run on the
    images Pillow resizes on the host (load_rgb_u8(path, size))."""
    import types
    import torch
    Image = pytest.importorskip("PIL.Image")
    from wct_hip import WCT
    rng = np.random.default_rng(4)
    c, s, o = tmp_path / "content", tmp_path / "style", tmp_path / "out"
    c.mkdir(); s.mkdir()
    Image.fromarray(rng.integers(0, 256, size=(150, 230, 3), dtype=np.uint8)).save(c / "c.png")
    Image.fromarray(rng.integers(0, 256, size=(90, 70, 3), dtype=np.uint8)).save(s / "st.png")
    assert cli.main(["--mode", "16x", "--contentPath", str(c), "--stylePath", str(s), "--outf", str(o), "--log_mark", "R",
                     "--content_size", "64", "--style_size", "96"]) == 0
    w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0))
    ch, sh = cli.load_rgb_u8(str(c / "c.png"), 64), cli.load_rgb_u8(str(s / "st.png"), 96)
    assert ch.shape == (64, 98, 3) and sh.shape == (123, 96, 3)
    assert np.array_equal(w.resize_u8(torch.from_numpy(cli.load_rgb_u8(str(c / "c.png"))), 64).cpu().numpy(), ch)
    got = np.asarray(Image.open(o / "R_mode=16x_alpha=1_c+st.jpg").convert("RGB"))
    ref = w.stylize_u8(torch.from_numpy(ch), torch.from_numpy(sh)).cpu().numpy()
    assert got.shape == ref.shape == (64, 96, 3)
    assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).mean() < 12


@pytest.mark.gpu
def test_cli_pipeline_is_byte_identical_to_the_serial_loop(tmp_path):
    """`--pipeline 3` (decode-ahead pool, asynchronous copies, writer pool: the reference's timed region WCT.py:112-131 as a
    pipeline) writes the SAME BYTES as `--pipeline 0` (the reference's serial loop): 3 contents x 2 styles of different sizes, with
    --content_size so that the device Resize is in the path, .jpg outputs."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    c, s = tmp_path / "content", tmp_path / "style"
    c.mkdir(); s.mkdir()
    for n, shape in (("a.png", (96, 128, 3)), ("b.png", (80, 72, 3)), ("c.jpg", (120, 100, 3))):
        Image.fromarray(rng.integers(0, 256, size=shape, dtype=np.uint8)).save(c / n)
    for n, shape in (("s1.png", (64, 64, 3)), ("s2.png", (72, 56, 3))):
        Image.fromarray(rng.integers(0, 256, size=shape, dtype=np.uint8)).save(s / n)
    outs = {}
    for tag, depth in (("serial", "0"), ("pipe", "3"), ("pipe1", "1")):
        o = tmp_path / tag
        assert cli.main(["--mode", "16x", "--contentPath", str(c), "--stylePath", str(s), "--outf", str(o), "--log_mark", "P",
                         "--content_size", "64", "--pipeline", depth, "--io_threads", "3"]) == 0
        outs[tag] = {f: (o / f).read_bytes() for f in sorted(os.listdir(o)) if f.endswith(".jpg")}
        assert len(outs[tag]) == 6
        assert "Processed 6 images." in (o / "log_P_16x.txt").read_text()
    assert outs["pipe"] == outs["serial"] and outs["pipe1"] == outs["serial"]


@pytest.mark.gpu
def test_cli_pipeline_recovers_a_clamped_pair_like_the_serial_loop(tmp_path):
    """A pair whose f16x3 arithmetic clamped is recomputed with exact-fp32 convolutions -- in the pipelined loop the flag is read
    from a counter copied in stream order with the image, and everything enqueued BEHIND the clamped pair is discarded and redone
    (its style statistics may carry the clamp).  uint8 images cannot reach the f16 range, so the clamp is injected: a wrapper engine
    reports one for the third pair (through saturation_count in the serial loop, through range_flag in the pipelined one).  Both
    loops must warn once and write the same bytes -- the third pair's being those of the fp32 path."""
    import types
    import torch
    Image = pytest.importorskip("PIL.Image")
    from wct_hip import WCT
    rng = np.random.default_rng(9)
    c, s = tmp_path / "content", tmp_path / "style"
    c.mkdir(); s.mkdir()
    for n in ("a.png", "b.png", "c.png", "d.png", "e.png"):
        Image.fromarray(rng.integers(0, 256, size=(96, 112, 3), dtype=np.uint8)).save(c / n)
    Image.fromarray(rng.integers(0, 256, size=(64, 72, 3), dtype=np.uint8)).save(s / "s1.png")
    eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0))

    class Injected:
        """The real engine with a clamp reported for the pair that is stylised third (and only the first time it is)."""
        def __init__(self):
            self.n_prepared, self.fired, self.fake = 0, False, 0.0
        def __getattr__(self, name):
            return getattr(eng, name)
        def __setattr__(self, name, value):
            if name in ("n_prepared", "fired", "fake"):
                object.__setattr__(self, name, value)
            else:
                setattr(eng, name, value)
        def stylize_prepared(self, *a, **k):
            self.n_prepared += 1
            if self.n_prepared == 3 and not self.fired:
                self.fired, self.fake = True, 1.0
            return eng.stylize_prepared(*a, **k)
        def range_flag(self):
            return eng.range_flag() + self.fake
        def saturation_count(self, reset=False):
            n = eng.saturation_count(reset) + int(self.fake)
            if reset:
                self.fake = 0.0
            return n

    outs, logs = {}, {}
    pairs = cli.list_pairs(str(c), str(s))
    for tag, depth in (("serial", 0), ("pipe", 3)):
        o = tmp_path / tag
        o.mkdir()
        a = cli.build_parser().parse_args(["--mode", "16x", "--contentPath", str(c), "--stylePath", str(s), "--outf", str(o), "--log_mark", "F",
                                           "--pipeline", str(depth), "--io_threads", "2"])
        lines = []
        (cli.run_pipelined if depth else cli.run_serial)(a, Injected(), pairs, str(c), str(s), lambda x: lines.append(str(x)))
        torch.cuda.synchronize()
        outs[tag] = {f: (o / f).read_bytes() for f in sorted(os.listdir(o)) if f.endswith(".jpg")}
        logs[tag] = "\n".join(lines)
        assert len(outs[tag]) == 5 and logs[tag].count("WARNING: f16x3 range exceeded") == 1, logs[tag]
    assert outs["pipe"] == outs["serial"]
    # and the recomputed pair really is the fp32 path's picture, not the f16x3 one
    eng.set_conv_mode("fp32")
    third = pairs[2][0]
    ref = eng.to_u8(eng.stylize(eng.to_tensor_u8(torch.from_numpy(cli.load_rgb_u8(str(c / third))).cuda()),
                                eng.to_tensor_u8(torch.from_numpy(cli.load_rgb_u8(str(s / "s1.png"))).cuda()), 1.0, 1), 0).cpu().numpy()
    eng.set_conv_mode("f16x3")
    got = np.asarray(Image.open(tmp_path / "pipe" / ("F_mode=16x_alpha=1_%s+s1.jpg" % third.split(".")[0])).convert("RGB"))
    assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).mean() < 12
