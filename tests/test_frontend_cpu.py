"""Host logic of the calibration-JSON front end (no GPU): schema parsing, transform literal formats, the
reference's parse-time errors.  The numerical part (pose initialisation refinement, solve) is in test_gpu_frontend."""
import json

import numpy as np
import pytest

from visgeom_amd import capi, synthetic as S
from visgeom_amd.calibration import GenericCameraCalibration, transform_from_values


def test_transform_literal_formats():
    """transformFromData, include/json.h:36-67 (README.md:160-173)"""
    assert np.array_equal(transform_from_values([1.0, 2.0, 0.5]), [1, 2, 0, 0, 0, 0.5])                 # x, y, theta
    v6 = [0.1, -0.2, 0.3, 0.3, -0.4, 0.1]
    assert np.array_equal(transform_from_values(v6), v6)                                                  # t, rotvec
    r = np.array([0.3, -0.4, 0.1])
    th = np.linalg.norm(r)
    q = np.concatenate([r / th * np.sin(th / 2), [np.cos(th / 2)]])
    out = transform_from_values(np.concatenate([[1, 2, 3], q]))                                           # t, quaternion xyzw
    assert np.allclose(out[:3], [1, 2, 3]) and np.allclose(out[3:], r, atol=1e-15)
    R = S.rodrigues(r)
    m = np.concatenate([np.concatenate([R, [[1], [2], [3]]], axis=1).ravel()])                            # row-major [R | t]
    out = transform_from_values(m)
    assert np.allclose(out[:3], [1, 2, 3]) and np.allclose(out[3:], r, atol=1e-14)
    with pytest.raises(capi.VisgeomError) as e:
        transform_from_values([1, 2, 3, 4])
    assert "invalid trasformation format" in str(e.value)   # the reference's message, typo included


def _write(tmp_path, prior=True, **kw):
    d = S.make_mono("eucm", 5, 0)
    return d, S.write_calibration_json(str(tmp_path), d, "eucm", prior=prior, **kw)


def test_parse_with_priors_needs_no_gpu(tmp_path):
    d, path = _write(tmp_path, prior=True, flags=["_check_extraction", "do_not_solve"], skip=(2,))
    c = GenericCameraCalibration()
    assert c.addResiduals(path)
    assert c.num_datasets() == 1
    assert np.array_equal(c.intrinsics("cam"), d["init_intrinsics"])
    assert np.array_equal(c.transform("xiCamBoard"), d["init_poses"])
    log = c.log()
    assert "Model : EUCM" in log and "Camera : cam" in log and "Transformations : xiCamBoard" in log
    assert "WARNING : UNKNOWN FLAG -- _check_extraction" in log   # disabled flags are tolerated
    c.close()


def _mutate(path, fn):
    root = json.load(open(path))
    fn(root)
    json.dump(root, open(path, "w"))


@pytest.mark.parametrize("what,msg", [
    ("constant_no_prior", "is constant but there is no prior"),
    ("bad_model", "invalid camera model name"),
    ("bad_count", "invalid number of intrinsic parameters"),
    ("two_sequences", "not one sequences in a transform chain"),
    ("unknown_type", "is not supported"),
    ("images_without_corners", "corner detector is out of scope"),
    ("bad_literal", "invalid trasformation format"),
    ("init_has_prior", "has a prior value"),
    ("missing_key", "No such node"),
])
def test_parse_errors_mirror_the_reference(tmp_path, what, msg):
    d, path = _write(tmp_path, prior=True)

    def fn(r):
        if what == "constant_no_prior":
            r["transformations"][0].update(prior=False, constant=True)
        elif what == "bad_model":
            r["cameras"][0]["type"] = "pinhole"
        elif what == "bad_count":
            r["cameras"][0]["value"] = r["cameras"][0]["value"][:5]
        elif what == "two_sequences":
            r["transformations"].append({"name": "other", "global": False, "prior": True, "constant": False,
                                         "value": r["transformations"][0]["value"]})
            r["data"][0]["transform_chain"].append({"name": "other", "direct": True})
        elif what == "unknown_type":
            r["data"][0]["type"] = "stereo_pairs"
        elif what == "images_without_corners":
            r["data"][0].update(type="images", object={"cols": 12, "rows": 8, "size": 0.1})
        elif what == "bad_literal":
            r["transformations"][0]["value"][1] = [1, 2, 3, 4, 5]
        elif what == "init_has_prior":
            r["data"][0]["init"] = "xiCamBoard"
        elif what == "missing_key":
            del r["cameras"][0]["constant"]

    _mutate(path, fn)
    c = GenericCameraCalibration()
    with pytest.raises(capi.VisgeomError) as e:
        c.addResiduals(path)
    assert msg in str(e.value)
    c.close()


def test_images_entry_with_corners_file(tmp_path):
    d = S.make_mono("eucm", 4, 0)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=True, as_images=True)
    c = GenericCameraCalibration()
    c.addResiduals(path)
    assert c.num_datasets() == 1
    c.close()


def test_cli_is_built_and_prints_usage():
    import subprocess

    from visgeom_amd import _build

    _build.build()
    r = subprocess.run([_build.CLI], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: calib file1.json" in r.stderr


def test_transformation_prior_entries(tmp_path):
    """parseData :808-829: the transform must exist and have a prior value; global, or a sequence (element 0)"""
    d, path = _write(tmp_path, prior=True)

    def add(r, name="xiRig", with_prior=True, is_global=True):
        tf = {"name": name, "global": is_global, "prior": with_prior, "constant": False}
        if with_prior:
            tf["value"] = [0.1, 0, 0, 0, 0, 0] if is_global else [[0, 0, 1, 0, 0, 0]]
        r["transformations"].append(tf)
        r["data"].append({"type": "transformation_prior", "transform": name, "stiffness": [10, 10, 10, 5, 5, 5]})

    _mutate(path, lambda r: add(r))
    c = GenericCameraCalibration()
    c.addResiduals(path)
    c.close()
    d, path = _write(tmp_path, prior=True)
    _mutate(path, lambda r: add(r, with_prior=False))
    c = GenericCameraCalibration()
    with pytest.raises(capi.VisgeomError) as e:
        c.addResiduals(path)
    assert "must have a prior value" in str(e.value)
    c.close()
    # a sequence with a prior value is accepted: the block goes on its element 0 (getTransformData default index)
    d, path = _write(tmp_path, prior=True)
    _mutate(path, lambda r: add(r, is_global=False))
    c = GenericCameraCalibration()
    assert c.addResiduals(path)
    c.close()
    d, path = _write(tmp_path, prior=True)
    _mutate(path, lambda r: r["data"].append({"type": "transformation_prior", "transform": "xiCamBoard", "stiffness": [1] * 6}))
    c = GenericCameraCalibration()
    assert c.addResiduals(path)
    c.close()
    d, path = _write(tmp_path, prior=True)
    _mutate(path, lambda r: r["data"].append({"type": "transformation_prior", "transform": "nope", "stiffness": [1] * 6}))
    c = GenericCameraCalibration()
    with pytest.raises(capi.VisgeomError) as e:
        c.addResiduals(path)
    assert "has not been declared" in str(e.value)


def test_odometry_entry_initialises_the_sequence(tmp_path):
    """data type "odometry" (unified_calibration.cpp:743-807): "init": true fills the sequence from the odometry
    values; parsing needs no GPU"""
    d = S.make_handeye(6)
    path = S.write_handeye_json(str(tmp_path), d)
    c = GenericCameraCalibration()
    assert c.addResiduals(path)
    assert np.array_equal(c.transform("xiOdomBase"), d["odometry"])
    assert np.array_equal(c.transform("xiBaseCam").ravel(), d["init_xi_base_cam"])
    assert c.num_datasets() == 1
    c.close()


@pytest.mark.parametrize("what,msg", [
    ("undeclared", "has not been declared"),
    ("global", "is global. Odometry must be a sequence"),
    ("twice", "has already been initialized"),
    ("missing", "No such node"),
    ("bad_literal", "invalid trasformation format"),
])
def test_odometry_parse_errors_mirror_the_reference(tmp_path, what, msg):
    d = S.make_handeye(6)
    path = S.write_handeye_json(str(tmp_path), d)

    def fn(r):
        odo = r["data"][0]
        if what == "undeclared":
            odo["transform"] = "nope"
        elif what == "global":
            odo["transform"] = "xiBaseCam"
        elif what == "twice":
            r["data"].append(dict(odo))
        elif what == "missing":
            del odo["anchor"]
        elif what == "bad_literal":
            odo["value"][2] = [1, 2]
    _mutate(path, fn)
    c = GenericCameraCalibration()
    with pytest.raises(capi.VisgeomError) as e:
        c.addResiduals(path)
    assert msg in str(e.value)
    c.close()


def test_json_parser_survives_mutated_input(tmp_path):
    """truncations, random byte flips and structural edits of a valid calibration file must come back as an error
    code (or parse), never crash the host library"""
    d, path = _write(tmp_path, prior=True)
    text = open(path).read()
    rng = np.random.default_rng(7)
    variants = [text[:k] for k in rng.integers(0, len(text), 60)]
    for _ in range(120):
        b = bytearray(text.encode())
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, len(b)))
            op = int(rng.integers(0, 3))
            if op == 0:
                b[pos] = int(rng.integers(32, 127))
            elif op == 1:
                del b[pos]
            else:
                b.insert(pos, int(rng.choice(list(b'{}[]",:0-9eE.tfn \n'))))
        variants.append(b.decode(errors="replace"))
    variants += ["", "[]", "{}", "nul", '{"a":', '{"transformations": 3, "cameras": [], "data": []}', "[" * 200000, '{"a":' * 100000,
                 '{"transformations": [], "cameras": [], "data": [{"type": "ir_data"}]}', "1e999", '"\\u12"', '{"x": "\\q"}']
    ok = bad = 0
    for i, v in enumerate(variants):
        f = tmp_path / ("mut_%d.json" % i)
        f.write_text(v)
        c = GenericCameraCalibration()
        try:
            c.addResiduals(str(f))
            ok += 1
        except capi.VisgeomError:
            bad += 1
        finally:
            c.close()
    assert ok + bad == len(variants) and bad > len(variants) // 2


def test_odometry_intrinsic_entry_parses_and_initialises_the_sequence(tmp_path):
    """parse side of "odometry_intrinsic" (unified_calibration.cpp:660-742), no GPU: the wheel geometry lands in
    intrinsicMap[transform], the sequence is xi_0 = 0, xi_i+1 = xi_i o zetaPrior_i with zetaPrior_i the chained wheel
    increments under the prior geometry (checked against the oracle's OdometryCost constructor)."""
    import numpy as np

    from oracle import vgo

    n = 6
    d = S.make_wheeled(n)
    path = S.write_wheeled_json(str(tmp_path), d)
    c = GenericCameraCalibration()
    c.addResiduals(path)
    assert np.array_equal(c.intrinsics("xiOdomBase"), d["init_wheels"])
    seq = c.transform("xiOdomBase")
    assert seq.shape == (n, 6) and np.array_equal(seq[0], np.zeros(6))
    xi = np.zeros(6)
    for i in range(n - 1):
        xi = vgo.compose(xi, vgo.OdometryCost(0.05, 0.05, 0.05, d["delta_q"][i], d["init_wheels"]).zeta)
        assert np.max(np.abs(seq[i + 1] - xi)) < 1e-14
    c.close()


def test_text_to_double_conversion_equals_strtod(tmp_path):
    """the corner files are read without strtod (vg_json.hpp, Cursor::number: exact 128-bit integer arithmetic for the common
    shapes); a compiled host check holds every conversion to strtod's bits on 600 000 generated numbers incl. exact ties"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "json_number_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(root, "tests", "host", "json_number_check.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert b"bad 0" in r.stdout


def test_parallel_cut_of_a_corner_file_equals_the_serial_one(tmp_path):
    """element_spans_parallel (three sweeps over ranges of the text side by side: string state, depth, top-level separators) against
    element_spans_serial on random arrays whose strings carry brackets, commas, quotes and backslash runs, with range boundaries
    inside strings and escapes, and on damaged copies: equal, or declined whenever the serial cut throws"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "json_spans_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(root, "tests", "host", "json_spans_check.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert b" bad 0" in r.stdout


def test_report_numbers_print_as_percent_g(tmp_path):
    """the residual report's numbers (writeImageResidual unified_calibration.cpp:1210-1213 through operator<<: "%g") come from
    vgtext::fmt_g6, one multiplication by an exact power of ten with the exact conversion behind every near-tie: a compiled host
    check holds its text to snprintf("%g") on 21 M values incl. exact ties of the sixth digit, decade boundaries and specials"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "format_g6_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(root, "tests", "host", "format_g6_check.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()
    assert b" bad 0" in r.stdout


@pytest.mark.parametrize("n_frames", [300, 600])
def test_corner_file_is_read_exactly_and_frames_keep_their_order(tmp_path, n_frames):
    """readCorners (unified_calibration.cpp:252-277) through the streaming reader: every corner equals the value written
    (repr round trip), frames without an entry for the camera are empty, entries of other cameras and unknown keys are skipped,
    the FIRST matching entry of a frame counts; 300 frames so that several host threads share the file, 600 (6 MB) so that the
    file is also CUT into frames by several threads (element_spans_parallel, from 4 MB)"""
    import json
    import os

    d = S.make_mono("eucm", n_frames, 1)
    skip = {3, 17, n_frames - 1}
    frames = []
    for i in range(n_frames):
        # entries of other cameras: their points are never converted (readCorners only touches the matching entry), whether the
        # `points` key stands behind or in front of `camera` -- a one-coordinate corner / a non-number there is not an error
        fr = [{"camera": "other", "points": [[1.0, 2.0], [3.0]], "extra": {"a": [1, {"b": "]"}], "s": 'x"y\\'}},
              {"points": [["u", None]], "camera": "another"}]
        if i not in skip:
            fr.append({"points": d["corners"][i].tolist(), "note": "first", "camera": "cam"})   # keys in another order
            fr.append({"camera": "cam", "points": (d["corners"][i] + 1).tolist()})                # a second match is ignored
        frames.append(fr)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=True)
    json.dump(frames, open(tmp_path / "calib_corners.json", "w"), indent=1)
    assert (os.path.getsize(tmp_path / "calib_corners.json") >= 4 << 20) == (n_frames == 600)
    c = GenericCameraCalibration()
    c.addResiduals(path)
    got = c.corners(0)
    assert len(got) == n_frames
    for i in range(n_frames):
        if i in skip:
            assert got[i] is None
        else:
            assert np.array_equal(got[i], d["corners"][i]), i
    assert c.timings()["json_bytes"] > 0 and c.timings()["parse_json_s"] > 0
    c.close()


@pytest.mark.parametrize("what, msg", [
    ("short", "a frame has 95 corners, the board has 96"),
    ("one_coordinate", "a corner needs two coordinates"),
    ("no_camera_key", "No such node (camera)"),
    ("no_points_key", "No such node (points)"),
    ("trailing", "JSON parse error"),
    ("unterminated", "JSON parse error"),
])
def test_corner_file_errors(tmp_path, what, msg):
    import json

    d = S.make_mono("eucm", 130, 1)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=True)
    frames = [[{"camera": "cam", "points": d["corners"][i].tolist()}] for i in range(130)]
    text = None
    if what == "short":
        frames[77][0]["points"] = frames[77][0]["points"][:-1]
    elif what == "one_coordinate":
        frames[101][0]["points"][5] = [3.0]
    elif what == "no_camera_key":
        del frames[5][0]["camera"]
    elif what == "no_points_key":
        del frames[129][0]["points"]
    elif what == "trailing":
        text = json.dumps(frames) + " ]"
    elif what == "unterminated":
        text = json.dumps(frames)[:-1]
    with open(tmp_path / "calib_corners.json", "w") as f:
        f.write(text if text is not None else json.dumps(frames))
    c = GenericCameraCalibration()
    with pytest.raises(capi.VisgeomError) as e:
        c.addResiduals(path)
    assert msg in str(e.value), str(e.value)
    c.close()


def test_corner_file_survives_mutated_input(tmp_path):
    """byte flips, deletions and insertions in a corner file: an error code or a parse, never a crash -- whichever thread meets it"""
    import json

    d = S.make_mono("eucm", 200, 1)
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=True)
    text = json.dumps([[{"camera": "cam", "points": d["corners"][i].tolist()}] for i in range(200)])
    rng = np.random.default_rng(11)
    ok = bad = 0
    for trial in range(80):
        b = bytearray(text.encode())
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            op = int(rng.integers(0, 3))
            if op == 0:
                b[pos] = int(rng.integers(32, 127))
            elif op == 1:
                del b[pos:pos + int(rng.integers(1, 50))]
            else:
                b.insert(pos, int(rng.choice(list(b'{}[]",:0-9eE.tfn \\'))))
        if trial % 10 == 0:
            b = b[:int(rng.integers(0, len(b)))]
        with open(tmp_path / "calib_corners.json", "wb") as f:
            f.write(bytes(b))
        c = GenericCameraCalibration()
        try:
            c.addResiduals(path)
            ok += 1
        except capi.VisgeomError:
            bad += 1
        finally:
            c.close()
    assert ok + bad == 80 and bad > 20


def test_report_numbers_are_printed_like_the_stream_default(tmp_path):
    """the report and image_error files print numbers the way `ostream << double` does (6 significant digits, "%g"); the fast
    conversion behind them (std::to_chars) must give printf's text for every magnitude"""
    rng = np.random.default_rng(5)
    d = S.make_mono("eucm", 400, 1)
    vals = rng.standard_normal((400, 6)) * 10.0 ** rng.integers(-12, 12, (400, 6))
    vals[0] = [0.0, -0.0, 1e15, 999999.5, 0.0001, 0.00001234]
    vals[1] = [123456.5, 1234567.0, 1e-300, -1e300, 0.1, 100000.0]
    d["init_poses"] = vals
    path = S.write_calibration_json(str(tmp_path), d, "eucm", prior=True)
    c = GenericCameraCalibration()
    c.addResiduals(path)
    lines = [ln for ln in c.report().splitlines() if " : " in ln and ln.split(" : ")[0].isdigit()]
    assert len(lines) == 400

    def row(v):
        s = ["%g" % x for x in v]
        w = max(len(x) for x in s)
        return " ".join(x.rjust(w) for x in s)

    for i, ln in enumerate(lines):
        x = c.transform("xiCamBoard")[i]
        assert ln == "%d : %s %s" % (i, row(x[:3]), row(x[3:])), (i, ln)
    c.close()
