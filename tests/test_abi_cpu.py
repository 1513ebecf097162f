"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/vo_mi355.h declares; the product never touches the oracle; host logic of the drop-in."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vo_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from visual_slam_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(_lib.SO_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.EXPORTS) == syms


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from visual_slam_amd import _lib
    with pytest.raises(_lib.VoError) as e:
        _lib.Context()
    assert e.value.code == _lib.VO_ERR_HIP


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "visual_slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle/ ", ""), os.path.join(dirpath, f)


def test_dropin_host_logic():
    from visual_slam_amd.vo_utils import VisualOdometry
    K, P = VisualOdometry.load_calib(os.path.join(GOLDEN, "kitti02", "calib.txt"))
    assert K.shape == (3, 3) and P.shape == (3, 4) and K[0, 0] == 718.856 and K[0, 2] == 607.1928 and K[1, 2] == 185.2157
    poses = VisualOdometry.load_poses(os.path.join(GOLDEN, "kitti02", "poses.txt"))
    assert len(poses) == 4 and poses[0].shape == (4, 4) and np.allclose(poses[0], np.eye(4), atol=1e-8)
    assert VisualOdometry.load_poses(os.path.join(GOLDEN, "nope.txt")) is None
    paths = VisualOdometry.load_image_paths(os.path.join(GOLDEN, "kitti02"))
    assert [os.path.basename(p) for p in paths] == ["%06d.png" % i for i in range(4)]
    T = VisualOdometry.form_transf(np.eye(3) * 2, np.array([1, 2, 3]))
    assert T.dtype == np.float64 and T[3].tolist() == [0, 0, 0, 1] and T[:3, 3].tolist() == [1, 2, 3]
    vo = VisualOdometry(os.path.join(GOLDEN, "kitti02"), False)
    # reprojection error: points projected with the same pose have zero error
    rng = np.random.default_rng(0)
    X = np.c_[rng.uniform(-5, 5, (20, 2)), rng.uniform(5, 30, 20)]
    rvec = np.array([0.01, -0.02, 0.03]); tvec = np.array([0.1, 0.2, -0.3])
    th = np.linalg.norm(rvec); k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    Xc = X @ R.T + tvec
    uv = (vo.K @ Xc.T).T; uv = uv[:, :2] / uv[:, 2:]
    err = vo.calc_reprojection_error(vo.K, vo.D, rvec, tvec, X, uv + np.array([3.0, 4.0]))
    assert err.shape == (20,) and np.allclose(err, 5.0, atol=1e-9)
    params = np.hstack([rvec, tvec, X.ravel()])
    assert np.allclose(vo.bundle_adjustment_residuals(params, vo.K, vo.D, uv), 0, atol=1e-9)


def test_synthetic_generator_is_seeded():
    from visual_slam_amd import synth
    a1, b1, K, R, t = synth.make_pair(3, 320, 200)
    a2, b2, *_ = synth.make_pair(3, 320, 200)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and a1.dtype == np.uint8 and a1.shape == (200, 320)
    a3, *_ = synth.make_pair(4, 320, 200)
    assert not np.array_equal(a1, a3)
    assert abs(np.linalg.norm(t) - 1) < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12


def test_orb_pattern_table_hash():
    """include/vo_orb_pattern_31.h is shared by the product and the oracle, so a typo in it would pass every GPU-vs-oracle
    test: pin the 1024 values to the sha256 its header quotes (the scikit-image / Rublee et al. table)."""
    import hashlib
    import re
    src = open(os.path.join(ROOT, "include", "vo_orb_pattern_31.h")).read()
    quoted = re.search(r"\b([0-9a-f]{64})\b", src).group(1)
    body = src[src.index("{", src.index("VO_ORB_PATTERN_31[")) + 1: src.rindex("}")]
    vals = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int8)
    assert vals.shape == (1024,)
    assert vals[:4].tolist() == [8, -3, 9, 5] and vals[-4:].tolist() == [-1, -6, 0, -11]
    assert hashlib.sha256(vals.tobytes()).hexdigest() == quoted == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def test_bench_geometry_matches_the_survey_table():
    """bench.py derives its algorithmic byte counts from the level geometry, not from constants."""
    import bench
    assert bench.level_sizes(1241, 376) == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert bench.level_sizes(1226, 370)[1:] == [(1022, 308), (851, 257), (709, 214), (591, 178), (493, 149), (411, 124), (342, 103)]
    assert bench.level_sizes(3840, 2160) == [(3840, 2160), (3200, 1800), (2667, 1500), (2222, 1250), (1852, 1042), (1543, 868), (1286, 723), (1072, 603)]
    algo, P = bench.algo_bytes_per_frame(1241, 376, 3000)
    assert P == 1444097 and algo["k_blur"] == 2888194 and algo["k_orient"] == 2247000 and algo["k_harris"] == 486000 and algo["k_rbrief"] == 1632000
    assert bench.algo_bytes_per_frame(3840, 2160, 20000)[1] == 25677702
    # PMC summaries taken at the KITTI size are not applied to another frame size
    assert bench.measured_traffic("k_fast_tiles", 64, 3840, 2160) is None
    t = bench.measured_traffic("k_fast_tiles", 64, 1241, 376)
    assert t is None or 0.5 * 64 * P < t < 2.0 * 64 * P


def test_header_is_plain_c_and_matches_the_python_layouts(tmp_path):
    """include/vo_mi355.h compiles as pedantic C99 (the boundary is a C ABI, not C++), and the struct layouts ctypes / numpy
    assume in visual_slam_amd/_lib.py are the ones the C compiler produces."""
    import ctypes as C
    import subprocess
    from visual_slam_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vo_mi355.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vo_orb_params), sizeof(vo_keypoint), sizeof(vo_pair_result),\n'
                   '  offsetof(vo_pair_result, n_kp1), offsetof(vo_pair_result, status), offsetof(vo_keypoint, octave), offsetof(vo_orb_params, trig_mode)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_lib.OrbParams), _lib.KP_DTYPE.itemsize, _lib.PAIR_DTYPE.itemsize, _lib.PAIR_DTYPE.fields["n_kp1"][1],
                   _lib.PAIR_DTYPE.fields["status"][1], _lib.KP_DTYPE.fields["octave"][1], _lib.OrbParams.trig_mode.offset]


def test_c_example_builds_against_the_library():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    out = subprocess.run([os.path.join(ROOT, "examples", "vo_pairs_demo")], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr        # loads (rpath) and parses arguments; no GPU needed for that
