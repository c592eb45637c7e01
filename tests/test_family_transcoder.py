"""apriltag family description -> this build's family file (robot_camera_calibration_amd/data/family_from_apriltag.py).

The reference's detector is apriltag (real_preprocessing/README.md:15-16,30-36); its users hold apriltag's own family
source (bit order = the layout generator's spiral with bit_x / bit_y, apriltag 3; or row-major, apriltag 2).  The real
tag36h11 table is not in the image and is not reproduced here: the tests use a SYNTHETIC family laid out in an
apriltag-3-style permuted bit order, written as apriltag-style C source, and check (i) the bit mapping cell by cell,
(ii) that tags DRAWN from the permuted description (a renderer of the test's own: cell (bit_x[i], bit_y[i]) of the
8 x 8 tag = bit i of the code, MSB first, exactly how apriltag's quad_decode reads them) decode to the right ids with the
transcoded table -- oracle on CPU, HIP path on the GPU -- with the corners bl, br, tr, tl of camera_pose.cpp:152-155."""
import os
import subprocess
import sys

import numpy as np
import pytest

from robot_camera_calibration_amd import abi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robot_camera_calibration_amd", "data"))
import family_from_apriltag as FA   # noqa: E402


def _spiral_order():
    """an apriltag-3-like bit order for a 6 x 6 payload: quadrant by quadrant, each walking inwards (NOT row-major);
    coordinates count from the border's top-left corner, so payload cells are 1 .. 6"""
    cells, seen = [], set()
    lo, hi = 1, 6
    while lo <= hi:
        ring = [(x, lo) for x in range(lo, hi)] + [(hi, y) for y in range(lo, hi)] + \
               [(x, hi) for x in range(hi, lo, -1)] + [(lo, y) for y in range(hi, lo, -1)]
        if lo == hi:
            ring = [(lo, lo)]
        for c in ring:
            if c not in seen:
                seen.add(c); cells.append(c)
        lo += 1; hi -= 1
    assert len(cells) == 36
    return [c[0] for c in cells], [c[1] for c in cells]


def _c_source(codes, bx, by):
    """the shape of an apriltag 3 family source file (tagXXhYY.c), with comments to be ignored"""
    s = ["/* synthetic test family -- apriltag-3-style source */", "#include <stdlib.h>", '#include "tagTest36.h"', "",
         "static uint64_t codedata[%d] = {" % len(codes)]
    s += ["   0x%016xUL," % c for c in codes]
    s += ["};", "apriltag_family_t *tagTest36_create()", "{", "   apriltag_family_t *tf = calloc(1, sizeof(apriltag_family_t));",
          '   tf->name = strdup("tagTest36");', "   tf->h = 10;", "   tf->ncodes = %d;" % len(codes), "   tf->codes = codedata;",
          "   tf->nbits = 36;", "   tf->bit_x = calloc(36, sizeof(uint32_t));", "   tf->bit_y = calloc(36, sizeof(uint32_t));"]
    for i, (x, y) in enumerate(zip(bx, by)):
        s += ["   tf->bit_x[%d] = %d;   // bit %d" % (i, x, i), "   tf->bit_y[%d] = %d;" % (i, y)]
    s += ["   tf->width_at_border = 8;", "   tf->total_width = 10;", "   tf->reversed_border = false;", "   return tf;", "}"]
    return "\n".join(s) + "\n"


def _draw(code_a3, bx, by, cell):
    """one upright tag straight from the apriltag-style description: 8 x 8 cells in a white quiet zone of 2 cells"""
    t = np.full((12 * cell, 12 * cell), 235, np.uint8)
    t[2 * cell:10 * cell, 2 * cell:10 * cell] = 20                      # border + payload background black
    for i, (x, y) in enumerate(zip(bx, by)):
        if (code_a3 >> (35 - i)) & 1:                                   # quad_decode: bit i arrives i-th, MSB first
            t[(2 + y) * cell:(3 + y) * cell, (2 + x) * cell:(3 + x) * cell] = 235
    return t


def test_bit_mapping_cell_by_cell(tmp_path):
    fam = [int(c) for c in abi.load_family()]
    bx, by = _spiral_order()
    a3 = FA.to_description(fam, bx, by)
    assert a3 != fam                                                    # the order really differs
    src = tmp_path / "tagTest36.c"
    src.write_text(_c_source(a3, bx, by))
    desc = FA.load_description(str(src))
    assert desc["nbits"] == 36 and desc["width_at_border"] == 8 and desc["total_width"] == 10 and desc["ncodes"] == len(fam)
    assert desc["bit_x"] == bx and desc["bit_y"] == by and desc["codes"] == a3
    out = FA.transcode(desc)
    assert out == fam                                                   # round trip to the build's own family
    for w, c in zip(out, a3):                                           # and cell by cell, straight from the two definitions
        for i, (x, y) in enumerate(zip(bx, by)):
            assert (c >> (35 - i)) & 1 == (w >> (35 - ((y - 1) * 6 + (x - 1)))) & 1
    # the command-line tool writes the node's family_file format
    r = subprocess.run([sys.executable, os.path.join(ROOT, "robot_camera_calibration_amd", "data", "family_from_apriltag.py"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    assert [int(l, 16) for l in lines] == fam
    # apriltag 2 descriptions are row-major already
    v2 = "tf->d = 6; tf->black_border = 1; tf->ncodes = 2;\ntf->codes[0] = 0x%xUL; tf->codes[1] = 0x%xUL;\n" % (fam[0], fam[1])
    assert FA.transcode(FA.parse_c_source(v2)) == fam[:2]


def test_refuses_what_the_build_cannot_decode():
    bx, by = _spiral_order()
    good = dict(nbits=36, bit_x=bx, bit_y=by, width_at_border=8, total_width=10, reversed_border=0, codes=[1, 2])
    assert len(FA.transcode(good)) == 2
    for key, val in (("nbits", 25), ("width_at_border", 9), ("total_width", 8), ("reversed_border", 1), ("codes", [1 << 36])):
        bad = dict(good); bad[key] = val
        with pytest.raises(FA.FamilyError):
            FA.transcode(bad)
    twice = dict(good); twice["bit_x"] = [bx[0]] + bx[:-1]               # a cell covered twice
    with pytest.raises(FA.FamilyError):
        FA.transcode(twice)
    with pytest.raises(FA.FamilyError):
        FA.parse_c_source("int x = 3;")


def _scene_from_description():
    """three upright tags of different size drawn from the permuted description into one 640 x 480 mono frame; the family
    handed to the detector is the TRANSCODED one"""
    fam = [int(c) for c in abi.load_family()]
    bx, by = _spiral_order()
    a3 = FA.to_description(fam, bx, by)
    table = np.array(FA.transcode(dict(nbits=36, bit_x=bx, bit_y=by, width_at_border=8, total_width=10, reversed_border=0, codes=a3)), np.uint64)
    frame = np.full((480, 640), 235, np.uint8)
    want = {}
    for tid, cell, (x0, y0) in ((5, 12, (30, 40)), (17, 16, (230, 30)), (40, 20, (60, 230))):
        t = _draw(a3[tid], bx, by, cell)
        frame[y0:y0 + t.shape[0], x0:x0 + t.shape[1]] = t
        bx0, by0, s = x0 + 2 * cell - 0.5, y0 + 2 * cell - 0.5, 8 * cell        # pixel-centre coordinates of the black square's outline
        want[tid] = np.array([[bx0, by0 + s], [bx0 + s, by0 + s], [bx0 + s, by0], [bx0, by0]])    # bl, br, tr, tl
    return frame, table, want


def _fid_cfg(factory, table):
    cfg = factory()
    abi.set_geometry(cfg, 640, 480, abi.RCC_PIX_MONO8)
    abi.set_distortion(cfg, abi.RCC_DIST_NONE, ())
    cfg.undistort = 0
    cfg.batch_capacity = 1
    abi.set_fiducial_target(cfg, table, tag_size=0.1, max_targets=8)
    return cfg


def _check(dets, want):
    assert sorted(int(d.id) for d in dets) == sorted(want)
    for d in dets:
        c = np.array([[d.corners[i][0], d.corners[i][1]] for i in range(4)])
        assert d.hamming == 0 and np.abs(c - want[int(d.id)]).max() <= 0.75, (int(d.id), c, want[int(d.id)])


def test_oracle_decodes_tags_drawn_from_the_permuted_description(oracle):
    frame, table, want = _scene_from_description()
    ctx = oracle.Context(_fid_cfg(oracle.default_config, table))
    n, det, fc = ctx.detect(frame.reshape(-1), 0)
    _check([det[q] for q in range(n)], want)
    ctx.close()


@pytest.mark.gpu
def test_hip_decodes_tags_drawn_from_the_permuted_description():
    frame, table, want = _scene_from_description()
    det = api.Detector(_fid_cfg(api.default_config, table))
    dets, _ = det.detect(np.ascontiguousarray(frame.reshape(1, -1)), 1)
    _check(list(dets), want)
    det.close()
