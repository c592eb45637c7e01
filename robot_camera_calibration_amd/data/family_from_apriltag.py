#!/usr/bin/env python3
"""family_from_apriltag.py -- an AprilTag family description -> the family file of this build.

The reference's detector is the apriltag library behind an apriltag_ros fork (real_preprocessing/README.md:15-16,
30-36); a user of the reference therefore holds apriltag's OWN description of the tag family in use (e.g. tag36h11.c of
AprilRobotics/apriltag).  This build takes the family as data (rcc_config.family_codes; the ROS node's `family_file`):
one code word per line, hexadecimal, the 6 x 6 payload ROW-MAJOR from the top-left cell, MSB first, white = 1, inside a
one-cell black border (8 x 8 cells) with a white quiet zone around it -- geometrically the footprint of apriltag's
36-bit families (width_at_border 8, total_width 10).  apriltag orders the payload bits differently:

  * apriltag 3 (`apriltag_family_t` with nbits / bit_x[] / bit_y[] / width_at_border / total_width / reversed_border):
    quad_decode shifts the sampled bits in MSB first in the order i = 0 .. nbits-1, bit i sitting at cell
    (bit_x[i], bit_y[i]) counted from the top-left corner of the BORDER (so payload cells run 1 .. 6) -- the layout
    generator's spiral, not row-major;
  * apriltag 2 (`d` = 6, `black_border` = 1): row-major, MSB first -- already this build's order.

Input: the family's C source as apriltag ships it (the assignments are read by pattern, nothing is compiled or
executed), or a JSON file {"nbits": 36, "codes": [...], "bit_x": [...], "bit_y": [...], "width_at_border": 8,
"total_width": 10, "reversed_border": false}.  Nothing here contains apriltag's tables: they are the user's input.

usage: family_from_apriltag.py tag36h11.c > family.txt
"""
import json
import re
import sys

PAYLOAD = 6          # cells per side of the payload this build decodes
NBITS = PAYLOAD * PAYLOAD


class FamilyError(ValueError):
    pass


def parse_c_source(text):
    """The fields of an apriltag family out of its C source (apriltag 3: tf->nbits, tf->bit_x[i] = v, tf->codes[i] = 0x..UL
    or a static codedata[] initialiser; apriltag 2: tf->d, tf->black_border).  Comments are dropped first."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    fam = {}

    def scalar(name):
        m = re.search(r"\b%s\s*=\s*(0[xX][0-9a-fA-F]+|\d+|true|false)\b" % name, text)
        if not m:
            return None
        v = m.group(1)
        return {"true": 1, "false": 0}.get(v, int(v, 0) if v not in ("true", "false") else 0)

    for key in ("nbits", "width_at_border", "total_width", "reversed_border", "d", "black_border", "ncodes", "h"):
        v = scalar(key)
        if v is not None:
            fam[key] = v

    def indexed(name):
        items = {int(i): int(v, 0) for i, v in re.findall(r"\b%s\s*\[\s*(\d+)\s*\]\s*=\s*(0[xX][0-9a-fA-F]+|\d+)" % name, text)}
        return [items[k] for k in sorted(items)] if items else None

    for key in ("bit_x", "bit_y"):
        v = indexed(key)
        if v is not None:
            fam[key] = v
    codes = indexed("codes")
    if codes is None:          # static initialiser: uint64_t codedata[587] = { 0x...UL, ... };
        m = re.search(r"\b\w*code\w*\s*\[\s*\d*\s*\]\s*=\s*\{(.*?)\}", text, flags=re.S)
        if m:
            codes = [int(v, 16) for v in re.findall(r"0[xX]([0-9a-fA-F]+)", m.group(1))]
    if not codes:
        raise FamilyError("no code words found in the family description")
    fam["codes"] = codes
    return fam


def load_description(path):
    text = open(path).read()
    if path.endswith(".json") or text.lstrip().startswith("{"):
        return json.loads(text)
    return parse_c_source(text)


def transcode(fam):
    """apriltag family description (dict) -> list of this build's code words (row-major 6 x 6 payload, MSB first)."""
    codes = [int(c) for c in fam["codes"]]
    if "ncodes" in fam and fam["ncodes"] != len(codes):
        raise FamilyError("ncodes = %d but %d code words given" % (fam["ncodes"], len(codes)))
    if "bit_x" in fam or "bit_y" in fam:          # apriltag 3
        nbits = int(fam.get("nbits", len(fam.get("bit_x", []))))
        bx, by = list(fam.get("bit_x", [])), list(fam.get("bit_y", []))
        if nbits != NBITS or len(bx) != NBITS or len(by) != NBITS:
            raise FamilyError("this build decodes 36-bit (6 x 6) payloads; the description has nbits = %d, %d / %d bit positions" % (nbits, len(bx), len(by)))
        wab, tw = int(fam.get("width_at_border", 8)), int(fam.get("total_width", 10))
        if wab != PAYLOAD + 2:
            raise FamilyError("width_at_border = %d: the payload is not enclosed by a one-cell border (this build: 8)" % wab)
        if tw < wab + 2:
            raise FamilyError("total_width = %d leaves no quiet zone around the border" % tw)
        if int(fam.get("reversed_border", 0)):
            raise FamilyError("reversed_border families (white border on black) are not decoded by this build")
        cells = [(x - 1, y - 1) for x, y in zip(bx, by)]          # border-relative -> payload-relative
        if sorted(cells) != [(x, y) for x in range(PAYLOAD) for y in range(PAYLOAD)]:
            raise FamilyError("bit_x / bit_y do not cover the 6 x 6 payload exactly once")
        out = []
        for c in codes:
            if c >> NBITS:
                raise FamilyError("code word 0x%x has more than %d bits" % (c, NBITS))
            w = 0
            for i, (x, y) in enumerate(cells):
                bit = (c >> (NBITS - 1 - i)) & 1          # quad_decode: bit i is shifted in i-th, MSB first
                w |= bit << (NBITS - 1 - (y * PAYLOAD + x))
            out.append(w)
        return out
    d = int(fam.get("d", 0))                          # apriltag 2: row-major already
    if d != PAYLOAD:
        raise FamilyError("neither bit_x / bit_y (apriltag 3) nor d = 6 (apriltag 2) found: not a 36-bit family description")
    if int(fam.get("black_border", 1)) != 1:
        raise FamilyError("black_border = %d: this build decodes a one-cell border" % int(fam["black_border"]))
    for c in codes:
        if c >> NBITS:
            raise FamilyError("code word 0x%x has more than %d bits" % (c, NBITS))
    return codes


def to_description(codes, bit_x, bit_y):
    """The inverse (tests): this build's code words laid out in a given apriltag-3 bit order."""
    out = []
    for w in codes:
        c = 0
        for i, (x, y) in enumerate(zip(bit_x, bit_y)):
            bit = (int(w) >> (NBITS - 1 - ((y - 1) * PAYLOAD + (x - 1)))) & 1
            c |= bit << (NBITS - 1 - i)
        out.append(c)
    return out


def write_family(codes, fh, origin=""):
    fh.write("# 36-bit fiducial family for rcc_config.family_codes / the ROS node's family_file: %d codes, payload row-major, MSB first\n" % len(codes))
    if origin:
        fh.write("# transcoded from %s by family_from_apriltag.py\n" % origin)
    for w in codes:
        fh.write("%09x\n" % w)


def main(argv):
    if len(argv) != 2:
        sys.stderr.write(__doc__)
        return 2
    try:
        codes = transcode(load_description(argv[1]))
    except (FamilyError, OSError, ValueError) as e:
        sys.stderr.write("family_from_apriltag.py: %s\n" % e)
        return 1
    write_family(codes, sys.stdout, origin=argv[1])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
