#!/usr/bin/env python3
"""Generates data/family36b.txt: a 36-bit (6x6 payload) fiducial family for the build's own tests and
benchmarks.  The reference's detector uses AprilTag tag36h11, whose 587-entry table is not in this
image and cannot be regenerated from memory; the decoder therefore takes the table
as data, and this family stands in for it: greedy search with a fixed LCG, minimum Hamming
distance 10 between any two codes under all four rotations (and between a code and its own
rotations), at least 10 bits of each colour.  Deterministic."""
import os

def rot(c):
    # M'[r][c] = M[c][5-r]  (bit index 35 - (r*6+c), MSB first)
    o = 0
    for r in range(6):
        for cc in range(6):
            b = (c >> (35 - (cc * 6 + (5 - r)))) & 1
            o |= b << (35 - (r * 6 + cc))
    return o

def main(n=48, dmin=10):
    fam, x = [], 0x2545F4914F6CDD1D
    while len(fam) < n:
        x = (x * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        c = (x >> 20) & ((1 << 36) - 1)
        if not (10 <= bin(c).count("1") <= 26):
            continue
        rs = [c]
        for _ in range(3):
            rs.append(rot(rs[-1]))
        if min(bin(rs[0] ^ rs[k]).count("1") for k in (1, 2, 3)) < dmin:
            continue
        if all(bin(r ^ f).count("1") >= dmin for f in fam for r in rs):
            fam.append(c)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "family36b.txt"), "w") as fh:
        fh.write("# build-generated 36-bit fiducial family, %d codes, min Hamming distance %d under rotation\n" % (n, dmin))
        for c in fam:
            fh.write("%09x\n" % c)
    print(len(fam), "codes")

if __name__ == "__main__":
    main()
