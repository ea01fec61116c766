#!/usr/bin/env python3
"""Stem conv (7x7 s2, NHWC4 -> 64) timing across image shapes and tiles: looks for pitch effects."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadre_amd import hip  # noqa: E402
from tools.gemm_bench import conv_case  # noqa: E402


def main():
    hip.lib()
    for H, W, F in ((288, 288, 512), (144, 256, 1024), (144, 288, 1024), (144, 264, 1024), (144, 256, 256), (84, 84, 4096)):
        row = []
        for tl in (0, 2, 3, 6):
            tf, t = conv_case(F, H, W, 4, 64, 7, 2, 3, False, tile=tl)
            row.append("tile %d %6.1f TF %7.2f ms" % (tl, tf, t * 1e3))
        print("stem %dx%d F=%d: %s" % (H, W, F, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
