"""Closed-form known answers for the IoU of rotated rectangles (x, y, w, h, yaw) -- arithmetic anyone can redo by hand, so that the three-way
referee of tests/test_postprocess_ref_cpu.py (product clip, oracle vertex collection, raster count) is anchored to something outside this repo.
Two concentric unit squares turned by t against each other intersect in an octagon of area 2 / (1 + sin t + cos t) (t in [0, pi/2])."""
import math


def _concentric(t):
    a = 2.0 / (1.0 + math.sin(t) + math.cos(t))
    return a / (2.0 - a)


S2 = math.sqrt(2.0)
IOU_KATS = [
    # (box a, box b, IoU, what)
    ((0, 0, 1, 1, 0), (0, 0, 1, 1, 0), 1.0, "identical"),
    ((3, -2, 2, 4, 0.7), (3, -2, 2, 4, 0.7 + math.pi), 1.0, "identical up to a half turn"),
    ((0, 0, 1, 1, 0), (0, 0, 1, 1, math.pi / 4), 1.0 / S2, "concentric unit squares at 45 degrees: octagon 2(sqrt2 - 1), IoU = 1/sqrt2"),
    ((0, 0, 1, 1, 0), (0, 0, 1, 1, math.pi / 6), _concentric(math.pi / 6), "concentric unit squares at 30 degrees"),
    ((5, 5, 2, 2, 0.4), (5, 5, 2, 2, 0.4 + math.pi / 3), _concentric(math.pi / 3), "the same at 60 degrees, side 2, off-origin, both turned"),
    ((0, 0, 1, 1, 0), (0.5, 0, 1, 1, 0), 1.0 / 3.0, "axis-aligned, shifted by half a side: 0.5 / 1.5"),
    ((0, 0, 1, 1, 0), (0.5, 0.5, 1, 1, 0), 1.0 / 7.0, "shifted diagonally by half a side: 0.25 / 1.75"),
    ((0, 0, 4, 1, 0), (0, 0, 4, 1, math.pi / 2), 1.0 / 7.0, "a 4 x 1 cross: 1 / 7"),
    ((0, 0, 4, 2, 0), (0, 0, 2, 1, 0), 0.25, "contained: area ratio 2 / 8"),
    ((0, 0, 4, 4, 0), (0, 0, S2, S2, math.pi / 4), 2.0 / 16.0, "a diamond of diagonal 2 inside a 4 x 4 square"),
    ((0, 0, 2, 2, 0), (2, 0, 2, 2, 0), 0.0, "sharing one edge"),
    ((0, 0, 2, 2, 0), (2, 2, 2, 2, 0), 0.0, "sharing one corner"),
    ((0, 0, 2, 2, 0), (1 + S2, 0, 2, 2, math.pi / 4), 0.0, "a diamond's corner touching an edge"),
    ((0, 0, 1, 1, 0), (10, 10, 1, 1, 1.0), 0.0, "disjoint"),
    ((0, 0, 2, 2, 0), (1, 1, S2, S2, math.pi / 4), 1.0 / 11.0, "a diamond of area 2 centred on a corner of a 2 x 2 square: a quarter of it inside, 0.5 / 5.5"),
]
