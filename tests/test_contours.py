"""Host contour boxes (mf_contour_boxes, Suzuki-Abe border following) against an
independent construction with scipy.ndimage: outer borders = bounding boxes of the
8-connected components; hole borders = bounding boxes of the enclosed 4-connected
background regions grown by one pixel.  cv2 itself is absent: parity with OpenCV is
unpinned (see contours.cpp); this pins the restated algorithm's box SET."""
import numpy as np
import pytest
from scipy import ndimage

from mass_amd import _lib


def contour_boxes(img, reverse=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = max(16, h * w)
    boxes = np.empty((cap, 4), np.int32)
    n = _lib.check(_lib.lib.mf_contour_boxes(img.ctypes.data, h, w, int(reverse), boxes.ctypes.data, cap))
    return boxes[:n]


def expected_boxes(img):
    img = np.asarray(img) != 0
    out = []
    lab, n = ndimage.label(img, structure=np.ones((3, 3)))
    for sl in ndimage.find_objects(lab):
        out.append((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
    bg, nb = ndimage.label(~np.pad(img, 1), structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outer = bg[0, 0]
    for k, sl in enumerate(ndimage.find_objects(bg), start=1):
        if k == outer:
            continue
        y0, y1, x0, x1 = sl[0].start - 1, sl[0].stop - 1, sl[1].start - 1, sl[1].stop - 1      # un-pad
        out.append((x0 - 1, y0 - 1, x1 - x0 + 2, y1 - y0 + 2))
    return sorted(out)


def test_simple_shapes_and_order():
    img = np.zeros((12, 14), np.uint8)
    img[1:4, 2:6] = 1                      # solid block, found first
    img[6:11, 3:10] = 1
    img[8, 5:8] = 0                        # a hole
    b = contour_boxes(img, reverse=False)
    assert b.tolist() == [[2, 1, 4, 3], [3, 6, 7, 5], [4, 7, 5, 3]]
    assert contour_boxes(img, reverse=True).tolist() == b[::-1].tolist()
    assert contour_boxes(np.zeros((5, 5), np.uint8)).shape == (0, 4)
    assert contour_boxes(np.ones((1, 1), np.uint8)).tolist() == [[0, 0, 1, 1]]
    assert contour_boxes(np.ones((4, 6), np.uint8)).tolist() == [[0, 0, 6, 4]]


@pytest.mark.parametrize("density", [0.08, 0.3, 0.5, 0.7, 0.92])
def test_box_set_matches_connected_components(density):
    rng = np.random.default_rng(int(density * 100))
    for trial in range(60):
        h, w = rng.integers(1, 40, 2)
        img = (rng.random((h, w)) < density).astype(np.uint8)
        if trial % 3 == 0:
            img = ndimage.binary_dilation(img, iterations=1).astype(np.uint8)       # blobs with holes
        got = sorted(map(tuple, contour_boxes(img).tolist()))
        assert got == expected_boxes(img), (h, w, img.tolist())
